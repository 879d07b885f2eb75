// taco_fused.h -- interface of the fused two-kernel Tacotron2 decoder step (taco_fused.hip): batches of 3 .. 8 rows.
#pragma once
#include "engine.h"

constexpr int FUSED_CHUNK = 64;          // decoder steps per enqueued chunk (= per hipGraph replay); even
constexpr int FUSED_MAX_B = 8;

// Device-resident loop state of one call (32 bytes, one scalar load per kernel).  Zero-initialised = before step 0.
struct FusedState {
    int t0;                              // first step of the current chunk
    int n_fin;                           // rows whose stop token has fired (counted after the last computed stop token)
    int steps_run;                       // decoder steps fully executed
    int exec_t;                          // t + 1 of the step whose first half (kernel X) decided to run
    int B, max_len, early_stop;
    int pad;
};

// Everything one call needs; all pointers are device memory that stays valid (and in place) while the chunk graphs of the
// call's shape bucket are replayed.  Buffers marked (zeroed) must be zero when the first chunk starts.
struct FusedCall {
    int B, Tin, max_len, early_stop, win_len, win_off;
    bool half_w;                         // LSTM matrices in fp16
    const float* memory;                 // [B * Tin][enc]
    const float* pm;                     // [B * Tin][128]
    const uint8_t* mask;                 // [B * Tin]
    const int* enc_len;                  // [B]
    const float* masks;                  // prenet dropout masks [B][max_len][2][256] or null
    unsigned long long* xch;             // (zeroed) fused_xch_u64(B, Tin, enc) entries
    int* flags;                          // (zeroed) [0] abort code
    const int* bl_err;                   // encoder status word (copied into the chunk report)
    int* report;                         // [16] chunk report written at the end of every chunk: FusedState, abort code, bl_err
    FusedState* state;                   // (zeroed, then fused_init)
    float* hatt; float* hdec;            // (zeroed) [2][B][1024] ping-pong hidden states
    float* catt; float* cdec;            // (zeroed) [B][1024]
    float* ctx;                          // (zeroed) [B][enc]
    float* wprev; float* wcum;           // (zeroed) [B][Tin]
    int* mainatt;                        // (zeroed) [2][B]
    float* dec_out;                      // (zeroed) [B][max_len][80]
    float* stop_out;                     // (zeroed) [B][max_len]
    float* attn_hist;                    // (zeroed) [B][max_len][Tin] or null
    int* lengths; int* finished;         // (zeroed) [B]
    long long* trace = nullptr;          // debug builds only: [128 steps][2 kernels][4 blocks][16 slots] timestamps
};

size_t fused_xch_u64(int B, int Tin, int enc);
// true if this call shape can run on the fused kernels on the engine's device
bool fused_applicable(const tts_hip_engine* e, int B, int Tin);
// sets the loop state for a new call (enqueued on st)
int fused_init(tts_hip_engine* e, hipStream_t st, const FusedCall& c);
// enqueues FUSED_CHUNK decoder steps (2 kernels each), the chunk's tail projection and the chunk advance on st
// (capturable: no synchronisation, no host reads)
int fused_enqueue_chunk(tts_hip_engine* e, hipStream_t st, const FusedCall& c);
