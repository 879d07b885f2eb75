// taco_persist.h -- interface of the persistent, weight-stationary Tacotron2 decoder loop (taco_persist.hip).
#pragma once
#include "engine.h"

// Everything the persistent decoder needs for one call; all pointers are device memory that stays valid until the call
// returns.  Buffers marked (zeroed) must be zero when the kernel starts.
struct PersistCall {
    int B, Tin, max_len, early_stop, win_len, win_off;
    bool half_w;                    // LSTM matrices in fp16 (tts_hip_tacotron2_infer_f16)
    const float* memory;            // [B * Tin][enc]   encoder outputs, zero at padded tokens
    const float* pm;                // [B * Tin][128]   processed memory
    const uint8_t* mask;            // [B * Tin]
    const int* enc_len;             // [B]
    const float* masks;             // prenet dropout masks [B][max_len][2][256] or null
    const float* pm_fold;           // [B * Tin][PERSIST_NPM]: memory folded through every consumer of the context, column
                                    // blocks PERSIST_COL_* (computed by the caller: tacotron2.hip, fold_memory)
    unsigned long long* xch;        // (zeroed) exchange buffers, persist_xch_u64(B, Tin) entries
    int* flags;                     // (zeroed) [0] abort code, [1] rendezvous counter, [2] steps run
    float* dec_out;                 // (zeroed) [B][max_len][80]
    float* stop_out;                // (zeroed) [B][max_len]
    float* attn_hist;               // (zeroed) [B][max_len][Tin]
    int* lengths;                   // (zeroed) [B]
    int* finished;                  // (zeroed) [B]
};

constexpr int PERSIST_NPM = 4096 + 4096 + 256 + 128;     // columns of pm_fold: att LSTM | dec LSTM | folded prenet | projection
constexpr int PERSIST_COL_ATT = 0, PERSIST_COL_DEC = 4096, PERSIST_COL_F = 8192, PERSIST_COL_P = 8448;
constexpr int PERSIST_MAX_B = 4;

// number of 8-byte entries of the exchange area for a call
size_t persist_xch_u64(int B, int Tin);
// true if this call shape can run on the persistent kernel on the engine's device (batch, LDS footprint, CU count)
bool persist_applicable(const tts_hip_engine* e, int B, int Tin);
// Runs the whole decoder loop (all steps, device-side early stop) in ONE launch on `st`.
// Returns TTS_HIP_OK and *steps_run; 1 if the blocks could not all become resident (nothing was modified) or 2 if a hop
// timed out in mid-loop (outputs partial: the caller re-zeroes them) -- in both cases the caller falls back to the per-step
// graph; or a negative TTS_HIP_E* code.  Synchronizes `st`.
int persist_decode(tts_hip_engine* e, hipStream_t st, const PersistCall& c, int* steps_run);
// load-time part: folded prenet-1 matrix etc. (called from tacotron2_finalize)
int persist_finalize(tts_hip_engine* e, const HostTensor* prenet0, const HostTensor* proj_k, const HostTensor* proj_b,
                     int enc, std::vector<void*>& allocs);
