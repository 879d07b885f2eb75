// engine.h -- internal state behind the tts_hip C ABI (include/tts_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/tts_hip.h"

#include "ttsw_host.h"      // HostTensor + the TTSW parser (host-only code, also built under ASan / UBSan)

// Growable device buffer (workspace).  Never shrinks; reallocated only when a larger request arrives.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    float* f() const { return (float*)p; }
};

// ---------------------------------------------------------------- WaveGlow (fixed reference geometry)
struct WgLayerDev {
    float* in_Bt = nullptr;     // [1024 (tanh/sigmoid interleaved per 128-tile)][3*512 taps] (first layer of a flow: [3*16])
    float* cond_Bt = nullptr;   // [32 phases][1024][320] conditioning conv folded with the upsampling kernel
    float* in_bias = nullptr;   // [1024] in_conv bias + cond bias (+ upsampling bias pushed through), same row order
    _Float16* in_Bt16 = nullptr;    // fp16 operands of the optional fp16 path (built on first use): [1024][1536] (taps in
    _Float16* cond_Bt16 = nullptr;  //   chunks of 32), [32][1024][4*96], [512][512]
    _Float16* rs_Bt16 = nullptr;
    _Float16* in_Bt_x3 = nullptr;   // split-fp16 mode: the same three matrices as [2 planes][...] (hi, lo), built on first use
    _Float16* cond_Bt_x3 = nullptr;
    _Float16* rs_Bt_x3 = nullptr;
    float* wino_G = nullptr;    // Winograd form (wn_wino.hip; built on first use): [6][1024][512] tap combinations and the
    float* wino_V = nullptr;    //   conditioning planes of the phase / mixed groups ([8][6][1024][224] / [16][4][1024][320])
    float* wino_Vf = nullptr;   //   three-pass form only: column-selected copies for the frame groups [32][6][1024][224]
    float* rs_Bt = nullptr;     // [512][512] residual half of res_skip (layers 0..6)
    float* rs_bias = nullptr;   // [512]
    int rs_n = 0;
};
struct WgFlowDev {
    int n_rem = 0, n_half = 0;
    float* start_w = nullptr;   // [n_half][512]
    float* start_b = nullptr;   // [512]
    float* end_w = nullptr;     // [8 layers][8][512] skip halves folded into the end conv (rows >= 2*n_half are zero)
    float* end_b = nullptr;     // [8]
    float* inv = nullptr;       // [n_rem][n_rem]  out = audio @ inv
    WgLayerDev layer[8];
};
struct WaveGlowDev {
    bool ready = false;
    WgFlowDev flow[12];
    std::vector<void*> allocs;
    DevBuf x, acts, audio, a0p, io_mel, io_z, io_out, io_zgen;
    bool f16_ready = false, x3_ready = false;
    DevBuf x16, acts16, a0p16, mel16;        // fp16 path: shadow of x, activations, first-layer operand, mel
    int form_mode = 1, last_form = -1;       // tts_hip_set_waveglow_form / tts_hip_last_waveglow_form
    int probe_flow = -1, probe_layer = -1;   // tts_hip_waveglow_probe_acts (test hook): stop after this layer and copy
    float* probe_out = nullptr;              //   its gated activations to this device buffer [B][T * 32][512]
    bool wino_ready = false;                 // Winograd form of the fp32 in-layer GEMM (wn_wino.hip)
    bool wino_legacy_ready = false;          //   ... and the three-pass form's extra weight copies
    DevBuf wino_U, wino_P, wino_mel;         // mel planes; forms 2 / 3 only: transformed inputs [6][M/4][512], products [6][M/4][1024]
};

// ---------------------------------------------------------------- Tacotron2
struct ConvBnDev {              // conv k5 with batch-norm folded in
    float* Bt = nullptr;        // [cout][5 * cin_pad]
    float* bias = nullptr;      // [cout]  folded bias at unmasked rows
    float* altbias = nullptr;   // [cout]  BN(0) = value at masked rows
    int cin = 0, cin_pad = 0, cout = 0;
};
struct LstmDev {                // weights packed gate-interleaved: row r = 4*u + gate
    float* W = nullptr;         // [4u][kin_total]  (input kernel rows then recurrent rows, K contiguous)
    float* b = nullptr;         // [4u]
    _Float16* W16 = nullptr;    // fp16 copy of W for tts_hip_tacotron2_infer_f16 (built on first use)
    int units = 0, kin = 0;
};
// Output of the Tacotron2 encoder for one batch (tts_hip_tacotron2_encode): what the decoder loop needs, in its own device
// buffer so that several encoded batches can be alive at once.
struct tts_hip_encoded {
    int B = 0, Tin = 0, enc = 0;
    DevBuf buf;
    uint8_t* mask = nullptr;            // [B * Tin]  token != pad
    int* enc_len = nullptr;             // [B]
    int* bl_err = nullptr;              // BiLSTM block-exchange status (0 = ok), checked at the decoder's first synchronization
    float* memory = nullptr;            // [B * Tin][enc]
    float* pm = nullptr;                // [B * Tin][128]  processed memory
};

// Identity of an instantiated decoder-chunk hipGraph: every pointer and scalar its 225 kernel nodes have baked in.
struct DecGraphKey {
    const void* ws;
    const void* enc_buf;
    int B, Tin, max_len_bucket, masks, win_len, win_off, half_w, persist_layout;
    bool operator<(const DecGraphKey& o) const {
        return std::tie(ws, enc_buf, B, Tin, max_len_bucket, masks, win_len, win_off, half_w, persist_layout) <
               std::tie(o.ws, o.enc_buf, o.B, o.Tin, o.max_len_bucket, o.masks, o.win_len, o.win_off, o.half_w, o.persist_layout);
    }
};

struct Tacotron2Dev {
    bool ready = false;
    int enc_dim = 512, spk_dim = 0;
    int vocab = 148;                    // rows of the embedding table (the checkpoint's vocabulary)
    float* embeddings = nullptr;        // [vocab][512]
    ConvBnDev enc_conv[3];
    float* bl_in_Bt[2] = {nullptr, nullptr};   // BiLSTM input kernels [1024][512]
    float* bl_in_b[2] = {nullptr, nullptr};    // [1024]
    float* bl_rec[2] = {nullptr, nullptr};     // recurrent kernels transposed [1024][256]
    float* prenet_w0 = nullptr;         // [20][256][4]  (k / 4, output, k % 4)
    float* prenet_w1 = nullptr;         // [256][256]
    LstmDev att, dec;
    float* query_w = nullptr;           // [128][1024]
    float* memory_Bt = nullptr;         // [128][enc]
    float* value_w = nullptr;           // [128]
    float* loc_dense = nullptr;         // [128][32]   (transposed)
    float* proj_w = nullptr;            // [81][1024 + enc]  (80 mel rows + gate row)
    float* proj_b = nullptr;            // [81]
    float* pfold_w = nullptr;           // [256][1024 + enc]  prenet layer 1 folded with the frame projection (taco_persist.hip)
    float* pfold_b = nullptr;           // [256]
    int persist_mode = 3;               // tts_hip_set_decoder_mode: 0 per-step graph only, 1 persistent kernel when allowed, 2 fused
                                        // two-kernel step when allowed, 3 auto (persistent for 1 - 2 rows, fused above)
    int last_path = -1;                 // how the last call ran its loop: 2 fused step, 1 persistent kernel, 0 per-step graph
    bool persist_timed = true;          // persistent kernel: timed optimistic polls on (switched off if they mostly miss)
    int fused_backoff = 0;              // calls to keep off the fused step after one of its exchanges timed out (shared GPU)
    int fused_fail_streak = 0;
    ConvBnDev post_conv[5];
    std::vector<void*> allocs;
    DevBuf ws;                          // per-call workspace arena
    DevBuf io;                          // staging for host callers
    tts_hip_encoded* enc_cache = nullptr;                 // encoder output of tts_hip_tacotron2_infer* calls (reused: stable pointers)
    std::map<DecGraphKey, hipGraphExec_t> graphs;         // instantiated decoder-chunk graphs, replayed across calls
    std::vector<DecGraphKey> graph_order;                 // insertion order (oldest evicted first)
    void* pinned = nullptr;                               // pinned host ring for the decoder loop's chunk reports (2 x 64 bytes)
    hipEvent_t chunk_ev[2] = {nullptr, nullptr};
};

struct MelStftDev {
    bool ready = false;
    float* basis_Bt = nullptr;          // [1056 (= 2*513 padded to 33*32)][1024]
    float* mel_Bt = nullptr;            // [80][544]
    std::vector<void*> allocs;
    DevBuf frames, mag, io_in, io_out;
};

struct TimedLaunch {
    hipEvent_t a, b;
    int kind;
};

struct tts_hip_engine {
    int device = 0;
    int n_cu = 0;                       // compute units of `device`
    hipStream_t stream = nullptr;
    mutable std::string err;
    std::map<std::string, HostTensor> host;
    WaveGlowDev wg;
    Tacotron2Dev taco;
    MelStftDev stft;
    // timing hooks
    bool timing = false;
    std::vector<TimedLaunch> timed;
    std::vector<TimedLaunch> ev_pool;
    double time_sum_us[4] = {0, 0, 0, 0};
    int64_t time_cnt[4] = {0, 0, 0, 0};
};

int set_err(const tts_hip_engine* e, int code, const char* fmt, ...);

// Calls on one handle are serialised by the caller; for the duration of a call its work goes to `stream` when the caller
// passed one (the *_async entry points, tacotron2 encode / decode), else to the handle's own stream.
struct StreamScope {
    tts_hip_engine* e;
    hipStream_t saved;
    StreamScope(tts_hip_engine* eng, void* stream) : e(eng), saved(eng->stream) {
        if (stream) e->stream = (hipStream_t)stream;
    }
    ~StreamScope() { e->stream = saved; }
};
void tacotron2_graphs_clear(tts_hip_engine* e);

#define HIPCHK(e, call)                                                                                         \
    do {                                                                                                        \
        hipError_t _err = (call);                                                                               \
        if (_err != hipSuccess)                                                                                 \
            return set_err((e), TTS_HIP_EHIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_err)); \
    } while (0)

// Winograd form of the WN in-layer GEMM (wn_wino.hip)
int waveglow_build_wino(tts_hip_engine* e, bool legacy_frames);
int waveglow_wino_begin(tts_hip_engine* e, const float* d_mel, int PR, int BT, int T, int form);
int waveglow_wino_layer(tts_hip_engine* e, const WgLayerDev& ly, int i, const float* x, float* acts_i, int PR, int BT, int T);
// timing helpers (engine.hip)
void timing_begin(tts_hip_engine* e, int kind);
void timing_end(tts_hip_engine* e);
void timing_collect(tts_hip_engine* e);

// model entry points (device pointers only)
int waveglow_finalize(tts_hip_engine* e);
int waveglow_run(tts_hip_engine* e, const float* d_mel, int B, int T, const float* d_z, float sigma, float* d_audio,
                 int precision);
void waveglow_free(tts_hip_engine* e);

int tacotron2_finalize(tts_hip_engine* e);
void tacotron2_free(tts_hip_engine* e);

int melstft_finalize(tts_hip_engine* e);
int melstft_run(tts_hip_engine* e, const float* d_audio, int B, int N, float* d_mel);
void melstft_free(tts_hip_engine* e);

// shared helpers
// n floats of device-side samples into `out` on `st` (engine.hip: Philox4x32-10; kind = TTS_HIP_RANDOM_*)
int philox_fill(tts_hip_engine* e, float* out, long long n, uint64_t seed, uint64_t offset, int kind, hipStream_t st);
const HostTensor* find_tensor(const tts_hip_engine* e, const std::string& name);
// uploads a host tensor to a fresh device allocation tracked in `allocs`
int upload(tts_hip_engine* e, const float* src, size_t n, float** dst, std::vector<void*>& allocs);
int dev_alloc(tts_hip_engine* e, size_t n_floats, float** dst, std::vector<void*>& allocs, bool zero);
