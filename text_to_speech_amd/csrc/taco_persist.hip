// taco_persist.hip -- persistent, weight-stationary Tacotron2 decoder loop for gfx950 (MI355X: 256 CUs).
//
// Replaces the per-step kernel chain of tacotron2.hip (7 dependent launches per decoder step, every LSTM weight re-read
// from HBM every step: 72.7 MB / step, 33 us / step at batch 1) for small batches.  Same mathematics as
// /root/reference/architectures/tacotron2_arch.py:629-689 (loop body), :422-486 (cell), :188-203 (prenet) and
// architectures/layers/location_sensitive_attention.py:104-186, re-associated as described below.
//
// ONE launch runs the whole loop.  256 blocks x 4 waves; wave w of block b owns unit u = 4 b + w of BOTH
// LSTMs and keeps its 4 + 4 gate rows in registers for the entire utterance (fp32: 208 VGPRs per lane), so a step
// streams no weights at all.  What a step costs instead is the exchange of small vectors between the CUs.  There is no
// grid barrier (7.7 us on this part): every exchanged value is published as ONE 8-byte (step tag, fp32) agent-scope store
// into a parity double buffer and consumers poll exactly the words they need with 16-byte sc1 loads until the tags match
// (scripts/micro/tagged_exchange.cpp: 1.6 - 1.9 us per hop at 256 blocks).  Six hops per step:
//
//   h_dec(t-1) --A--> p1 --B--> p2 --C--> h_att --D--> q --E--> energies --F--> h_dec(t)
//
// Linear algebra that removes hops and register pressure (all exact up to fp32 re-association, parity-tested):
//   * The attention context ctx = sum_tau w[tau] memory[tau] is never materialised.  Its three consumers are linear in it
//     (attention-LSTM and decoder-LSTM input kernels, projection / gate, and through them the folded prenet), so
//     W ctx = sum_tau w[tau] (W memory[tau]); PM = memory x [W_att_ctx | W_dec_ctx | F_ctx | P_ctx] is one GEMM per
//     utterance, each block keeps its 34 columns of PM in LDS, and every wave turns the Tin energies into softmax weights
//     itself.  That also removes the ctx columns from the register-resident LSTM rows (272 -> 208 registers).
//   * Prenet layer 1 is folded into the frame projection: p1 = relu(W0^T (P c + b)) = relu(F c + W0^T b) with
//     F = W0^T P (256 x (1024 + enc)), so the fed-back frame needs no hop of its own; the frame itself (an output) is
//     computed off the critical path.  Step 0 uses the all-zero go frame: p1 = p2 = 0.
//   * Location conv + dense are one 62 x 128 map (as in tacotron2.hip); the location term of a position only depends on
//     the previous step's alignment, so the wave that owns the position computes it before the query arrives.
//
// Roles inside a block (all blocks run the same code): every wave = one unit of each LSTM; wave 0 = prenet-1 output
// `blk`; wave 1 = projection row `blk` (blk <= 80: 80 mel rows + the gate row, whose owner also does the stop / lengths
// bookkeeping and publishes the finished count) or query dim `blk - 128` (blk >= 128); wave 2 = prenet-2 output `blk`;
// wave 3 = energies of the flattened (row, position) indices blk, blk + 256, ...
//
// Every wait is bounded.  A start-up rendezvous makes sure all 256 blocks are resident before anything is modified; if
// they are not (another kernel owns CUs for too long), the kernel gives up cleanly and the host falls back to the
// per-step graph.
#include "taco_persist.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "xch_util.h"

using namespace ttsgemm;
using namespace ttsxch;

namespace {


constexpr int NBLK = 256;                 // blocks = CUs of an MI355X; 4 units per block x 256 = 1024 units
constexpr int PRE = 256, RNN = 1024, ATT = 128, NMEL = 80, LOCK = 31;
constexpr int PMW = 36;                   // floats per (row, position) of the LDS copy of PM: 16 att | 16 dec | F | proj | pad
constexpr int PM_ATT = PERSIST_COL_ATT, PM_DEC = PERSIST_COL_DEC, PM_F = PERSIST_COL_F, PM_P = PERSIST_COL_P;
constexpr long long SPIN_LIMIT = 1 << 19; // polls of one hop (~1 us each) before giving up
constexpr long long RDV_LIMIT = 1 << 16;  // start-up rendezvous (s_sleep'd polls, ~0.1 s): other kernels may have to drain first
constexpr int ABORT_RENDEZVOUS = 1, ABORT_TIMEOUT = 2;

struct Xch {                              // offsets (in 8-byte entries) inside one parity half of the exchange area
    unsigned hdec, hatt, p1, p2, q, e, ctrl, half;
};
__host__ __device__ inline Xch xch_layout(int B, int Tin) {
    Xch x;
    unsigned o = 0;
    x.hdec = o; o += B * RNN;
    x.hatt = o; o += B * RNN;
    x.p1 = o;   o += B * PRE;
    x.p2 = o;   o += B * PRE;
    x.q = o;    o += B * ATT;
    x.e = o;    o += (unsigned)(B * Tin + 1) & ~1u;
    x.ctrl = o; o += 2;
    x.half = o;
    return x;
}

struct PersistArgs {
    int B, Tin, enc, max_len, early_stop, win_len, win_off;
    const void* Wa; const void* Wd;       // packed LSTM rows [4 u + gate][K]  (fp32, or fp16 when HW)
    int KA, KD;
    const float* ba; const float* bd;     // [4 u + gate]
    const float* Ff; const float* fb;     // folded prenet-1 [256][1024 + enc], bias [256]
    const float* W1t;                     // prenet-2 [out 256][in 256]
    const float* Pw; const float* Pb;     // projection rows [81][1024 + enc], bias [81]
    const float* Wq;                      // [128][1024]
    const float* wloc;                    // [62][128]
    const float* vw;                      // [128]
    const float* PM;                      // [B * Tin][PERSIST_NPM]
    const float* pm;                      // [B * Tin][128]
    const uint8_t* mask;
    const int* enc_len;
    const float* masks;
    u64* xch;
    int* flags;
    float* dec_out; float* stop_out; float* attn_hist;
    int* lengths; int* finished;
    long long* trace;                     // debug builds (-DTTS_DEBUG_HOOKS) only: per-phase timestamps, else null
    int delay[6];                         // hops A..F: first (optimistic) poll this many 10-ns ticks after the hop's anchor; 0 = off
};

// Phase timestamps for scripts/persist_probe.py: only in a build made with -DTTS_DEBUG_HOOKS (csrc/build.sh never passes it).
// Blocks 0, 80, 200 and 255 record wall_clock64() (100 MHz) at TR_SLOTS points of the first TR_STEPS iterations.
#ifdef TTS_DEBUG_HOOKS
constexpr int TR_STEPS = 256, TR_SLOTS = 20;
#define TR(slot)                                                                                                  \
    do {                                                                                                          \
        if (a.trace && lane == 0 && t < TR_STEPS) {                                                               \
            const int tb_ = blk == 0 ? 0 : blk == 80 ? 1 : blk == 200 ? 2 : blk == 255 ? 3 : -1;                  \
            if (tb_ >= 0) a.trace[((size_t)tb_ * TR_STEPS + t) * TR_SLOTS + (slot)] = (long long)wall_clock64(); \
        }                                                                                                         \
    } while (0)
#else
#define TR(slot) do { } while (0)
#endif

// Lane-halving reduction of V (= 4, 8 or 16) per-lane partial sums.  Halving step s pairs lane with lane ^ (1 << s) inside
// its row of 16 (DPP): lanes with bit s clear keep the lower half of the values and receive the partner's, the others the
// upper half -- so after log2 V steps a lane holds ONE value, index = bit reversal of its low log2 V lane bits.  The rest
// of the row is folded with rotations (which preserve those bits), the four rows with two ds_bpermute steps.  Every lane
// whose low bits are bitrev(i) ends up with the wave total of value i.
template <int V>
__device__ __forceinline__ float reduce_multi(float (&acc)[V], int lane) {
    constexpr int LOGV = V == 4 ? 2 : V == 8 ? 3 : 4;
    auto halve = [&](auto S, int half) {
        const bool hi = (lane >> decltype(S)::value) & 1;
#pragma unroll
        for (int i = 0; i < V / 2; ++i) {
            if (i < half) {
                float a_lo = acc[i], a_hi = acc[i + half];
                asm volatile("" : "+v"(a_lo), "+v"(a_hi));      // keeps select(load, load) from becoming an indexed load
                const float send = hi ? a_lo : a_hi;
                const float keep = hi ? a_hi : a_lo;
                acc[i] = keep + row_xor<decltype(S)::value>(send, lane);
            }
        }
    };
    halve(std::integral_constant<int, 0>{}, V / 2);
    halve(std::integral_constant<int, 1>{}, V / 4);
    if constexpr (LOGV >= 3) halve(std::integral_constant<int, 2>{}, V / 8);
    if constexpr (LOGV >= 4) halve(std::integral_constant<int, 3>{}, V / 16);
    float v = acc[0];
    if constexpr (LOGV == 2) v += dpp<DPP_ROR4>(v);
    if constexpr (LOGV <= 3) v += dpp<DPP_ROR8>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// V = 4 (batch 1): the four totals broadcast to every lane through scalar registers instead (no LDS round trips at all)
__device__ __forceinline__ void reduce4_bcast(float (&acc)[4], int lane, float& t0, float& t1, float& t2, float& t3) {
    auto halve = [&](auto S, int half) {
        const bool hi = (lane >> decltype(S)::value) & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (i < half) {
                float a_lo = acc[i], a_hi = acc[i + half];
                asm volatile("" : "+v"(a_lo), "+v"(a_hi));
                const float send = hi ? a_lo : a_hi;
                const float keep = hi ? a_hi : a_lo;
                acc[i] = keep + row_xor<decltype(S)::value>(send, lane);
            }
    };
    halve(std::integral_constant<int, 0>{}, 2);
    halve(std::integral_constant<int, 1>{}, 1);
    float v = acc[0];                                    // lane & 3 = bit reversal of the value index: 0, 2, 1, 3
    v += dpp<DPP_ROR4>(v);
    v += dpp<DPP_ROR8>(v);
    t0 = (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
    t1 = (lane_bcast(v, 2) + lane_bcast(v, 18)) + (lane_bcast(v, 34) + lane_bcast(v, 50));
    t2 = (lane_bcast(v, 1) + lane_bcast(v, 17)) + (lane_bcast(v, 33) + lane_bcast(v, 49));
    t3 = (lane_bcast(v, 3) + lane_bcast(v, 19)) + (lane_bcast(v, 35) + lane_bcast(v, 51));
}
// lane that holds value `idx` after reduce_multi<V> (row 0)
template <int V>
__device__ __forceinline__ int reduced_lane(int idx) {
    constexpr int LOGV = V == 4 ? 2 : V == 8 ? 3 : 4;
    return (int)(__brev((unsigned)idx) >> (32 - LOGV));
}

struct Poller {
    __amdgpu_buffer_rsrc_t rs;            // the whole exchange area
    int* flags;                           // global: [0] abort code
    int* abort_s;                         // LDS: set by any thread of the block that gave up
    mutable int timed_tries = 0, timed_misses = 0;   // block-level timed polls and how many needed the fallback

    __device__ __forceinline__ bool should_stop(long long spins) const {
        if ((spins & 255) == 0 && *(volatile int*)abort_s) return true;
        if ((spins & 2047) == 0 && __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
        return false;
    }
    __device__ __forceinline__ void give_up() const {
        __hip_atomic_store(flags, ABORT_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *(volatile int*)abort_s = 1;
    }

    // Block-cooperative: thread `tid` fetches tagged pairs tid, tid + 256, ... (< npairs) starting at entry `base` and
    // stores their values to dst[2 * pair .. 2 * pair + 1].  All loads of a thread are in flight together.
    // Waiting is done by ONE lane on a sentinel (the last pair, published by the last blocks), sleeping between polls:
    // 65 536 lanes spinning on the exchange area slow every hop down, the one in flight included (scripts/micro/
    // stride_exchange.cpp: 2.00 -> 1.56 us per hop; in this kernel waves reach a hop microseconds before its data exists).
    // `also` (optional): a second pair that thread 255 (another wave) awaits at the same time; its first value goes to *also_dst.
    // Timed optimistic poll first (t_ready != 0): the blocks run in lock step within ~50 ns and a hop's latency is stable,
    // so instead of sentinel + barrier + full poll (two memory round trips after the data became visible) every thread
    // sleeps until the data is due, loads its pairs ONCE and the block votes; only if something was missing does it fall
    // back to the sentinel wait.  At most two such rounds, so the exchange area is never hammered.
    static __device__ __forceinline__ void sleep_until(long long t_ready) {
        while ((long long)wall_clock64() - t_ready < 0) __builtin_amdgcn_s_sleep(1);
    }
    template <int PPT>
    __device__ __forceinline__ void pairs_to_lds(unsigned base, unsigned tag, int npairs, float* dst, int tid,
                                                 long long t_ready = 0, unsigned also = 0xffffffffu,
                                                 int* also_dst = nullptr) const {
        if (t_ready != 0) {
            ++timed_tries;
            sleep_until(t_ready);
            unsigned off[PPT];
            bool need[PPT];
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const int pair = tid + j * 256;
                need[j] = pair < npairs;
                off[j] = (base + 2u * (unsigned)pair) * 8u;
            }
            const bool do_also = tid == 255 && also != 0xffffffffu;
            for (int attempt = 0; attempt < 2; ++attempt) {
                asm volatile("" ::: "memory");
                u32x4 v[PPT], w = {0u, tag, 0u, tag};
                bool ok = true;
#pragma unroll
                for (int j = 0; j < PPT; ++j)
                    if (need[j]) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[j], 0, 16);
                if (do_also) w = __builtin_amdgcn_raw_buffer_load_b128(rs, also * 8u, 0, 16);
#pragma unroll
                for (int j = 0; j < PPT; ++j)
                    if (need[j]) ok = ok && v[j][1] == tag && v[j][3] == tag;
                ok = ok && w[1] == tag && w[3] == tag;
                if (__syncthreads_and(ok)) {
#pragma unroll
                    for (int j = 0; j < PPT; ++j)
                        if (need[j]) {
                            const int pair = tid + j * 256;
                            *reinterpret_cast<f32x2*>(dst + 2 * pair) = f32x2{bitsf(v[j][0]), bitsf(v[j][2])};
                        }
                    if (do_also) *also_dst = (int)w[0];
                    return;
                }
                __builtin_amdgcn_s_sleep(3);
            }
            ++timed_misses;
        }
        if (tid == 0) wait_pair(base + 2u * (unsigned)(npairs - 1), tag);
        if (tid == 255 && also != 0xffffffffu) {
            wait_pair(also, tag);
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, also * 8u, 0, 16);
            *also_dst = (int)w[0];
        }
        __syncthreads();
        unsigned off[PPT];
        bool need[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int pair = tid + j * 256;
            need[j] = pair < npairs;
            off[j] = (base + 2u * (unsigned)pair) * 8u;
        }
        u32x4 v[PPT];
        long long spins = 0;
        while (true) {
            asm volatile("" ::: "memory");                       // the loads must be re-issued every iteration
            bool ok = true;
#pragma unroll
            for (int j = 0; j < PPT; ++j)
                if (need[j]) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[j], 0, 16);       // aux 16 = sc1 (agent scope)
#pragma unroll
            for (int j = 0; j < PPT; ++j)
                if (need[j]) ok = ok && v[j][1] == tag && v[j][3] == tag;
            if (ok) break;
            ++spins;
            if (spins > SPIN_LIMIT) { give_up(); break; }
            if (should_stop(spins)) break;
        }
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            if (need[j]) {
                const int pair = tid + j * 256;
                *reinterpret_cast<f32x2*>(dst + 2 * pair) = f32x2{bitsf(v[j][0]), bitsf(v[j][2])};
            }
    }

    // One pair, the same address in every active lane (a single 16-byte request per poll), sleeping between polls.
    __device__ __forceinline__ void wait_pair(unsigned entry, unsigned tag) const {
        long long spins = 0;
        while (true) {
            asm volatile("" ::: "memory");
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, entry * 8u, 0, 16);
            if (w[1] == tag && w[3] == tag) break;
            ++spins;
            if (spins > SPIN_LIMIT) { give_up(); break; }
            if (should_stop(spins)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }

    // Wave-local: N pairs per lane straight into registers (no LDS, no block barrier); `sentinel` (an entry the wave reads
    // anyway, published late) is awaited first with one request per poll instead of 64 lanes x N.
    template <int N>
    __device__ __forceinline__ void pairs_to_regs(const unsigned (&entry)[N], unsigned tag, f32x2 (&out)[N], unsigned sentinel,
                                                  long long t_ready = 0) const {
        if (t_ready != 0) {
            sleep_until(t_ready);
            for (int attempt = 0; attempt < 2; ++attempt) {
                asm volatile("" ::: "memory");
                u32x4 v[N];
                bool ok = true;
#pragma unroll
                for (int j = 0; j < N; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, entry[j] * 8u, 0, 16);
#pragma unroll
                for (int j = 0; j < N; ++j) ok = ok && v[j][1] == tag && v[j][3] == tag;
                if (__all(ok)) {
#pragma unroll
                    for (int j = 0; j < N; ++j) out[j] = f32x2{bitsf(v[j][0]), bitsf(v[j][2])};
                    return;
                }
                __builtin_amdgcn_s_sleep(3);
            }
        }
        wait_pair(sentinel, tag);
        u32x4 v[N];
        long long spins = 0;
        while (true) {
            asm volatile("" ::: "memory");
            bool ok = true;
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, entry[j] * 8u, 0, 16);
#pragma unroll
            for (int j = 0; j < N; ++j) ok = ok && v[j][1] == tag && v[j][3] == tag;
            if (__all(ok)) break;
            ++spins;
            if (spins > SPIN_LIMIT) { give_up(); break; }
            if (should_stop(spins)) break;
        }
#pragma unroll
        for (int j = 0; j < N; ++j) out[j] = f32x2{bitsf(v[j][0]), bitsf(v[j][2])};
    }
};

// acc[g * NBT + b] += W[g][I0 + i] . x[b][i * 256 + lane * 4 ..] for i in [0, NI): lanes over k, weights in registers
template <int NBT, int I0, int NI, int NW, class WV>
__device__ __forceinline__ void gemv_acc(float (&acc)[4 * NBT], const WV (&W)[4][NW], const float* x, int ldx, int lane) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int b = 0; b < NBT; ++b) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + b * ldx + i * 256 + lane * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const WV wr = W[g][I0 + i];
                const f32x4 w = f32x4{(float)wr[0], (float)wr[1], (float)wr[2], (float)wr[3]};
                float a = acc[g * NBT + b];
                a = fmaf(xv[0], w[0], a);
                a = fmaf(xv[1], w[1], a);
                a = fmaf(xv[2], w[2], a);
                a = fmaf(xv[3], w[3], a);
                acc[g * NBT + b] = a;
            }
        }
}

template <class WV, class EL>
__device__ __forceinline__ WV load_w(const void* base, long long elem) {
    return *reinterpret_cast<const WV*>((const EL*)base + elem);
}

// NBT: batch rows carried (B <= NBT, padded rows compute zeros); KT: ceil(Tin / 64) bound; HW: fp16 LSTM matrices
template <int NBT, int KT, bool HW>
__global__ __launch_bounds__(256) void decoder_persist_kernel(const PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TP = KT * 64;                      // padded position count
    constexpr int WS = TP + 32;                      // alignment rows with a 16-entry zero halo on each side
    constexpr int NPOS = (NBT * TP + NBLK - 1) / NBLK;      // (row, position) pairs a wave 3 may own
    typedef typename std::conditional<HW, f16x4, f32x4>::type wv_t;
    typedef typename std::conditional<HW, _Float16, float>::type wel_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x;
    const int B = a.B, Tin = a.Tin, enc = a.enc, max_len = a.max_len;
    float* hd = lds;                                 // [NBT][1024]  h_dec of the previous step
    float* ha = hd + NBT * RNN;                      // [NBT][1024]  h_att of this step
    float* p2s = ha + NBT * RNN;                     // [NBT][256]
    float* es = p2s + NBT * PRE;                     // [NBT][TP]    energies
    float* wpv = es + NBT * TP;                      // [NBT][WS]    previous alignment   (wave 3's copy)
    float* wcm = wpv + NBT * WS;                     // [NBT][WS]    cumulative alignment (wave 3's copy)
    float* wl = wcm + NBT * WS;                      // [62][128]    folded location map
    float* pms = wl + 2 * LOCK * ATT;                // [NBT * Tin][PMW]
    int* ctl = (int*)(pms + (size_t)NBT * Tin * PMW);    // [0] abort seen by this block, [1] finished count of the step

    // ------------------------------------------------------------------------------------------------ one-time setup
    for (int i = tid; i < NBT * (2 * RNN + PRE + TP + 2 * WS); i += 256) lds[i] = 0.f;
    for (int i = tid; i < 2 * LOCK * ATT; i += 256) wl[i] = a.wloc[i];
    for (int i = tid; i < NBT * Tin * PMW; i += 256) {
        const int r = i / PMW, c = i % PMW;          // r = b * Tin + tau
        float v = 0.f;
        if (r < B * Tin && c < 34) {
            const int col = c < 16 ? PM_ATT + 16 * blk + c : c < 32 ? PM_DEC + 16 * blk + (c - 16) : c == 32 ? PM_F + blk
                                                                                                   : PM_P + min(blk, NMEL);
            v = a.PM[(size_t)r * PERSIST_NPM + col];
        }
        pms[i] = v;
    }
    if (tid < 2) ctl[tid] = 0;

    const int u = blk * 4 + wave;                    // the unit this wave owns in both LSTMs
    // attention LSTM rows: inputs [p2 (256) | ctx (enc, folded away) | h_att (1024)] -> 1 + 4 float4 per gate per lane
    // decoder LSTM rows:   inputs [h_att (1024) | ctx (folded away) | h_dec (1024)]  -> 4 + 4
    wv_t WA[4][5], WD[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const long long ra = (long long)(4 * u + g) * a.KA, rd = (long long)(4 * u + g) * a.KD;
        WA[g][0] = load_w<wv_t, wel_t>(a.Wa, ra + lane * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            WA[g][1 + i] = load_w<wv_t, wel_t>(a.Wa, ra + PRE + enc + i * 256 + lane * 4);
            WD[g][i] = load_w<wv_t, wel_t>(a.Wd, rd + i * 256 + lane * 4);
            WD[g][4 + i] = load_w<wv_t, wel_t>(a.Wd, rd + RNN + enc + i * 256 + lane * 4);
        }
    }
    const f32x4 biasA = *reinterpret_cast<const f32x4*>(a.ba + 4 * u);
    const f32x4 biasD = *reinterpret_cast<const f32x4*>(a.bd + 4 * u);
    // role row of this wave (see the file header)
    const bool is_proj = wave == 1 && blk <= NMEL, is_query = wave == 1 && blk >= 128;
    f32x4 R[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const float* src = wave == 0 ? a.Ff + (size_t)blk * (RNN + enc) + i * 256 + lane * 4
                         : is_proj   ? a.Pw + (size_t)blk * (RNN + enc) + i * 256 + lane * 4
                         : is_query  ? a.Wq + (size_t)(blk - 128) * RNN + i * 256 + lane * 4
                         : wave == 2 && i == 0 ? a.W1t + (size_t)blk * PRE + lane * 4 : nullptr;
        R[i] = src ? *reinterpret_cast<const f32x4*>(src) : zero;
    }
    const float role_bias = wave == 0 ? a.fb[blk] : is_proj ? a.Pb[blk] : 0.f;
    const f32x2 vv = *reinterpret_cast<const f32x2*>(a.vw + lane * 2);      // value vector dims 2 l, 2 l + 1 (wave 3)

    // per-row constants: token mask bits of the positions lane + 64 k, encoder length
    unsigned mbits[NBT];
    int elen[NBT];
#pragma unroll
    for (int b = 0; b < NBT; ++b) {
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int tau = lane + 64 * k;
            if (b < B && tau < Tin && a.mask[(size_t)b * Tin + tau]) m |= 1u << k;
        }
        mbits[b] = m;
        elen[b] = b < B ? a.enc_len[b] : 0;
    }
    // wave 3: the (row, position) pairs it owns and their processed-memory rows (dims 2 l, 2 l + 1)
    f32x2 pmv[NPOS], pmloc[NPOS];
#pragma unroll
    for (int j = 0; j < NPOS; ++j) {
        const int idx = blk + NBLK * j;              // flattened b * Tin + tau
        const f32x2 zero2 = {0.f, 0.f};
        pmv[j] = (wave == 3 && idx < B * Tin) ? *reinterpret_cast<const f32x2*>(a.pm + (size_t)idx * ATT + lane * 2) : zero2;
        pmloc[j] = pmv[j];                           // step 0: both alignments are zero, so the location term is zero
    }

    const Xch X = xch_layout(B, Tin);
    Poller P;
    P.rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, 0x80000000u, 0x00020000);
    P.flags = a.flags;
    P.abort_s = ctl;
    __syncthreads();

    // ------------------------------------------------------------------------------------------------ rendezvous
    if (tid == 0) {
        __hip_atomic_fetch_add(a.flags + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long spins = 0;
        while (__hip_atomic_load(a.flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x) {
            if (++spins > RDV_LIMIT || __hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                int expected = 0;
                __hip_atomic_compare_exchange_strong(a.flags, &expected, ABORT_RENDEZVOUS, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                ctl[0] = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
    if (ctl[0]) return;                              // nothing has been modified: the host falls back

    // ------------------------------------------------------------------------------------------------ loop state
    float accA[4 * NBT], accD[4 * NBT];              // per-lane partial gate sums (attention / decoder LSTM)
#pragma unroll
    for (int i = 0; i < 4 * NBT; ++i) accA[i] = accD[i] = 0.f;
    float c_att = 0.f, c_dec = 0.f;                  // cell states of (unit u, row b) live in lane b of their wave
    float rolec[NBT];                                // wave 0 / 1: context part of the role row's dot product (per lane)
#pragma unroll
    for (int b = 0; b < NBT; ++b) rolec[b] = 0.f;
    int main_att[NBT];                               // arg max of the previous alignment (attention window)
#pragma unroll
    for (int b = 0; b < NBT; ++b) main_att[b] = 0;
    int fin = 0, len = 0;                            // gate owner (block 80, wave 1): lane b keeps row b's bookkeeping
    constexpr int V = 4 * NBT;
    const int wb = lane & (NBT - 1);                 // lane b (< B) of every wave does the cell update of (unit u, row b)
    const bool writer = lane < B;
    const int src_i = reduced_lane<V>(0 * NBT + wb), src_f = reduced_lane<V>(1 * NBT + wb);
    const int src_c = reduced_lane<V>(2 * NBT + wb), src_o = reduced_lane<V>(3 * NBT + wb);
    int steps = 0;
    // anchors of the timed polls (100 MHz wall clock): this block's own h_dec / h_att publish and its A / D arrivals
    // (compiled out of the 4-row variant, which has no register to spare and is not tuned)
    constexpr bool TIMED = NBT <= 2;
    long long t_hdec = 0, t_hatt = 0, t_async = 0, t_dsync = 0;
    auto due = [&](long long anchor, int hop) -> long long {
        if constexpr (!TIMED) return 0;
        return a.delay[hop] > 0 && anchor != 0 ? anchor + a.delay[hop] : 0;
    };
    auto now = [&]() -> long long {
        if constexpr (!TIMED) return 0;
        return (long long)wall_clock64();
    };

    for (int t = 0; t <= max_len; ++t) {
        const unsigned tag = (unsigned)t + 1;        // values produced in iteration t carry tag t + 1
        const unsigned cur = (t & 1) * X.half, prv = ((t + 1) & 1) * X.half;   // parity halves of this / the previous step
        // ---------------------------------------------------------------- A: h_dec(t - 1) -> p1(t), frame(t - 1), stop(t - 1)
        if (wave == 0) TR(0);
        if (t > 0) {
            P.template pairs_to_lds<(NBT * RNN / 2 + 255) / 256>(prv + X.hdec, (unsigned)t, B * RNN / 2, hd, tid, due(t_hdec, 0));
            __syncthreads();
            if (ctl[0]) break;
        }
        t_async = now();
        if (wave == 0) TR(1);
        if (wave == 0 || is_proj) {
            float s[NBT];
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                float acc = rolec[b];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(hd + b * RNN + i * 256 + lane * 4);
                    acc = fmaf(xv[0], R[i][0], acc);
                    acc = fmaf(xv[1], R[i][1], acc);
                    acc = fmaf(xv[2], R[i][2], acc);
                    acc = fmaf(xv[3], R[i][3], acc);
                }
                s[b] = wave_sum(acc) + role_bias;
            }
            if (wave == 0) {                         // prenet layer 1 (folded with the projection); go frame at t = 0
                if (lane < B) {
                    float v = 0.f;
#pragma unroll
                    for (int b = 0; b < NBT; ++b) v = lane == b ? s[b] : v;
                    v = t == 0 ? 0.f : fmaxf(v, 0.f);
                    if (a.masks && t < max_len) v *= a.masks[((size_t)lane * max_len + t) * 2 * PRE + blk];
                    publish(a.xch + cur + X.p1 + lane * PRE + blk, tag, v);
                }
            } else if (t > 0) {                      // frame / stop token of step t - 1
                if (lane < B) {
                    float v = 0.f;
#pragma unroll
                    for (int b = 0; b < NBT; ++b) v = lane == b ? s[b] : v;
                    if (blk < NMEL) {
                        a.dec_out[((size_t)lane * max_len + (t - 1)) * NMEL + blk] = v;
                    } else {
                        const float sp = sigmoid_exact(v);
                        a.stop_out[(size_t)lane * max_len + (t - 1)] = sp;
                        if (!fin && sp > 0.5f) fin = 1;          // finished |= stop > 0.5 ; lengths += !finished  (:664-665)
                        if (!fin) len += 1;
                    }
                }
            }
            if (blk == NMEL && wave == 1) {          // the gate owner tells everybody how many rows have finished
                const int nf = __popcll(__ballot(fin != 0 && lane < B));
                if (lane < 2) publish(a.xch + cur + X.ctrl + lane, tag, bitsf((unsigned)nf));   // both halves of the 16-byte poll
            }
        }
        if (wave == 0) TR(2);
        if (t >= max_len) { steps = max_len; break; }
        // decoder LSTM: recurrent part of step t (off the critical path)
#pragma unroll
        for (int i = 0; i < V; ++i) accD[i] = 0.f;
        gemv_acc<NBT, 4, 4>(accD, WD, hd, RNN, lane);
        // ---------------------------------------------------------------- B: p1(t) -> p2(t)   (wave 2, registers only)
        if (wave == 2) {
            TR(10);
            float s[NBT];
            unsigned ent[2 * NBT];                   // rows beyond B re-read row 0 (their values are not used)
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                const int bb = b < B ? b : 0;
                ent[2 * b] = cur + X.p1 + bb * PRE + lane * 4;
                ent[2 * b + 1] = ent[2 * b] + 2;
            }
            f32x2 pv[2 * NBT];
            P.template pairs_to_regs<2 * NBT>(ent, tag, pv, cur + X.p1 + (B - 1) * PRE + PRE - 2, due(t_async, 1));
            TR(11);
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                const float acc = pv[2 * b][0] * R[0][0] + pv[2 * b][1] * R[0][1] + pv[2 * b + 1][0] * R[0][2] + pv[2 * b + 1][1] * R[0][3];
                s[b] = wave_sum(b < B ? acc : 0.f);
            }
            if (lane < B) {
                float v = 0.f;
#pragma unroll
                for (int b = 0; b < NBT; ++b) v = lane == b ? s[b] : v;
                v = fmaxf(v, 0.f);
                if (a.masks) v *= a.masks[((size_t)lane * max_len + t) * 2 * PRE + PRE + blk];
                publish(a.xch + cur + X.p2 + lane * PRE + blk, tag, v);
            }
            TR(12);
        }
        if (wave == 0) TR(3);
        // ---------------------------------------------------------------- C: p2(t) + finished count -> h_att(t)
        P.template pairs_to_lds<(NBT * PRE / 2 + 255) / 256>(cur + X.p2, tag, B * PRE / 2, p2s, tid, due(t_async, 2), cur + X.ctrl, ctl + 1);
        __syncthreads();
        if (ctl[0]) break;
        if (a.early_stop && ctl[1] >= B) { steps = t; break; }
        if (wave == 0) TR(4);
        {
            gemv_acc<NBT, 0, 1>(accA, WA, p2s, PRE, lane);
            float gi, gf, gg, go;
            if constexpr (NBT == 1) {
                reduce4_bcast(accA, lane, gi, gf, gg, go);
            } else {
                const float v = reduce_multi<V>(accA, lane);
                gi = __shfl(v, src_i, 64), gf = __shfl(v, src_f, 64), gg = __shfl(v, src_c, 64), go = __shfl(v, src_o, 64);
            }
            if (writer) {
                const float ig = sigmoid_fast(gi + biasA[0]), fg = sigmoid_fast(gf + biasA[1]);
                const float cg = tanh_fast(gg + biasA[2]), og = sigmoid_fast(go + biasA[3]);
                c_att = fg * c_att + ig * cg;
                publish(a.xch + cur + X.hatt + wb * RNN + u, tag, og * tanh_fast(c_att));
            }
            t_hatt = now();
        }
        if (wave == 0) TR(5);
        // ---------------------------------------------------------------- D: h_att(t) -> q(t); LSTM partial sums
        P.template pairs_to_lds<(NBT * RNN / 2 + 255) / 256>(cur + X.hatt, tag, B * RNN / 2, ha, tid, due(t_hatt, 3));
        __syncthreads();
        if (ctl[0]) break;
        t_dsync = now();
        if (wave == 0) TR(6);
        if (is_query) {
            float s[NBT];
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(ha + b * RNN + i * 256 + lane * 4);
                    acc = fmaf(xv[0], R[i][0], acc);
                    acc = fmaf(xv[1], R[i][1], acc);
                    acc = fmaf(xv[2], R[i][2], acc);
                    acc = fmaf(xv[3], R[i][3], acc);
                }
                s[b] = wave_sum(acc);
            }
            if (lane < B) {
                float v = 0.f;
#pragma unroll
                for (int b = 0; b < NBT; ++b) v = lane == b ? s[b] : v;
                publish(a.xch + cur + X.q + lane * ATT + (blk - 128), tag, v);
            }
        }
        {   // h_att part of the decoder LSTM (this step) and recurrent part of the attention LSTM (next step)
            gemv_acc<NBT, 0, 4>(accD, WD, ha, RNN, lane);
#pragma unroll
            for (int i = 0; i < V; ++i) accA[i] = 0.f;
            gemv_acc<NBT, 1, 4>(accA, WA, ha, RNN, lane);
        }
        if (wave == 0) TR(7);
        // ---------------------------------------------------------------- E: q(t) -> energies(t)   (wave 3, registers only)
        if (wave == 3) {
            TR(13);
            f32x2 qv[NBT];
            {
                unsigned ent[NBT];
#pragma unroll
                for (int b = 0; b < NBT; ++b) ent[b] = cur + X.q + (b < B ? b : 0) * ATT + lane * 2;
                P.template pairs_to_regs<NBT>(ent, tag, qv, cur + X.q + (B - 1) * ATT + ATT - 2, due(t_dsync, 4));
            }
            TR(14);
#pragma unroll
            for (int j = 0; j < NPOS; ++j) {
                const int idx = blk + NBLK * j;
                if (idx < B * Tin) {                 // wave-uniform
                    const int b = idx / Tin;
                    f32x2 qb = qv[0];
#pragma unroll
                    for (int bb = 1; bb < NBT; ++bb) qb = b == bb ? qv[bb] : qb;
                    float e = vv[0] * tanh_fast(qb[0] + pmloc[j][0]);
                    e = fmaf(vv[1], tanh_fast(qb[1] + pmloc[j][1]), e);
                    e = wave_sum(e);
                    if (lane == 0) publish(a.xch + cur + X.e + idx, tag, e);
                    // an odd number of energies: the last 16-byte poll also covers one padding entry, which must carry the tag
                    if (lane == 1 && idx == B * Tin - 1 && ((B * Tin) & 1)) publish(a.xch + cur + X.e + idx + 1, tag, 0.f);
                }
            }
            TR(15);
        }
        // ---------------------------------------------------------------- F: energies(t) -> alignment -> h_dec(t)
        P.template pairs_to_lds<(NBT * TP / 2 + 255) / 256>(cur + X.e, tag, (B * Tin + 1) / 2, es, tid, due(t_dsync, 5));
        __syncthreads();
        if (ctl[0]) break;
        if (wave == 0) TR(8);
        float wgt[NBT][KT];                          // softmax weights of the positions lane + 64 k
#pragma unroll
        for (int b = 0; b < NBT; ++b) {
            // attention window (tacotron2_arch.py:630-638); inclusive upper bound
            int lo = 0, hi = Tin;
            if (a.win_len > 0) {
                int center = max(main_att[b], a.win_off);
                center = min(center, elen[b] - a.win_len + a.win_off);
                lo = center - a.win_off;
                hi = center - a.win_off + a.win_len;
            }
            float ev[KT];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int tau = lane + 64 * k;
                bool on = (mbits[b] >> k) & 1u;
                if (a.win_len > 0) on = on && tau >= lo && tau <= hi;
                ev[k] = on ? es[b * Tin + tau] : -INFINITY;      // es is packed [B][Tin] like the exchange area
                mx = fmaxf(mx, ev[k]);
            }
            mx = wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                ev[k] = b < B ? __builtin_amdgcn_exp2f((ev[k] - mx) * 1.4426950408889634f) : 0.f;   // 0 at masked positions
                sum += ev[k];
            }
            const float rsum = __builtin_amdgcn_rcpf(wave_sum(sum));
#pragma unroll
            for (int k = 0; k < KT; ++k) wgt[b][k] = b < B ? ev[k] * rsum : 0.f;
            if (a.win_len > 0) {                     // arg max, first index on ties
                float best = -1.f;
                int besti = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < KT; ++k)
                    if (wgt[b][k] > best) { best = wgt[b][k]; besti = lane + 64 * k; }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    const float ob = __shfl_xor(best, m, 64);
                    const int oi = __shfl_xor(besti, m, 64);
                    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
                }
                main_att[b] = besti;
            }
        }
        // context part of the decoder LSTM, then its gates
#pragma unroll
        for (int b = 0; b < NBT; ++b)
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int tau = min(lane + 64 * k, Tin - 1);     // lanes beyond Tin carry weight 0
                const f32x4 pd = *reinterpret_cast<const f32x4*>(pms + ((size_t)b * Tin + tau) * PMW + 16 + 4 * wave);
#pragma unroll
                for (int g = 0; g < 4; ++g) accD[g * NBT + b] = fmaf(wgt[b][k], pd[g], accD[g * NBT + b]);
            }
        {
            float gi, gf, gg, go;
            if constexpr (NBT == 1) {
                reduce4_bcast(accD, lane, gi, gf, gg, go);
            } else {
                const float v = reduce_multi<V>(accD, lane);
                gi = __shfl(v, src_i, 64), gf = __shfl(v, src_f, 64), gg = __shfl(v, src_c, 64), go = __shfl(v, src_o, 64);
            }
            if (writer) {
                const float ig = sigmoid_fast(gi + biasD[0]), fg = sigmoid_fast(gf + biasD[1]);
                const float cg = tanh_fast(gg + biasD[2]), og = sigmoid_fast(go + biasD[3]);
                c_dec = fg * c_dec + ig * cg;
                publish(a.xch + cur + X.hdec + wb * RNN + u, tag, og * tanh_fast(c_dec));
            }
            t_hdec = now();
        }
        if (wave == 0) TR(9);
        // ---- everything below is off the critical path (it overlaps the wait for hop A of the next step) ----
        // context parts of: the attention LSTM (next step), the folded prenet row (wave 0), the projection row (wave 1)
#pragma unroll
        for (int b = 0; b < NBT; ++b) {
            float rc = 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int tau = min(lane + 64 * k, Tin - 1);
                const float* row = pms + ((size_t)b * Tin + tau) * PMW;
                const f32x4 pa = *reinterpret_cast<const f32x4*>(row + 4 * wave);
#pragma unroll
                for (int g = 0; g < 4; ++g) accA[g * NBT + b] = fmaf(wgt[b][k], pa[g], accA[g * NBT + b]);
                if (wave < 2) rc = fmaf(wgt[b][k], row[32 + wave], rc);
            }
            rolec[b] = rc;
        }
        if (wave == 3) {
            // this wave's copy of the alignments, then the location term of the positions it owns for the next step
#pragma unroll
            for (int b = 0; b < NBT; ++b)
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const int tau = lane + 64 * k;
                    if (b < B && tau < Tin) {
                        wpv[b * WS + 16 + tau] = wgt[b][k];
                        wcm[b * WS + 16 + tau] += wgt[b][k];
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave: LDS operations complete in order
#pragma unroll
            for (int j = 0; j < NPOS; ++j) {
                const int idx = blk + NBLK * j;
                if (idx < B * Tin) {
                    const int b = idx / Tin, tau = idx - b * Tin;
                    const float* pw = wpv + b * WS + tau + 1;    // entry i: position tau + i - 15
                    const float* cw = wcm + b * WS + tau + 1;
                    f32x2 loc = pmv[j];
#pragma unroll
                    for (int i = 0; i < LOCK; ++i) {
                        const float sp = pw[i], sc = cw[i];      // wave-uniform addresses: LDS broadcast
                        const f32x2 w0 = *reinterpret_cast<const f32x2*>(wl + (2 * i) * ATT + lane * 2);
                        const f32x2 w1 = *reinterpret_cast<const f32x2*>(wl + (2 * i + 1) * ATT + lane * 2);
                        loc[0] = fmaf(sp, w0[0], loc[0]);
                        loc[1] = fmaf(sp, w0[1], loc[1]);
                        loc[0] = fmaf(sc, w1[0], loc[0]);
                        loc[1] = fmaf(sc, w1[1], loc[1]);
                    }
                    pmloc[j] = loc;
                }
            }
            TR(16);
        }
        if (a.attn_hist && blk == NBLK - 1 && wave == 2) {       // alignment history (an output)
#pragma unroll
            for (int b = 0; b < NBT; ++b)
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const int tau = lane + 64 * k;
                    if (b < B && tau < Tin) a.attn_hist[((size_t)b * max_len + t) * Tin + tau] = wgt[b][k];
                }
        }
    }
    if (blk == NMEL && wave == 1) {
        if (lane < B) {
            a.lengths[lane] = len;
            a.finished[lane] = fin;
        }
        if (lane == 0) a.flags[2] = steps;
    }
    if (blk == 0 && tid == 0) {                      // how the timed polls of the two long hops (C, F) fared (host: keep or drop them)
        a.flags[3] = P.timed_tries;
        a.flags[4] = P.timed_misses;
    }
}

size_t persist_lds_bytes(int NBT, int KT, int Tin) {
    const size_t TP = (size_t)KT * 64, WS = TP + 32;
    return ((size_t)NBT * (2 * RNN + PRE + TP + 2 * WS) + 2 * LOCK * ATT + (size_t)NBT * Tin * PMW + 4) * sizeof(float);
}

template <int NBT, int KT, bool HW>
hipError_t launch_persist(hipStream_t st, const PersistArgs& args, size_t lds) {
    auto kern = decoder_persist_kernel<NBT, KT, HW>;
    static PerDeviceOnce attr;
    if (hipError_t er = set_max_dyn_lds_once((const void*)kern, lds, attr); er != hipSuccess) return er;
    // A plain launch: the kernel never calls grid.sync() -- it has its own start-up rendezvous, which gives up cleanly when
    // the blocks cannot all be resident -- so the cooperative launch API bought nothing except its occupancy check (done
    // here), and it makes HIP create a separate cooperative queue whose teardown crashed rocprofv3 at process exit.
    int per_cu = 0, dev = 0, n_cu = 0;
    if (hipError_t er = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, lds); er != hipSuccess) return er;
    if (hipError_t er = hipGetDevice(&dev); er != hipSuccess) return er;
    if (hipError_t er = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); er != hipSuccess) return er;
    if ((long long)per_cu * n_cu < NBLK) return hipErrorCooperativeLaunchTooLarge;      // the caller falls back to the per-step graph
    hipLaunchKernelGGL(kern, dim3(NBLK), dim3(256), lds, st, args);
    return hipGetLastError();
}

template <bool HW>
hipError_t dispatch_persist(hipStream_t st, const PersistArgs& args, int NBT, int KT, size_t lds) {
    switch (NBT * 16 + KT) {
        case 1 * 16 + 2: return launch_persist<1, 2, HW>(st, args, lds);
        case 1 * 16 + 4: return launch_persist<1, 4, HW>(st, args, lds);
        case 1 * 16 + 8: return launch_persist<1, 8, HW>(st, args, lds);
        case 2 * 16 + 2: return launch_persist<2, 2, HW>(st, args, lds);
        case 2 * 16 + 4: return launch_persist<2, 4, HW>(st, args, lds);
        case 4 * 16 + 2: return launch_persist<4, 2, HW>(st, args, lds);
        default: return hipErrorInvalidValue;
    }
}

void pick_shape(int B, int Tin, int* NBT, int* KT) {
    *NBT = B <= 1 ? 1 : B <= 2 ? 2 : 4;
    *KT = Tin <= 128 ? 2 : Tin <= 256 ? 4 : 8;
}

}  // namespace

size_t persist_xch_u64(int B, int Tin) { return 2 * (size_t)xch_layout(B, Tin).half; }

bool persist_applicable(const tts_hip_engine* e, int B, int Tin) {
    if (!e->taco.pfold_w) return false;
    if (e->n_cu < NBLK || B < 1 || B > PERSIST_MAX_B || Tin < 1 || Tin > 512) return false;
    int NBT, KT;
    pick_shape(B, Tin, &NBT, &KT);
    if ((NBT == 2 && KT > 4) || (NBT == 4 && KT > 2)) return false;
    return persist_lds_bytes(NBT, KT, Tin) <= 160 * 1024;
}

int persist_finalize(tts_hip_engine* e, const HostTensor* prenet0, const HostTensor* proj_k, const HostTensor* proj_b,
                     int enc, std::vector<void*>& allocs) {
    // F[o][k] = sum_m W0[m][o] P[k][m]   (W0: Keras [80][256], P: Keras [1024 + enc][80]);  fb[o] = sum_m W0[m][o] pb[m]
    const int K = RNN + enc;
    std::vector<float> F((size_t)PRE * K), fb(PRE);
    for (int o = 0; o < PRE; ++o) {
        for (int k = 0; k < K; ++k) {
            double s = 0;
            for (int m = 0; m < NMEL; ++m) s += (double)prenet0->data[(size_t)m * PRE + o] * (double)proj_k->data[(size_t)k * NMEL + m];
            F[(size_t)o * K + k] = (float)s;
        }
        double s = 0;
        for (int m = 0; m < NMEL; ++m) s += (double)prenet0->data[(size_t)m * PRE + o] * (double)proj_b->data[m];
        fb[o] = (float)s;
    }
    int rc;
    if ((rc = upload(e, F.data(), F.size(), &e->taco.pfold_w, allocs))) return rc;
    return upload(e, fb.data(), fb.size(), &e->taco.pfold_b, allocs);
}

int persist_decode(tts_hip_engine* e, hipStream_t st, const PersistCall& c, int* steps_run) {
    Tacotron2Dev& tc = e->taco;
    const int enc = tc.enc_dim;
    int NBT, KT;
    pick_shape(c.B, c.Tin, &NBT, &KT);
    PersistArgs a{};
    a.B = c.B; a.Tin = c.Tin; a.enc = enc; a.max_len = c.max_len; a.early_stop = c.early_stop;
    a.win_len = c.win_len; a.win_off = c.win_off;
    a.Wa = c.half_w ? (const void*)tc.att.W16 : (const void*)tc.att.W;
    a.Wd = c.half_w ? (const void*)tc.dec.W16 : (const void*)tc.dec.W;
    a.KA = PRE + enc + RNN; a.KD = RNN + enc + RNN;
    a.ba = tc.att.b; a.bd = tc.dec.b;
    a.Ff = tc.pfold_w; a.fb = tc.pfold_b;
    a.W1t = tc.prenet_w1;
    a.Pw = tc.proj_w; a.Pb = tc.proj_b;
    a.Wq = tc.query_w; a.wloc = tc.loc_dense; a.vw = tc.value_w;
    a.PM = c.pm_fold; a.pm = c.pm; a.mask = c.mask; a.enc_len = c.enc_len; a.masks = c.masks;
    a.xch = c.xch; a.flags = c.flags;
    a.dec_out = c.dec_out; a.stop_out = c.stop_out; a.attn_hist = c.attn_hist;
    a.lengths = c.lengths; a.finished = c.finished;
    // First-poll delays (10-ns ticks after the hop's anchor; hops A..F), found with scripts/persist_sweep.py on 100-token
    // utterances: batch 1 10.7 -> 9.3 us / step, batch 2 14.1 -> 12.8.  The two all-to-all LSTM hops (A, D) stay on the
    // sentinel path (their 1024 values only become visible ~0.9 us after the publish: a timed full poll cannot beat sentinel +
    // poll there); the 4-row variant is not tuned.  A call whose timed polls mostly missed (other clocks, another part)
    // switches them off for the engine's later calls.
    static const int kDelay[2][6] = {{0, 100, 205, 0, 70, 170}, {0, 115, 230, 0, 82, 190}};
    for (int i = 0; i < 6; ++i) a.delay[i] = (NBT <= 2 && e->taco.persist_timed) ? kDelay[NBT - 1][i] : 0;
#ifdef TTS_DEBUG_HOOKS
    if (const char* dl = getenv("TTS_PERSIST_DELAYS")) {          // "a,b,c,d,e,f"
        int v[6] = {0, 0, 0, 0, 0, 0};
        if (sscanf(dl, "%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) == 6)
            for (int i = 0; i < 6; ++i) a.delay[i] = v[i];
    }
    const char* trace_file = getenv("TTS_PERSIST_TRACE_FILE");
    const size_t trace_n = (size_t)4 * TR_STEPS * TR_SLOTS;
    if (trace_file) {
        HIPCHK(e, hipMalloc((void**)&a.trace, trace_n * sizeof(long long)));
        HIPCHK(e, hipMemsetAsync(a.trace, 0, trace_n * sizeof(long long), st));
    }
#endif
    const size_t lds = persist_lds_bytes(NBT, KT, c.Tin);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (e->timing) {
        HIPCHK(e, hipEventCreate(&ev0));
        HIPCHK(e, hipEventCreate(&ev1));
        HIPCHK(e, hipEventRecord(ev0, st));
    }
    hipError_t er = c.half_w ? dispatch_persist<true>(st, a, NBT, KT, lds) : dispatch_persist<false>(st, a, NBT, KT, lds);
    if (er == hipErrorCooperativeLaunchTooLarge || er == hipErrorInvalidConfiguration) {
        (void)hipGetLastError();
        return 1;                                    // this device cannot hold the grid: per-step graph instead
    }
    HIPCHK(e, er);
    if (e->timing) HIPCHK(e, hipEventRecord(ev1, st));
    int h[5] = {0, 0, 0, 0, 0};
    HIPCHK(e, hipMemcpyAsync(h, c.flags, sizeof h, hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    if (h[3] >= 64 && h[4] * 4 > h[3]) e->taco.persist_timed = false;     // more than a quarter missed: not worth it here
    if (e->timing) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess && h[0] == 0 && h[2] > 0) {
            e->time_sum_us[2] += 1e3 * ms;           // kind 2 = decoder step: total / steps gives the per-step average
            e->time_cnt[2] += h[2];
        }
        (void)hipEventDestroy(ev0);
        (void)hipEventDestroy(ev1);
    }
#ifdef TTS_DEBUG_HOOKS
    if (a.trace) {
        std::vector<long long> ht(trace_n);
        (void)hipMemcpy(ht.data(), a.trace, trace_n * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(a.trace);
        if (FILE* f = fopen(trace_file, "wb")) {
            fwrite(ht.data(), sizeof(long long), trace_n, f);
            fclose(f);
        }
    }
#endif
    if (h[0] == ABORT_RENDEZVOUS) return 1;
    if (h[0] != 0) {
        // a hop timed out in mid-loop (never seen; e.g. a block lost its CU for a second): the outputs are partial -- the
        // caller clears them and runs the per-step graph instead
        set_err(e, TTS_HIP_EHIP, "tacotron2 persistent decoder: exchange timed out at a hop (code %d); fell back to the per-step graph", h[0]);
        return 2;
    }
    *steps_run = h[2];
    return TTS_HIP_OK;
}
