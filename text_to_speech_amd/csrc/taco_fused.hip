// taco_fused.hip -- Tacotron2 decoder step as TWO kernels with in-kernel exchanges, for batches of 3 .. 8 rows on gfx950.
//
// Same mathematics as /root/reference/architectures/tacotron2_arch.py:629-689 (loop body), :422-486 (cell), :188-203
// (prenet) and architectures/layers/location_sensitive_attention.py:104-186.
//
// The per-step chain of tacotron2.hip (7 dependent launches) spends half of a batch-8 step in five small kernels that move
// < 1.4 MB each and are pure launch / dependency latency, and its two LSTM kernels only start streaming their weights once
// the chain has reached them.  The persistent kernel of taco_persist.hip (weights in registers, nothing streamed) stops
// paying at 4 rows: its all-to-all hops carry B x 1024 tagged values to every CU.  Here:
//
//   * the two all-to-all edges of a step (h_dec -> everyone, h_att -> everyone: B x 1024 values each) are KERNEL BOUNDARIES
//     (3.5 - 4.2 us from the last wave of one kernel to the landed first loads of the next, plain loads afterwards); the light edges (p1, p2, finished count, q, energies: <= B x 256 values) and the
//     context (B x enc values) are tagged 8-byte (step, value) exchanges INSIDE a kernel, as in taco_persist.hip;
//   * every block is 8 waves: 4 LSTM waves (unit u = 4 blk + w) that stream their 4 gate rows through a small register
//     window (PF slices of 256 columns in flight, consumed as they arrive; the slices that wait for the chain are requested
//     first and held), and 4 role waves that run the dependency chain meanwhile.  Separate waves because vector-memory
//     results return in issue order: a poll issued behind weight loads would wait for all of them.  A shallow window because
//     a hop's price sits in the consumer CU's own memory queue: with the whole stream requested up front (first version:
//     8 waves x 20 KiB per CU) even the staging loads only landed when the stream had drained (5.7 / 7.9 us into X / Y).
//
//   X(t):  [role] frame(t-1), stop(t-1), finished / lengths   (projection rows; off the critical path)
//          [role] p1 = relu(F [h_dec | ctx] + fb) -> p2 = relu(W1 p1)     (F = W0^T P folded at load, taco_persist.hip)
//          [LSTM] attention LSTM: gates = Wa [p2 | ctx(t-1) | h_att(t-1)]; only the p2 slice waits for the chain
//   Y(t):  [role] q = Wq h_att(t) -> energies (location term precomputed) -> softmax, context (8 columns per wave)
//          [LSTM] decoder LSTM: gates = Wd [h_att(t) | ctx(t) | h_dec(t-1)]; only the ctx slices wait for the chain
//
// Loop control lives on the device (FusedState): X(t) runs if step t - 1 ran; it computes the stop tokens of step t - 1,
// publishes the finished count with p2, and every block decides "all rows have fired" from that one value; Y(t) runs if
// X(t) did.  A chunk = FUSED_CHUNK (64) steps + a tail projection (the frame of the chunk's last step) is one hipGraph.
// Every wait is bounded; a timeout raises flags[0], later kernels return at once and the host re-runs the call on the
// 7-kernel graph.
//
// Round 4: the weight stream pauses around the hops (quiet windows, below); everything a kernel's entry needs --
// loop state, abort flag, staging rows, the role waves' operands -- is requested before the first wait; a hop's time stamp is
// left by the block's LAST publisher (`stamp_last`), so partially filled shapes (batch 5 - 7, fewer than 1 024 (row, position)
// pairs) do not look early; first-look delays per kernel shape (`kDelay`, scripts/fused_sweep.py).
#include "taco_fused.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "xch_util.h"

using namespace ttsgemm;
using namespace ttsxch;

namespace {

constexpr int NBLK = 256;                 // blocks = CUs of an MI355X; 4 units per block
constexpr int NTHR = 512;                 // 4 LSTM waves + 4 role waves
#ifndef TTS_FUSED_PF
#define TTS_FUSED_PF 2
#endif
#ifndef TTS_FUSED_PRE
#define TTS_FUSED_PRE 1
#endif
constexpr int PF = TTS_FUSED_PF;          // K slices (4 x 1 KiB per wave) of the weight stream in flight per LSTM wave
constexpr int PRE_SL = TTS_FUSED_PRE;     // ... of which this many are requested before the operands are staged (barrier #1)
constexpr int PRE = 256, RNN = 1024, ATT = 128, NMEL = 80, LOCK = 31;
constexpr long long SPIN_LIMIT = 1 << 18; // polls (with s_sleep) of one hop before giving up: ~0.1 s
constexpr int ABORT_TIMEOUT = 2;

struct FX {                               // offsets (8-byte entries) of the exchange area; every offset is even
    unsigned p1, p2, ctrl, q, e, ctx, total;
    int TinP;                             // energies row pitch (Tin rounded up to even)
};
__host__ __device__ inline FX fx_layout(int B, int Tin, int enc) {
    FX x;
    unsigned o = 0;
    x.TinP = (Tin + 1) & ~1;
    x.p1 = o;   o += B * PRE;
    x.p2 = o;   o += B * PRE;
    x.ctrl = o; o += 2;
    x.q = o;    o += B * ATT;
    x.e = o;    o += B * x.TinP;
    x.ctx = o;  o += B * enc;
    x.total = o;
    return x;
}

struct FusedArgs {
    int B, Tin, win_len, win_off;
    const void* Wa; const void* Wd;       // packed LSTM rows [4 u + gate][K]  (fp32, or fp16 when HW)
    const float* ba; const float* bd;
    const float* Ff; const float* fb;     // folded prenet-1 [256][1024 + enc], bias [256]
    const float* W1t;                     // prenet-2 [out 256][in 256]
    const float* Pw; const float* Pb;     // projection rows [81][1024 + enc], bias [81]
    const float* Wq;                      // [128][1024]
    const float* wloc;                    // [62][128]
    const float* vw;                      // [128]
    const float* memory; const float* pm; const uint8_t* mask; const int* enc_len; const float* masks;
    u64* xch; int* flags; FusedState* st;
    float* hatt; float* hdec; float* catt; float* cdec; float* ctx; float* wprev; float* wcum; int* mainatt;
    float* dec_out; float* stop_out; float* attn_hist; int* lengths; int* finished;
    long long* trace;                     // debug builds (-DTTS_DEBUG_HOOKS) only: per-phase timestamps, else null
    int delay[5];                         // hops p1, p2, q, energies, context: first look this many 10-ns ticks after the local publish
};

// Phase timestamps for scripts/fused_trace.py: only in a build made with -DTTS_DEBUG_HOOKS (csrc/build.sh never passes it).
// Blocks 0, 80, 200 and 255 record wall_clock64() (100 MHz) at FTR_SLOTS points of kernels X (kind 0) and Y (kind 1) of
// the first FTR_STEPS steps.
#ifdef TTS_DEBUG_HOOKS
constexpr int FTR_STEPS = 128, FTR_SLOTS = 16;
#define FTR(kind, slot)                                                                                                   \
    do {                                                                                                                  \
        if (a.trace && lane == 0 && t < FTR_STEPS) {                                                                      \
            const int tb_ = blk == 0 ? 0 : blk == 80 ? 1 : blk == 200 ? 2 : blk == 255 ? 3 : -1;                          \
            if (tb_ >= 0) a.trace[(((size_t)t * 2 + (kind)) * 4 + tb_) * FTR_SLOTS + (slot)] = (long long)wall_clock64(); \
        }                                                                                                                 \
    } while (0)
#else
#define FTR(kind, slot) do { } while (0)
#endif

// LDS words that waves of a block signal each other through.  An explicit LDS pointer type: through a generic `volatile int*`
// the compiler emits flat_load / flat_store + s_waitcnt vmcnt(0), i.e. every look at such a word also waited for the wave's
// outstanding global stores (a publish costs ~1 us to be acknowledged under the weight stream).
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ int lds_peek(const lds_int* p) { return *(const volatile lds_int*)p; }
__device__ __forceinline__ void lds_poke(lds_int* p, int v) { *(volatile lds_int*)p = v; }

// ---------------------------------------------------------------------------------------------------- wave-level polls
struct WavePoll {
    __amdgpu_buffer_rsrc_t rs;            // the whole exchange area
    int* flags;                           // global: [0] abort code
    lds_int* abort_s;                     // LDS: set by any wave of the block that gave up

    __device__ __forceinline__ bool should_stop(long long spins) const {
        if ((spins & 63) == 0 && lds_peek(abort_s)) return true;
        if ((spins & 1023) == 0 && __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            lds_poke(abort_s, 1);
            return true;
        }
        return false;
    }
    __device__ __forceinline__ void give_up() const {
        __hip_atomic_store(flags, ABORT_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_poke(abort_s, 1);
    }
    // One pair, the same address in every lane (a single 16-byte request per poll), sleeping between polls.
    __device__ __forceinline__ u32x4 wait_pair(unsigned entry, unsigned tag) const {
        long long spins = 0;
        u32x4 w;
        while (true) {
            asm volatile("" ::: "memory");
            w = __builtin_amdgcn_raw_buffer_load_b128(rs, entry * 8u, 0, 16);       // aux 16 = sc1 (agent scope)
            if (w[1] == tag && w[3] == tag) break;
            ++spins;
            if (spins > SPIN_LIMIT) { give_up(); break; }
            if (should_stop(spins)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        return w;
    }
    // N pairs per lane straight into registers; `sentinel` (a pair published late) is awaited first with one request per poll
    // instead of 64 lanes x N, so that waves that arrive early do not hammer the exchange area.
    template <int N>
    __device__ __forceinline__ void pairs_to_regs(const unsigned (&entry)[N], unsigned tag, f32x2 (&out)[N], unsigned sentinel) const {
        wait_pair(sentinel, tag);
        u32x4 v[N];
        long long spins = 0;
        while (true) {
            asm volatile("" ::: "memory");
            bool ok = true;
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, entry[i] * 8u, 0, 16);
#pragma unroll
            for (int i = 0; i < N; ++i) ok = ok && v[i][1] == tag && v[i][3] == tag;
            if (__all(ok)) break;
            ++spins;
            if (spins > SPIN_LIMIT) { give_up(); break; }
            if (should_stop(spins)) break;
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) out[i] = f32x2{bitsf(v[i][0]), bitsf(v[i][2])};
    }
};

// LDS-DMA staging: one wave instruction moves 1 KiB (lane l: 16 bytes from `src` + 16 l to `dst` + 16 l) without touching
// VGPRs.  `dst` must be wave-uniform.  The issuing wave has to drain vmcnt before the barrier that publishes the bytes.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000u, 0x00020000);
}
__device__ __forceinline__ void dma_1k(const __amdgpu_buffer_rsrc_t& rs, float* dst, unsigned byte_off, int lane) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, byte_off + 16u * (unsigned)lane, 0, 0, 0);
}

// Lane-halving reduction of V (16 or 32) per-lane partial sums: in-row DPP steps for lane bits 0..3, bits 4 / 5 through
// ds_bpermute.  Afterwards every lane holds the wave total of value index bitrev_LOGV(lane & (V - 1)).
template <int V>
__device__ __forceinline__ float reduce_v(float (&acc)[V], int lane) {
    static_assert(V == 16 || V == 32, "V");
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
    halve(std::integral_constant<int, 2>{}, V / 8);
    halve(std::integral_constant<int, 3>{}, V / 16);
    float v;
    if constexpr (V == 32) {
        const bool hi = (lane >> 4) & 1;
        float a_lo = acc[0], a_hi = acc[1];
        asm volatile("" : "+v"(a_lo), "+v"(a_hi));
        const float send = hi ? a_lo : a_hi;
        const float keep = hi ? a_hi : a_lo;
        v = keep + __shfl_xor(send, 16, 64);
    } else {
        v = acc[0];
        v += __shfl_xor(v, 16, 64);
    }
    v += __shfl_xor(v, 32, 64);
    return v;
}
template <int V>
__device__ __forceinline__ int reduced_lane(int idx) {
    constexpr int LOGV = V == 16 ? 4 : 5;
    return (int)(__brev((unsigned)idx) >> (32 - LOGV));
}

template <bool HW>
struct WT {
    typedef typename std::conditional<HW, f16x4, f32x4>::type vec;
    typedef typename std::conditional<HW, _Float16, float>::type el;
};
template <bool HW, bool NT>
__device__ __forceinline__ typename WT<HW>::vec load_w(const void* base, long long elem) {
    const typename WT<HW>::vec* p = reinterpret_cast<const typename WT<HW>::vec*>((const typename WT<HW>::el*)base + elem);
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// acc[g * NBT + b] += W[g] . x[b][lane * 4 ..]   for one 256-wide K slice; x rows are `ldx` floats apart in LDS
template <int NBT, class WV>
__device__ __forceinline__ void fma_slice(float (&acc)[4 * NBT], const WV& w0, const WV& w1, const WV& w2, const WV& w3,
                                          const float* x, int ldx, int lane) {
    const f32x4 wf[4] = {f32x4{(float)w0[0], (float)w0[1], (float)w0[2], (float)w0[3]},
                         f32x4{(float)w1[0], (float)w1[1], (float)w1[2], (float)w1[3]},
                         f32x4{(float)w2[0], (float)w2[1], (float)w2[2], (float)w2[3]},
                         f32x4{(float)w3[0], (float)w3[1], (float)w3[2], (float)w3[3]}};
#pragma unroll
    for (int b = 0; b < NBT; ++b) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + b * ldx + lane * 4);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float a = acc[g * NBT + b];
            a = fmaf(xv[0], wf[g][0], a);
            a = fmaf(xv[1], wf[g][1], a);
            a = fmaf(xv[2], wf[g][2], a);
            a = fmaf(xv[3], wf[g][3], a);
            acc[g * NBT + b] = a;
        }
    }
    // pin the sums here: otherwise the scheduler sinks the whole slice below the caller's next barrier (only the LDS reads
    // have to stay above it) and parks the operands in scratch -- the arithmetic must overlap the wait, not follow it
#pragma unroll
    for (int i = 0; i < 4 * NBT; ++i) asm volatile("" : "+v"(acc[i]));
}

// role wave: s[b] = row . x[b][cols]   (NI slices of 256 columns; slice i of the row multiplies x columns col[i] ..)
template <int NBT, int NI>
__device__ __forceinline__ void role_dots(float (&s)[NBT], const f32x4 (&R)[NI], const float* xs, int ldx, const int (&col)[NI], int lane) {
#pragma unroll
    for (int b = 0; b < NBT; ++b) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + b * ldx + col[i] + lane * 4);
            acc = fmaf(xv[0], R[i][0], acc);
            acc = fmaf(xv[1], R[i][1], acc);
            acc = fmaf(xv[2], R[i][2], acc);
            acc = fmaf(xv[3], R[i][3], acc);
        }
        s[b] = wave_sum(acc);
    }
}
template <int NBT>
__device__ __forceinline__ float pick_row(const float (&s)[NBT], int lane) {
    float v = 0.f;
#pragma unroll
    for (int b = 0; b < NBT; ++b) v = lane == b ? s[b] : v;
    return v;
}

// LSTM tail shared by both kernels: gates and cell update from the reduced sums.
template <int NBT>
__device__ __forceinline__ void lstm_finish(float v, int lane, int B, const f32x4& bias4, float c_old, float* c_state, float* h_new, int u) {
    constexpr int V = 4 * NBT;
    const int wb = lane & (NBT - 1);
    const float gi = __shfl(v, reduced_lane<V>(0 * NBT + wb), 64);
    const float gf = __shfl(v, reduced_lane<V>(1 * NBT + wb), 64);
    const float gg = __shfl(v, reduced_lane<V>(2 * NBT + wb), 64);
    const float go = __shfl(v, reduced_lane<V>(3 * NBT + wb), 64);
    if (lane < B) {
        const float ig = sigmoid_fast(gi + bias4[0]), fg = sigmoid_fast(gf + bias4[1]);
        const float cg = tanh_fast(gg + bias4[2]), og = sigmoid_fast(go + bias4[3]);
        const float cn = fg * c_old + ig * cg;
        c_state[(size_t)lane * RNN + u] = cn;
        h_new[(size_t)lane * RNN + u] = og * tanh_fast(cn);
    }
}

// One K slice (256 columns) of a unit's four gate rows: lane l holds columns 4 l .. 4 l + 3 of each row.
template <bool HW>
struct WSlice {
    typename WT<HW>::vec g[4];
};
template <bool HW, bool NT>
__device__ __forceinline__ void load_slice(WSlice<HW>& w, const void* base, long long row0, int K, int koff, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) w.g[g] = load_w<HW, NT>(base, (row0 + g) * K + koff + lane * 4);
}

// Every block stages the same rows at the same moment: starting each block at its own offset keeps the 32 CUs of an XCD from
// walking the L2 channels in lock step.  (TTS_FUSED_ROT=0: all blocks in the same order.)
#ifndef TTS_FUSED_ROT
#define TTS_FUSED_ROT 0
#endif
__device__ __forceinline__ int stage_index(int idx, int blk, int total) {
#if TTS_FUSED_ROT
    const int r = idx + ((blk >> 3) * 160) % total;      // blocks b, b + 8, ... share an XCD: 32 different offsets, 2.5 KiB apart
    return r >= total ? r - total : r;
#else
    return idx;
#endif
}

// ---------------------------------------------------------------------------------------------------- timed polls
// The 256 blocks run in lock step (entry within ~0.3 us), and every block holds a producer of every hop.  So a consumer
// does not watch the exchange area while it waits: it spins on an LDS word in which the block's own producer leaves the
// (100 MHz) time of its publish, sleeps until that time + the hop's latency, and only then loads its pairs -- usually once.
// Tags decide, time only chooses when to look.  Hops: 0 p1, 1 p2, 2 q, 3 energies, 4 context.
constexpr long long LDS_SPIN_LIMIT = 1 << 22;
__device__ __forceinline__ void stamp(lds_int* ts, int later = 0) {
    lds_poke(ts, (int)(((unsigned)wall_clock64() + (unsigned)later) | 1u));
}
// One stamp per hop and block, left by whichever of its `n` producing waves publishes LAST (they count themselves on an LDS word).
__device__ __forceinline__ void stamp_last(lds_int* count, int n, lds_int* ts) {
    if (__hip_atomic_fetch_add(count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == n - 1) stamp(ts);
}
__device__ __forceinline__ void wait_stamp(const WavePoll& P, const lds_int* ts, int delay) {
    long long spins = 0;
    int v;
    while ((v = lds_peek(ts)) == 0) {
        if (++spins > LDS_SPIN_LIMIT) { P.give_up(); return; }
        if ((spins & 255) == 0 && lds_peek(P.abort_s)) return;
        __builtin_amdgcn_s_sleep(1);
    }
    while ((int)((unsigned)wall_clock64() - (unsigned)(v + delay)) < 0) __builtin_amdgcn_s_sleep(1);
}
#ifndef TTS_POLL_SLEEP
#define TTS_POLL_SLEEP 4              // 64-cycle units between two looks of a poll that missed
#endif
template <int N>
__device__ __forceinline__ void poll_pairs(const WavePoll& P, const unsigned (&entry)[N], unsigned tag, f32x2 (&out)[N]) {
    u32x4 v[N];
    long long spins = 0;
    while (true) {
        asm volatile("" ::: "memory");
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(P.rs, entry[i] * 8u, 0, 16);      // aux 16 = sc1
#pragma unroll
        for (int i = 0; i < N; ++i) ok = ok && v[i][1] == tag && v[i][3] == tag;
        if (__all(ok)) break;
        ++spins;
        if (spins > SPIN_LIMIT) { P.give_up(); break; }
        if (P.should_stop(spins)) break;
        __builtin_amdgcn_s_sleep(TTS_POLL_SLEEP);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = f32x2{bitsf(v[i][0]), bitsf(v[i][2])};
}

// Loop state and abort flag of a kernel's entry, requested TOGETHER and only then looked at.  (Read as `*a.st` followed by the
// flag behind the first early return, hipcc fetched the fields in two rounds and the flag in a third: three dependent memory
// latencies between dispatch and the first useful instruction of every wave.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct EntryLoads { i32x4 lo, hi; int flag; };
struct EntryState { FusedState s; int flag; };
// issue: right behind the first staging loads, in front of every other request of the entry (results return in issue order)
__device__ __forceinline__ EntryLoads request_entry(const FusedState* st, const int* flags) {
    EntryLoads l;
    l.lo = *reinterpret_cast<const i32x4*>(st);
    l.hi = *(reinterpret_cast<const i32x4*>(st) + 1);
    l.flag = __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return l;
}
// look: all three at once
__device__ __forceinline__ EntryState look_entry(EntryLoads l) {
    asm volatile("" : "+v"(l.lo), "+v"(l.hi), "+v"(l.flag));
    EntryState e;
    e.s.t0 = l.lo[0]; e.s.n_fin = l.lo[1]; e.s.steps_run = l.lo[2]; e.s.exec_t = l.lo[3];
    e.s.B = l.hi[0]; e.s.max_len = l.hi[1]; e.s.early_stop = l.hi[2]; e.s.pad = l.hi[3];
    e.flag = l.flag;
    return e;
}

// Quiet windows (round 4): a hop's price sits in the consumer's memory queue and on a fabric that the weight stream
// keeps busy.  A role wave raises an LDS counter shortly before its timed first look at a hop (TTS_FUSED_QUIET_LEAD ticks before)
// and lowers it when its poll has succeeded; the LSTM waves do not request new weight slices while the counter is up (the 256
// blocks run in lock step, so the whole chip's stream pauses around every hop).  Measured per hop: the hops of kernel Y and the
// p1 hop of kernel X pay, the p2 hop does not (bits of TTS_FUSED_QUIET).  Bounded wait: the stream resumes on its own after
// 2 048 looks (~0.1 ms); the tags decide correctness, the counter only delays requests.
#ifndef TTS_FUSED_QUIET
#define TTS_FUSED_QUIET 3          // bit 0: the p1 hop of kernel X, bit 2: its p2 hop, bit 1: the hops of kernel Y
#endif
#ifndef TTS_FUSED_QUIET_LEAD
#define TTS_FUSED_QUIET_LEAD 30    // > 0: the window opens this many 10-ns ticks before the first look instead of at the publish
#endif
template <int KERNEL>
__device__ __forceinline__ void quiet_begin(lds_int* ctl, int lane) {
    if constexpr ((TTS_FUSED_QUIET >> KERNEL) & 1)
        if (lane == 0) __hip_atomic_fetch_add(ctl + 7, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int KERNEL>
__device__ __forceinline__ void quiet_end(lds_int* ctl, int lane) {
    if constexpr ((TTS_FUSED_QUIET >> KERNEL) & 1)
        if (lane == 0) __hip_atomic_fetch_add(ctl + 7, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the delay a role wave sleeps through BEFORE it opens its quiet window
__device__ __forceinline__ int quiet_pre(int delay) { return TTS_FUSED_QUIET_LEAD > 0 ? max(0, delay - TTS_FUSED_QUIET_LEAD) : 0; }
__device__ __forceinline__ void stream_gate(const lds_int* ctl) {
#if TTS_FUSED_QUIET
    int spins = 0;
    while (lds_peek(ctl + 7) > 0 && ++spins < 2048) __builtin_amdgcn_s_sleep(2);
#endif
}

// ================================================================================================== kernel X
// tail != 0: only the frame / stop-token part (end of a chunk).
template <int NBT, int ENC, bool HW>
__global__ __launch_bounds__(NTHR) void fused_x_kernel(const FusedArgs a, const int j, const int tail) {
    constexpr int NC = ENC / 256;                     // context slices
    constexpr int KX = 2 * RNN + ENC;                 // LDS row: [h_att | ctx | h_dec]
    constexpr int KA = PRE + ENC + RNN;               // attention-LSTM row: [p2 | ctx | h_att]
    constexpr int KP = RNN + ENC, NP = KP / 256;      // projection / folded prenet rows: [h_dec | ctx]
    constexpr int V = 4 * NBT, HB = NBT / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                                  // [NBT][KX]
    float* p2s = xs + NBT * KX;                       // [NBT][256]
    lds_int* ctl = (lds_int*)(p2s + NBT * PRE);       // [0] abort, [1] finished count, [2..4] publish times: p1 (two waves), p2

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), blk = blockIdx.x;
    const int B = a.B;
    const int par = j & 1;                            // = t & 1 (chunks are even): h(t - 1) lives in buffer t & 1, h(t) goes to the other one
    // Everything requested before the loop state is looked at: the state is the third load of a dependent chain (kernel
    // arguments -> state -> ...), and waiting for it first put ~1 us of latency in front of the staging loads.  A kernel that
    // turns out to have nothing to do returns with these loads outstanding.
    // ---- staging in two stages.  Every thread: its share of [ctx(t-1) | h_dec(t-1)] of rows < B -> LDS (zeros beyond B) in
    // front of barrier #1: that is what the chain (prenet 1, projection) and the first weight slices need.  h_att(t-1) is only
    // read by the LSTM waves' later slices: they fetch it themselves and meet on an LDS counter (ctl[5]), off the chain's path.
    constexpr int RW = (ENC + RNN) / 4;               // float4 per row of the first stage
    constexpr int NS1 = (NBT * RW + NTHR - 1) / NTHR;
    constexpr int NS2 = NBT * (RNN / 4) / 256;        // h_att float4 per LSTM thread
    f32x4 sv[NS1];
    {
        const float* hd = a.hdec + (size_t)par * B * RNN;
#pragma unroll
        for (int i = 0; i < NS1; ++i) {
            const int idx = tid + i * NTHR;
            const int b = idx / RW, k = (idx - b * RW) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < NBT * RW && b < B) v = *reinterpret_cast<const f32x4*>(k < ENC ? a.ctx + (size_t)b * ENC + k : hd + (size_t)b * RNN + (k - ENC));
            sv[i] = v;
        }
    }
    const EntryLoads el = request_entry(a.st, a.flags);
    constexpr int NC_ = ENC / 256;
    auto ekoff = [](int e) { return e < NC_ ? PRE + e * 256 : PRE + ENC + (e - NC_) * 256; };      // column inside the weight row
    f32x4 hv[NS2];                                    // LSTM waves: this thread's share of h_att(t-1)
    WSlice<HW> win[PF];
    // role waves: the operands of their roles (rows of the folded prenet-1 / projection / prenet-2 matrices) do not depend on
    // the loop state either; requested here they have landed when barrier #1 opens (requested behind the state they arrived
    // 0.4 us after it)
    constexpr int KP_ = RNN + ENC, NP_ = KP_ / 256;
    f32x4 R[NP_];
    float rbias = 0.f;
    int fin_old = 0;
    if (wave >= 4) {
        const int r_ = wave - 4;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const bool p1w = r_ == 0 || r_ == 3, projw = r_ == 2 && blk <= NMEL;
#pragma unroll
        for (int i = 0; i < NP_; ++i) {
            const float* src = p1w ? a.Ff + (size_t)blk * KP_ + i * 256 + lane * 4
                             : projw ? a.Pw + (size_t)blk * KP_ + i * 256 + lane * 4
                             : (r_ == 1 && i == 0) ? a.W1t + (size_t)blk * PRE + lane * 4 : nullptr;
            R[i] = src ? *reinterpret_cast<const f32x4*>(src) : zero;
        }
        rbias = p1w ? a.fb[blk] : projw ? a.Pb[blk] : 0.f;
        if (projw && blk == NMEL && lane < B) fin_old = a.finished[lane];
    }
    if (wave < 4) {
        const float* ha = a.hatt + (size_t)par * B * RNN;
#pragma unroll
        for (int i = 0; i < NS2; ++i) {
            const int idx = tid + i * 256;
            const int b = idx / (RNN / 4), k = (idx % (RNN / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b < B) v = *reinterpret_cast<const f32x4*>(ha + (size_t)b * RNN + k);
            hv[i] = v;
        }
#pragma unroll
        for (int e = 0; e < PRE_SL; ++e) load_slice<HW, !HW>(win[e], a.Wa, 4ll * (blk * 4 + wave), KA, ekoff(e), lane);
    }
    asm volatile("" ::: "memory");
    const EntryState es = look_entry(el);
    const FusedState& s = es.s;
    const int t = s.t0 + j;
    if (s.steps_run != t) return;                     // the loop ended before this step
    if (es.flag != 0) return;                         // an earlier kernel gave up
    const bool frame_on = t >= 1 && (j >= 1 || tail != 0);       // frame / stop token of step t - 1 (j == 0: the previous chunk's tail did it)
    const bool step_on = tail == 0 && t < s.max_len;
    if (!frame_on && !step_on) return;
    const int max_len = s.max_len;
    const unsigned tag = (unsigned)t + 1;
    const FX X = fx_layout(B, a.Tin, ENC);
    if (wave == 0) FTR(0, 8);
    if (wave == 4) FTR(0, 0);


    auto store_staged = [&]() {
#pragma unroll
        for (int i = 0; i < NS1; ++i) {
            const int idx = tid + i * NTHR;
            const int b = idx / RW, k = (idx - b * RW) * 4;
            if (idx < NBT * RW) *reinterpret_cast<f32x4*>(xs + (size_t)b * KX + RNN + k) = sv[i];
        }
        for (int i = tid; i < NBT * PRE; i += NTHR) p2s[i] = 0.f;
        if (tid < 8) ctl[tid] = tid == 1 ? s.n_fin : 0;
    };

    if (wave < 4) {
        // ------------------------------------------------------------------------------------------ LSTM waves
        // early slices (operands known at kernel start): ctx(t-1) [NC], h_att(t-1) [4]; late slice: p2 (waits for the chain)
        constexpr int NE = NC + 4;
        const int u = blk * 4 + wave;
        if (!step_on) {
            store_staged();
            __syncthreads();                          // #1
            __syncthreads();                          // #2
            return;
        }
        const long long row0 = 4ll * u;
        auto excol = [](int e) { return e < NC ? RNN + e * 256 : (e - NC) * 256; };                 // column inside the LDS row
        WSlice<HW> wl;
        const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.ba + 4 * u);
        const float c_old = lane < B ? a.catt[(size_t)lane * RNN + u] : 0.f;
        asm volatile("" ::: "memory");
        store_staged();
        __syncthreads();                              // #1: ctx, h_dec staged
        if (wave == 0) FTR(0, 9);
#pragma unroll
        for (int i = 0; i < NS2; ++i) {
            const int idx = tid + i * 256;
            *reinterpret_cast<f32x4*>(xs + (size_t)(idx / (RNN / 4)) * KX + (idx % (RNN / 4)) * 4) = hv[i];
        }
        if (lane == 0) __hip_atomic_fetch_add(ctl + 5, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int e = PRE_SL; e < PF; ++e) load_slice<HW, !HW>(win[e], a.Wa, row0, KA, ekoff(e), lane);
        asm volatile("" ::: "memory");
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            WSlice<HW>& w = win[e % PF];
            if (e == NC) {                            // first h_att slice: every LSTM wave has stored its share
                long long spins = 0;
                while (lds_peek(ctl + 5) < 4 && ++spins < LDS_SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
            }
            fma_slice<NBT>(acc, w.g[0], w.g[1], w.g[2], w.g[3], xs + excol(e), KX, lane);
            asm volatile("" ::: "memory");            // the refill is requested here, not hoisted to the top
            stream_gate(ctl);
            if (e + PF < NE) load_slice<HW, !HW>(w, a.Wa, row0, KA, ekoff(e + PF), lane);
            else if (e + PF == NE) load_slice<HW, !HW>(wl, a.Wa, row0, KA, 0, lane);       // the p2 slice last: it waits for the chain
            asm volatile("" ::: "memory");
            if (e == 0 && wave == 0) FTR(0, 13);
        }
        if (wave == 0) FTR(0, 10);
        __syncthreads();                              // #2: p2 and the finished count are in LDS
        if (wave == 0) FTR(0, 11);
        if (ctl[0] != 0) return;
        if (s.early_stop && ctl[1] >= B) return;      // every row has fired: the loop ends here, nothing is modified
        fma_slice<NBT>(acc, wl.g[0], wl.g[1], wl.g[2], wl.g[3], p2s, PRE, lane);
        const float v = reduce_v<V>(acc, lane);
        if (wave == 0) FTR(0, 12);
        lstm_finish<NBT>(v, lane, B, bias4, c_old, a.catt, a.hatt + (size_t)(par ^ 1) * B * RNN, u);
        if (blk == 0 && tid == 0) a.st->exec_t = t + 1;
        if (wave == 0) FTR(0, 14);
        return;
    }
    // ---------------------------------------------------------------------------------------------- role waves
    // r = 0 / 3: prenet-1 output blk of rows [0, HB) / [HB, NBT); r = 1: prenet-2 output blk; r = 2: projection row blk
    const int r = wave - 4;
    __builtin_amdgcn_s_setprio(3);                    // the chain is the critical path; the LSTM waves fill the gaps
    WavePoll P;
    P.rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, 0x80000000u, 0x00020000);
    P.flags = a.flags;
    P.abort_s = ctl;
    const bool is_p1 = (r == 0 || r == 3) && step_on, is_p2 = r == 1 && step_on, is_proj = r == 2 && frame_on && blk <= NMEL;
    const int row_lo = r == 3 ? HB : 0;               // first row of a prenet-1 wave
    float dmask = 1.f;                                // prenet dropout mask of (row, output blk); row = row_lo + lane for prenet 1
    {
        const int mrow = is_p1 ? row_lo + lane : lane;
        if (a.masks && mrow < B && t < max_len && (is_p1 || is_p2))
            dmask = a.masks[((size_t)mrow * max_len + t) * 2 * PRE + (is_p2 ? PRE : 0) + blk];
    }
    store_staged();
    if (r == 0) FTR(0, 1);
    __syncthreads();                                  // #1
    if (r == 0) FTR(0, 2);
    int pcol[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) pcol[i] = i < 4 ? RNN + ENC + i * 256 : RNN + (i - 4) * 256;      // [h_dec | ctx] inside the LDS row
    if (is_proj) {                                    // frame / stop token of step t - 1
        float sm[NBT];
        role_dots<NBT, NP>(sm, R, xs, KX, pcol, lane);
        int fin = fin_old, fired = 0;
        if (lane < B) {
            const float v = pick_row<NBT>(sm, lane) + rbias;
            if (blk < NMEL) {
                a.dec_out[((size_t)lane * max_len + (t - 1)) * NMEL + blk] = v;
            } else {
                const float sp = sigmoid_exact(v);
                a.stop_out[(size_t)lane * max_len + (t - 1)] = sp;
                // finished |= stop > 0.5 ; lengths += !finished      (tacotron2_arch.py:664-665)
                if (!fin && sp > 0.5f) {
                    fin = 1;
                    fired = 1;
                    a.finished[lane] = 1;
                }
                if (!fin) a.lengths[lane] += 1;
            }
        }
        if (blk == NMEL) {                            // the gate owner tells everybody how many rows have finished
            const int nf = s.n_fin + __popcll(__ballot(fired != 0));
            if (lane == 0) a.st->n_fin = nf;
            if (tail == 0 && lane < 2) publish(a.xch + X.ctrl + lane, tag, bitsf((unsigned)nf));
        }
    }
    if (is_p1) {                                      // prenet layer 1 folded with the projection; go frame at t = 0
        float sm[HB];
        role_dots<HB, NP>(sm, R, xs + (size_t)row_lo * KX, KX, pcol, lane);
        if (r == 0) FTR(0, 15);
        if (lane < HB && row_lo + lane < B) {
            float v = pick_row<HB>(sm, lane) + rbias;
            v = t == 0 ? 0.f : fmaxf(v, 0.f) * dmask;
            publish(a.xch + X.p1 + (row_lo + lane) * PRE + blk, tag, v);
        }
        if (lane == 0) stamp(ctl + (r == 0 ? 2 : 3));
        if (r == 0) FTR(0, 3);
    }
    if (is_p2) {                                      // prenet layer 2: p1 of every row -> output blk
        unsigned ent[2 * NBT];
#pragma unroll
        for (int b = 0; b < NBT; ++b) {
            const int bb = b < B ? b : 0;             // rows beyond B re-read row 0 (not used)
            ent[2 * b] = X.p1 + bb * PRE + lane * 4;
            ent[2 * b + 1] = ent[2 * b] + 2;
        }
        wait_stamp(P, ctl + 2, quiet_pre(a.delay[0]));
        wait_stamp(P, ctl + 3, quiet_pre(a.delay[0]));
        quiet_begin<0>(ctl, lane);
        wait_stamp(P, ctl + 2, a.delay[0]);
        wait_stamp(P, ctl + 3, a.delay[0]);
        f32x2 pv[2 * NBT];
        poll_pairs<2 * NBT>(P, ent, tag, pv);
        quiet_end<0>(ctl, lane);
        float sm[NBT];
#pragma unroll
        for (int b = 0; b < NBT; ++b) {
            const float acc = pv[2 * b][0] * R[0][0] + pv[2 * b][1] * R[0][1] + pv[2 * b + 1][0] * R[0][2] + pv[2 * b + 1][1] * R[0][3];
            sm[b] = wave_sum(b < B ? acc : 0.f);
        }
        FTR(0, 4);
        if (lane < B) publish(a.xch + X.p2 + lane * PRE + blk, tag, fmaxf(pick_row<NBT>(sm, lane), 0.f) * dmask);
        if (lane == 0) stamp(ctl + 4);
        FTR(0, 5);
    }
    if (step_on) {                                    // every role wave fetches a quarter of p2 into LDS
        constexpr int NQ = NBT * PRE / 2 / 256;       // pairs per lane
        unsigned ent[NQ];
        int pair[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            pair[i] = r * (NQ * 64) + i * 64 + lane;
            ent[i] = X.p2 + 2u * (unsigned)(pair[i] < B * PRE / 2 ? pair[i] : 0);
        }
        wait_stamp(P, ctl + 4, quiet_pre(a.delay[1]));
        quiet_begin<2>(ctl, lane);
        wait_stamp(P, ctl + 4, a.delay[1]);
        f32x2 pv[NQ];
        poll_pairs<NQ>(P, ent, tag, pv);
        quiet_end<2>(ctl, lane);
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (pair[i] < B * PRE / 2) *reinterpret_cast<f32x2*>(p2s + 2 * pair[i]) = pv[i];
        if (r == 3 && j >= 1 && t >= 1) {             // finished count after the stop tokens of step t - 1 (published long ago)
            const u32x4 w = P.wait_pair(X.ctrl, tag);
            if (lane == 0) ctl[1] = (int)w[0];
        }
        if (r == 0) FTR(0, 6);
    }
    __syncthreads();                                  // #2
    if (r == 0) FTR(0, 7);
}

// ================================================================================================== kernel Y
// KT = ceil(Tin / 128): positions of a row per lane pair-load.
template <int NBT, int ENC, int KT, bool HW>
__global__ __launch_bounds__(NTHR) void fused_y_kernel(const FusedArgs a, const int j) {
    constexpr int NC = ENC / 256;
    constexpr int KX = 2 * RNN + ENC;                 // LDS row = decoder-LSTM row: [h_att | ctx | h_dec]
    constexpr int V = 4 * NBT, HB = NBT / 2;
    constexpr int TP = KT * 128;                      // padded positions per row
    constexpr int NPOS = (NBT * TP + 1023) / 1024;    // (row, position) pairs a role wave may own
    constexpr int MPF = 16 * KT;                      // positions per lane of a context unit (8 time slices x 8 columns per wave)
    constexpr int CU_PER_ROW = ENC / 8;
    constexpr int NL4 = 2 * LOCK * ATT / 4, NSL = (NL4 + 255) / 256;             // location map: float4 per role thread
    constexpr int NSL3 = (NL4 + 191) / 192;                                       // ... per thread of three role waves (two_pairs)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                                  // [NBT][KX]
    float* wl = xs + NBT * KX;                        // [62][128]
    float* wsm = wl + 2 * LOCK * ATT;                 // [4 role waves][TP] softmax weights
    float* msl = wsm + 4 * TP;                        // [4 role waves][TP][8] encoder outputs: 8 columns of one row
    lds_int* ctl = (lds_int*)(msl + 4 * TP * 8);      // [0] abort, [2..4] publish times: q, energies, context; [5..7] counters of
                                                      // the staging / quiet windows, [8] [9] energies / context publishers done

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), blk = blockIdx.x;
    const int B = a.B, Tin = a.Tin;
    const int par = j & 1;                            // = t & 1: h_att(t) is in buffer par ^ 1, h_dec(t - 1) in buffer par
    // everything requested before the loop state is looked at (see kernel X)
    // ---- staging in two stages.  Every thread: its share of h_att(t) (rows < B) -> LDS in front of barrier #1 (the query and
    // the first weight slices need it).  h_dec(t-1) is only read by the LSTM waves' later slices: they fetch it themselves and
    // meet on an LDS counter (ctl[6]), off the chain's path.
    constexpr int NS1 = NBT * (RNN / 4) / NTHR, NS2 = NBT * (RNN / 4) / 256;
    f32x4 sv[NS1];
    {
        const float* ha = a.hatt + (size_t)(par ^ 1) * B * RNN;
#pragma unroll
        for (int i = 0; i < NS1; ++i) {
            const int idx = tid + i * NTHR;           // float4 index in [NBT][1024]
            const int b = idx / (RNN / 4), k = (idx % (RNN / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b < B) v = *reinterpret_cast<const f32x4*>(ha + (size_t)b * RNN + k);
            sv[i] = v;
        }
    }
    const EntryLoads el = request_entry(a.st, a.flags);
    auto ecol = [](int e) { return e < 4 ? e * 256 : RNN + ENC + (e - 4) * 256; };
    f32x4 hv[NS2];                                    // LSTM waves: this thread's share of h_dec(t-1)
    WSlice<HW> win[PF];
    f32x4 RQ[4];                                      // query wave (role wave 3) only: its row of the query matrix (see kernel X)
    if (wave == 7) {                                  // (no zeros for the other waves: the merge cost the query wave a full wait)
#pragma unroll
        for (int i = 0; i < 4; ++i) RQ[i] = *reinterpret_cast<const f32x4*>(a.Wq + (size_t)(blk & (ATT - 1)) * RNN + i * 256 + lane * 4);
    }
    if (wave < 4) {
        const float* hd = a.hdec + (size_t)par * B * RNN;
#pragma unroll
        for (int i = 0; i < NS2; ++i) {
            const int idx = tid + i * 256;
            const int b = idx / (RNN / 4), k = (idx % (RNN / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b < B) v = *reinterpret_cast<const f32x4*>(hd + (size_t)b * RNN + k);
            hv[i] = v;
        }
#pragma unroll
        for (int e = 0; e < PRE_SL; ++e) load_slice<HW, true>(win[e], a.Wd, 4ll * (blk * 4 + wave), KX, ecol(e), lane);
    }
    // --- operands of the role waves' roles: none of them depends on the loop state, so they are requested in front of it too
    // (round 4: behind it they cost barrier #1 one more memory latency; the token mask, looked at on the spot, two more)
    const int r = wave - 4;
    const bool role = wave >= 4;
    const int g_id = r * NBLK + blk;                  // 0 .. 1023 (role waves)
    // query: r == 3 of block blk computes attention dim blk & 127 of rows [0, HB) (blk < 128) or [HB, NBT) (blk >= 128)
    const bool is_q = r == 3;
    const int qdim = blk & (ATT - 1), qrow = blk < ATT ? 0 : HB;
    f32x2 vv = {0.f, 0.f};
    if (role) vv = *reinterpret_cast<const f32x2*>(a.vw + lane * 2);
    f32x2 pmv[NPOS];
    float cp[NPOS], cc[NPOS];                         // alignment windows: lane i < 31 holds position tau + i - 15
#pragma unroll
    for (int p = 0; p < NPOS; ++p) {
        const int idx = g_id + 1024 * p;
        const f32x2 zero2 = {0.f, 0.f};
        pmv[p] = zero2;
        cp[p] = cc[p] = 0.f;
        if (role && idx < B * Tin) {
            const int b = idx / Tin, tau = idx - b * Tin;
            pmv[p] = *reinterpret_cast<const f32x2*>(a.pm + (size_t)idx * ATT + lane * 2);
            const int tw = tau + lane - LOCK / 2;
            if (lane < LOCK && tw >= 0 && tw < Tin) {
                cp[p] = a.wprev[(size_t)b * Tin + tw];
                cc[p] = a.wcum[(size_t)b * Tin + tw];
            }
        }
    }
    const bool has_ctx = role && g_id < B * CU_PER_ROW;
    const int n_c = min(4, max(0, (B * CU_PER_ROW - blk + NBLK - 1) / NBLK));      // role waves of this block that own a context unit
    // some role wave owns two (row, position) pairs (8 rows of more than 128 tokens): the batched energies path below.  With fp16
    // LSTM weights only: there it takes 27.4 -> 26.1 us per step at 256 tokens; with fp32 weights (a longer stream under the
    // hops) it measured +0.3 us, and shapes without a second pair lose 0.3 - 1.0 us to the three-wave staging of the map.
    const bool two_pairs = NPOS > 1 && HW && B * Tin > 4 * NBLK;
    const int cb = has_ctx ? g_id / CU_PER_ROW : 0, c8 = has_ctx ? g_id % CU_PER_ROW : 0;
    unsigned mraw[2 * KT];                            // token mask bytes of positions 128 k + 2 lane (+ 1): requested here, looked at behind barrier #1
    float wc_old[2 * KT];
    int elen = 0, matt_old = 0;
#pragma unroll
    for (int i = 0; i < 2 * KT; ++i) { wc_old[i] = 0.f; mraw[i] = 0; }
    if (has_ctx) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tau = 128 * k + 2 * lane + h;
                if (tau < Tin) {
                    mraw[2 * k + h] = a.mask[(size_t)cb * Tin + tau];
                    if (c8 == 0) wc_old[2 * k + h] = a.wcum[(size_t)cb * Tin + tau];
                }
            }
        elen = a.enc_len[cb];
        matt_old = a.mainatt[par * B + cb];
    }
    asm volatile("" ::: "memory");
    const EntryState es = look_entry(el);
    const FusedState& s = es.s;
    const int t = s.t0 + j;
    if (s.exec_t != t + 1) return;                    // X(t) decided that the loop has ended (or never ran)
    if (es.flag != 0) return;
    const int max_len = s.max_len;
    const unsigned tag = (unsigned)t + 1;
    const FX X = fx_layout(B, Tin, ENC);
    if (wave == 0) FTR(1, 10);
    if (wave == 4) FTR(1, 0);


    auto store_staged = [&]() {
#pragma unroll
        for (int i = 0; i < NS1; ++i) {
            const int idx = tid + i * NTHR;
            *reinterpret_cast<f32x4*>(xs + (size_t)(idx / (RNN / 4)) * KX + (idx % (RNN / 4)) * 4) = sv[i];
        }
        for (int i = tid; i < (NBT - B) * ENC; i += NTHR) xs[(size_t)(B + i / ENC) * KX + RNN + i % ENC] = 0.f;
        if (tid < 16) ctl[tid] = 0;
    };

    if (wave < 4) {
        // ------------------------------------------------------------------------------------------ LSTM waves
        // early slices: h_att(t) [4], h_dec(t-1) [4]; late slices: ctx(t) [NC] (wait for the chain); row = LDS row order
        constexpr int NE = 8;
        const int u = blk * 4 + wave;
        const long long row0 = 4ll * u;
        WSlice<HW> wlate[NC];
        const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.bd + 4 * u);
        const float c_old = lane < B ? a.cdec[(size_t)lane * RNN + u] : 0.f;
        asm volatile("" ::: "memory");
        store_staged();
        __syncthreads();                              // #1: h_att(t) staged
        if (wave == 0) FTR(1, 11);
#pragma unroll
        for (int i = 0; i < NS2; ++i) {
            const int idx = tid + i * 256;
            *reinterpret_cast<f32x4*>(xs + (size_t)(idx / (RNN / 4)) * KX + RNN + ENC + (idx % (RNN / 4)) * 4) = hv[i];
        }
        if (lane == 0) __hip_atomic_fetch_add(ctl + 6, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int e = PRE_SL; e < PF; ++e) load_slice<HW, true>(win[e], a.Wd, row0, KX, ecol(e), lane);
        asm volatile("" ::: "memory");
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            WSlice<HW>& w = win[e % PF];
            if (e == 4) {                             // first h_dec slice: every LSTM wave has stored its share
                long long spins = 0;
                while (lds_peek(ctl + 6) < 4 && ++spins < LDS_SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
            }
            fma_slice<NBT>(acc, w.g[0], w.g[1], w.g[2], w.g[3], xs + ecol(e), KX, lane);
            asm volatile("" ::: "memory");            // the refill is requested here, not hoisted to the top
            stream_gate(ctl);
            if (e + PF < NE) load_slice<HW, true>(w, a.Wd, row0, KX, ecol(e + PF), lane);
            else if (e + PF - NE < NC) load_slice<HW, true>(wlate[e + PF - NE], a.Wd, row0, KX, RNN + (e + PF - NE) * 256, lane);   // ctx slices last
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int c = PF; c < NC; ++c) load_slice<HW, true>(wlate[c], a.Wd, row0, KX, RNN + c * 256, lane);      // (only if NC > PF)
        if (wave == 0) FTR(1, 12);
        __syncthreads();                              // #2: ctx(t) is in LDS
        if (wave == 0) FTR(1, 13);
        if (ctl[0] != 0) return;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            fma_slice<NBT>(acc, wlate[c].g[0], wlate[c].g[1], wlate[c].g[2], wlate[c].g[3], xs + RNN + c * 256, KX, lane);
        const float v = reduce_v<V>(acc, lane);
        lstm_finish<NBT>(v, lane, B, bias4, c_old, a.cdec, a.hdec + (size_t)(par ^ 1) * B * RNN, u);
        if (blk == 0 && tid == 0) a.st->steps_run = t + 1;
        if (wave == 0) FTR(1, 14);
        return;
    }
    // ---------------------------------------------------------------------------------------------- role waves
    __builtin_amdgcn_s_setprio(3);
    WavePoll P;
    P.rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xch, 0, 0x80000000u, 0x00020000);
    P.flags = a.flags;
    P.abort_s = ctl;
    store_staged();
    if (r == 0) FTR(1, 1);
    __syncthreads();                                  // #1
    if (r == 0) FTR(1, 2);
    unsigned on_bits = 0;                             // bits 2 k, 2 k + 1
#pragma unroll
    for (int i = 0; i < 2 * KT; ++i) {
        asm volatile("" : "+v"(mraw[i]));
        if (mraw[i]) on_bits |= 1u << i;
    }
    if (two_pairs) {
        // two (row, position) pairs per wave (8 rows of more than 128 tokens): the location terms are on the chain, so the map
        // is staged by the three role waves that do not compute the query (the query wave stores its share only behind its
        // publish, ~0.9 us later) and the terms start as soon as those three have met on the LDS counter
        if (!is_q) {
            const int rt = r * 64 + lane;             // 0 .. 191
            f32x4 lv[NSL3];
#pragma unroll
            for (int i = 0; i < NSL3; ++i) {
                const int idx = rt + i * 192;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                lv[i] = idx < NL4 ? *reinterpret_cast<const f32x4*>(a.wloc + (size_t)idx * 4) : zero;
            }
#pragma unroll
            for (int i = 0; i < NSL3; ++i) {
                const int idx = rt + i * 192;
                if (idx < NL4) *reinterpret_cast<f32x4*>(wl + (size_t)idx * 4) = lv[i];
            }
            if (lane == 0) __hip_atomic_fetch_add(ctl + 5, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            float sm[HB];
            const int qcol[4] = {0, 256, 512, 768};
            role_dots<HB, 4>(sm, RQ, xs + (size_t)qrow * KX, KX, qcol, lane);
            FTR(1, 15);
            if (lane < HB && qrow + lane < B) publish(a.xch + X.q + (qrow + lane) * ATT + qdim, tag, pick_row<HB>(sm, lane));
            if (lane == 0) stamp(ctl + 2);
            FTR(1, 3);
        }
    } else
    {   // the location map (31 KiB, only the energies need it): fetched by the role waves now, while the query is computed
        // and published, instead of sitting in front of barrier #1; the four waves meet on an LDS counter
        const int rt = tid - 256;
        f32x4 lv[NSL];
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int idx = rt + i * 256;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            lv[i] = idx < NL4 ? *reinterpret_cast<const f32x4*>(a.wloc + (size_t)idx * 4) : zero;
        }
        if (!is_q) {
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                const int idx = rt + i * 256;
                if (idx < NL4) *reinterpret_cast<f32x4*>(wl + (size_t)idx * 4) = lv[i];
            }
            if (lane == 0) __hip_atomic_fetch_add(ctl + 5, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            // the query wave stores its part after the publish (below)
#pragma unroll
            for (int i = 0; i < NSL; ++i) asm volatile("" : "+v"(lv[i]));
            float sm[HB];
            const int qcol[4] = {0, 256, 512, 768};
            role_dots<HB, 4>(sm, RQ, xs + (size_t)qrow * KX, KX, qcol, lane);
            FTR(1, 15);
            if (lane < HB && qrow + lane < B) publish(a.xch + X.q + (qrow + lane) * ATT + qdim, tag, pick_row<HB>(sm, lane));
            if (lane == 0) stamp(ctl + 2);
            FTR(1, 3);
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                const int idx = rt + i * 256;
                if (idx < NL4) *reinterpret_cast<f32x4*>(wl + (size_t)idx * 4) = lv[i];
            }
            if (lane == 0) __hip_atomic_fetch_add(ctl + 5, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (has_ctx) {                                    // this unit's slice memory[cb][0 .. Tin)[c8 * 8 .. + 8) -> LDS (32 bytes per position);
                                                      // needed three hops from now: requested here, drained before the softmax
        const __amdgpu_buffer_rsrc_t rs_m = rsrc_of(a.memory + (size_t)cb * Tin * ENC + c8 * 8);
#pragma unroll
        for (int i = 0; i < TP / 32; ++i) {
            const int tt = min(32 * i + (lane >> 1), Tin - 1);       // positions beyond Tin carry weight 0
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m, (lds_ptr_t)(msl + (r * TP + 32 * i) * 8), 16,
                                                     (unsigned)tt * (ENC * 4u) + (lane & 1) * 16u, 0, 0, 0);
        }
    }
    {   // all parts of the location map are in LDS
        long long spins = 0;
        while (lds_peek(ctl + 5) < (two_pairs ? 3 : 4)) {
            if (++spins > LDS_SPIN_LIMIT) { P.give_up(); break; }
            if ((spins & 255) == 0 && lds_peek(ctl)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (two_pairs) {
        // the location terms of BOTH positions first, then ONE look at the query rows they need (term, look, term, look put a
        // second round trip on the chain -- and such a wave may be a context unit, whose slice every block then waits for)
        f32x2 loc[NPOS];
#pragma unroll
        for (int p = 0; p < NPOS; ++p) {
            loc[p] = pmv[p];
            if (g_id + 1024 * p < B * Tin) {          // wave-uniform
#pragma unroll
                for (int i = 0; i < LOCK; ++i) {
                    const float sp = lane_bcast(cp[p], i), sc = lane_bcast(cc[p], i);
                    const f32x2 w0 = *reinterpret_cast<const f32x2*>(wl + (2 * i) * ATT + lane * 2);
                    const f32x2 w1 = *reinterpret_cast<const f32x2*>(wl + (2 * i + 1) * ATT + lane * 2);
                    loc[p][0] = fmaf(sp, w0[0], loc[p][0]);
                    loc[p][1] = fmaf(sp, w0[1], loc[p][1]);
                    loc[p][0] = fmaf(sc, w1[0], loc[p][0]);
                    loc[p][1] = fmaf(sc, w1[1], loc[p][1]);
                }
            }
        }
        if (g_id < B * Tin) {                         // this wave owns at least one position
            wait_stamp(P, ctl + 2, quiet_pre(a.delay[2]));
            quiet_begin<1>(ctl, lane);
            wait_stamp(P, ctl + 2, a.delay[2]);
            unsigned ent[NPOS];
#pragma unroll
            for (int p = 0; p < NPOS; ++p) {
                const int idx = g_id + 1024 * p < B * Tin ? g_id + 1024 * p : g_id;
                ent[p] = X.q + (unsigned)(idx / Tin) * ATT + (unsigned)lane * 2;
            }
            f32x2 qv[NPOS];
            poll_pairs<NPOS>(P, ent, tag, qv);
            quiet_end<1>(ctl, lane);
            if (r == 0) FTR(1, 4);
#pragma unroll
            for (int p = 0; p < NPOS; ++p) {
                const int idx = g_id + 1024 * p;
                if (idx < B * Tin) {
                    const int b = idx / Tin, tau = idx - b * Tin;
                    float e = vv[0] * tanh_fast(qv[p][0] + loc[p][0]);
                    e = fmaf(vv[1], tanh_fast(qv[p][1] + loc[p][1]), e);
                    e = wave_sum(e);
                    if (lane == 0) publish(a.xch + X.e + (unsigned)b * X.TinP + tau, tag, e);
                    if (lane == 1 && tau == Tin - 1 && (Tin & 1)) publish(a.xch + X.e + (unsigned)b * X.TinP + tau + 1, tag, 0.f);
                }
            }
            if (r == 0) FTR(1, 5);
        }
    } else {
    // energies of the positions this wave owns: the location term first (it only needs the previous alignments)
#pragma unroll
    for (int p = 0; p < NPOS; ++p) {
        const int idx = g_id + 1024 * p;
        if (idx < B * Tin) {                          // wave-uniform
            const int b = idx / Tin, tau = idx - b * Tin;
            f32x2 loc = pmv[p];
#pragma unroll
            for (int i = 0; i < LOCK; ++i) {
                const float sp = lane_bcast(cp[p], i), sc = lane_bcast(cc[p], i);
                const f32x2 w0 = *reinterpret_cast<const f32x2*>(wl + (2 * i) * ATT + lane * 2);
                const f32x2 w1 = *reinterpret_cast<const f32x2*>(wl + (2 * i + 1) * ATT + lane * 2);
                loc[0] = fmaf(sp, w0[0], loc[0]);
                loc[1] = fmaf(sp, w0[1], loc[1]);
                loc[0] = fmaf(sc, w1[0], loc[0]);
                loc[1] = fmaf(sc, w1[1], loc[1]);
            }
            if (p == 0) {
                wait_stamp(P, ctl + 2, quiet_pre(a.delay[2]));
                quiet_begin<1>(ctl, lane);
                wait_stamp(P, ctl + 2, a.delay[2]);
            }
            unsigned ent[1] = {X.q + (unsigned)b * ATT + (unsigned)lane * 2};
            f32x2 qv[1];
            poll_pairs<1>(P, ent, tag, qv);
            if (p == 0) quiet_end<1>(ctl, lane);
            if (r == 0 && p == 0) FTR(1, 4);
            float e = vv[0] * tanh_fast(qv[0][0] + loc[0]);
            e = fmaf(vv[1], tanh_fast(qv[0][1] + loc[1]), e);
            e = wave_sum(e);
            if (lane == 0) publish(a.xch + X.e + (unsigned)b * X.TinP + tau, tag, e);
            // an odd Tin: the row's last 16-byte poll also covers one padding entry, which must carry the tag
            if (lane == 1 && tau == Tin - 1 && (Tin & 1)) publish(a.xch + X.e + (unsigned)b * X.TinP + tau + 1, tag, 0.f);
            if (r == 0 && p == 0) FTR(1, 5);
        }
    }
    }
    // the stamp of the energies hop: left by the last of the block's waves that published any (with fewer than 1 024 (row,
    // position) pairs the query wave owns none and used to stamp ~2.5 us before the first energy existed: every context unit
    // then looked early and kept polling).  A block without a producer estimates: query hop + location term.
    {
        const int n_e = min(4, max(0, (B * Tin - blk + NBLK - 1) / NBLK));
        if (g_id < B * Tin) {
            if (lane == 0) stamp_last(ctl + 8, n_e, ctl + 3);
        } else if (n_e == 0 && r == 3 && lane == 0) {
            stamp(ctl + 3, 200);
        }
    }
    if (has_ctx) {                                    // softmax of row cb, 8 context columns
        unsigned ent[KT];
        const unsigned rowe = X.e + (unsigned)cb * X.TinP;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int tau = 128 * k + 2 * lane;
            ent[k] = rowe + (unsigned)(tau < X.TinP ? tau : 0);
        }
        wait_stamp(P, ctl + 3, quiet_pre(a.delay[3]));
        quiet_begin<1>(ctl, lane);
        wait_stamp(P, ctl + 3, a.delay[3]);
        f32x2 ev[KT];
        poll_pairs<KT>(P, ent, tag, ev);              // (also drains this wave's LDS-DMA of the memory slice: vmcnt is in order)
        quiet_end<1>(ctl, lane);
        if (r == 0) FTR(1, 6);
        // attention window (tacotron2_arch.py:630-638); inclusive upper bound
        int lo = 0, hi = Tin;
        if (a.win_len > 0) {
            int center = max(matt_old, a.win_off);
            center = min(center, elen - a.win_len + a.win_off);
            lo = center - a.win_off;
            hi = center - a.win_off + a.win_len;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tau = 128 * k + 2 * lane + h;
                bool on = (on_bits >> (2 * k + h)) & 1u;
                if (a.win_len > 0) on = on && tau >= lo && tau <= hi;
                ev[k][h] = on ? ev[k][h] : -INFINITY;
                mx = fmaxf(mx, ev[k][h]);
            }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                ev[k][h] = __builtin_amdgcn_exp2f((ev[k][h] - mx) * 1.4426950408889634f);      // 0 at masked positions
                sum += ev[k][h];
            }
        const float rsum = __builtin_amdgcn_rcpf(wave_sum(sum));
        float* wrow = wsm + r * TP;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            ev[k][0] *= rsum;
            ev[k][1] *= rsum;
            *reinterpret_cast<f32x2*>(wrow + 128 * k + 2 * lane) = ev[k];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the memory slice has landed (normally long ago)
        float acc = 0.f;                              // lane = (time slice ts = lane >> 3, column lane & 7)
        const float* mrow = msl + r * TP * 8 + lane;
#pragma unroll
        for (int i = 0; i < MPF; ++i) {
            const int tt = (lane >> 3) + 8 * i;
            const float wv = wrow[tt];                // same wave: LDS operations complete in order; 0 beyond Tin (masked)
            acc = fmaf(wv, mrow[64 * i], acc);
        }
        acc += dpp<DPP_ROR8>(acc);                    // time slices ts ^ 1 (lane ^ 8 inside a row of 16)
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        if (lane < 8) {
            const unsigned o = (unsigned)cb * ENC + c8 * 8 + lane;
            publish(a.xch + X.ctx + o, tag, acc);
            a.ctx[o] = acc;                           // for X(t + 1) / the tail projection (next kernel: plain load)
        }
        if (lane == 0) stamp_last(ctl + 9, n_c, ctl + 4);
        if (r == 0) FTR(1, 7);
        if (c8 == 0) {                                // the row's bookkeeping: alignments, history, arg max
            float best = -1.f;
            int besti = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tau = 128 * k + 2 * lane + h;
                    if (tau < Tin) {
                        const float p = ev[k][h];
                        a.wprev[(size_t)cb * Tin + tau] = p;
                        a.wcum[(size_t)cb * Tin + tau] = wc_old[2 * k + h] + p;
                        if (a.attn_hist) a.attn_hist[((size_t)cb * max_len + t) * Tin + tau] = p;
                        if (p > best) { best = p; besti = tau; }      // ascending order keeps the lowest index per lane
                    }
                }
            if (a.win_len > 0) {
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    const float ob = __shfl_xor(best, m, 64);
                    const int oi = __shfl_xor(besti, m, 64);
                    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
                }
                if (lane == 0) a.mainatt[(par ^ 1) * B + cb] = besti;
            }
        }
    } else if (n_c == 0 && r == 0 && lane == 0) {
        stamp(ctl + 4, 250);                          // a block without a context unit (fewer than 4 rows): energies hop + softmax from here
    }
    {   // every role wave fetches a quarter of the context of all rows into the LDS rows
        constexpr int NQ = NBT * ENC / 2 / 256;       // pairs per lane
        unsigned ent[NQ];
        int pair[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            pair[i] = r * (NQ * 64) + i * 64 + lane;
            ent[i] = X.ctx + 2u * (unsigned)(pair[i] < B * ENC / 2 ? pair[i] : 0);
        }
        wait_stamp(P, ctl + 4, quiet_pre(a.delay[4]));
        quiet_begin<1>(ctl, lane);
        wait_stamp(P, ctl + 4, a.delay[4]);
        f32x2 cv[NQ];
        poll_pairs<NQ>(P, ent, tag, cv);
        quiet_end<1>(ctl, lane);
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (pair[i] < B * ENC / 2) {
                const int o = 2 * pair[i], b = o / ENC, c = o - b * ENC;
                *reinterpret_cast<f32x2*>(xs + (size_t)b * KX + RNN + c) = cv[i];
            }
        if (r == 0) FTR(1, 8);
    }
    __syncthreads();                                  // #2
    if (r == 0) FTR(1, 9);
}

__global__ void fused_init_kernel(FusedState* st, int B, int max_len, int early_stop) {
    st->t0 = 0;
    st->n_fin = 0;
    st->steps_run = 0;
    st->exec_t = 0;
    st->B = B;
    st->max_len = max_len;
    st->early_stop = early_stop;
    st->pad = 0;
}
// end of a chunk: one 64-byte report for the host (loop state, abort code, encoder status), then the next chunk's base step
__global__ void fused_advance_kernel(FusedState* st, const int* flags, const int* bl_err, int* report) {
    const int* sp = (const int*)st;
    for (int i = 0; i < 8; ++i) report[i] = sp[i];
    report[8] = flags[0];
    report[9] = bl_err ? bl_err[0] : 0;
    st->t0 += FUSED_CHUNK;
}

size_t lds_x(int NBT, int ENC) { return ((size_t)NBT * (2 * RNN + ENC) + (size_t)NBT * PRE + 8) * sizeof(float); }
size_t lds_y(int NBT, int ENC, int KT) {
    return ((size_t)NBT * (2 * RNN + ENC) + 2 * LOCK * ATT + 4 * KT * 128 * 9 + 16) * sizeof(float);
}

template <int NBT, int ENC, bool HW>
hipError_t launch_x(hipStream_t st, const FusedArgs& a, int j, int tail) {
    auto kern = fused_x_kernel<NBT, ENC, HW>;
    const size_t lds = lds_x(NBT, ENC);
    static PerDeviceOnce attr;
    if (hipError_t er = set_max_dyn_lds_once((const void*)kern, lds, attr); er != hipSuccess) return er;
    hipLaunchKernelGGL(kern, dim3(NBLK), dim3(NTHR), lds, st, a, j, tail);
    return hipGetLastError();
}
template <int NBT, int ENC, int KT, bool HW>
hipError_t launch_y(hipStream_t st, const FusedArgs& a, int j) {
    auto kern = fused_y_kernel<NBT, ENC, KT, HW>;
    const size_t lds = lds_y(NBT, ENC, KT);
    static PerDeviceOnce attr;
    if (hipError_t er = set_max_dyn_lds_once((const void*)kern, lds, attr); er != hipSuccess) return er;
    hipLaunchKernelGGL(kern, dim3(NBLK), dim3(NTHR), lds, st, a, j);
    return hipGetLastError();
}

template <int NBT, int ENC, bool HW>
hipError_t launch_y_kt(hipStream_t st, const FusedArgs& a, int j, int KT) {
    switch (KT) {
        case 1: return launch_y<NBT, ENC, 1, HW>(st, a, j);
        case 2: return launch_y<NBT, ENC, 2, HW>(st, a, j);
        default: return hipErrorInvalidValue;
    }
}

template <int NBT, int ENC, bool HW>
hipError_t chunk_t(hipStream_t st, const FusedArgs& a, int KT, const int* bl_err, int* report) {
    for (int j = 0; j < FUSED_CHUNK; ++j) {
        if (hipError_t er = launch_x<NBT, ENC, HW>(st, a, j, 0); er != hipSuccess) return er;
        if (hipError_t er = launch_y_kt<NBT, ENC, HW>(st, a, j, KT); er != hipSuccess) return er;
    }
    if (hipError_t er = launch_x<NBT, ENC, HW>(st, a, FUSED_CHUNK, 1); er != hipSuccess) return er;
    hipLaunchKernelGGL(fused_advance_kernel, dim3(1), dim3(1), 0, st, a.st, (const int*)a.flags, bl_err, report);
    return hipGetLastError();
}

void pick_shape(int B, int Tin, int* NBT, int* KT) {
    *NBT = B <= 4 ? 4 : 8;
    *KT = Tin <= 128 ? 1 : 2;
}

}  // namespace

size_t fused_xch_u64(int B, int Tin, int enc) { return fx_layout(B, Tin, enc).total; }

bool fused_applicable(const tts_hip_engine* e, int B, int Tin) {
    const int enc = e->taco.enc_dim;
    if (!e->taco.pfold_w || e->n_cu < NBLK || B < 1 || B > FUSED_MAX_B || Tin < 2 || Tin > 256) return false;
    if (enc != 512 && enc != 768) return false;
    int NBT, KT;
    pick_shape(B, Tin, &NBT, &KT);
    return lds_y(NBT, enc, KT) <= 160 * 1024 && lds_x(NBT, enc) <= 160 * 1024;
}

int fused_init(tts_hip_engine* e, hipStream_t st, const FusedCall& c) {
    hipLaunchKernelGGL(fused_init_kernel, dim3(1), dim3(1), 0, st, c.state, c.B, c.max_len, c.early_stop);
    HIPCHK(e, hipGetLastError());
    return TTS_HIP_OK;
}

int fused_enqueue_chunk(tts_hip_engine* e, hipStream_t st, const FusedCall& c) {
    Tacotron2Dev& tc = e->taco;
    const int enc = tc.enc_dim;
    FusedArgs a{};
    a.B = c.B; a.Tin = c.Tin; a.win_len = c.win_len; a.win_off = c.win_off;
    a.Wa = c.half_w ? (const void*)tc.att.W16 : (const void*)tc.att.W;
    a.Wd = c.half_w ? (const void*)tc.dec.W16 : (const void*)tc.dec.W;
    a.ba = tc.att.b; a.bd = tc.dec.b;
    a.Ff = tc.pfold_w; a.fb = tc.pfold_b;
    a.W1t = tc.prenet_w1;
    a.Pw = tc.proj_w; a.Pb = tc.proj_b;
    a.Wq = tc.query_w; a.wloc = tc.loc_dense; a.vw = tc.value_w;
    a.memory = c.memory; a.pm = c.pm; a.mask = c.mask; a.enc_len = c.enc_len; a.masks = c.masks;
    a.xch = c.xch; a.flags = c.flags; a.st = c.state;
    a.hatt = c.hatt; a.hdec = c.hdec; a.catt = c.catt; a.cdec = c.cdec; a.ctx = c.ctx;
    a.wprev = c.wprev; a.wcum = c.wcum; a.mainatt = c.mainatt;
    a.dec_out = c.dec_out; a.stop_out = c.stop_out; a.attn_hist = c.attn_hist; a.lengths = c.lengths; a.finished = c.finished;
    a.trace = c.trace;
    // first look at a hop this long after the block's own producer published (10-ns ticks): the latency of a tagged publish
    // next to the (pausing) weight stream.  Found per kernel shape with scripts/fused_sweep.py on the debug build (round 4: the
    // sweep now really changes the delays of the graph it times); a wrong value costs time, never correctness.
    static const int kDelay[2][2][5] = {
        {{60, 80, 100, 40, 60}, {20, 40, 80, 20, 60}},         // 4-row kernels: fp32 weights, fp16 weights
        {{30, 90, 150, 40, 60}, {40, 40, 120, 20, 70}}};       // 8-row kernels
    for (int i = 0; i < 5; ++i) a.delay[i] = kDelay[c.B > 4 ? 1 : 0][c.half_w ? 1 : 0][i];
#ifdef TTS_DEBUG_HOOKS
    if (const char* dl = getenv("TTS_FUSED_DELAYS")) {            // "a,b,c,d,e"
        int v[5];
        if (sscanf(dl, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]) == 5)
            for (int i = 0; i < 5; ++i) a.delay[i] = v[i];
    }
#endif
    int NBT, KT;
    pick_shape(c.B, c.Tin, &NBT, &KT);
    hipError_t er;
    const int key = (NBT == 8 ? 4 : 0) | (enc == 768 ? 2 : 0) | (c.half_w ? 1 : 0);
    switch (key) {
        case 0: er = chunk_t<4, 512, false>(st, a, KT, c.bl_err, c.report); break;
        case 1: er = chunk_t<4, 512, true>(st, a, KT, c.bl_err, c.report); break;
        case 2: er = chunk_t<4, 768, false>(st, a, KT, c.bl_err, c.report); break;
        case 3: er = chunk_t<4, 768, true>(st, a, KT, c.bl_err, c.report); break;
        case 4: er = chunk_t<8, 512, false>(st, a, KT, c.bl_err, c.report); break;
        case 5: er = chunk_t<8, 512, true>(st, a, KT, c.bl_err, c.report); break;
        case 6: er = chunk_t<8, 768, false>(st, a, KT, c.bl_err, c.report); break;
        default: er = chunk_t<8, 768, true>(st, a, KT, c.bl_err, c.report); break;
    }
    HIPCHK(e, er);
    return TTS_HIP_OK;
}
