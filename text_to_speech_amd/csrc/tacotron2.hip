// tacotron2.hip -- Tacotron2 inference on gfx950: encoder, autoregressive decoder, postnet.
//
// Replaces /root/reference/architectures/tacotron2_arch.py:866-925 (Tacotron2.infer), :609-749 (Tacotron2Decoder.infer
// and its while_loop body), :422-486 (Tacotron2DecoderCell.call), :188-203 (prenet), :235-333 (encoder), :214-232
// (postnet) and architectures/layers/location_sensitive_attention.py:96-186.
//
// Decoder step = 6 launches, weights streamed once per step (HBM/Infinity-Cache bound):
//   prenet        frame[B,80] -> p2[B,256]                      grid (B, 8)
//   lstm_step<KS> attention LSTM: x = [p2 | ctx | h_att] (K = 256*KS), 4 gate rows per wave, weights in registers,
//                 x staged in LDS, lane-halving shuffle reduction, fused cell update                 grid 256
//   energies      q = h_att Wq (per block), location features, e = v . tanh(q + pm + loc)            grid (B, Tin/16)
//   softmax_ctx   masked softmax over Tin (wave shuffles), ctx = w @ memory, cum += w, history       grid (B, enc/128)
//   lstm_step<KS> decoder LSTM: x = [h_att | ctx | h_dec]                                             grid 256
//   project       frame / stop token, finished / lengths bookkeeping, output scatter                 grid (21) x B rows
// The step index is `*t0 + j` (j baked into the launch), so 32 steps form one hipGraph that is replayed per chunk;
// every kernel returns immediately once all rows are finished (device flag), which keeps the reference's
// "stop as soon as every row has fired" semantics (:625-627) without a host round trip per step.
#include "engine.h"
#include "taco_persist.h"
#include "taco_fused.h"

#include <atomic>
#include <mutex>
#include <type_traits>
#include "gemm_f32.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

using namespace ttsgemm;

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NB = 8;                 // batch rows processed together by the LSTM kernel (zero padded)
constexpr int PRE = 256, ARNN = 1024, DRNN = 1024, ATT = 128, NMEL = 80, LOCK = 31;
constexpr int CHUNK = 32;             // decoder steps per hipGraph replay

struct __attribute__((aligned(32))) DecState {   // device-resident loop state (one per call); one 32-byte scalar load
    int t0;                           // first step of the current chunk
    int n_fin[2];                     // rows whose stop token has fired, in two parity slots: the kernels of step t read
                                      // slot t & 1 and the stop-token wave of step t writes slot (t + 1) & 1, so the value a
                                      // launch decides `done` from cannot change while that launch is running
    int steps_run;                    // loop iterations executed so far
    int B, max_len, early_stop;
    int pad;
};

// ------------------------------------------------------------------------------------------------ weight packing
// LSTM: Wp[4*u + g][k] = k < kin ? kernel[k][g*U + u] : recurrent[k - kin][g*U + u]
__global__ void pack_lstm_kernel(const float* __restrict__ kernel, const float* __restrict__ rec,
                                 const float* __restrict__ bias, float* __restrict__ Wp, float* __restrict__ bp,
                                 int kin, int U) {
    const int K = kin + U;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long long)4 * U * K) {
        const int r = (int)(idx / K), k = (int)(idx % K);
        const int u = r >> 2, gt = r & 3;
        const int col = gt * U + u;
        Wp[idx] = k < kin ? kernel[(long long)k * 4 * U + col] : rec[(long long)(k - kin) * 4 * U + col];
    }
    if (idx < 4 * U) {
        const int u = (int)idx >> 2, gt = (int)idx & 3;
        bp[idx] = bias[gt * U + u];
    }
}

// ------------------------------------------------------------------------------------------------ encoder kernels
__global__ void embed_kernel(const int* __restrict__ tok, const float* __restrict__ emb, float* __restrict__ x,
                             uint8_t* __restrict__ mask, int n_rows, int vocab) {
    const int row = blockIdx.x;
    if (row >= n_rows) return;
    int id = tok[row];
    const bool on = id != 0;
    if (id < 0 || id >= vocab) id = 0;
    if (threadIdx.x == 0) mask[row] = on ? 1 : 0;
    for (int c = threadIdx.x; c < 512; c += blockDim.x) x[(long long)row * 512 + c] = on ? emb[(long long)id * 512 + c] : 0.f;
}

// Masked (Bi)LSTM recurrence.  xproj already holds x @ kernel + bias for every step (one GEMM), so a step is
// h @ recurrent + pointwise.  The recurrent kernel of one direction is 256 x 1024 floats = 1 MiB: streamed from L2 by a
// single CU it costs 6.9 us per step (the first version: 0.88 ms for 128 tokens).  Here BL_Q = 4 blocks of 1024 threads
// share a (direction, batch row): block q owns 64 units (256 gate columns) and keeps its 256 x 256 slice in registers
// (64 per thread: column c = tid & 255, k-quarter kq = tid >> 8), so a step is 64 FMAs per thread, an LDS reduction over
// the four k-quarters, the cell update of the block's units and an exchange of the 4 x 64 new h values through global
// memory: every value is published as one 8-byte (step tag, value) agent-scope store into a parity double buffer and
// polled by the thread that needs it.  The four blocks of a group have consecutive block ids, so they are dispatched
// together; every wait is bounded (BL_SPIN) and reports through `err`.
constexpr int BL_Q = 4;
constexpr long long BL_SPIN = 1ll << 22;
__global__ __launch_bounds__(1024) void bilstm_kernel(const float* __restrict__ xproj, const float* __restrict__ rec_f,
                                                      const float* __restrict__ rec_b,
                                                      const uint8_t* __restrict__ mask, float* __restrict__ memory,
                                                      unsigned long long* hx, int* err, int Tin, int enc) {
    __shared__ float h_s[256];
    __shared__ float g_s[4][256];
    __shared__ int stop_s;
    const int q = blockIdx.x, dir = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int c = tid & 255, kq = tid >> 8;
    const int gate = c >> 6, ul = c & 63;
    const int col = gate * 256 + q * 64 + ul;                  // Keras column of (gate, unit q * 64 + ul)
    const float* U = dir == 0 ? rec_f : rec_b;
    float w[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) w[i] = U[(long long)(kq * 64 + i) * 1024 + col];
    unsigned long long* hgrp = hx + ((long long)(dir * gridDim.z + b) * 2) * 256;        // [parity][256] (step tag, h)
    float cstate = 0.f, h_own = 0.f;                           // threads tid < 64: unit q * 64 + tid
    if (tid < 256) h_s[tid] = 0.f;
    if (tid == 0) stop_s = 0;
    __syncthreads();
    for (int s = 0; s < Tin; ++s) {
        const int t = dir == 0 ? s : Tin - 1 - s;
        const long long row = (long long)b * Tin + t;
        const bool on = mask[row] != 0;                        // block- and group-uniform
        float xg[4] = {0.f, 0.f, 0.f, 0.f};
        if (on && tid < 64) {                                  // requested before the dot products
#pragma unroll
            for (int g2 = 0; g2 < 4; ++g2) xg[g2] = xproj[row * 2048 + dir * 1024 + g2 * 256 + q * 64 + tid];
        }
        if (on) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 64; ++i) acc = fmaf(h_s[kq * 64 + i], w[i], acc);
            g_s[kq][c] = acc;
        }
        __syncthreads();
        if (tid < 64) {
            float hv = 0.f;
            if (on) {
                float pre[4];
#pragma unroll
                for (int g2 = 0; g2 < 4; ++g2) {
                    const int cc = g2 * 64 + tid;
                    pre[g2] = xg[g2] + ((g_s[0][cc] + g_s[1][cc]) + (g_s[2][cc] + g_s[3][cc]));
                }
                const float ig = sigmoid_exact(pre[0]), fg = sigmoid_exact(pre[1]);
                const float gg = tanhf(pre[2]), og = sigmoid_exact(pre[3]);
                cstate = fg * cstate + ig * gg;
                h_own = og * tanhf(cstate);
                hv = h_own;
            }
            // publish (step tag, value) as ONE 8-byte agent-scope store: the tag travels with the data, so the readers
            // need no flag, no fence and no second round trip
            const unsigned long long packed = ((unsigned long long)(unsigned)(s + 1) << 32) | __builtin_bit_cast(unsigned, h_own);
            __hip_atomic_store(hgrp + (s & 1) * 256 + q * 64 + tid, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            memory[row * enc + dir * 256 + q * 64 + tid] = hv;     // padded positions: 0 (state carried through)
        }
        if (s + 1 == Tin) break;                               // nobody needs the last exchange
        if (tid < 256) {
            float hn = h_own;                                  // own slice: no round trip (threads q*64 .. q*64+63 below)
            if ((tid >> 6) != q) {
                const unsigned long long* src = hgrp + (s & 1) * 256 + tid;
                long long spins = 0;
                unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while ((unsigned)(v >> 32) != (unsigned)(s + 1)) {
                    if (++spins > BL_SPIN) {
                        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        stop_s = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                hn = __builtin_bit_cast(float, (unsigned)(v & 0xffffffffull));
                h_s[tid] = hn;
            }
        }
        if (tid < 64) h_s[q * 64 + tid] = h_own;
        __syncthreads();
        if (stop_s) return;                                    // a partner never arrived: give up (reported by the host)
    }
}

__global__ void speaker_concat_kernel(const float* __restrict__ spk, const uint8_t* __restrict__ mask,
                                      float* __restrict__ memory, int Tin, int enc, int spk_dim, long long rows) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * spk_dim) return;
    const long long row = idx / spk_dim;
    const int c = (int)(idx % spk_dim);
    const int b = (int)(row / Tin);
    memory[row * enc + 512 + c] = mask[row] ? spk[(long long)b * spk_dim + c] : 0.f;
}

__global__ void enc_len_kernel(const uint8_t* __restrict__ mask, int* __restrict__ enc_len, int Tin) {
    const int b = blockIdx.x;
    int n = 0;
    for (int t = threadIdx.x; t < Tin; t += 64) n += mask[(long long)b * Tin + t];
    for (int s = 32; s >= 1; s >>= 1) n += __shfl_xor(n, s, 64);
    if (threadIdx.x == 0) enc_len[b] = n;
}

// ------------------------------------------------------------------------------------------------ decoder kernels
// Branch-free on purpose: with short-circuit evaluation every field became its own dependent scalar load + wait + branch
// (4-5 sequential ~0.3 us round trips at the head or tail of every step kernel); this way all fields arrive with one load.
__device__ __forceinline__ bool step_done(const DecState* st, int j, int& t) {
    const int t0 = st->t0, nf0 = st->n_fin[0], nf1 = st->n_fin[1], nb = st->B, ml = st->max_len, es = st->early_stop;
    const int nf = (j & 1) ? nf1 : nf0;           // CHUNK is even: the parity of t equals the parity of j
    t = t0 + j;
    return (t >= ml) | ((es != 0) & (nf >= nb));
}

// prenet: grid (B, 8).  Every block recomputes layer 1 (80 -> 256) for its row, then its 32 outputs of layer 2.
// All weight loads are issued before anything else (one memory round trip), the loop state is only needed for the
// dropout-mask index and the final store.
__global__ __launch_bounds__(256) void prenet_kernel(const DecState* __restrict__ st, int j,
                                                     const float* __restrict__ frame, const float* __restrict__ w0,
                                                     const float* __restrict__ w1, const float* __restrict__ masks,
                                                     float* __restrict__ p2) {
    __shared__ __attribute__((aligned(16))) float f_s[NMEL];
    __shared__ __attribute__((aligned(16))) float p1_s[PRE];
    const int b = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    const int o = part * 32 + (tid >> 3), sub = tid & 7;
    f32x4 wa[NMEL / 4], wb[8];
#pragma unroll
    // both weight streams are lane-contiguous: w0 is packed [k / 4][256 outputs][4] (one 16-byte load per lane, 1 KiB per
    // wave instruction); the 8 lanes of an output read 128 contiguous bytes of its w1 row per instruction.  (Row-major
    // [256][80] rows made every lane of a wave hit its own cache line: 2 us of the kernel's 4.5.)
    for (int i = 0; i < NMEL / 4; ++i) wa[i] = *reinterpret_cast<const f32x4*>(w0 + ((long long)i * PRE + tid) * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) wb[i] = *reinterpret_cast<const f32x4*>(w1 + o * PRE + i * 32 + sub * 4);
    if (tid < NMEL) f_s[tid] = frame[b * NMEL + tid];
    int t;
    const bool done = step_done(st, j, t);
    float m0 = 1.f, m1 = 1.f;
    if (masks && !done) {
        const long long base = ((long long)b * st->max_len + t) * 2 * PRE;
        m0 = masks[base + tid];
        m1 = masks[base + PRE + o];
    }
    __syncthreads();
    {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NMEL / 4; ++i) {
            const f32x4 fv = *reinterpret_cast<const f32x4*>(f_s + i * 4);
            acc = fmaf(fv[0], wa[i][0], acc);
            acc = fmaf(fv[1], wa[i][1], acc);
            acc = fmaf(fv[2], wa[i][2], acc);
            acc = fmaf(fv[3], wa[i][3], acc);
        }
        p1_s[tid] = fmaxf(acc, 0.f) * m0;
    }
    __syncthreads();
    {   // 32 outputs per block, 8 lanes per output
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 pv = *reinterpret_cast<const f32x4*>(p1_s + i * 32 + sub * 4);
            acc = fmaf(pv[0], wb[i][0], acc);
            acc = fmaf(pv[1], wb[i][1], acc);
            acc = fmaf(pv[2], wb[i][2], acc);
            acc = fmaf(pv[3], wb[i][3], acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (sub == 0 && !done) p2[b * PRE + o] = fmaxf(acc, 0.f) * m1;
    }
}

// One LSTM step.  K = 256 * KS inputs = [seg0 | seg1 | seg2(recurrent h)]; block = 4 waves = 4 hidden units; each wave
// keeps its 4 gate rows (i, f, c, o of one unit) in registers (KS float4 per row per lane), x for NBT batch rows is
// staged in LDS (all loads of the staging pass in flight at once), and the 4 x NBT partial sums are reduced with a
// lane-halving exchange (V - 1 + log2(64 / V) shuffles for V = 4 * NBT values instead of 6 V).
// HW: the weight rows are fp16 in memory (half the bytes of the stream that bounds this kernel); they are widened to fp32
// once per register and everything else -- x, accumulation, gates, cell state -- stays fp32.
template <int KS, int NBT, bool HW>
__global__ __launch_bounds__(256) void lstm_step_kernel(const DecState* __restrict__ st, int j,
                                                        const void* __restrict__ Wp_v, const float* __restrict__ bp,
                                                        const float* __restrict__ s0, int n0,
                                                        const float* __restrict__ s1, int n1,
                                                        const float* __restrict__ h_old, float* __restrict__ h_new,
                                                        float* __restrict__ c_state, int B, int U) {
    constexpr int K = 256 * KS;
    constexpr int V = 4 * NBT;                                       // partial sums per lane
    constexpr int NX4 = NBT * (K / 4);                               // float4 of x to stage
    constexpr int NST = (NX4 + 255) / 256;                           // staging float4 per thread
    extern __shared__ __attribute__((aligned(16))) float xs[];      // [NBT][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = blockIdx.x * 4 + wave;
    // weights -> registers (the longest-latency stream)
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    typedef typename std::conditional<HW, f16x4, f32x4>::type wvec_t;
    typedef typename std::conditional<HW, _Float16, float>::type wel_t;
    wvec_t wraw[4][KS];
    const wel_t* wrow = (const wel_t*)Wp_v + (long long)(4 * u) * K + lane * 4;
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const wvec_t* wp = reinterpret_cast<const wvec_t*>(wrow + (long long)gt * K + i * 256);
            // decoder LSTM (KS >= 10): stream past the L2, so that the attention LSTM's rows -- 1.8 MB per XCD in fp16, the
            // same rows on the same XCD every step -- stay resident in the 4 MB L2 (fp16 mode: 35.1 -> 33.8 us/step;
            // fp32, 3.7 MB per XCD: neutral)
            if constexpr (KS >= 10 || !HW) wraw[gt][i] = __builtin_nontemporal_load(wp);
            else wraw[gt][i] = *wp;
        }

    // stage x[b][:] = [s0[b] | s1[b] | h_old[b]] for NBT rows (zeros beyond B): every load of a pass is issued before
    // any LDS store, and the first pass is issued right behind the weight loads so both streams are in flight together
    f32x4 sv[NST];
    auto issue_stage = [&](int b0) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = tid + i * 256;
            const int bb = idx / (K / 4), k = (idx % (K / 4)) * 4;
            const int b = b0 + bb;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < NX4 && b < B) {
                const float* src = k < n0 ? s0 + (long long)b * n0 + k
                                 : k < n0 + n1 ? s1 + (long long)b * n1 + (k - n0)
                                               : h_old + (long long)b * U + (k - n0 - n1);
                v = *reinterpret_cast<const f32x4*>(src);
            }
            sv[i] = v;
        }
    };
    constexpr int b0 = 0;                       // B <= NBT per launch (the host loops over batch chunks)
    issue_stage(0);
    // Everything the epilogue needs that does not depend on this step's arithmetic (loop state, bias, old cell state) is
    // requested here, right behind the weight and x streams, so that the tail of the kernel is arithmetic + one store
    // instead of a second dependent memory round trip.
    int t;
    const bool done = step_done(st, j, t);
    constexpr int LOGV0 = NBT == 1 ? 2 : NBT == 2 ? 3 : NBT == 4 ? 4 : 5;
    constexpr int SH0 = 6 - LOGV0;
    const bool writer = lane < (NBT << SH0) && (lane & ((1 << SH0) - 1)) == 0 && (lane >> SH0) < B;
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(bp + 4 * u);
    const float c_old = writer ? c_state[(long long)(lane >> SH0) * U + u] : 0.f;
    // pin the weights: without this the compiler sinks each weight load next to its first use inside the FMA loop
    // (serialising HBM round trips) when register pressure is high (NBT = 8)
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
        for (int i = 0; i < KS; ++i) asm volatile("" : "+v"(wraw[gt][i]));
    f32x4 w[4][KS];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            if constexpr (HW) w[gt][i] = f32x4{(float)wraw[gt][i][0], (float)wraw[gt][i][1], (float)wraw[gt][i][2], (float)wraw[gt][i][3]};
            else w[gt][i] = wraw[gt][i];
        }

    {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = tid + i * 256;
            if (idx < NX4) *reinterpret_cast<f32x4*>(xs + (idx / (K / 4)) * K + (idx % (K / 4)) * 4) = sv[i];
        }
        __syncthreads();
        // Packed accumulation: each (gate, row) sum is kept as an (even-k, odd-k) pair so that one v_pk_fma_f32 takes two
        // adjacent registers of the x float4 and of the weight float4.  x is read from LDS in groups of 4 float4, one
        // group ahead of the FMAs that consume it (explicit double buffer; the compiler barrier keeps it from hoisting
        // every LDS read to the top, which spilled at NBT = 8).
        f32x2 acc2[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc2[i] = f32x2{0.f, 0.f};
        constexpr int P = KS * NBT, GSZ = 4, G = (P + GSZ - 1) / GSZ;
        f32x4 xbuf[2][GSZ];
        const float* xl = xs + lane * 4;
#pragma unroll
        for (int q = 0; q < GSZ; ++q)
            if (q < P) xbuf[0][q] = *reinterpret_cast<const f32x4*>(xl + (q % NBT) * K + (q / NBT) * 256);
#pragma clang loop unroll(full)
        for (int g = 0; g < G; ++g) {
            if (g + 1 < G) {
#pragma unroll
                for (int q = 0; q < GSZ; ++q) {
                    const int pp = (g + 1) * GSZ + q;
                    if (pp < P) xbuf[(g + 1) & 1][q] = *reinterpret_cast<const f32x4*>(xl + (pp % NBT) * K + (pp / NBT) * 256);
                }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < GSZ; ++q) {
                const int pp = g * GSZ + q;
                if (pp < P) {
                    const int i = pp / NBT, bb = pp % NBT;
                    const f32x4 xv = xbuf[g & 1][q];
                    const f32x2 xlo = {xv[0], xv[1]}, xhi = {xv[2], xv[3]};
#pragma unroll
                    for (int gt = 0; gt < 4; ++gt) {
                        const f32x2 wlo = {w[gt][i][0], w[gt][i][1]}, whi = {w[gt][i][2], w[gt][i][3]};
                        f32x2 a = acc2[gt * NBT + bb];
                        a = __builtin_elementwise_fma(xlo, wlo, a);
                        a = __builtin_elementwise_fma(xhi, whi, a);
                        acc2[gt * NBT + bb] = a;
                    }
                }
            }
        }
        float acc[V];                                                // index gate * NBT + b
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = acc2[i][0] + acc2[i][1];
        // lane-halving reduction over masks 32, 16, ...: afterwards lane l holds output index l >> SH, summed over the
        // lanes that share its top bits; the remaining low-bit masks are a plain butterfly.
        constexpr int LOGV = NBT == 1 ? 2 : NBT == 2 ? 3 : NBT == 4 ? 4 : 5;
        constexpr int SH = 6 - LOGV;
#pragma unroll
        for (int half = V / 2, m = 32; half >= 1; half >>= 1, m >>= 1) {
            const bool hi = (lane & m) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                float a_lo = acc[i], a_hi = acc[i + half];
                // opaque copies: otherwise instcombine turns select(load, load) into a dynamically indexed load of the
                // register array, which lowers to a compare/select chain over every element (seen for <7, 8>: 3.3 k extra VALU)
                asm volatile("" : "+v"(a_lo), "+v"(a_hi));
                const float send = hi ? a_lo : a_hi;
                const float keep = hi ? a_hi : a_lo;
                acc[i] = keep + __shfl_xor(send, m, 64);
            }
        }
        float v = acc[0];
#pragma unroll
        for (int m = (1 << SH) >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        // lane (gate * NBT + bb) << SH holds the gate pre-activation (without bias): gather the 4 gates per bb
        const int src_b = (lane >> SH) & (NBT - 1);
        const float gi = __shfl(v, (0 * NBT + src_b) << SH, 64);
        const float gf = __shfl(v, (1 * NBT + src_b) << SH, 64);
        const float gg = __shfl(v, (2 * NBT + src_b) << SH, 64);
        const float go = __shfl(v, (3 * NBT + src_b) << SH, 64);
        static_assert(SH == SH0, "writer lanes");
        if (!done && writer) {                      // only the final stores depend on the loop state
            const int b = b0 + (lane >> SH);
            const float ig = sigmoid_exact(gi + bias4[0]);
            const float fg = sigmoid_exact(gf + bias4[1]);
            const float cg = tanhf(gg + bias4[2]);
            const float og = sigmoid_exact(go + bias4[3]);
            const float cn = fg * c_old + ig * cg;
            c_state[(long long)b * U + u] = cn;
            h_new[(long long)b * U + u] = og * tanhf(cn);
        }
    }
}

// q[b][a] = sum_k h_att[b][k] * Wq[k][a]: one block per 2 attention dims, one float4 of k per thread, block reduction.
__global__ __launch_bounds__(256) void query_kernel(const DecState* __restrict__ st, int j,
                                                    const float* __restrict__ h_att, const float* __restrict__ wq_a,
                                                    float* __restrict__ q, int B) {
    __shared__ float red_s[2][4][NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a0 = blockIdx.x * 2;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wq_a + (long long)a0 * ARNN + tid * 4);        // wq_a is [128][1024]
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(wq_a + (long long)(a0 + 1) * ARNN + tid * 4);
    int t;
    const bool done = step_done(st, j, t);        // requested up front, consulted before the store
    for (int b0 = 0; b0 < B; b0 += NB) {
        f32x4 hv[NB];
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            hv[bb] = b0 + bb < B ? *reinterpret_cast<const f32x4*>(h_att + (long long)(b0 + bb) * ARNN + tid * 4) : zero;
        }
        float acc[2 * NB];
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            acc[bb] = hv[bb][0] * w0[0] + hv[bb][1] * w0[1] + hv[bb][2] * w0[2] + hv[bb][3] * w0[3];
            acc[NB + bb] = hv[bb][0] * w1[0] + hv[bb][1] * w1[1] + hv[bb][2] * w1[2] + hv[bb][3] * w1[3];
        }
        // 16 values: halving over masks 32..4, then masks 2, 1
#pragma unroll
        for (int half = NB, m = 32; half >= 1; half >>= 1, m >>= 1) {
            const bool hi = (lane & m) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                float a_lo = acc[i], a_hi = acc[i + half];
                // opaque copies: otherwise instcombine turns select(load, load) into a dynamically indexed load of the
                // register array, which lowers to a compare/select chain over every element (seen for <7, 8>: 3.3 k extra VALU)
                asm volatile("" : "+v"(a_lo), "+v"(a_hi));
                const float send = hi ? a_lo : a_hi;
                const float keep = hi ? a_hi : a_lo;
                acc[i] = keep + __shfl_xor(send, m, 64);
            }
        }
        float v = acc[0];
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        if ((lane & 3) == 0) red_s[(lane >> 2) >> 3][wave][(lane >> 2) & 7] = v;     // idx = lane >> 2 = a_local * NB + bb
        __syncthreads();
        if (tid < 2 * NB && !done) {
            const int al = tid >> 3, bb = tid & 7;
            if (b0 + bb < B)
                q[(long long)(b0 + bb) * ATT + a0 + al] = (red_s[al][0][bb] + red_s[al][1][bb]) + (red_s[al][2][bb] + red_s[al][3][bb]);
        }
        __syncthreads();
    }
}

// energies: grid (B, ceil(Tin / 4)), one wave per input position, one lane per pair of attention dims.  The location conv
// (2 -> 32, k 31) and dense (32 -> 128) are folded at load time into one [62][128] map (no nonlinearity in between);
// e[t] = sum_a v[a] * tanh(q[a] + pm[t][a] + loc[t][a]).  The map is staged once per block in LDS (coalesced), each lane
// then streams its two columns with 62 ds_read_b64 while the previous / cumulative alignment window of the wave's position
// sits in the first 31 lanes and is broadcast with readlane.  (The first version gave a thread 8 dims of one of 16
// positions and re-read 2 x float4 of the map per tap and thread: 1.9 us of LDS traffic in a 5.4 us kernel.)
constexpr int EPB = 4;                     // positions per block
__global__ __launch_bounds__(256) void energies_kernel(const DecState* __restrict__ st, int j,
                                                       const float* __restrict__ q, const float* __restrict__ wloc,
                                                       const float* __restrict__ v_w, const float* __restrict__ pm,
                                                       const float* __restrict__ w_prev,
                                                       const float* __restrict__ w_cum, float* __restrict__ energies,
                                                       int Tin) {
    __shared__ __attribute__((aligned(16))) float wl_s[2 * LOCK * ATT];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tt = blockIdx.y * EPB + wave;                        // this wave's position
    const bool tok = tt < Tin;
    {
        constexpr int N4 = 2 * LOCK * ATT / 4;                     // 1984 float4
        f32x4 tmp[(N4 + 255) / 256];
#pragma unroll
        for (int i = 0; i < (N4 + 255) / 256; ++i) {
            const int idx = tid + i * 256;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            tmp[i] = idx < N4 ? *reinterpret_cast<const f32x4*>(wloc + idx * 4) : zero;
        }
#pragma unroll
        for (int i = 0; i < (N4 + 255) / 256; ++i) {
            const int idx = tid + i * 256;
            if (idx < N4) *reinterpret_cast<f32x4*>(wl_s + idx * 4) = tmp[i];
        }
    }
    // alignment window of this position: lane i < 31 holds w_prev / w_cum at tt + i - 15 (0 outside the sequence)
    const int tw = tt + lane - LOCK / 2;
    const bool win = lane < LOCK && tw >= 0 && tw < Tin;
    const float cp = win ? w_prev[(long long)b * Tin + tw] : 0.f;
    const float cc = win ? w_cum[(long long)b * Tin + tw] : 0.f;
    const f32x2 zero2 = {0.f, 0.f};
    const f32x2 pmv = tok ? *reinterpret_cast<const f32x2*>(pm + ((long long)b * Tin + tt) * ATT + lane * 2) : zero2;
    const f32x2 qv = *reinterpret_cast<const f32x2*>(q + (long long)b * ATT + lane * 2);
    const f32x2 vv = *reinterpret_cast<const f32x2*>(v_w + lane * 2);
    int t;
    const bool done = step_done(st, j, t);        // requested up front, consulted before the store
    __syncthreads();
    f32x2 loc = {0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < LOCK; ++jj) {
        const float sp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cp), jj));
        const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cc), jj));
        const f32x2 w0 = *reinterpret_cast<const f32x2*>(wl_s + (jj * 2 + 0) * ATT + lane * 2);
        const f32x2 w1 = *reinterpret_cast<const f32x2*>(wl_s + (jj * 2 + 1) * ATT + lane * 2);
        loc[0] = fmaf(sp, w0[0], loc[0]);
        loc[1] = fmaf(sp, w0[1], loc[1]);
        loc[0] = fmaf(sc, w1[0], loc[0]);
        loc[1] = fmaf(sc, w1[1], loc[1]);
    }
    float e = vv[0] * tanhf(qv[0] + pmv[0] + loc[0]);
    e = fmaf(vv[1], tanhf(qv[1] + pmv[1] + loc[1]), e);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) e += __shfl_xor(e, m, 64);
    if (lane == 0 && tok && !done) energies[(long long)b * Tin + tt] = e;
}

// softmax + context: grid (B, enc / 32).  Every wave gets the masked softmax statistics (max, sum) over Tin with shuffles
// only; thread tt then writes weight tt to LDS once, and each thread accumulates 1/8 of one of the block's 32 context
// columns from operands it requested at the top of the kernel.  Two barriers instead of the five of the first version
// (4.8 -> 3.x us).  Block y == 0 also updates w_prev / w_cum, the alignment history and main_attention (argmax, first
// index on ties).
__global__ __launch_bounds__(256) void softmax_ctx_kernel(const DecState* __restrict__ st, int j,
                                                          const float* __restrict__ energies,
                                                          const uint8_t* __restrict__ mask,
                                                          const int* __restrict__ enc_len, int win_len, int win_off,
                                                          const int* __restrict__ main_att_old,
                                                          int* __restrict__ main_att, const float* __restrict__ memory,
                                                          float* __restrict__ w_prev, float* __restrict__ w_cum,
                                                          float* __restrict__ ctx, float* __restrict__ attn_hist,
                                                          int Tin, int enc) {
    extern __shared__ float w_s[];                // [Tin] softmax weights
    __shared__ float part[4][32];
    __shared__ float red_s[4];
    __shared__ int redi_s[4];
    const int b = blockIdx.x, ec = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The context operand (this thread's 32-column slice of `memory`, rows ts, ts + 8, ...) does not depend on the softmax:
    // request it first so that it arrives while the statistics are computed (Tin <= 8 * MPF; longer inputs: loop below).
    constexpr int MPF = 32;
    const int col = tid & 31, ts = tid >> 5;
    const float* mem = memory + (long long)b * Tin * enc + ec * 32 + col;
    float mv[MPF];
#pragma unroll
    for (int i = 0; i < MPF; ++i) {
        const int tt = min(ts + 8 * i, Tin - 1);              // clamped, not predicated: 32 independent loads, no branches
        mv[i] = mem[(long long)tt * enc];
    }
    const float wc_old = (ec == 0 && tid < Tin) ? w_cum[(long long)b * Tin + tid] : 0.f;
    int t;
    const bool done = step_done(st, j, t);        // consulted only before the stores below
    const int max_len = st->max_len;
    // attention window (tacotron2_arch.py:630-638)
    int lo = 0, hi = Tin;
    if (win_len > 0) {
        int center = max(main_att_old[b], win_off);      // previous step's argmax (ping-pong: block 0 writes the new one)
        center = min(center, enc_len[b] - win_len + win_off);
        lo = center - win_off;
        hi = center - win_off + win_len;          // inclusive upper bound
    }
    const float* eb = energies + (long long)b * Tin;
    const uint8_t* mb = mask + (long long)b * Tin;
    auto masked_energy = [&](int tt) -> float {          // -inf at padded tokens and outside the window
        bool on = mb[tt] != 0;
        if (win_len > 0) on = on && tt >= lo && tt <= hi;
        return on ? eb[tt] : -INFINITY;
    };
    // wave-level statistics: lane covers tt = lane + 64 k; the first four (Tin <= 256) stay in registers
    float ev[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int tt = lane + 64 * k;
        ev[k] = tt < Tin ? masked_energy(tt) : -INFINITY;
    }
    float mx = fmaxf(fmaxf(ev[0], ev[1]), fmaxf(ev[2], ev[3]));
    for (int tt = lane + 256; tt < Tin; tt += 64) mx = fmaxf(mx, masked_energy(tt));
#pragma unroll
    for (int s2 = 32; s2 >= 1; s2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s2, 64));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) sum += expf(ev[k] - mx);       // exp(-inf) = 0 at masked / out-of-range positions
    for (int tt = lane + 256; tt < Tin; tt += 64) sum += expf(masked_energy(tt) - mx);
#pragma unroll
    for (int s2 = 32; s2 >= 1; s2 >>= 1) sum += __shfl_xor(sum, s2, 64);
    // weight tt = tid (wave w holds it in ev[w]); longer inputs continue in strides of 256
    float best = -1.f;
    int besti = 0x7fffffff;
    {
        const float e_own = wave == 0 ? ev[0] : wave == 1 ? ev[1] : wave == 2 ? ev[2] : ev[3];
        for (int tt = tid; tt < Tin; tt += 256) {
            const float p = expf((tt < 256 ? e_own : masked_energy(tt)) - mx) / sum;
            w_s[tt] = p;
            if (ec == 0 && !done) {
                w_prev[(long long)b * Tin + tt] = p;
                w_cum[(long long)b * Tin + tt] = (tt == tid ? wc_old : w_cum[(long long)b * Tin + tt]) + p;
                if (attn_hist) attn_hist[((long long)b * max_len + t) * Tin + tt] = p;
                if (p > best) { best = p; besti = tt; }      // strided order keeps the lowest index per thread
            }
        }
    }
    if (ec == 0 && !done) {
#pragma unroll
        for (int s2 = 32; s2 >= 1; s2 >>= 1) {
            const float ob = __shfl_xor(best, s2, 64);
            const int oi = __shfl_xor(besti, s2, 64);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        if (lane == 0) { red_s[wave] = best; redi_s[wave] = besti; }
    }
    __syncthreads();
    {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < MPF; ++i) {
            const int tt = ts + 8 * i;
            const float wv = w_s[min(tt, Tin - 1)];
            acc = fmaf(tt < Tin ? wv : 0.f, mv[i], acc);
        }
        for (int tt = ts + 8 * MPF; tt < Tin; tt += 8) acc = fmaf(w_s[tt], mem[(long long)tt * enc], acc);
        acc += __shfl_xor(acc, 32, 64);                   // the wave's two time slices of this column
        if (lane < 32) part[wave][lane] = acc;
    }
    if (ec == 0 && !done && tid == 0) {
        for (int w2 = 1; w2 < 4; ++w2)
            if (red_s[w2] > best || (red_s[w2] == best && redi_s[w2] < besti)) { best = red_s[w2]; besti = redi_s[w2]; }
        main_att[b] = besti;
    }
    __syncthreads();
    if (tid < 32 && !done) ctx[(long long)b * enc + ec * 32 + tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
}

// projection + stop token + bookkeeping.  grid = 21 blocks x 4 waves = 84 waves >= 81 outputs; wave `o` computes
// output column o (0..79 mel, 80 gate) for every batch row from cell_out = [h_dec | ctx]: its weight row stays in
// registers, the x loads of NB rows are all in flight together, and the NB sums use the lane-halving reduction.
template <int KP>
__global__ __launch_bounds__(256) void project_kernel(DecState* __restrict__ st, int j, const float* __restrict__ h_dec,
                                                      const float* __restrict__ ctx, const float* __restrict__ pw,
                                                      const float* __restrict__ pb, float* __restrict__ frame,
                                                      float* __restrict__ dec_out, float* __restrict__ stop_out,
                                                      int* __restrict__ finished, int* __restrict__ lengths, int B) {
    constexpr int K = 256 * KP, enc = K - DRNN;
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o > NMEL) return;
    f32x4 w[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) w[i] = *reinterpret_cast<const f32x4*>(pw + (long long)o * K + i * 256 + lane * 4);
    const float bias = pb[o];
    int t;
    const bool done = step_done(st, j, t);        // requested up front (with max_len), consulted before the stores
    const int max_len = st->max_len;
    const int nf_cur = st->n_fin[j & 1];
    int newly = 0;                                           // rows whose stop token fires in this step (stop wave only)
    for (int b0 = 0; b0 < B; b0 += NB) {
        const int bw = b0 + (lane >> 3);                     // batch row this lane will write
        const int fin_old = (o == NMEL && (lane & 7) == 0 && bw < B) ? finished[bw] : 0;
        float acc[NB];
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const int b = b0 + bb;
            float a = 0.f;
            if (b < B) {
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const int k = i * 256 + lane * 4;
                    const f32x4 xv = k < DRNN ? *reinterpret_cast<const f32x4*>(h_dec + (long long)b * DRNN + k)
                                              : *reinterpret_cast<const f32x4*>(ctx + (long long)b * enc + (k - DRNN));
                    a = fmaf(xv[0], w[i][0], a);
                    a = fmaf(xv[1], w[i][1], a);
                    a = fmaf(xv[2], w[i][2], a);
                    a = fmaf(xv[3], w[i][3], a);
                }
            }
            acc[bb] = a;
        }
#pragma unroll
        for (int half = NB / 2, m = 32; half >= 1; half >>= 1, m >>= 1) {
            const bool hi = (lane & m) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                float a_lo = acc[i], a_hi = acc[i + half];
                // opaque copies: otherwise instcombine turns select(load, load) into a dynamically indexed load of the
                // register array, which lowers to a compare/select chain over every element (seen for <7, 8>: 3.3 k extra VALU)
                asm volatile("" : "+v"(a_lo), "+v"(a_hi));
                const float send = hi ? a_lo : a_hi;
                const float keep = hi ? a_hi : a_lo;
                acc[i] = keep + __shfl_xor(send, m, 64);
            }
        }
        float v = acc[0];
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        const int b = bw;                                    // lane l holds batch row l >> 3 of this chunk
        bool fired = false;
        if ((lane & 7) == 0 && b < B && !done) {
            v += bias;
            if (o < NMEL) {
                frame[b * NMEL + o] = v;
                dec_out[((long long)b * max_len + t) * NMEL + o] = v;
            } else {
                const float sp = sigmoid_exact(v);
                stop_out[(long long)b * max_len + t] = sp;
                // finished |= stop > 0.5 ; lengths += !finished      (tacotron2_arch.py:664-665)
                int fin = fin_old;
                if (!fin && sp > 0.5f) {
                    fin = 1;
                    finished[b] = 1;
                    fired = true;
                }
                if (!fin) lengths[b] += 1;
                if (b == B - 1) st->steps_run = t + 1;
            }
        }
        if (o == NMEL) newly += __popcll(__ballot(fired));   // o is wave-uniform
    }
    // hand the count to the next step through the other parity slot (also when this step did nothing, so that the slot the
    // next step reads is never stale): nobody reads that slot during this launch
    if (o == NMEL && lane == 0) st->n_fin[(j & 1) ^ 1] = nf_cur + newly;
}

__global__ void advance_chunk_kernel(DecState* st) { st->t0 += CHUNK; }

__global__ void init_state_kernel(DecState* st, int B, int max_len, int early_stop) {
    st->t0 = 0;
    st->n_fin[0] = 0;
    st->n_fin[1] = 0;
    st->steps_run = 0;
    st->B = B;
    st->max_len = max_len;
    st->early_stop = early_stop;
}

// dec_mask[b][t] = t <= lengths[b]   (tacotron2_arch.py:745, per row); masked copy of decoder_output for the postnet
__global__ void dec_mask_kernel(const int* __restrict__ lengths, const float* __restrict__ dec_out,
                                uint8_t* __restrict__ dmask, float* __restrict__ xm, int max_len, long long rows) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * NMEL) return;
    const long long row = idx / NMEL;
    const int b = (int)(row / max_len), t = (int)(row % max_len);
    const bool on = t <= lengths[b];
    if (idx % NMEL == 0) dmask[row] = on ? 1 : 0;
    xm[idx] = on ? dec_out[idx] : 0.f;
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) o[idx] = a[idx] + b[idx];
}

// ------------------------------------------------------------------------------------------------ host helpers
struct Arena {
    char* base = nullptr;
    size_t off = 0, cap = 0;
    template <class T>
    T* take(size_t n) {
        off = (off + 255) / 256 * 256;
        T* p = (T*)(base + off);
        off += n * sizeof(T);
        return p;
    }
};

int fold_conv_bn(tts_hip_engine* e, const std::string& conv, const std::string& norm, int cin, int cout, ConvBnDev* out,
                 std::vector<void*>& allocs) {
    const HostTensor* k = find_tensor(e, conv + "/kernel");
    const HostTensor* b = find_tensor(e, conv + "/bias");
    const HostTensor* ga = find_tensor(e, norm + "/gamma");
    const HostTensor* be = find_tensor(e, norm + "/beta");
    const HostTensor* mu = find_tensor(e, norm + "/moving_mean");
    const HostTensor* va = find_tensor(e, norm + "/moving_variance");
    if (!k || !b || !ga || !be || !mu || !va) return set_err(e, TTS_HIP_ENOTREADY, "missing tensors of %s", conv.c_str());
    const std::vector<int64_t> vec{cout};
    if (k->dims != std::vector<int64_t>{5, cin, cout} || b->dims != vec || ga->dims != vec || be->dims != vec ||
        mu->dims != vec || va->dims != vec)
        return set_err(e, TTS_HIP_EINVAL, "unexpected shape for %s / %s (kernel [5, %d, %d], vectors [%d])", conv.c_str(),
                       norm.c_str(), cin, cout, cout);
    const int cin_pad = (cin + 31) / 32 * 32;
    out->cin = cin;
    out->cin_pad = cin_pad;
    out->cout = cout;
    std::vector<float> Bt((size_t)cout * 5 * cin_pad, 0.f), bias(cout), alt(cout);
    for (int o = 0; o < cout; ++o) {
        // BatchNormalization inference: (x - mean) / sqrt(var + eps) * gamma + beta, eps = 1e-5 (tacotron2_arch.py:69,100)
        const double s = (double)ga->data[o] / std::sqrt((double)va->data[o] + 1e-5);
        bias[o] = (float)(((double)b->data[o] - (double)mu->data[o]) * s + (double)be->data[o]);
        alt[o] = (float)((0.0 - (double)mu->data[o]) * s + (double)be->data[o]);      // BN(0): masked rows
        for (int tap = 0; tap < 5; ++tap)
            for (int c = 0; c < cin; ++c)
                Bt[((size_t)o * 5 + tap) * cin_pad + c] = (float)((double)k->data[((size_t)tap * cin + c) * cout + o] * s);
    }
    int rc;
    if ((rc = upload(e, Bt.data(), Bt.size(), &out->Bt, allocs))) return rc;
    if ((rc = upload(e, bias.data(), bias.size(), &out->bias, allocs))) return rc;
    return upload(e, alt.data(), alt.size(), &out->altbias, allocs);
}

int upload_transposed(tts_hip_engine* e, const HostTensor* t, int K, int N, int ldd, float** dst,
                      std::vector<void*>& allocs) {
    std::vector<float> tr((size_t)N * ldd, 0.f);
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) tr[(size_t)n * ldd + k] = t->data[(size_t)k * N + n];
    return upload(e, tr.data(), tr.size(), dst, allocs);
}

// out[m][n] = act(mask(sum_z part[z][m][n] + bias[n]))  -- second half of a conv computed as one K slice per tap
__global__ void conv_reduce_kernel(const float* __restrict__ part, long long zstride, int nz,
                                   const float* __restrict__ bias, const float* __restrict__ altbias,
                                   const uint8_t* __restrict__ rowmask, int mask_out, int act, float* __restrict__ out,
                                   int N, long long total4) {
    const long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= total4) return;
    const long long idx = i4 * 4;
    const long long m = idx / N;
    const int n = (int)(idx % N);
    f32x4 v = *reinterpret_cast<const f32x4*>(bias + n);
    for (int zz = 0; zz < nz; ++zz) v += *reinterpret_cast<const f32x4*>(part + zz * zstride + idx);
    if (rowmask && rowmask[m] == 0) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        v = mask_out ? zero : *reinterpret_cast<const f32x4*>(altbias + n);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = act == ACT_RELU ? fmaxf(v[k], 0.f) : act == ACT_TANH ? tanhf(v[k]) : v[k];
    *reinterpret_cast<f32x4*>(out + idx) = v;
}

// k = 5 "same" conv + folded batch-norm (+ activation, + masking) as an implicit GEMM over the 5 taps.  Few output tiles
// (an utterance's encoder or postnet: M = 100..800 rows) leave most CUs idle and make the K = 5 * 512 MFMA chain of one
// 32 x 32 tile the critical path (1280 x 64 cycles = 34 us), so small problems run one K slice per tap (blockIdx.z = tap:
// 5x the blocks, 1/5 of the chain) into `scratch` [5][M][cout] and a vectorised pass adds the slices, bias, mask and
// activation.  Large ones (>= 512 tiles) keep the single-pass kernel with the fused epilogue.
int conv_gemm(tts_hip_engine* e, const ConvBnDev& cv, const float* x, int ldx, float* out, int M, int L,
              const uint8_t* rowmask, int act, int mask_out, float* scratch, size_t scratch_floats) {
    GemmArgs g{};
    g.M = M;
    g.N = cv.cout;
    g.L = L;
    g.Bt = cv.Bt;
    g.ldb = 5ll * cv.cin_pad;
    g.mode = EPI_LINEAR;
    g.split = cv.cout;
    g.ld0 = cv.cout;
    const long long tiles = (long long)((M + 63) / 64) * ((cv.cout + 63) / 64);
    const size_t need = (size_t)5 * M * cv.cout;
    if (tiles < 512 && scratch && need <= scratch_floats && cv.cout % 4 == 0) {
        g.nseg = 1;
        g.seg[0] = ASeg{x, ldx, -2, cv.cin, cv.cin_pad};
        g.shift_z = 1;                               // tap z reads rows m + z - 2
        g.strideBz = cv.cin_pad;                     // and the z-th K slice of the weight rows
        g.act = ACT_NONE;
        g.out0 = scratch;
        g.strideOutZ = (long long)M * cv.cout;
        HIPCHK(e, gemm_small(g, 5, e->stream));
        const long long total4 = (long long)M * cv.cout / 4;
        hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, e->stream, scratch,
                           (long long)M * cv.cout, 5, cv.bias, cv.altbias, rowmask, mask_out, act, out, cv.cout, total4);
        HIPCHK(e, hipGetLastError());
        return TTS_HIP_OK;
    }
    g.nseg = 5;
    for (int tap = 0; tap < 5; ++tap) g.seg[tap] = ASeg{x, ldx, tap - 2, cv.cin, cv.cin_pad};
    g.bias = cv.bias;
    g.act = act;
    g.out0 = out;
    g.rowmask = rowmask;
    g.altbias = cv.altbias;
    g.mask_out = mask_out;
    HIPCHK(e, gemm_small(g, 1, e->stream));
    return TTS_HIP_OK;
}

template <int KS, int NBT, bool HW>
hipError_t launch_lstm(hipStream_t s, const DecState* st, int j, const LstmDev& L, const float* s0, int n0,
                       const float* s1, int n1, const float* h_old, float* h_new, float* c_state, int B) {
    const size_t lds = (size_t)NBT * 256 * KS * sizeof(float);
    auto kern = lstm_step_kernel<KS, NBT, HW>;
    static PerDeviceOnce attr;
    if (hipError_t er = set_max_dyn_lds_once((const void*)kern, lds, attr); er != hipSuccess) return er;
    const void* W = HW ? (const void*)L.W16 : (const void*)L.W;
    hipLaunchKernelGGL(kern, dim3(L.units / 4), dim3(256), lds, s, st, j, W, L.b, s0, n0, s1, n1, h_old, h_new,
                       c_state, B, L.units);
    return hipGetLastError();
}

template <int KS, bool HW>
hipError_t lstm_by_batch(hipStream_t s, const DecState* st, int j, const LstmDev& L, const float* s0, int n0,
                         const float* s1, int n1, const float* h_old, float* h_new, float* c_state, int B) {
    // one launch per chunk of <= 8 batch rows (B > 8 re-streams the weights from L2 / Infinity Cache per chunk)
    for (int b0 = 0; b0 < B; b0 += NB) {
        const int nb = std::min(NB, B - b0);
        const float* a0 = s0 + (size_t)b0 * n0;
        const float* a1 = s1 + (size_t)b0 * n1;
        const float* ho = h_old + (size_t)b0 * L.units;
        float* hn = h_new + (size_t)b0 * L.units;
        float* cs = c_state + (size_t)b0 * L.units;
        hipError_t er;
        if (nb == 1) er = launch_lstm<KS, 1, HW>(s, st, j, L, a0, n0, a1, n1, ho, hn, cs, nb);
        else if (nb == 2) er = launch_lstm<KS, 2, HW>(s, st, j, L, a0, n0, a1, n1, ho, hn, cs, nb);
        else if (nb <= 4) er = launch_lstm<KS, 4, HW>(s, st, j, L, a0, n0, a1, n1, ho, hn, cs, nb);
        else er = launch_lstm<KS, 8, HW>(s, st, j, L, a0, n0, a1, n1, ho, hn, cs, nb);
        if (er != hipSuccess) return er;
    }
    return hipSuccess;
}

template <bool HW>
hipError_t lstm_dispatch_p(hipStream_t s, const DecState* st, int j, const LstmDev& L, const float* s0, int n0,
                           const float* s1, int n1, const float* h_old, float* h_new, float* c_state, int B) {
    switch ((n0 + n1 + L.units) / 256) {
        case 7: return lstm_by_batch<7, HW>(s, st, j, L, s0, n0, s1, n1, h_old, h_new, c_state, B);
        case 8: return lstm_by_batch<8, HW>(s, st, j, L, s0, n0, s1, n1, h_old, h_new, c_state, B);
        case 10: return lstm_by_batch<10, HW>(s, st, j, L, s0, n0, s1, n1, h_old, h_new, c_state, B);
        case 11: return lstm_by_batch<11, HW>(s, st, j, L, s0, n0, s1, n1, h_old, h_new, c_state, B);
        default: return hipErrorInvalidValue;
    }
}

hipError_t lstm_dispatch(hipStream_t s, const DecState* st, int j, const LstmDev& L, const float* s0, int n0,
                         const float* s1, int n1, const float* h_old, float* h_new, float* c_state, int B, bool half_w) {
    return half_w ? lstm_dispatch_p<true>(s, st, j, L, s0, n0, s1, n1, h_old, h_new, c_state, B)
                  : lstm_dispatch_p<false>(s, st, j, L, s0, n0, s1, n1, h_old, h_new, c_state, B);
}

__global__ void cvt_w16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (_Float16)src[i];
}

}  // namespace

void tacotron2_free(tts_hip_engine* e) {
    for (void* p : e->taco.allocs) (void)hipFree(p);
    e->taco.allocs.clear();
    e->taco.att.W16 = nullptr;
    e->taco.dec.W16 = nullptr;
    e->taco.pfold_w = nullptr;
    e->taco.pfold_b = nullptr;
    tacotron2_graphs_clear(e);
    if (e->taco.pinned) {
        (void)hipHostFree(e->taco.pinned);
        e->taco.pinned = nullptr;
        for (auto& ev : e->taco.chunk_ev) {
            if (ev) (void)hipEventDestroy(ev);
            ev = nullptr;
        }
    }
    if (e->taco.enc_cache) {
        e->taco.enc_cache->buf.release();
        delete e->taco.enc_cache;
        e->taco.enc_cache = nullptr;
    }
    e->taco.ws.release();
    e->taco.io.release();
    e->taco.ready = false;
}

int tacotron2_finalize(tts_hip_engine* e) {
    Tacotron2Dev& tc = e->taco;
    tacotron2_free(e);
    auto& al = tc.allocs;
    int rc;
    auto get = [&](const char* name) { return find_tensor(e, std::string("tacotron2/") + name); };
    // every tensor is indexed with fixed extents below, so its dims are checked here (like `need` in waveglow_finalize)
    auto dims_str = [](const std::vector<int64_t>& d) {
        std::string r = "[";
        for (size_t i = 0; i < d.size(); ++i) r += (i ? ", " : "") + std::to_string((long long)d[i]);
        return r + "]";
    };
#define NEED(var, name, ...)                                                                \
    const HostTensor* var = get(name);                                                      \
    if (!var) { tacotron2_free(e); return set_err(e, TTS_HIP_ENOTREADY, "missing tensor tacotron2/%s", name); }      \
    if (var->dims != std::vector<int64_t>{__VA_ARGS__}) {                                   \
        tacotron2_free(e);                                                                  \
        return set_err(e, TTS_HIP_EINVAL, "tacotron2/%s has shape %s, expected %s", name, dims_str(var->dims).c_str(), \
                       dims_str(std::vector<int64_t>{__VA_ARGS__}).c_str());                \
    }
#define TCHK(x) if ((rc = (x))) { tacotron2_free(e); return rc; }
    // the vocabulary is the checkpoint's: 148 symbols for the reference's English models, 70 for the French table, ...
    const HostTensor* emb = get("encoder/embeddings");
    if (!emb) { tacotron2_free(e); return set_err(e, TTS_HIP_ENOTREADY, "missing tensor tacotron2/encoder/embeddings"); }
    if (emb->dims.size() != 2 || emb->dims[0] < 2 || emb->dims[0] > (1 << 20) || emb->dims[1] != 512) {
        tacotron2_free(e);
        return set_err(e, TTS_HIP_EINVAL, "tacotron2/encoder/embeddings has shape %s, expected [vocab >= 2, 512]", dims_str(emb->dims).c_str());
    }
    tc.vocab = (int)emb->dims[0];
    TCHK(upload(e, emb->data.data(), emb->numel(), &tc.embeddings, al));
    for (int i = 0; i < 3; ++i) {
        const std::string s = std::to_string(i + 1);
        TCHK(fold_conv_bn(e, "tacotron2/encoder/conv_" + s, "tacotron2/encoder/norm_" + s, 512, 512, &tc.enc_conv[i], al));
    }
    {   // BiLSTM: one N = 2048 input projection (forward | backward), recurrent kernels kept in Keras layout
        NEED(kf, "encoder/bi_lstm/forward/kernel", 512, 1024);
        NEED(kb, "encoder/bi_lstm/backward/kernel", 512, 1024);
        NEED(rf, "encoder/bi_lstm/forward/recurrent_kernel", 256, 1024);
        NEED(rb, "encoder/bi_lstm/backward/recurrent_kernel", 256, 1024);
        NEED(bf, "encoder/bi_lstm/forward/bias", 1024);
        NEED(bb, "encoder/bi_lstm/backward/bias", 1024);
        std::vector<float> Bt((size_t)2048 * 512), bias(2048);
        for (int n = 0; n < 1024; ++n) {
            for (int k = 0; k < 512; ++k) {
                Bt[(size_t)n * 512 + k] = kf->data[(size_t)k * 1024 + n];
                Bt[(size_t)(1024 + n) * 512 + k] = kb->data[(size_t)k * 1024 + n];
            }
            bias[n] = bf->data[n];
            bias[1024 + n] = bb->data[n];
        }
        TCHK(upload(e, Bt.data(), Bt.size(), &tc.bl_in_Bt[0], al));
        TCHK(upload(e, bias.data(), bias.size(), &tc.bl_in_b[0], al));
        TCHK(upload(e, rf->data.data(), rf->numel(), &tc.bl_rec[0], al));
        TCHK(upload(e, rb->data.data(), rb->numel(), &tc.bl_rec[1], al));
    }
    NEED(p0, "decoder/prenet/layer_0/kernel", NMEL, PRE);
    NEED(p1, "decoder/prenet/layer_1/kernel", PRE, PRE);
    {   // layer 0: Keras [80][256] -> [k / 4][256][4]
        std::vector<float> w0p((size_t)NMEL * PRE);
        for (int k = 0; k < NMEL; ++k)
            for (int o = 0; o < PRE; ++o) w0p[((size_t)(k / 4) * PRE + o) * 4 + (k & 3)] = p0->data[(size_t)k * PRE + o];
        TCHK(upload(e, w0p.data(), w0p.size(), &tc.prenet_w0, al));
    }
    TCHK(upload_transposed(e, p1, PRE, PRE, PRE, &tc.prenet_w1, al));
    const HostTensor* ak0 = get("decoder/attention_rnn/kernel");
    if (!ak0) { tacotron2_free(e); return set_err(e, TTS_HIP_ENOTREADY, "missing tensor tacotron2/decoder/attention_rnn/kernel"); }
    const int enc = ak0->dims.size() == 2 ? (int)ak0->dims[0] - PRE : -1;    // 512, or 768 with a 256-d speaker embedding
    if (enc != 512 && enc != 768) { tacotron2_free(e); return set_err(e, TTS_HIP_EINVAL, "encoder width %d unsupported (512 or 768)", enc); }
    tc.enc_dim = enc;
    tc.spk_dim = enc - 512;
    NEED(ak, "decoder/attention_rnn/kernel", PRE + enc, 4 * ARNN);
    NEED(ar, "decoder/attention_rnn/recurrent_kernel", ARNN, 4 * ARNN);
    NEED(ab, "decoder/attention_rnn/bias", 4 * ARNN);
    NEED(dk, "decoder/decoder_rnn/cell_0/kernel", ARNN + enc, 4 * DRNN);
    NEED(dr, "decoder/decoder_rnn/cell_0/recurrent_kernel", DRNN, 4 * DRNN);
    NEED(db, "decoder/decoder_rnn/cell_0/bias", 4 * DRNN);
    {
        DevBuf s1, s2, s3;
        auto pack = [&](const HostTensor* k, const HostTensor* r, const HostTensor* b, int kin, LstmDev* L) -> int {
            const int U = 1024, K = kin + U;
            HIPCHK(e, s1.ensure(k->numel() * 4));
            HIPCHK(e, s2.ensure(r->numel() * 4));
            HIPCHK(e, s3.ensure(b->numel() * 4));
            HIPCHK(e, hipMemcpyAsync(s1.p, k->data.data(), k->numel() * 4, hipMemcpyHostToDevice, e->stream));
            HIPCHK(e, hipMemcpyAsync(s2.p, r->data.data(), r->numel() * 4, hipMemcpyHostToDevice, e->stream));
            HIPCHK(e, hipMemcpyAsync(s3.p, b->data.data(), b->numel() * 4, hipMemcpyHostToDevice, e->stream));
            int rc2;
            if ((rc2 = dev_alloc(e, (size_t)4 * U * K, &L->W, al, false))) return rc2;
            if ((rc2 = dev_alloc(e, (size_t)4 * U, &L->b, al, false))) return rc2;
            const long long total = 4ll * U * K;
            hipLaunchKernelGGL(pack_lstm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, s1.f(),
                               s2.f(), s3.f(), L->W, L->b, kin, U);
            HIPCHK(e, hipGetLastError());
            HIPCHK(e, hipStreamSynchronize(e->stream));
            L->units = U;
            L->kin = kin;
            return 0;
        };
        rc = pack(ak, ar, ab, PRE + enc, &tc.att);
        if (!rc) rc = pack(dk, dr, db, ARNN + enc, &tc.dec);
        s1.release();
        s2.release();
        s3.release();
        TCHK(rc);
    }
    NEED(qk, "decoder/lsa/query_layer/kernel", ARNN, ATT);
    NEED(mk, "decoder/lsa/memory_layer/kernel", enc, ATT);
    NEED(vk, "decoder/lsa/value_layer/kernel", ATT, 1);
    NEED(lc, "decoder/lsa/location_conv/kernel", LOCK, 2, 32);
    NEED(ld, "decoder/lsa/location_dense/kernel", 32, ATT);
    TCHK(upload_transposed(e, qk, ARNN, ATT, ARNN, &tc.query_w, al));      // [128][1024]: one attention dim per row
    TCHK(upload_transposed(e, mk, enc, ATT, enc, &tc.memory_Bt, al));
    TCHK(upload(e, vk->data.data(), vk->numel(), &tc.value_w, al));
    {   // fold conv (no bias) and dense (no bias): wloc[(j*2 + c)][a] = sum_f conv[j][c][f] * dense[f][a]
        std::vector<float> wl((size_t)2 * LOCK * ATT);
        for (int jc = 0; jc < 2 * LOCK; ++jc)
            for (int a = 0; a < ATT; ++a) {
                double s = 0;
                for (int f = 0; f < 32; ++f) s += (double)lc->data[(size_t)jc * 32 + f] * (double)ld->data[(size_t)f * ATT + a];
                wl[(size_t)jc * ATT + a] = (float)s;
            }
        TCHK(upload(e, wl.data(), wl.size(), &tc.loc_dense, al));
    }
    NEED(lk, "decoder/linear_projection/kernel", DRNN + enc, NMEL);
    NEED(lb, "decoder/linear_projection/bias", NMEL);
    NEED(gk, "decoder/gate_output/kernel", DRNN + enc, 1);
    NEED(gb, "decoder/gate_output/bias", 1);
    {
        const int K = DRNN + enc;
        std::vector<float> pw((size_t)(NMEL + 1) * K), pb(NMEL + 1);
        for (int o = 0; o < NMEL; ++o) {
            for (int k = 0; k < K; ++k) pw[(size_t)o * K + k] = lk->data[(size_t)k * NMEL + o];
            pb[o] = lb->data[o];
        }
        for (int k = 0; k < K; ++k) pw[(size_t)NMEL * K + k] = gk->data[k];
        pb[NMEL] = gb->data[0];
        TCHK(upload(e, pw.data(), pw.size(), &tc.proj_w, al));
        TCHK(upload(e, pb.data(), pb.size(), &tc.proj_b, al));
    }
    TCHK(persist_finalize(e, p0, lk, lb, enc, al));       // prenet layer 1 folded with the projection (taco_persist.hip)
    {
        int cin = NMEL;
        for (int i = 0; i < 5; ++i) {
            const int cout = i < 4 ? 512 : NMEL;
            const std::string s = std::to_string(i + 1);
            TCHK(fold_conv_bn(e, "tacotron2/postnet/conv_" + s, "tacotron2/postnet/norm_" + s, cin, cout, &tc.post_conv[i], al));
            cin = cout;
        }
    }
#undef NEED
#undef TCHK
    HIPCHK(e, hipStreamSynchronize(e->stream));
    tc.ready = true;
    return TTS_HIP_OK;
}

// fp16 copies of the two decoder LSTM weight matrices (98 % of the bytes a decoder step streams), built on first use
static int tacotron2_build_f16(tts_hip_engine* e) {
    Tacotron2Dev& tc = e->taco;
    for (LstmDev* L : {&tc.att, &tc.dec}) {
        if (L->W16) continue;
        const long long n = 4ll * L->units * (L->kin + L->units);
        void* p = nullptr;
        HIPCHK(e, hipMalloc(&p, (size_t)n * sizeof(_Float16)));
        tc.allocs.push_back(p);
        hipLaunchKernelGGL(cvt_w16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, L->W,
                           (_Float16*)p, n);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipStreamSynchronize(e->stream));
        L->W16 = (_Float16*)p;
    }
    return TTS_HIP_OK;
}

void tacotron2_graphs_clear(tts_hip_engine* e) {
    for (auto& kv : e->taco.graphs) (void)hipGraphExecDestroy(kv.second);
    e->taco.graphs.clear();
    e->taco.graph_order.clear();
}

// ---------------------------------------------------------------------------------------------------------- encoder
// tokens -> (mask, lengths, memory, processed memory) in `out` (its own device buffer, so several encoded utterances can
// be alive); temporaries live in the engine workspace.  Everything is enqueued on e->stream; nothing is synchronized.
static int tacotron2_encode_impl(tts_hip_engine* e, const int32_t* tokens, int B, int Tin, const float* speaker, int mem,
                                 tts_hip_encoded* out) {
    Tacotron2Dev& tc = e->taco;
    if (!tc.ready) return set_err(e, TTS_HIP_ENOTREADY, "tacotron2 weights not finalized");
    if (!tokens || !out || B <= 0 || Tin <= 0 || Tin > 4096 || B > 1024)
        return set_err(e, TTS_HIP_EINVAL, "tacotron2_encode: bad argument");
    if (tc.spk_dim > 0 && !speaker) return set_err(e, TTS_HIP_EINVAL, "tacotron2_encode: this model needs a speaker embedding");
    if (mem != TTS_HIP_MEM_HOST && mem != TTS_HIP_MEM_DEVICE) return set_err(e, TTS_HIP_EINVAL, "bad mem kind %d", mem);
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = e->stream;
    const int enc = tc.enc_dim;
    const long long R = (long long)B * Tin;

    // result buffer
    {
        size_t need = 0;
        auto sz = [&](size_t n, size_t el) { need = (need + 255) / 256 * 256 + n * el; };
        sz(R, 1); sz(B, 4); sz(16, 4); sz(R * enc, 4); sz(R * ATT, 4);
        need += 1024;
        const void* before = out->buf.p;
        HIPCHK(e, out->buf.ensure(need));
        if (out->buf.p != before) tacotron2_graphs_clear(e);      // captured graphs hold pointers into this buffer
        Arena A;
        A.base = (char*)out->buf.p;
        A.cap = out->buf.bytes;
        out->mask = A.take<uint8_t>(R);
        out->enc_len = A.take<int>(B);
        out->bl_err = A.take<int>(16);
        out->memory = A.take<float>(R * enc);
        out->pm = A.take<float>(R * ATT);
        out->B = B;
        out->Tin = Tin;
        out->enc = enc;
    }
    // temporaries
    size_t need = 0;
    auto sz = [&](size_t n, size_t el) { need = (need + 255) / 256 * 256 + n * el; };
    sz(R, 4); sz(R * 512, 4); sz(R * 512, 4); sz(R * 2048, 4); sz((size_t)B * tc.spk_dim + 1, 4); sz((size_t)2 * B * 2 * 256, 8);
    const size_t conv_rows = (size_t)std::min<long long>(R, 32768);
    sz(5 * conv_rows * 512, 4);
    need += 4096;
    {
        const void* before = tc.ws.p;
        HIPCHK(e, tc.ws.ensure(need));
        if (tc.ws.p != before) tacotron2_graphs_clear(e);
    }
    Arena A;
    A.base = (char*)tc.ws.p;
    A.cap = tc.ws.bytes;
    int* d_tok = A.take<int>(R);
    float* d_x0 = A.take<float>(R * 512);
    float* d_x1 = A.take<float>(R * 512);
    float* d_xproj = A.take<float>(R * 2048);
    float* d_spk = A.take<float>((size_t)B * tc.spk_dim + 1);
    unsigned long long* d_blh = A.take<unsigned long long>((size_t)2 * B * 2 * 256);
    float* d_convtmp = A.take<float>(5 * conv_rows * 512);
    const size_t convtmp_n = 5 * conv_rows * 512;
    if (A.off > A.cap) return set_err(e, TTS_HIP_ENOMEM, "tacotron2 workspace accounting error");
    uint8_t* d_mask = out->mask;
    int* d_enc_len = out->enc_len;
    float* d_memory = out->memory;
    float* d_pm = out->pm;

    const hipMemcpyKind kin = mem == TTS_HIP_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    HIPCHK(e, hipMemcpyAsync(d_tok, tokens, R * 4, kin, st));
    if (tc.spk_dim) HIPCHK(e, hipMemcpyAsync(d_spk, speaker, (size_t)B * tc.spk_dim * 4, kin, st));
    HIPCHK(e, hipMemsetAsync(d_blh, 0, (size_t)2 * B * 2 * 256 * 8, st));
    HIPCHK(e, hipMemsetAsync(out->bl_err, 0, 16 * 4, st));

    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)R), dim3(128), 0, st, d_tok, tc.embeddings, d_x0, d_mask, (int)R, tc.vocab);
    HIPCHK(e, hipGetLastError());
    hipLaunchKernelGGL(enc_len_kernel, dim3(B), dim3(64), 0, st, d_mask, d_enc_len, Tin);
    int rc;
    float* xin = d_x0;
    float* xout = d_x1;
    for (int i = 0; i < 3; ++i) {
        // MaskedConv1D -> BN -> relu; rows at padded tokens are stored as zeros (they are only ever consumed masked)
        if ((rc = conv_gemm(e, tc.enc_conv[i], xin, 512, xout, (int)R, Tin, d_mask, ACT_RELU, 1, d_convtmp, convtmp_n))) return rc;
        std::swap(xin, xout);
    }
    {
        GemmArgs g{};
        g.M = (int)R;
        g.N = 2048;
        g.L = (int)R;
        g.nseg = 1;
        g.seg[0] = ASeg{xin, 512, 0, 512, 512};
        g.Bt = tc.bl_in_Bt[0];
        g.ldb = 512;
        g.bias = tc.bl_in_b[0];
        g.mode = EPI_LINEAR;
        g.split = 2048;
        g.out0 = d_xproj;
        g.ld0 = 2048;
        HIPCHK(e, gemm_small(g, 1, st));
    }
    hipLaunchKernelGGL(bilstm_kernel, dim3(BL_Q, 2, B), dim3(1024), 0, st, d_xproj, tc.bl_rec[0], tc.bl_rec[1], d_mask,
                       d_memory, d_blh, out->bl_err, Tin, enc);
    HIPCHK(e, hipGetLastError());
    if (tc.spk_dim) {
        const long long n = R * tc.spk_dim;
        hipLaunchKernelGGL(speaker_concat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_spk, d_mask,
                           d_memory, Tin, enc, tc.spk_dim, R);
        HIPCHK(e, hipGetLastError());
    }
    {   // processed_memory = memory(masked) @ memory_layer   (location_sensitive_attention.py:96-102)
        GemmArgs g{};
        g.M = (int)R;
        g.N = ATT;
        g.L = (int)R;
        g.nseg = 1;
        g.seg[0] = ASeg{d_memory, enc, 0, enc, enc};
        g.Bt = tc.memory_Bt;
        g.ldb = enc;
        g.mode = EPI_LINEAR;
        g.split = ATT;
        g.out0 = d_pm;
        g.ld0 = ATT;
        HIPCHK(e, gemm_small(g, 1, st));
    }
    return TTS_HIP_OK;
}

static std::mutex& whole_gpu_mutex(int device) {
    static std::mutex m[PerDeviceOnce::kMaxDevices];
    return m[device >= 0 && device < PerDeviceOnce::kMaxDevices ? device : 0];
}

// ---------------------------------------------------------------------------------------------------------- decoder
// Autoregressive loop + postnet from an encoded batch.  Enqueued on e->stream; synchronizes it before returning (the
// loop's exit is data dependent and `steps_run` / the BiLSTM status are host values).
static int tacotron2_decode_impl(tts_hip_engine* e, const tts_hip_encoded* en, int max_len, int early_stop,
                                 const float* prenet_masks, int win_len, int win_offset, float* mel,
                                 float* decoder_output, float* stop_tokens, float* attention, int32_t* lengths,
                                 int32_t* steps_run, int mem, bool half_w, const uint64_t* mask_seed = nullptr) {
    Tacotron2Dev& tc = e->taco;
    if (!tc.ready) return set_err(e, TTS_HIP_ENOTREADY, "tacotron2 weights not finalized");
    if (!en || !en->buf.p || en->B <= 0 || max_len <= 0) return set_err(e, TTS_HIP_EINVAL, "tacotron2_decode: bad argument");
    if (en->enc != tc.enc_dim) return set_err(e, TTS_HIP_EINVAL, "tacotron2_decode: encoded batch belongs to other weights");
    if (mem != TTS_HIP_MEM_HOST && mem != TTS_HIP_MEM_DEVICE) return set_err(e, TTS_HIP_EINVAL, "bad mem kind %d", mem);
    if (half_w) {
        int rc16 = tacotron2_build_f16(e);
        if (rc16) return rc16;
    }
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = e->stream;
    const int B = en->B, Tin = en->Tin, enc = tc.enc_dim;
    const long long R = (long long)B * Tin;          // encoder rows
    const long long RD = (long long)B * max_len;     // decoder rows
    // Buffers the step kernels address are sized for max_len rounded up to a bucket, so that every call of a bucket has
    // the same workspace layout and can replay the same instantiated hipGraph (rows are indexed with the real max_len,
    // which the kernels read from the device-side loop state).
    const long long max_len_b = (max_len + 255) / 256 * 256;
    const long long RB = (long long)B * max_len_b;
    uint8_t* d_mask = en->mask;
    int* d_enc_len = en->enc_len;
    float* d_memory = en->memory;
    float* d_pm = en->pm;

    // ---------------- workspace arena
    // Which machine runs the loop (tts_hip_set_decoder_mode): 0 = the 7-kernel per-step graph only, 1 = persistent kernel when its
    // shape rule allows, 2 = fused two-kernel step when its shape rule allows, 3 (default) = persistent for 1 - 2 rows, fused
    // above, whichever is applicable otherwise.  The per-step graph is the fallback of both.
    bool try_persist = (tc.persist_mode == 1 || tc.persist_mode == 3) && persist_applicable(e, B, Tin);
    bool try_fused = (tc.persist_mode == 2 || tc.persist_mode == 3) && !e->timing && fused_applicable(e, B, Tin);
    if (try_persist && try_fused) {
        if (B <= 2) try_fused = false;
        else try_persist = false;
    }
    const size_t n_xch = std::max(try_persist ? persist_xch_u64(B, Tin) : (size_t)0, try_fused ? fused_xch_u64(B, Tin, enc) : (size_t)0);
    const bool with_masks = prenet_masks != nullptr || mask_seed != nullptr;
    const size_t n_masks = with_masks ? (size_t)RB * 2 * PRE : 1;
    const size_t conv_rows = (size_t)std::min<long long>(RD, 32768);     // 512 tiles x 64 rows at most
    size_t need = 0;
    auto sz = [&](size_t n, size_t el) { need = (need + 255) / 256 * 256 + n * el; };
    sz(n_masks, 4); sz(try_persist ? (size_t)R * PERSIST_NPM : 1, 4);
    sz(64, 4); sz(8, 4); sz(16, 4); sz(n_xch + 2, 8); sz(16, 4);
    sz(2 * B * ARNN, 4); sz(B * ARNN, 4); sz(2 * B * DRNN, 4); sz(B * DRNN, 4); sz(B * enc, 4); sz(B * PRE, 4); sz(B * ATT, 4);
    sz(B * NMEL, 4); sz(R, 4); sz(R, 4); sz(R, 4); sz(B, 4); sz(B, 4); sz(2 * B, 4);
    sz(RB * NMEL, 4); sz(RB, 4); sz(RB * Tin, 4); sz(RD, 1); sz(RD * NMEL, 4); sz(RD * 512, 4); sz(RD * 512, 4);
    sz(RD * NMEL, 4); sz(RD * NMEL, 4); sz(5 * conv_rows * 512, 4);
    need += 4096;
    {
        const void* before = tc.ws.p;
        HIPCHK(e, tc.ws.ensure(need));
        if (tc.ws.p != before) tacotron2_graphs_clear(e);
    }
    Arena A;
    A.base = (char*)tc.ws.p;
    A.cap = tc.ws.bytes;
    float* d_masks = A.take<float>(n_masks);
    float* d_pmfold = A.take<float>(try_persist ? (size_t)R * PERSIST_NPM : 1);
    DecState* d_state = A.take<DecState>(1);
    FusedState* d_fstate = A.take<FusedState>(1);
    int* d_freport = A.take<int>(16);
    unsigned long long* d_xch = A.take<unsigned long long>(n_xch + 2);
    int* d_pflags = A.take<int>(16);
    float* d_hatt = A.take<float>(2 * B * ARNN);
    float* d_catt = A.take<float>(B * ARNN);
    float* d_hdec = A.take<float>(2 * B * DRNN);
    float* d_cdec = A.take<float>(B * DRNN);
    float* d_ctx = A.take<float>(B * enc);
    float* d_p2 = A.take<float>(B * PRE);
    float* d_q = A.take<float>(B * ATT);
    float* d_frame = A.take<float>(B * NMEL);
    float* d_energy = A.take<float>(R);
    float* d_wprev = A.take<float>(R);
    float* d_wcum = A.take<float>(R);
    int* d_finished = A.take<int>(B);
    int* d_lengths = A.take<int>(B);
    int* d_mainatt = A.take<int>(2 * B);
    float* d_decout = A.take<float>(RB * NMEL);
    float* d_stop = A.take<float>(RB);
    float* d_attn = A.take<float>(RB * Tin);
    uint8_t* d_dmask = A.take<uint8_t>(RD);
    float* d_xm = A.take<float>(RD * NMEL);
    float* d_pa = A.take<float>(RD * 512);
    float* d_pb = A.take<float>(RD * 512);
    float* d_post = A.take<float>(RD * NMEL);
    float* d_mel = A.take<float>(RD * NMEL);
    float* d_convtmp = A.take<float>(5 * conv_rows * 512);
    const size_t convtmp_n = 5 * conv_rows * 512;
    if (A.off > A.cap) return set_err(e, TTS_HIP_ENOMEM, "tacotron2 workspace accounting error");
    // all recurrent state, loop state, exchange area, histories and outputs of the real rows start at zero
    const size_t zero_from = (char*)d_state - A.base, zero_mid = (char*)d_decout - A.base;
    auto zero_state = [&]() -> int {
        HIPCHK(e, hipMemsetAsync(A.base + zero_from, 0, zero_mid - zero_from, st));
        HIPCHK(e, hipMemsetAsync(d_decout, 0, (size_t)RD * NMEL * 4, st));
        HIPCHK(e, hipMemsetAsync(d_stop, 0, (size_t)RD * 4, st));
        HIPCHK(e, hipMemsetAsync(d_attn, 0, (size_t)RD * Tin * 4, st));
        return TTS_HIP_OK;
    };
    int rc;
    if ((rc = zero_state())) return rc;

    // prenet dropout masks always travel through the workspace: a captured graph must not hold a caller's pointer
    const hipMemcpyKind kin = mem == TTS_HIP_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const float* masks_dev = nullptr;
    if (prenet_masks) {
        HIPCHK(e, hipMemcpyAsync(d_masks, prenet_masks, (size_t)RD * 2 * PRE * 4, kin, st));
        masks_dev = d_masks;
    } else if (mask_seed) {                                 // drawn on the device, straight into the workspace
        if ((rc = philox_fill(e, d_masks, (long long)RD * 2 * PRE, mask_seed[0], mask_seed[1], TTS_HIP_RANDOM_PRENET_MASK, st))) return rc;
        masks_dev = d_masks;
    }

    // ---------------- decoder loop
    hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(1), 0, st, d_state, B, max_len, early_stop ? 1 : 0);
    HIPCHK(e, hipGetLastError());
    const size_t sm_lds = (size_t)Tin * sizeof(float);
    // Measurement hook (results are garbage when set): TTS_HIP_DEBUG_ONLY_KERNEL=k launches only step kernel k (0 prenet,
    // 1 attention LSTM, 2 query, 3 energies, 4 softmax_ctx, 5 decoder LSTM, 6 project) seven times per step, which gives
    // that kernel's cost inside the graph without the other six around it (scripts/run_taco.py prints the step time).
    // Only exists in a build made with -DTTS_DEBUG_HOOKS (csrc/build.sh never passes it).
#ifdef TTS_DEBUG_HOOKS
    const char* only_env = getenv("TTS_HIP_DEBUG_ONLY_KERNEL");
    const int only = only_env ? atoi(only_env) : -1;
#else
    constexpr int only = -1;
#endif
    auto enqueue_step = [&](int j) -> int {
        const int par = j & 1;                       // CHUNK is even, so the parity of t equals the parity of j
        float* hatt_old = d_hatt + (size_t)par * B * ARNN;
        float* hatt_new = d_hatt + (size_t)(par ^ 1) * B * ARNN;
        float* hdec_old = d_hdec + (size_t)par * B * DRNN;
        float* hdec_new = d_hdec + (size_t)(par ^ 1) * B * DRNN;
        timing_begin(e, 2);
        auto k_prenet = [&]() -> int {
            hipLaunchKernelGGL(prenet_kernel, dim3(B, 8), dim3(256), 0, st, d_state, j, d_frame, tc.prenet_w0,
                               tc.prenet_w1, masks_dev, d_p2);
            HIPCHK(e, hipGetLastError());
            return TTS_HIP_OK;
        };
        auto k_att = [&]() -> int {
            HIPCHK(e, lstm_dispatch(st, d_state, j, tc.att, d_p2, PRE, d_ctx, enc, hatt_old, hatt_new, d_catt, B, half_w));
            return TTS_HIP_OK;
        };
        auto k_query = [&]() -> int {
            hipLaunchKernelGGL(query_kernel, dim3(ATT / 2), dim3(256), 0, st, d_state, j, hatt_new, tc.query_w, d_q, B);
            HIPCHK(e, hipGetLastError());
            return TTS_HIP_OK;
        };
        auto k_energies = [&]() -> int {
            hipLaunchKernelGGL(energies_kernel, dim3(B, (Tin + EPB - 1) / EPB), dim3(256), 0, st, d_state, j, d_q, tc.loc_dense,
                               tc.value_w, d_pm, d_wprev, d_wcum, d_energy, Tin);
            HIPCHK(e, hipGetLastError());
            return TTS_HIP_OK;
        };
        auto k_softmax = [&]() -> int {
            hipLaunchKernelGGL(softmax_ctx_kernel, dim3(B, enc / 32), dim3(256), sm_lds, st, d_state, j, d_energy, d_mask,
                               d_enc_len, win_len, win_offset, d_mainatt + par * B, d_mainatt + (par ^ 1) * B, d_memory,
                               d_wprev, d_wcum, d_ctx, d_attn, Tin, enc);
            HIPCHK(e, hipGetLastError());
            return TTS_HIP_OK;
        };
        auto k_dec = [&]() -> int {
            HIPCHK(e, lstm_dispatch(st, d_state, j, tc.dec, hatt_new, ARNN, d_ctx, enc, hdec_old, hdec_new, d_cdec, B, half_w));
            return TTS_HIP_OK;
        };
        auto k_project = [&]() -> int {
            if (enc == 512)
                hipLaunchKernelGGL(project_kernel<6>, dim3(21), dim3(256), 0, st, d_state, j, hdec_new, d_ctx, tc.proj_w,
                                   tc.proj_b, d_frame, d_decout, d_stop, d_finished, d_lengths, B);
            else
                hipLaunchKernelGGL(project_kernel<7>, dim3(21), dim3(256), 0, st, d_state, j, hdec_new, d_ctx, tc.proj_w,
                                   tc.proj_b, d_frame, d_decout, d_stop, d_finished, d_lengths, B);
            HIPCHK(e, hipGetLastError());
            return TTS_HIP_OK;
        };
        int rcs = TTS_HIP_OK;
        for (int k = 0; k < 7 && rcs == TTS_HIP_OK; ++k) {
            switch (only >= 0 ? only : k) {
                case 0: rcs = k_prenet(); break;
                case 1: rcs = k_att(); break;
                case 2: rcs = k_query(); break;
                case 3: rcs = k_energies(); break;
                case 4: rcs = k_softmax(); break;
                case 5: rcs = k_dec(); break;
                default: rcs = k_project(); break;
            }
        }
        if (rcs) return rcs;
        timing_end(e);
        return TTS_HIP_OK;
    };
    int host_steps = 0;
    bool persisted = false, fused = false;
    // the workspace layout (and with it every pointer a captured graph holds) depends on which exchange areas were sized in
    const int layout_id = (try_persist ? 1 : 0) | (try_fused ? 2 : 0);
    // instantiated chunk graphs are kept and replayed by every later call with the same shape bucket: the workspace and the
    // encoded batch are stable allocations, so the kernel nodes hold valid pointers
    auto cached_graph = [&](const DecGraphKey& key, auto&& enqueue, hipGraphExec_t* out) -> int {
        auto it = tc.graphs.find(key);
        if (it != tc.graphs.end()) {
            *out = it->second;
            return TTS_HIP_OK;
        }
        hipGraph_t graph = nullptr;
        hipGraphExec_t gexec = nullptr;
        HIPCHK(e, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int crc = enqueue();
        hipError_t ce = hipStreamEndCapture(st, &graph);
        if (crc != TTS_HIP_OK || ce != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            if (crc != TTS_HIP_OK) return crc;
            HIPCHK(e, ce);
        }
        hipError_t ie = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);             // the executable graph is self-contained
        HIPCHK(e, ie);
        constexpr size_t kMaxGraphs = 16;
        if (tc.graph_order.size() >= kMaxGraphs) {
            auto old = tc.graphs.find(tc.graph_order.front());
            if (old != tc.graphs.end()) {
                (void)hipGraphExecDestroy(old->second);
                tc.graphs.erase(old);
            }
            tc.graph_order.erase(tc.graph_order.begin());
        }
        tc.graphs[key] = gexec;
        tc.graph_order.push_back(key);
        *out = gexec;
        return TTS_HIP_OK;
    };
    int bl_err = 0;
    bool bl_checked = false;
    bool try_fused_now = try_fused;
    if (try_persist) {
        // Persistent weight-stationary loop (taco_persist.hip): the attention context is folded through the four linear maps
        // that consume it -- PM = memory x [W_att[:, ctx] | W_dec[:, ctx] | F[:, ctx] | P[:, ctx]] -- once per utterance.
        struct Part { const float* Bt; long long ldb; int N, col; };
        const Part parts[4] = {{tc.att.W + PRE, (long long)PRE + enc + ARNN, 4 * ARNN, PERSIST_COL_ATT},
                               {tc.dec.W + ARNN, (long long)ARNN + enc + DRNN, 4 * DRNN, PERSIST_COL_DEC},
                               {tc.pfold_w + DRNN, (long long)DRNN + enc, PRE, PERSIST_COL_F},
                               {tc.proj_w + DRNN, (long long)DRNN + enc, NMEL + 1, PERSIST_COL_P}};
        for (const Part& p : parts) {
            GemmArgs g{};
            g.M = (int)R;
            g.N = p.N;
            g.L = (int)R;
            g.nseg = 1;
            g.seg[0] = ASeg{d_memory, enc, 0, enc, enc};
            g.Bt = p.Bt;
            g.ldb = p.ldb;
            g.mode = EPI_LINEAR;
            g.split = p.N;
            g.out0 = d_pmfold + p.col;
            g.ld0 = PERSIST_NPM;
            HIPCHK(e, gemm_small(g, 1, st));
        }
        PersistCall pc{};
        pc.B = B; pc.Tin = Tin; pc.max_len = max_len; pc.early_stop = early_stop ? 1 : 0;
        pc.win_len = win_len; pc.win_off = win_offset; pc.half_w = half_w;
        pc.memory = d_memory; pc.pm = d_pm; pc.mask = d_mask; pc.enc_len = d_enc_len; pc.masks = masks_dev;
        pc.pm_fold = d_pmfold; pc.xch = d_xch; pc.flags = d_pflags;
        pc.dec_out = d_decout; pc.stop_out = d_stop; pc.attn_hist = d_attn; pc.lengths = d_lengths; pc.finished = d_finished;
        // The persistent kernel and the fused step each need every CU of the device at once: two of them from two handles of
        // one process would hold half the CUs each and wait for the other half (until their bounded waits give up).  One at a
        // time per device; the work is the same, and nothing deadlocks.  (Another process is handled by the timeouts.)
        std::unique_lock<std::mutex> whole_gpu(whole_gpu_mutex(e->device));
        const int prc = persist_decode(e, st, pc, &host_steps);
        whole_gpu.unlock();
        if (prc < 0) return prc;
        persisted = prc == TTS_HIP_OK;                  // 1: the grid could not become resident -> per-step graph below
        if (prc == 2) {                                 // gave up in mid-loop: start over on the per-step graph
            if ((rc = zero_state())) return rc;
            hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(1), 0, st, d_state, B, max_len, early_stop ? 1 : 0);
            HIPCHK(e, hipGetLastError());
        }
    }
    if (try_fused && tc.fused_backoff > 0) {
        --tc.fused_backoff;                              // a recent call timed out (the GPU is shared): not this time
        try_fused_now = false;
    }
    if (try_fused_now) {
        std::unique_lock<std::mutex> whole_gpu(whole_gpu_mutex(e->device));      // see the persistent section above
        // Fused two-kernel step (taco_fused.hip): chunks of FUSED_CHUNK steps, one hipGraph per shape bucket; after each chunk
        // the host reads the loop state (32 bytes) and the abort flag.
        FusedCall fc{};
        fc.B = B; fc.Tin = Tin; fc.max_len = max_len; fc.early_stop = early_stop ? 1 : 0;
        fc.win_len = win_len; fc.win_off = win_offset; fc.half_w = half_w;
        fc.memory = d_memory; fc.pm = d_pm; fc.mask = d_mask; fc.enc_len = d_enc_len; fc.masks = masks_dev;
        fc.xch = d_xch; fc.flags = d_pflags; fc.state = d_fstate; fc.bl_err = en->bl_err; fc.report = d_freport;
        fc.hatt = d_hatt; fc.hdec = d_hdec; fc.catt = d_catt; fc.cdec = d_cdec; fc.ctx = d_ctx;
        fc.wprev = d_wprev; fc.wcum = d_wcum; fc.mainatt = d_mainatt;
        fc.dec_out = d_decout; fc.stop_out = d_stop; fc.attn_hist = d_attn; fc.lengths = d_lengths; fc.finished = d_finished;
        if ((rc = fused_init(e, st, fc))) return rc;
        hipGraphExec_t gexec = nullptr;
#ifdef TTS_DEBUG_HOOKS
        const bool fgraph = getenv("TTS_HIP_NO_GRAPH") == nullptr;
        const char* trace_file = getenv("TTS_FUSED_TRACE_FILE");       // phase timestamps of the first 128 steps (scripts/fused_trace.py)
        const size_t trace_n = (size_t)128 * 2 * 4 * 16;
        const bool own_graph = trace_file || getenv("TTS_FUSED_DELAYS");   // (the delays are kernel arguments of the cached graph)
        if (own_graph) {
            if (trace_file) {
                HIPCHK(e, hipMalloc((void**)&fc.trace, trace_n * sizeof(long long)));
                HIPCHK(e, hipMemsetAsync(fc.trace, 0, trace_n * sizeof(long long), st));
            }
            hipGraph_t graph = nullptr;                                 // a graph of its own: the trace pointer / delays are baked in
            HIPCHK(e, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int crc = fused_enqueue_chunk(e, st, fc);
            HIPCHK(e, hipStreamEndCapture(st, &graph));
            if (crc) return crc;
            HIPCHK(e, hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
        }
#else
        constexpr bool fgraph = true;
        constexpr bool own_graph = false;
#endif
        if (fgraph && !own_graph) {
            const DecGraphKey key{tc.ws.p, en->buf.p, B, Tin, (int)max_len_b, with_masks ? 1 : 0, win_len, win_offset,
                                  half_w ? 1 : 0, layout_id | 4};
            if ((rc = cached_graph(key, [&]() { return fused_enqueue_chunk(e, st, fc); }, &gexec))) return rc;
        }
        // Chunk k + 1 is enqueued BEFORE the host looks at chunk k's loop state, so the GPU never waits for the host between
        // chunks (a sync + relaunch per 32 steps cost ~1.2 us per step); the state travels through a pinned ring.  When chunk k
        // turns out to have ended the loop, chunk k + 1 is 66 kernels that return at once (~0.1 ms, once per call).
        struct ChunkReport { FusedState st; int abort_code; int bl_err; int pad[6]; };
        static_assert(sizeof(ChunkReport) == 64, "one report per 64-byte slot");
        if (!tc.pinned) {
            HIPCHK(e, hipHostMalloc(&tc.pinned, 2 * sizeof(ChunkReport), hipHostMallocDefault));
            for (auto& ev : tc.chunk_ev) HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        ChunkReport* rep = (ChunkReport*)tc.pinned;
        fused = true;
        const int n_chunks = (max_len + FUSED_CHUNK - 1) / FUSED_CHUNK;
        auto enqueue = [&](int k) -> int {
            if (fgraph) HIPCHK(e, hipGraphLaunch(gexec, st));
            else if (int rc2 = fused_enqueue_chunk(e, st, fc)) return rc2;
            HIPCHK(e, hipMemcpyAsync(rep + (k & 1), d_freport, sizeof(ChunkReport), hipMemcpyDeviceToHost, st));
            HIPCHK(e, hipEventRecord(tc.chunk_ev[k & 1], st));
            return TTS_HIP_OK;
        };
        if ((rc = enqueue(0))) return rc;
        for (int k = 0; k < n_chunks; ++k) {
            if (k + 1 < n_chunks && (rc = enqueue(k + 1))) return rc;
            HIPCHK(e, hipEventSynchronize(tc.chunk_ev[k & 1]));
            const ChunkReport h = rep[k & 1];
            bl_checked = true;
            bl_err = h.bl_err;
            bool stop = false;
            if (bl_err) {
                stop = true;
            } else if (h.abort_code != 0) {             // an exchange timed out (e.g. the GPU was shared and a block lost its CU):
                fused = false;                          // start over on the per-step graph
                stop = true;
            } else {
                host_steps = h.st.steps_run;
                if (h.st.steps_run < std::min((k + 1) * FUSED_CHUNK, max_len)) stop = true;      // the loop ended inside this chunk
                if (early_stop && h.st.n_fin >= B) stop = true;                                  // ... or with the stop tokens of its last step
            }
            if (stop) {
                if (k + 1 < n_chunks) HIPCHK(e, hipEventSynchronize(tc.chunk_ev[(k + 1) & 1]));     // the chunk already in flight (it does nothing)
                break;
            }
        }
        if (bl_err) return set_err(e, TTS_HIP_EHIP, "tacotron2 encoder: BiLSTM block exchange timed out");
        if (fused) {
            tc.fused_fail_streak = 0;
        } else {
            // back off exponentially (1, 2, 4 ... 64 calls on the per-step graph) while another tenant keeps the CUs busy
            tc.fused_fail_streak = std::min(tc.fused_fail_streak + 1, 7);
            tc.fused_backoff = 1 << (tc.fused_fail_streak - 1);
            set_err(e, TTS_HIP_EHIP, "tacotron2 fused decoder: an exchange timed out; fell back to the per-step graph");
            if ((rc = zero_state())) return rc;
            hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(1), 0, st, d_state, B, max_len, early_stop ? 1 : 0);
            HIPCHK(e, hipGetLastError());
        }
#ifdef TTS_DEBUG_HOOKS
        if (own_graph && !fc.trace) (void)hipGraphExecDestroy(gexec);
        if (fc.trace) {
            std::vector<long long> ht(trace_n);
            (void)hipMemcpy(ht.data(), fc.trace, trace_n * sizeof(long long), hipMemcpyDeviceToHost);
            (void)hipFree(fc.trace);
            (void)hipGraphExecDestroy(gexec);
            if (FILE* f = fopen(trace_file, "wb")) {
                fwrite(ht.data(), sizeof(long long), trace_n, f);
                fclose(f);
            }
        }
#endif
    }
    tc.last_path = persisted ? 1 : fused ? 2 : 0;
    if (!persisted && !fused) {
        // chunks of CHUNK steps; after each chunk the host reads the loop state (one 32-byte copy)
        DecState h{};
#ifdef TTS_DEBUG_HOOKS
        const bool use_graph = !e->timing && getenv("TTS_HIP_NO_GRAPH") == nullptr;
#else
        const bool use_graph = !e->timing;           // per-step HIP events (tts_hip_kernel_timing) cannot be captured
#endif
        hipGraphExec_t gexec = nullptr;
        if (use_graph) {
            const DecGraphKey key{tc.ws.p, en->buf.p, B, Tin, (int)max_len_b, with_masks ? 1 : 0, win_len, win_offset,
                                  half_w ? 1 : 0, layout_id};
            auto enqueue = [&]() -> int {
                int crc = TTS_HIP_OK;
                for (int j = 0; j < CHUNK && crc == TTS_HIP_OK; ++j) crc = enqueue_step(j);
                if (crc == TTS_HIP_OK) hipLaunchKernelGGL(advance_chunk_kernel, dim3(1), dim3(1), 0, st, d_state);
                return crc;
            };
            if ((rc = cached_graph(key, enqueue, &gexec))) return rc;
        }
        for (int t0 = 0; t0 < max_len; t0 += CHUNK) {
            if (use_graph) {
                HIPCHK(e, hipGraphLaunch(gexec, st));
            } else {
                for (int j = 0; j < CHUNK; ++j)
                    if ((rc = enqueue_step(j))) return rc;
                hipLaunchKernelGGL(advance_chunk_kernel, dim3(1), dim3(1), 0, st, d_state);
            }
            HIPCHK(e, hipMemcpyAsync(&h, d_state, sizeof h, hipMemcpyDeviceToHost, st));
            if (t0 == 0) HIPCHK(e, hipMemcpyAsync(&bl_err, en->bl_err, sizeof bl_err, hipMemcpyDeviceToHost, st));
            HIPCHK(e, hipStreamSynchronize(st));
            bl_checked = true;
            if (bl_err) return set_err(e, TTS_HIP_EHIP, "tacotron2 encoder: BiLSTM block exchange timed out");
            host_steps = h.steps_run;
            if (early_stop && h.n_fin[0] >= B) break;      // the last step of a chunk (odd j) wrote slot 0
        }
    }

    // ---------------- postnet + residual
    {
        const long long n = RD * NMEL;
        hipLaunchKernelGGL(dec_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_lengths, d_decout,
                           d_dmask, d_xm, max_len, RD);
        HIPCHK(e, hipGetLastError());
        if ((rc = conv_gemm(e, tc.post_conv[0], d_xm, NMEL, d_pa, (int)RD, max_len, d_dmask, ACT_TANH, 1, d_convtmp, convtmp_n))) return rc;
        if ((rc = conv_gemm(e, tc.post_conv[1], d_pa, 512, d_pb, (int)RD, max_len, d_dmask, ACT_TANH, 1, d_convtmp, convtmp_n))) return rc;
        if ((rc = conv_gemm(e, tc.post_conv[2], d_pb, 512, d_pa, (int)RD, max_len, d_dmask, ACT_TANH, 1, d_convtmp, convtmp_n))) return rc;
        if ((rc = conv_gemm(e, tc.post_conv[3], d_pa, 512, d_pb, (int)RD, max_len, d_dmask, ACT_TANH, 1, d_convtmp, convtmp_n))) return rc;
        if ((rc = conv_gemm(e, tc.post_conv[4], d_pb, 512, d_post, (int)RD, max_len, d_dmask, ACT_NONE, 0, d_convtmp, convtmp_n))) return rc;
        hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_decout, d_post, d_mel, n);
        HIPCHK(e, hipGetLastError());
    }

    // ---------------- outputs
    const hipMemcpyKind kout = mem == TTS_HIP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (mel) HIPCHK(e, hipMemcpyAsync(mel, d_mel, RD * NMEL * 4, kout, st));
    if (decoder_output) HIPCHK(e, hipMemcpyAsync(decoder_output, d_decout, RD * NMEL * 4, kout, st));
    if (stop_tokens) HIPCHK(e, hipMemcpyAsync(stop_tokens, d_stop, RD * 4, kout, st));
    if (attention) HIPCHK(e, hipMemcpyAsync(attention, d_attn, RD * Tin * 4, kout, st));
    if (lengths) HIPCHK(e, hipMemcpyAsync(lengths, d_lengths, (size_t)B * 4, kout, st));
    if (!bl_checked) HIPCHK(e, hipMemcpyAsync(&bl_err, en->bl_err, sizeof bl_err, hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    if (bl_err) return set_err(e, TTS_HIP_EHIP, "tacotron2 encoder: BiLSTM block exchange timed out");
    if (steps_run) *steps_run = host_steps;
    return TTS_HIP_OK;
}

static int tacotron2_infer_impl(tts_hip_engine* e, const int32_t* tokens, int B, int Tin, const float* speaker,
                                int max_len, int early_stop, const float* prenet_masks, int win_len,
                                int win_offset, float* mel, float* decoder_output, float* stop_tokens,
                                float* attention, int32_t* lengths, int32_t* steps_run, int mem, bool half_w) {
    if (!e) return TTS_HIP_EINVAL;
    if (max_len <= 0) return set_err(e, TTS_HIP_EINVAL, "tacotron2_infer: bad argument");
    Tacotron2Dev& tc = e->taco;
    if (!tc.enc_cache) tc.enc_cache = new (std::nothrow) tts_hip_encoded();
    if (!tc.enc_cache) return set_err(e, TTS_HIP_ENOMEM, "out of host memory");
    int rc = tacotron2_encode_impl(e, tokens, B, Tin, speaker, mem, tc.enc_cache);
    if (rc) return rc;
    return tacotron2_decode_impl(e, tc.enc_cache, max_len, early_stop, prenet_masks, win_len, win_offset, mel,
                                 decoder_output, stop_tokens, attention, lengths, steps_run, mem, half_w);
}

extern "C" int tts_hip_tacotron2_infer(tts_hip_engine* e, const int32_t* tokens, int B, int Tin, const float* speaker,
                                       int max_len, int early_stop, const float* prenet_masks, int win_len,
                                       int win_offset, float* mel, float* decoder_output, float* stop_tokens,
                                       float* attention, int32_t* lengths, int32_t* steps_run, int mem) {
    return tacotron2_infer_impl(e, tokens, B, Tin, speaker, max_len, early_stop, prenet_masks, win_len, win_offset, mel,
                                decoder_output, stop_tokens, attention, lengths, steps_run, mem, false);
}

extern "C" int tts_hip_tacotron2_infer_f16(tts_hip_engine* e, const int32_t* tokens, int B, int Tin,
                                           const float* speaker, int max_len, int early_stop,
                                           const float* prenet_masks, int win_len, int win_offset, float* mel,
                                           float* decoder_output, float* stop_tokens, float* attention,
                                           int32_t* lengths, int32_t* steps_run, int mem) {
    return tacotron2_infer_impl(e, tokens, B, Tin, speaker, max_len, early_stop, prenet_masks, win_len, win_offset, mel,
                                decoder_output, stop_tokens, attention, lengths, steps_run, mem, true);
}

extern "C" int tts_hip_tacotron2_encode(tts_hip_engine* e, const int32_t* tokens, int B, int Tin, const float* speaker,
                                        int mem, void* stream, tts_hip_encoded** out) {
    if (!e || !out) return TTS_HIP_EINVAL;
    *out = nullptr;
    tts_hip_encoded* en = new (std::nothrow) tts_hip_encoded();
    if (!en) return set_err(e, TTS_HIP_ENOMEM, "out of host memory");
    StreamScope scope(e, stream);
    const int rc = tacotron2_encode_impl(e, tokens, B, Tin, speaker, mem, en);
    if (rc) {
        en->buf.release();
        delete en;
        return rc;
    }
    *out = en;
    return TTS_HIP_OK;
}

extern "C" int tts_hip_tacotron2_decode(tts_hip_engine* e, const tts_hip_encoded* encoded, int max_len, int early_stop,
                                        const float* prenet_masks, int win_len, int win_offset, int precision, float* mel,
                                        float* decoder_output, float* stop_tokens, float* attention, int32_t* lengths,
                                        int32_t* steps_run, int mem, void* stream) {
    if (!e) return TTS_HIP_EINVAL;
    if (precision != 0 && precision != 1) return set_err(e, TTS_HIP_EINVAL, "tacotron2_decode: precision must be 0 (f32) or 1 (f16 weights)");
    StreamScope scope(e, stream);
    return tacotron2_decode_impl(e, encoded, max_len, early_stop, prenet_masks, win_len, win_offset, mel, decoder_output,
                                 stop_tokens, attention, lengths, steps_run, mem, precision == 1);
}

extern "C" int tts_hip_tacotron2_decode_seeded(tts_hip_engine* e, const tts_hip_encoded* encoded, int max_len, int early_stop,
                                               uint64_t seed, uint64_t offset, int win_len, int win_offset, int precision,
                                               float* mel, float* decoder_output, float* stop_tokens, float* attention,
                                               int32_t* lengths, int32_t* steps_run, int mem, void* stream) {
    if (!e) return TTS_HIP_EINVAL;
    if (precision != 0 && precision != 1) return set_err(e, TTS_HIP_EINVAL, "tacotron2_decode_seeded: precision must be 0 (f32) or 1 (f16 weights)");
    StreamScope scope(e, stream);
    const uint64_t ms[2] = {seed, offset};
    return tacotron2_decode_impl(e, encoded, max_len, early_stop, nullptr, win_len, win_offset, mel, decoder_output, stop_tokens,
                                 attention, lengths, steps_run, mem, precision == 1, ms);
}

// cached decoder graphs whose kernel nodes point into `buf` (the graphs of other encoded batches stay valid)
static void tacotron2_graphs_drop(tts_hip_engine* e, const void* buf) {
    auto& tc = e->taco;
    for (auto it = tc.graphs.begin(); it != tc.graphs.end();) {
        if (it->first.enc_buf == buf) {
            (void)hipGraphExecDestroy(it->second);
            for (auto o = tc.graph_order.begin(); o != tc.graph_order.end();)
                o = (!(*o < it->first) && !(it->first < *o)) ? tc.graph_order.erase(o) : std::next(o);
            it = tc.graphs.erase(it);
        } else {
            ++it;
        }
    }
}

extern "C" int tts_hip_tacotron2_reencode(tts_hip_engine* e, tts_hip_encoded* encoded, const int32_t* tokens, int B, int Tin,
                                          const float* speaker, int mem, void* stream) {
    if (!e || !encoded) return TTS_HIP_EINVAL;
    StreamScope scope(e, stream);
    const void* before = encoded->buf.p;
    // (the encoder itself drops every cached graph if the buffer has to grow; a buffer that stays keeps its graphs: the next
    // sentence of the same shape bucket replays them)
    const int rc = tacotron2_encode_impl(e, tokens, B, Tin, speaker, mem, encoded);
    (void)before;
    return rc;
}

extern "C" int tts_hip_encoded_free(tts_hip_engine* e, tts_hip_encoded* encoded) {
    if (!e || !encoded) return TTS_HIP_EINVAL;
    (void)hipSetDevice(e->device);
    tacotron2_graphs_drop(e, encoded->buf.p);          // only the cached graphs that point into this buffer
    encoded->buf.release();                            // hipFree waits for work that still uses it
    delete encoded;
    return TTS_HIP_OK;
}
