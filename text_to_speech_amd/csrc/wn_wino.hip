// wn_wino.hip -- Winograd minimal filtering F(4,3) along the tap axis of the WN dilated convolution (fp32 path, calls of 144
// frames or more).
//
// The in-layer pre-activation of WaveGlow's WN (/root/reference/architectures/waveglow_arch.py:117-127) is a k = 3 dilated
// convolution plus the conditioning term,  y[l] = W- x[l - d] + W0 x[l] + W+ x[l + d] + c[l] + b.  The FOUR outputs
// y[l], y[l + d], y[l + 2d], y[l + 3d] share the six inputs x[l - d] ... x[l + 4d] and need only SIX K = 512 products
// instead of twelve (K per output 768 instead of 1536) on M / 4 "group rows":
//
//   input transform    U_p = sum_i BT[p][i] x_i                      (six combinations of the six inputs)
//   products           P_p = U_p G_p^T + melP_p V_p^T                (G: tap combinations, built at load)
//   output transform   y_j = sum_p AT[j][p] P_p + b,  acts = tanh(.) * sigmoid(.)  written to the four output rows
//
// Round 4: ONE kernel per layer (wino4_fused2_kernel, the default form).  A block owns 64 group rows x 128 pre-activation
// columns of ALL SIX products (a wave: 32 x 64 x 6 = 192 accumulator registers).  Its K loop walks the 512 tap columns chunk by
// chunk: the six INPUT tiles of a chunk arrive by LDS-DMA with per-lane row addresses (phase carries, frame groups, zeros
// outside an utterance through out-of-range offsets), a product's A fragment is three or four input fragments combined in
// registers under the other half step's MFMAs, and the epilogue applies the output transform, bias and gate in registers:
// no U planes, no P planes, no pre-pass or combine launch.  Measured at config 2 on one box: three passes (round 3) 433 ms
// per step, fused GEMM behind the pre-pass 415, this 401 (direct form 565); the kernel runs the 456 GFLOP a layer executes in
// 3.62 ms (80 % of the fp32 MFMA peak INCLUDING both transforms and the gate; the per-product GEMM of round 3 alone ran at
// 85 %, its layer -- 0.19 + 3.55 + 0.27 ms -- at 74 %).  What bounds the tile: six accumulator sets leave room for 64 x 128
// per four waves at two blocks per CU (8-wave 128 x 128 blocks measured 3 % slower: one barrier domain per CU and 12.5
// rounds of 256 blocks), i.e. 12 DMA pieces per 16 MFMAs and wave -- twice the direct kernel's bytes per MFMA.
// Forms 2 and 3 (tts_hip_set_waveglow_form; measurement only, bit-identical results) keep the earlier stages: the three
// passes (wino4_prepass_kernel, one z slice of gemm_f32_kernel per product, wino4_combine_kernel) and the fused GEMM behind
// the pre-pass (wino4_fused_kernel).
//
// The conditioning term (K = 320 per output) is spread over the products so that none idles: three K slices A = [0, 112),
// B = [112, 208), C = [208, 320), each carried by a product subset whose columns of the output transform AT have rank 4 --
// {0, 1, 2, 5}, {0, 3, 4, 5}, {1, 2, 3, 4} -- and combined with the inverse of those columns: K = 512 + 208 for products
// 0, 3, 4, 5 and 512 + 224 for products 1, 2 (the fused kernels skip the all-zero padding chunk of the 208-column products).
//
// Groups.  Dilation d <= 8 (sample groups): four PHASES p0 + j d of one frame, 8 group phases p0 = (gp / d) 4d + gp % d; the
// outputs share their mel rows and differ in the per-phase conditioning weights, so the slice combinations are WEIGHT
// combinations built at load.  d >= 32 (s = d / 32 frames): four FRAMES t0 + j s of one phase; the outputs share the weights
// -- the products' weights are chunk ranges of cond_Bt itself, no copies (round 3 kept six column-selected copies per phase:
// 6.3 GB) -- and the combinations are MEL combinations built once per call.  d = 16: two phases x two frames, sharing neither
// -- but every row of subset {1, 2, 3, 4}'s coefficient matrix is an outer product (over the two frames) x (over the two
// phases), so those four products carry the WHOLE conditioning as one mel combination times one weight combination each
// (K = 512 + 320) and products 0 and 5 run K = 512: the same K per output.  Frame groups are cut per utterance, so any
// utterance length works.
//
// Numerics: every operand stays fp32, weight / mel combinations are formed in fp64 and rounded once.  F(4,3)'s transform
// constants (4, 5, 8, 1/6, 1/24) cost accuracy: one layer's gated activations against the oracle 1.9e-6 relative RMS (direct
// form 1.4e-6; tests/test_waveglow_gpu.py), end to end 6.0e-7 waveform RMS (direct form 4.96e-7; tolerance 1e-4), with four
// times less `end` attenuation 6.5e-6 (4.9e-6).  Not bit-identical to the direct form.
#include "engine.h"
#include "gemm_f32.h"

#include <algorithm>

using namespace ttsgemm;

namespace {
constexpr int C = 512;
constexpr int NPH = 32;
constexpr int KMEL = 320;
constexpr int KCONV = 3 * C;

// x row of (phase ps -- may leave [0, 32): carried into the neighbouring frame -- , frame row f + df), zero outside the utterance
__device__ __forceinline__ f32x4 x_at(const float* __restrict__ x, int ps, long long f, int df, int c, int PR, int BT, int T) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (f >= BT) return zero;
    const int carry = (ps >> 5) + df;                  // arithmetic shift: floor
    const int t = (int)(f % T) + carry;
    if (t < 0 || t >= T) return zero;
    return *reinterpret_cast<const f32x4*>(x + ((long long)(ps & 31) * PR + f + carry) * C + c);
}

// mel window of frame f: [mel_t | mel_{t-1} | mel_{t-2} | mel_{t-3}][80], zeros before the start of the utterance; column k
__device__ __forceinline__ float melwin(const float* __restrict__ mel, long long f, int k, int BT, int T) {
    if (f >= BT) return 0.f;
    const int q = k / 80, j = k % 80;
    return (int)(f % T) - q >= 0 ? mel[(f - q) * 80 + j] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The transforms (Lavin & Gray's F(4,3); fp32 error ~3x the direct form's, far inside the tolerance), with the six inputs
// x_i = x[l + (i - 1) d] of the four outputs l, l + d, l + 2d, l + 3d:
//   U0 = 4 x0 - 5 x2 + x4            G0 = W- / 4                          y0 = P0 + P1 + P2 + P3 + P4
//   U1 = -4 x1 - 4 x2 + x3 + x4      G1 = -(W- + W0 + W+) / 6             y1 = P1 - P2 + 2 P3 - 2 P4
//   U2 = 4 x1 - 4 x2 - x3 + x4       G2 = -(W- - W0 + W+) / 6             y2 = P1 + P2 + 4 P3 + 4 P4
//   U3 = -2 x1 - x2 + 2 x3 + x4      G3 = W- / 24 + W0 / 12 + W+ / 6      y3 = P1 - P2 + 8 P3 - 8 P4 + P5
//   U4 = 2 x1 - x2 - 2 x3 + x4       G4 = W- / 24 - W0 / 12 + W+ / 6
//   U5 = 4 x1 - 5 x3 + x5            G5 = W+
// Phase and frame groups: the conditioning is cut into three K slices, each carried by a product subset whose columns of the
// output transform have rank 4 -- A = [0, 112) on {0, 1, 2, 5}, B = [112, 208) on {0, 3, 4, 5}, C = [208, 320) on {1, 2, 3, 4} --
// and combined by the inverse of those columns (W4_A / W4_B / W4_C below), so that every product runs K = 512 + 224 (208 padded
// to 224 for products 0, 3, 4, 5).  The bias is added in the combine pass.
constexpr int K4 = 224, SA = 112, SB = 96, SC = 112;       // conditioning K of a product; slice widths (A, B, C)

__device__ __forceinline__ int group_phase0(int gp, int d) { return (gp / d) * 4 * d + gp % d; }

// Frame groups (dilations >= 32, s = d / 32): every utterance owns G = 4 ceil(T / 16) group rows per phase for every s, so
// that no group straddles two utterances whatever T is; group g of an utterance starts at frame t0 = (g / s) 4s + g % s and
// covers t0 + j s (frames >= T: inputs read as zero, outputs are not written; t0 >= T: an empty group).
__host__ __device__ __forceinline__ int frame_groups_per_utt(int T) { return (T + 15) / 16 * 4; }
__device__ __forceinline__ bool frame_group(long long gf, int s, int BT, int T, int& b, int& t0) {
    const int G = frame_groups_per_utt(T);
    b = (int)(gf / G);
    const int g = (int)(gf % G);
    t0 = (g / s) * 4 * s + g % s;
    return (long long)b * T < BT && t0 < T;
}
// Dilation 16: the four outputs l + 16 j of a group are two phases x two frames -- (p0, t), (p0 + 16, t), (p0, t + 1),
// (p0 + 16, t + 1) for p0 < 16 and even t; every utterance owns ceil(T / 2) group rows per p0.
__host__ __device__ __forceinline__ int mixed_groups_per_utt(int T) { return (T + 1) / 2; }
__device__ __forceinline__ bool mixed_group(long long gm, int BT, int T, int& b, int& t0) {
    const int G = mixed_groups_per_utt(T);
    b = (int)(gm / G);
    t0 = 2 * (int)(gm % G);
    return (long long)b * T < BT;                          // (t0 < T always)
}
// x row of (phase p, utterance b, frame t), zero outside the utterance
__device__ __forceinline__ f32x4 x_bt(const float* __restrict__ x, int p, int b, int t, int c, int PR, int T) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (t < 0 || t >= T) return zero;
    return *reinterpret_cast<const f32x4*>(x + ((long long)p * PR + (long long)b * T + t) * C + c);
}

__global__ void wino4_prepass_kernel(const float* __restrict__ x, float* __restrict__ U, int d, int PR, int BT, int T, long long Mq) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Mq * (C / 4)) return;
    const long long mg = idx / (C / 4);
    const int c = (int)(idx % (C / 4)) * 4;
    f32x4 v[6];
    if (d == 16) {                                         // two phases x two frames (PRm group rows per p0)
        const int PRm = (int)(Mq / 16);
        const int p0 = (int)(mg / PRm);
        int b, t0;
        const bool ok = mixed_group(mg % PRm, BT, T, b, t0);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int ps = p0 + 16 * (i - 1);
            v[i] = ok ? x_bt(x, ps & 31, b, t0 + (ps >> 5), c, PR, T) : zero;
        }
    } else if (d < NPH) {                                  // four phases of one frame
        const int gp = (int)(mg / PR);
        const long long f = mg % PR;
        const int p0 = group_phase0(gp, d);
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = x_at(x, p0 + (i - 1) * d, f, 0, c, PR, BT, T);
    } else {                                               // four frames f0 + j s of one phase (PRq group rows per phase)
        const int s = d / NPH, PRq = (int)(Mq / NPH);
        const int p = (int)(mg / PRq);
        int b, t0;
        const bool ok = frame_group(mg % PRq, s, BT, T, b, t0);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = ok ? x_bt(x, p, b, t0 + (i - 1) * s, c, PR, T) : zero;
    }
    const long long o = mg * C + c, plane = Mq * C;
    *reinterpret_cast<f32x4*>(U + o) = 4.f * v[0] - 5.f * v[2] + v[4];
    *reinterpret_cast<f32x4*>(U + plane + o) = -4.f * (v[1] + v[2]) + v[3] + v[4];
    *reinterpret_cast<f32x4*>(U + 2 * plane + o) = 4.f * (v[1] - v[2]) - v[3] + v[4];
    *reinterpret_cast<f32x4*>(U + 3 * plane + o) = 2.f * (v[3] - v[1]) - v[2] + v[4];
    *reinterpret_cast<f32x4*>(U + 4 * plane + o) = 2.f * (v[1] - v[3]) - v[2] + v[4];
    *reinterpret_cast<f32x4*>(U + 5 * plane + o) = 4.f * v[1] - 5.f * v[3] + v[5];
}

__global__ void wino4_weights_kernel(const float* __restrict__ in_Bt, float* __restrict__ G) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * C * C) return;
    const int n = idx / C, c = idx % C;
    const float* row = in_Bt + (long long)n * KCONV + (c / 16) * 48 + c % 16;
    const double wm = row[0], w0 = row[16], wp = row[32];
    const long long plane = (long long)2 * C * C;
    G[idx] = (float)(wm / 4.0);
    G[plane + idx] = (float)(-(wm + w0 + wp) / 6.0);
    G[2 * plane + idx] = (float)(-(wm - w0 + wp) / 6.0);
    G[3 * plane + idx] = (float)(wm / 24.0 + w0 / 12.0 + wp / 6.0);
    G[4 * plane + idx] = (float)(wm / 24.0 - w0 / 12.0 + wp / 6.0);
    G[5 * plane + idx] = (float)wp;
}

// coefficient of output j's conditioning weights in product k for a column of slice A / B / C (rows: the subset's products)
__constant__ double W4_A[4][4] = {{1, 0, -1, 0}, {0, .5, .5, 0}, {0, -.5, .5, 0}, {0, -1, 0, 1}};                      // products 0, 1, 2, 5
__constant__ double W4_B[4][4] = {{1, 0, -.25, 0}, {0, .25, .125, 0}, {0, -.25, .125, 0}, {0, -4, 0, 1}};               // products 0, 3, 4, 5
__constant__ double W4_C[4][4] = {{2. / 3, 2. / 3, -1. / 6, -1. / 6}, {2. / 3, -2. / 3, -1. / 6, 1. / 6},
                                  {-1. / 6, -1. / 12, 1. / 6, 1. / 12}, {-1. / 6, 1. / 12, 1. / 6, -1. / 12}};           // products 1, 2, 3, 4

// V[8][6][1024][224] from cond_Bt [32][1024][320].  Column layout of a product's K = 224: products 0, 5: [A | B | 0 x 16];
// products 1, 2: [A | C]; products 3, 4: [B | C | 0 x 16]
__global__ void wino4_cond_weights_kernel(const float* __restrict__ cond_Bt, float* __restrict__ V, int d) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)8 * 6 * 2 * C * K4) return;
    const int kk = (int)(idx % K4), n = (int)((idx / K4) % (2 * C)), k = (int)((idx / ((long long)K4 * 2 * C)) % 6),
              gp = (int)(idx / ((long long)K4 * 2 * C * 6));
    // which slice column, and which row of that slice's coefficient matrix
    int col = -1, row = 0;
    const double (*cf)[4] = W4_A;
    if (k == 0 || k == 5) {
        if (kk < SA) { col = kk; cf = W4_A; row = k == 0 ? 0 : 3; }
        else if (kk < SA + SB) { col = SA + (kk - SA); cf = W4_B; row = k == 0 ? 0 : 3; }
    } else if (k == 1 || k == 2) {
        if (kk < SA) { col = kk; cf = W4_A; row = k; }
        else { col = SA + SB + (kk - SA); cf = W4_C; row = k - 1; }
    } else {
        if (kk < SB) { col = SA + kk; cf = W4_B; row = k - 2; }
        else if (kk < SB + SC) { col = SA + SB + (kk - SB); cf = W4_C; row = k - 1; }
    }
    double acc = 0.0;
    if (col >= 0) {
        const int p0 = group_phase0(gp, d);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc += cf[row][j] * (double)cond_Bt[((long long)(p0 + j * d) * 2 * C + n) * KMEL + col];
    }
    V[idx] = (float)acc;
}

// mel planes [6][rows][224] in the column layout above (rows = frames; the same for every group phase)
__global__ void wino4_mel_planes_kernel(const float* __restrict__ mel, float* __restrict__ P, int rows, int BT, int T) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)rows * K4) return;
    const long long r = idx / K4;
    const int kk = (int)(idx % K4);
    const long long plane = (long long)rows * K4;
    const float a = kk < SA ? melwin(mel, r, kk, BT, T) : 0.f;                                   // slice A column kk
    const float b = kk < SB ? melwin(mel, r, SA + kk, BT, T) : 0.f;                              // slice B column kk
    const float ab = kk < SA ? a : kk < SA + SB ? melwin(mel, r, kk, BT, T) : 0.f;               // [A | B | 0]
    const float ac = kk < SA ? a : melwin(mel, r, SA + SB + (kk - SA), BT, T);                   // [A | C]
    const float bc = kk < SB ? b : kk < SB + SC ? melwin(mel, r, SA + SB + (kk - SB), BT, T) : 0.f;   // [B | C | 0]
    P[idx] = ab;
    P[plane + idx] = ac;
    P[2 * plane + idx] = ac;
    P[3 * plane + idx] = bc;
    P[4 * plane + idx] = bc;
    P[5 * plane + idx] = ab;
}

// which slice column a product's conditioning column kk is (-1: padding) and the row of the slice's coefficient matrix
__device__ __forceinline__ int slice_col(int k, int kk, int& row, int& which) {
    if (k == 0 || k == 5) {
        row = k == 0 ? 0 : 3;
        if (kk < SA) { which = 0; return kk; }
        if (kk < SA + SB) { which = 1; return kk; }
    } else if (k == 1 || k == 2) {
        if (kk < SA) { which = 0; row = k; return kk; }
        which = 2;
        row = k - 1;
        return SA + SB + (kk - SA);
    } else {
        if (kk < SB) { which = 1; row = k - 2; return SA + kk; }
        if (kk < SB + SC) { which = 2; row = k - 1; return SA + SB + (kk - SB); }
    }
    which = 0;
    return -1;
}

// dilations >= 32 (four FRAMES of one phase): the four outputs share the weights, so the products' weights are plain column
// selections V[32][6][1024][224] of V_p in the column layout above, and the slice combinations are formed on the mel side
__global__ void wino4_cond_weights_frames_kernel(const float* __restrict__ cond_Bt, float* __restrict__ V) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)NPH * 6 * 2 * C * K4) return;
    const int kk = (int)(idx % K4), n = (int)((idx / K4) % (2 * C)), k = (int)((idx / ((long long)K4 * 2 * C)) % 6),
              p = (int)(idx / ((long long)K4 * 2 * C * 6));
    int row, which;
    const int col = slice_col(k, kk, row, which);
    V[idx] = col >= 0 ? cond_Bt[((long long)p * 2 * C + n) * KMEL + col] : 0.f;
}

// mel planes [6][rows][224] for s = d / 32: group row gf <-> frames t0 + j s of its utterance (frame_group); product k, slice
// column: sum_j coef[k][j] melwin(f_j) over the frames inside the utterance
__global__ void wino4_mel_planes_frames_kernel(const float* __restrict__ mel, float* __restrict__ P, int s, int rows, int BT, int T) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)6 * rows * K4) return;
    const int kk = (int)(idx % K4), k = (int)(idx / ((long long)rows * K4));
    int b, t0;
    const bool ok = frame_group((idx / K4) % rows, s, BT, T, b, t0);
    int row, which;
    const int col = slice_col(k, kk, row, which);
    double acc = 0.0;
    if (ok && col >= 0) {
        const double (*cf)[4] = which == 0 ? W4_A : which == 1 ? W4_B : W4_C;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (t0 + j * s < T) acc += cf[row][j] * (double)melwin(mel, (long long)b * T + t0 + j * s, col, BT, T);
    }
    P[idx] = (float)acc;
}

// Dilation 16.  With c_j = mel(t_j) V(p_j), product k of the subset {1, 2, 3, 4} needs sum_j coef[k][j] c_j, and every row of
// that subset's coefficient matrix is an outer product (over {t, t + 1}) x (over {p0, p0 + 16}) -- (2/3, -1/6) x (1, +-1) for
// products 1, 2 and (-1, 1) x (1/6, +-1/12) for products 3, 4 -- so each is ONE K = 320 product of a mel combination and a
// weight combination.  Products 0 and 5 carry no conditioning (their launch runs K = 512).
// Vm[16][4][1024][320] (products 1 .. 4): V(p0) + V(p1), V(p0) - V(p1), V(p0) / 6 + V(p1) / 12, V(p0) / 6 - V(p1) / 12
__global__ void wino4_cond_weights_mixed_kernel(const float* __restrict__ cond_Bt, float* __restrict__ V) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)16 * 2 * C * KMEL) return;
    const int k = (int)(idx % KMEL), n = (int)((idx / KMEL) % (2 * C)), p0 = (int)(idx / ((long long)KMEL * 2 * C));
    const double v0 = cond_Bt[((long long)p0 * 2 * C + n) * KMEL + k], v1 = cond_Bt[((long long)(p0 + 16) * 2 * C + n) * KMEL + k];
    const long long zs = (long long)2 * C * KMEL, o = (long long)p0 * 4 * zs + (long long)n * KMEL + k;
    V[o] = (float)(v0 + v1);
    V[o + zs] = (float)(v0 - v1);
    V[o + 2 * zs] = (float)(v0 / 6.0 + v1 / 12.0);
    V[o + 3 * zs] = (float)(v0 / 6.0 - v1 / 12.0);
}
// mel planes [4][rows][320] (products 1 .. 4): (2/3) m(t) - (1/6) m(t + 1) twice, m(t + 1) - m(t) twice (m(t + 1) = 0 past the end)
__global__ void wino4_mel_planes_mixed_kernel(const float* __restrict__ mel, float* __restrict__ P, int rows, int BT, int T) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)rows * KMEL) return;
    const int k = (int)(idx % KMEL);
    int b, t0;
    const bool ok = mixed_group(idx / KMEL, BT, T, b, t0);
    const double m0 = ok ? (double)melwin(mel, (long long)b * T + t0, k, BT, T) : 0.0;
    const double m1 = ok && t0 + 1 < T ? (double)melwin(mel, (long long)b * T + t0 + 1, k, BT, T) : 0.0;
    const long long plane = (long long)rows * KMEL;
    const float a = (float)(m0 * (2.0 / 3.0) - m1 / 6.0), e = (float)(m1 - m0);
    P[idx] = a;
    P[plane + idx] = a;
    P[2 * plane + idx] = e;
    P[3 * plane + idx] = e;
}

__global__ void wino4_combine_kernel(const float* __restrict__ P, const float* __restrict__ bias, float* __restrict__ acts, int d,
                                     int PR, int BT, int T, long long Mq) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Mq * (C / 4)) return;
    const long long mg = idx / (C / 4);
    const int ch = (int)(idx % (C / 4)) * 4;
    const int col = (ch >> 5) * 64 + (ch & 31);
    const long long plane = Mq * 2 * C, o = mg * 2 * C + col;
    f32x4 a[6], b[6];
#pragma unroll
    for (int z = 0; z < 6; ++z) {
        a[z] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P + z * plane + o));
        b[z] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P + z * plane + o + 32));
    }
    const f32x4 ba = *reinterpret_cast<const f32x4*>(bias + col), bb = *reinterpret_cast<const f32x4*>(bias + col + 32);
    f32x4 t[4], s[4];
    t[0] = a[0] + a[1] + a[2] + a[3] + a[4] + ba;
    t[1] = a[1] - a[2] + 2.f * (a[3] - a[4]) + ba;
    t[2] = a[1] + a[2] + 4.f * (a[3] + a[4]) + ba;
    t[3] = a[1] - a[2] + 8.f * (a[3] - a[4]) + a[5] + ba;
    s[0] = b[0] + b[1] + b[2] + b[3] + b[4] + bb;
    s[1] = b[1] - b[2] + 2.f * (b[3] - b[4]) + bb;
    s[2] = b[1] + b[2] + 4.f * (b[3] + b[4]) + bb;
    s[3] = b[1] - b[2] + 8.f * (b[3] - b[4]) + b[5] + bb;
    if (d == 16) {                                         // two phases x two frames
        const int PRm = (int)(Mq / 16);
        const int p0 = (int)(mg / PRm);
        int b, t0;
        if (!mixed_group(mg % PRm, BT, T, b, t0)) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (t0 + (j >> 1) >= T) break;
            f32x4 g;
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = gate_tanh_sigmoid(t[j][k], s[j][k]);
            *reinterpret_cast<f32x4*>(acts + ((long long)(p0 + 16 * (j & 1)) * PR + (long long)b * T + t0 + (j >> 1)) * C + ch) = g;
        }
        return;
    }
    long long r0, rstep;                                   // acts row of output 0 and the row step between outputs
    int nout = 4;                                          // outputs of this group that exist
    if (d < NPH) {
        const int gp = (int)(mg / PR);
        r0 = (long long)group_phase0(gp, d) * PR + mg % PR;
        rstep = (long long)d * PR;
    } else {
        const int sf = d / NPH, PRq = (int)(Mq / NPH);
        const int p = (int)(mg / PRq);
        int b, t0;
        if (!frame_group(mg % PRq, sf, BT, T, b, t0)) return;                     // padding / empty group
        r0 = (long long)p * PR + (long long)b * T + t0;
        rstep = sf;
        nout = (T - t0 + sf - 1) / sf;                     // frames t0 + j sf < T
        nout = nout > 4 ? 4 : nout;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j >= nout) break;
        f32x4 g;
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = gate_tanh_sigmoid(t[j][k], s[j][k]);
        *reinterpret_cast<f32x4*>(acts + (r0 + j * rstep) * C + ch) = g;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Fused form of the GEMM + combine passes.  One block owns BM group rows x BN pre-activation columns of ALL SIX products: its
// K loop walks product after product (K = 512 + the product's conditioning chunks) through one LDS-DMA pipeline, each product
// into its own accumulator set (a wave holds 32 rows x 64 columns x 6 products = 192 accumulator registers), and the epilogue
// applies the output transform, bias and gate in registers and stores the four output row sets of `acts` -- no P planes
// (1.26 GB written and read per layer at config 2), no combine launch.  Same products in the same k order as the three-pass
// form, so the two agree to the last bit of the fp32 sums (the output transform is written identically).
// The conditioning chunks of a product are a chunk range of its mel plane against one or two chunk ranges of the weight rows:
// for the frame groups (dilations >= 32) the products' weights are column selections of cond_Bt itself -- A = chunks 0 .. 6,
// B = 7 .. 12, C = 13 .. 19 of its 320 columns -- so no per-product copies exist, and the all-zero padding chunk of the
// 208-column products is skipped (45 K steps instead of 46 for four of the six products).
struct WinoFusedArgs {
    const float* U;   long long uplane;                       // transformed inputs [6][Mq][512]
    const float* G;   long long gplane;                       // tap combinations [6][1024][512]
    const float* mel; long long mplane; int ldm;              // conditioning operand planes (row stride ldm); plane p - pofs
    const float* V;   long long vplane, strideVp; int ldv;    // conditioning weights: V + phase * strideVp + (p - pofs) * vplane
    int pofs;
    unsigned long long cfg_lo, cfg_hi; // product p's conditioning chunks (16 columns each), 16 bits per product (p < 4: cfg_lo):
                                       // n1 | b1 << 5 | n2 << 9 | b2 << 12 -- operand chunks 0 .. n1 + n2 - 1 against weight chunks
                                       // b1 .. b1 + n1 - 1, then b2 .. b2 + n2 - 1
    const float* bias;                 // [1024] (gate-permuted like the weight rows)
    float* acts;                       // [32 PR][512]
    int Mq, phase_rows;                // group rows; group rows per (group) phase block
    int kind;                          // 0 phase groups (d <= 8), 1 frame groups (d >= 32), 2 mixed groups (d = 16)
    int d, PR, BT, T;
};

template <int WR, int WC, int NBUF, int OCC>
__global__ __launch_bounds__(WR * WC * 64, OCC) void wino4_fused_kernel(const WinoFusedArgs g) {
    constexpr int NW = WR * WC, BM = WR * 32, BN = WC * 64;
    constexpr int NPA = BM / 16, NPB = BN / 16, PPW = (NPA + NPB) / NW;     // 16-row DMA pieces: A side, B side, per wave
    static_assert((NPA + NPB) % NW == 0, "the pieces of a tile divide among the waves");
    constexpr int STAGE = (BM + BN) * 16;                                   // floats per LDS buffer: [A rows | B rows] x 16 k
    static_assert(NBUF * STAGE >= NW * 32 * 36, "the epilogue's transpose patches fit the pipeline buffers");
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware order (as gemm_f32_kernel): the numNt column blocks of an M tile run back to back on one XCD
    constexpr int numNt = 2 * C / BN;
    const int numMt = g.Mq / BM;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int mt = (slot / numNt) * 8 + xcd, nt = slot % numNt;
    if (mt >= numMt) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int ph = m0 / g.phase_rows, fr0 = m0 - ph * g.phase_rows;        // block-uniform (group) phase, first row inside it

    f32x16 acc[6][2];
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][h][r] = 0.f;

    // ---- tile stream.  Piece q of a tile = 16 rows x 64 B: q < NPA operand rows, else weight rows; lane l of a piece fetches
    // chunk (l & 3) ^ ((l >> 4) & 3) of row l >> 2 and lands at byte 16 l of the piece (XOR swizzle through the source address)
    const int prow = lane >> 2, chunk = (lane & 3) ^ ((lane >> 4) & 3);
    unsigned vo0[PPW], vo1[PPW];                                            // per-lane byte offsets: K = 512 part / conditioning part
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        const bool isA = q < NPA;
        const int row = (isA ? q : q - NPA) * 16 + prow;
        vo0[i] = (unsigned)((row * C + chunk * 4) * 4);
        vo1[i] = (unsigned)((row * (isA ? g.ldm : g.ldv) + chunk * 4) * 4);
    }
    // product p's conditioning chunks: a 16-bit field of two 64-bit words, extracted with shifts (pure scalar ALU: a select
    // chain over six kernel arguments is turned by hipcc into a scalar LOAD from a selected address inside the K loop -- or,
    // with the words held in variables, into a table of pointers in scratch)
    const unsigned long long cfg_lo = g.cfg_lo, cfg_hi = g.cfg_hi;
    auto cfg_of = [&](int p) -> unsigned {
        return (unsigned)((p < 4 ? cfg_lo >> (16 * (p & 3)) : cfg_hi >> (16 * (p & 3))) & 0xffffull);
    };
    int lp = 0, lseg = 0, lkc = 0;                                          // (product, part, chunk) of the next tile to request
    unsigned lcfg = cfg_of(0);
    const float *abase = nullptr, *bbase = nullptr;
    auto seg_setup = [&]() {
        if (lseg == 0) {
            abase = g.U + lp * g.uplane + (long long)m0 * C;
            bbase = g.G + lp * g.gplane + (long long)n0 * C;
        } else {
            abase = g.mel + (lp - g.pofs) * g.mplane + (long long)fr0 * g.ldm;
            bbase = g.V + ph * g.strideVp + (lp - g.pofs) * g.vplane + (long long)n0 * g.ldv;
        }
    };
    // The tile stream in two parts, as in gemm_f32_kernel's rotated loop: `prepare` does the (scalar) address math of the next
    // tile, `issue_piece` requests one 16-row piece of it.
    unsigned vo[PPW], ko[PPW];                                              // per-lane offsets / scalar K offsets of the prepared tile
    const float* pb[PPW];                                                   // (wave-uniform) descriptor bases of its pieces
    auto prepare = [&]() {
        const bool live = lp < 6;                                           // past the last tile the pieces fetch nothing
        const int n1 = lcfg & 31, b1 = (lcfg >> 5) & 15, n2 = (lcfg >> 9) & 7, b2 = lcfg >> 12;
        const unsigned ka = (unsigned)lkc * 64u;
        const unsigned kb = lseg == 0 ? ka : (unsigned)(lkc < n1 ? b1 + lkc : b2 + lkc - n1) * 64u;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const bool isA = wave * PPW + i < NPA;                          // wave-uniform
            pb[i] = isA ? abase : bbase;
            ko[i] = isA ? ka : kb;
            vo[i] = live ? (lseg == 0 ? vo0[i] : vo1[i]) : OOB;
        }
        if (live && ++lkc == (lseg == 0 ? C / 16 : n1 + n2)) {
            lkc = 0;
            if (lseg == 0 && n1 + n2 > 0) {
                lseg = 1;
            } else {
                lseg = 0;
                lcfg = cfg_of(++lp);
            }
            seg_setup();
        }
    };
    auto issue_piece = [&](int i, int buf) {
        // (by-value copies: with an element of a local array as the builtin's operand hipcc's HOST pass silently emits no stub
        //  for the kernel -- undefined symbol when the library is loaded)
        const unsigned voff = vo[i], koff = ko[i];
        const float* base = pb[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc_uniform(base), (lds_ptr_t)(smem + buf * STAGE + (wave * PPW + i) * 256), 16,
                                                 voff, koff, 0, 0);
    };
    const int xr = (li >> 2) & 3;
    struct Frag {
        f32x4 a, b0, b1;
    };
    auto read_frag = [&](int buf, int k8, Frag& f) {                        // the K = 8 half k8 of a tile: one A and two B fragments
        const int koff = ((2 * k8 + lh) ^ xr) * 4;
        const float* a = smem + buf * STAGE + (wr * 32 + li) * 16 + koff;
        const float* b = smem + buf * STAGE + BM * 16 + (wc * 64 + li) * 16 + koff;
        f.a = *reinterpret_cast<const f32x4*>(a);
        f.b0 = *reinterpret_cast<const f32x4*>(b);
        f.b1 = *reinterpret_cast<const f32x4*>(b + 32 * 16);
    };

    // K loop, rotated by half a step (gemm_f32_kernel: the plain loop left the matrix pipe idle after every barrier while all
    // waves issued their DMA pieces and waited for their operand reads: 71 % of peak).  Step t = [reads of tile t's second half |
    // MFMAs of its first half | wait for tile t + 1, barrier | reads of tile t + 1's first half | MFMAs of the second half with
    // the pieces of tile t + NBUF requested one per MFMA pair into tile t's buffer -- every wave holds tile t in registers].
    seg_setup();
#pragma unroll
    for (int b = 0; b < NBUF; ++b) {
        prepare();
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(i, b);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NBUF - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
    read_frag(0, 0, f0);
    int buf = 0;
    static_for<6>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        const unsigned c = cfg_of(P);
        const int nsteps = C / 16 + (int)(c & 31) + (int)((c >> 9) & 7);
        for (int s = 0; s < nsteps; ++s) {
            read_frag(buf, 1, f1);
            prepare();                                                      // tile t + NBUF (address math only)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.a[kk], f0.b0[kk], acc[P][0], 0, 0, 0);
                acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.a[kk], f0.b1[kk], acc[P][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                              // (keeps the MFMAs above the wait)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW * (NBUF - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            const int bufn = buf == NBUF - 1 ? 0 : buf + 1;
            read_frag(bufn, 0, f0);                                         // (past the last tile: zeros nothing uses)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f1.a[kk], f1.b0[kk], acc[P][0], 0, 0, 0);
                acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f1.a[kk], f1.b1[kk], acc[P][1], 0, 0, 0);
                if (kk < PPW) {
                    issue_piece(kk, buf);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            static_assert(PPW <= 4, "one DMA piece per MFMA pair of the second half");
            buf = bufn;
        }
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the fetch-nothing pieces of the last steps
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: output transform + bias + gate in registers (the arithmetic of wino4_combine_kernel), each gated 32 x 32
    // tile transposed through a wave-private LDS patch so that a lane stores 16 bytes of one acts row
    float* patch = smem + wave * (32 * 36);
    const int er = lane >> 3, ec4 = (lane & 7) * 4;
    const float bt = g.bias[n0 + wc * 64 + li], bs = g.bias[n0 + wc * 64 + 32 + li];
    const int ch0 = ((n0 + wc * 64) >> 6) * 32 + ec4;
    const int sfr = g.d / NPH;
    auto out_row = [&](int lr, int j) -> long long {                        // acts row of output j of local group row lr (-1: none)
        const int gl = fr0 + lr;                                            // group row inside the (group) phase block
        if (g.kind == 0) return (long long)(group_phase0(ph, g.d) + j * g.d) * g.PR + gl;
        int b, t0;
        if (g.kind == 1) {
            if (!frame_group(gl, sfr, g.BT, g.T, b, t0) || t0 + j * sfr >= g.T) return -1;
            return (long long)ph * g.PR + (long long)b * g.T + t0 + j * sfr;
        }
        if (!mixed_group(gl, g.BT, g.T, b, t0) || t0 + (j >> 1) >= g.T) return -1;
        return (long long)(ph + 16 * (j & 1)) * g.PR + (long long)b * g.T + t0 + (j >> 1);
    };
    static_for<4>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float tv, sv;
            if constexpr (j == 0) {
                tv = acc[0][0][r] + acc[1][0][r] + acc[2][0][r] + acc[3][0][r] + acc[4][0][r] + bt;
                sv = acc[0][1][r] + acc[1][1][r] + acc[2][1][r] + acc[3][1][r] + acc[4][1][r] + bs;
            } else if constexpr (j == 1) {
                tv = acc[1][0][r] - acc[2][0][r] + 2.f * (acc[3][0][r] - acc[4][0][r]) + bt;
                sv = acc[1][1][r] - acc[2][1][r] + 2.f * (acc[3][1][r] - acc[4][1][r]) + bs;
            } else if constexpr (j == 2) {
                tv = acc[1][0][r] + acc[2][0][r] + 4.f * (acc[3][0][r] + acc[4][0][r]) + bt;
                sv = acc[1][1][r] + acc[2][1][r] + 4.f * (acc[3][1][r] + acc[4][1][r]) + bs;
            } else {
                tv = acc[1][0][r] - acc[2][0][r] + 8.f * (acc[3][0][r] - acc[4][0][r]) + acc[5][0][r] + bt;
                sv = acc[1][1][r] - acc[2][1][r] + 8.f * (acc[3][1][r] - acc[4][1][r]) + acc[5][1][r] + bs;
            }
            patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = gate_tanh_sigmoid(tv, sv);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // wave-private patch: no barrier needed
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(patch + (er + 8 * qq) * 36 + ec4);
            const long long row = out_row(wr * 32 + er + 8 * qq, j);
            if (row >= 0) *reinterpret_cast<f32x4*>(g.acts + row * C + ch0) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // patch reads done before the next output overwrites it
    });
}

template <int WR, int WC, int NBUF, int OCC>
hipError_t launch_wino_fused(const WinoFusedArgs& a, hipStream_t st) {
    constexpr int BM = WR * 32, BN = WC * 64;
    const size_t lds = (size_t)NBUF * (BM + BN) * 16 * sizeof(float);
    if (a.Mq % BM != 0 || a.phase_rows % BM != 0 || a.Mq % a.phase_rows != 0) return hipErrorInvalidValue;
    auto kern = wino4_fused_kernel<WR, WC, NBUF, OCC>;
    static PerDeviceOnce attr_set;
    if (hipError_t e = set_max_dyn_lds_once((const void*)kern, lds, attr_set); e != hipSuccess) return e;
    const int numMt8 = (a.Mq / BM + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3(numMt8 * (2 * C / BN)), dim3(WR * WC * 64), lds, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------
// Fused form WITHOUT the pre-pass: the input transform moves into the operand reads.  The K loop of the tap part runs chunk
// by chunk (16 columns of the 512), and for every chunk the SIX INPUT tiles x[l + (i - 1) d] of the block's 64 group rows are
// fetched once (the same bytes as six transformed tiles) by LDS-DMA with per-lane row addresses -- the gather the pre-pass
// did: phase carries, frame groups, rows outside an utterance read as zero through out-of-range offsets -- and stay in LDS for
// the six products of that chunk; a product's A fragment is then 3 - 4 ds_read_b128 of input fragments combined in registers
// (U0 = 4 x0 - 5 x2 + x4, ...: the arithmetic of wino4_prepass_kernel) under the MFMAs of the other half step.  No U planes
// (0.63 GB of workspace and 1.26 GB of traffic per layer at config 2), no pre-pass launch.  The conditioning part follows as
// in wino4_fused_kernel (mel planes by DMA).  4 waves, 64 x 128 tile per product, two blocks per CU; LDS: two stages of six
// input tiles (48 KB; the conditioning part reuses them as its operand ring) + three weight tiles (24 KB).
struct WinoFused2Args {
    const float* x;                                           // residual stream [32 PR][512]
    const float* G;   long long gplane;                       // tap combinations [6][1024][512]
    const float* mel; long long mplane; int ldm;              // conditioning operand planes (row stride ldm); plane p - pofs
    const float* V;   long long vplane, strideVp; int ldv;    // conditioning weights: V + phase * strideVp + (p - pofs) * vplane
    int pofs;
    unsigned long long cfg_lo, cfg_hi; // conditioning chunks per product (see WinoFusedArgs)
    const float* bias;
    float* acts;
    int Mq, phase_rows, kind, d, PR, BT, T;
};

__global__ __launch_bounds__(256, 2) void wino4_fused2_kernel(const WinoFused2Args g) {
    constexpr int BM = 64, BN = 128, NBUF = 3, NG = C / 16;                 // NG = 32 chunks of the tap part
    constexpr int XT = BM * 16, XS = 6 * XT, BS = BN * 16;                  // floats: one input tile, one stage of six, one weight tile
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const xs = smem;                                                 // [2][6][64][16]; conditioning part: operand ring [3][64][16]
    float* const Bs = smem + 2 * XS;                                        // [3][128][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    constexpr int numNt = 2 * C / BN;
    const int numMt = g.Mq / BM;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int mt = (slot / numNt) * 8 + xcd, nt = slot % numNt;
    if (mt >= numMt) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int ph = m0 / g.phase_rows, fr0 = m0 - ph * g.phase_rows;

    f32x16 acc[6][2];
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][h][r] = 0.f;

    // ---- per-lane source offsets.  A DMA piece = 16 rows x 64 B; lane l fetches chunk (l & 3) ^ ((l >> 4) & 3) of row l >> 2.
    const int prow = lane >> 2, chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const int sfr = g.d / NPH;
    // input i of this lane's group row = (block-uniform row delta of input i) + (this lane's base row), or nothing: one base
    // offset and a validity mask per lane, the deltas go into the descriptor base
    unsigned xbase = 0, xvalid = 0;
    int xdelta[6];                                                          // (floats; < 2^31: checked at launch)
    {
        const int gl = fr0 + wave * 16 + prow;                              // group row inside the (group) phase block
        int b = 0, t0 = 0;
        bool ok;
        if (g.kind == 0) {                                                  // four phases of one frame: base = the frame row
            ok = gl < g.BT;
            t0 = gl % g.T;
            xbase = (unsigned)gl;
        } else if (g.kind == 1) {                                           // four frames of one phase: base = frame t0 of the utterance
            ok = frame_group(gl, sfr, g.BT, g.T, b, t0);
            xbase = (unsigned)(b * g.T + t0);
        } else {                                                            // two phases x two frames
            ok = mixed_group(gl, g.BT, g.T, b, t0);
            xbase = (unsigned)(b * g.T + t0);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int dt;                                                         // frame offset of input i from the base row
            if (g.kind == 0) {
                const int ps = group_phase0(ph, g.d) + (i - 1) * g.d;
                dt = ps >> 5;
                xdelta[i] = ((ps & 31) * g.PR + dt) * C;
            } else if (g.kind == 1) {
                dt = (i - 1) * sfr;
                xdelta[i] = (ph * g.PR + dt) * C;
            } else {
                const int ps = ph + 16 * (i - 1);
                dt = ps >> 5;
                xdelta[i] = ((ps & 31) * g.PR + dt) * C;
            }
            if (ok && t0 + dt >= 0 && t0 + dt < g.T) xvalid |= 1u << i;
        }
        xbase = xbase * (unsigned)(C * 4) + (unsigned)chunk * 16u;
    }
    // this wave's two weight pieces are 16 rows apart: one per-lane offset, the second piece's distance is a scalar
    const unsigned vb0 = (unsigned)(((2 * wave * 16 + prow) * C + chunk * 4) * 4);
    const unsigned vb1 = (unsigned)(((2 * wave * 16 + prow) * g.ldv + chunk * 4) * 4);
    const unsigned va1 = (unsigned)(((wave * 16 + prow) * g.ldm + chunk * 4) * 4);      // its piece of a mel-plane tile

    const unsigned long long cfg_lo = g.cfg_lo, cfg_hi = g.cfg_hi;
    auto cfg_of = [&](int p) -> unsigned {
        return (unsigned)((p < 4 ? cfg_lo >> (16 * (p & 3)) : cfg_hi >> (16 * (p & 3))) & 0xffffull);
    };
    auto nchunks = [&](int p) -> int {
        const unsigned c = cfg_of(p);
        return (int)(c & 31) + (int)((c >> 9) & 7);
    };

    // ---- DMA requests
    const float* const gb0 = g.G + (long long)n0 * C;
    auto issue_x = [&](auto ic, int grp) {                                  // input tile I of chunk group `grp` -> stage grp & 1
        constexpr int I = decltype(ic)::value;
        const unsigned voff = (grp < NG && ((xvalid >> I) & 1u)) ? xbase : OOB;      // (past the last group: fetch nothing, keep the counts)
        const float* base = g.x + (long long)xdelta[I];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc_uniform(base), (lds_ptr_t)(xs + (grp & 1) * XS + I * XT + wave * 256), 16, voff,
                                                 (unsigned)grp * 64u, 0, 0);
    };
    auto issue_b_taps = [&](int j, int prod, int grp, int buf) {            // weight piece j of (product, chunk group) -> Bs[buf]
        const float* base = gb0 + prod * (2 * C * C);                       // (the planes of G are 1024 x 512 floats apart)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc_uniform(base), (lds_ptr_t)(Bs + buf * BS + (2 * wave + j) * 256), 16, vb0,
                                                 (unsigned)grp * 64u + (unsigned)j * (16u * C * 4u), 0, 0);
    };
    // conditioning part: (product, chunk) iterator over the products that carry chunks
    int lp = 0, lkc = 0;
    while (lp < 6 && nchunks(lp) == 0) ++lp;
    const float *cab = nullptr, *cbb = nullptr;
    auto cond_setup = [&]() {
        cab = g.mel + (lp - g.pofs) * g.mplane + (long long)fr0 * g.ldm;
        cbb = g.V + ph * g.strideVp + (lp - g.pofs) * g.vplane + (long long)n0 * g.ldv;
    };
    cond_setup();
    bool clive = false;                                                     // the prepared conditioning tile exists
    unsigned cka = 0, ckb = 0;
    const float *ca = nullptr, *cb = nullptr;
    auto cond_prepare = [&]() {                                             // address math of the next conditioning tile
        clive = lp < 6;
        const unsigned c = cfg_of(lp < 6 ? lp : 5);
        const int n1 = c & 31, b1 = (c >> 5) & 15, n2 = (c >> 9) & 7, b2 = c >> 12;
        cka = (unsigned)lkc * 64u;
        ckb = (unsigned)(lkc < n1 ? b1 + lkc : b2 + lkc - n1) * 64u;
        ca = cab;
        cb = cbb;
        if (clive && ++lkc == n1 + n2) {
            lkc = 0;
            ++lp;
            while (lp < 6 && nchunks(lp) == 0) ++lp;
            cond_setup();
        }
    };
    auto cond_issue = [&](int i, int buf) {                                 // piece i (0: mel-plane rows, 1 - 2: weight rows) -> ring slot buf
        const unsigned voff = clive ? (i == 0 ? va1 : vb1) : OOB;
        const unsigned koff = i == 0 ? cka : ckb + (unsigned)(i - 1) * (16u * (unsigned)g.ldv * 4u);
        const float* base = i == 0 ? ca : cb;
        float* dst = i == 0 ? xs + buf * XT + wave * 256 : Bs + buf * BS + (2 * wave + i - 1) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc_uniform(base), (lds_ptr_t)dst, 16, voff, koff, 0, 0);
    };

    // ---- operand fragments
    const int xr = (li >> 2) & 3;
    struct Frag {
        f32x4 a, b0, b1;
    };
    struct Raw {
        f32x4 v[4];
    };
    auto read_b = [&](int buf, int h, Frag& f) {
        const float* b = Bs + buf * BS + (wc * 64 + li) * 16 + ((2 * h + lh) ^ xr) * 4;
        f.b0 = *reinterpret_cast<const f32x4*>(b);
        f.b1 = *reinterpret_cast<const f32x4*>(b + 32 * 16);
    };
    // product P's input fragments of half h: P = 0: x0 x2 x4, P = 1 .. 4: x1 x2 x3 x4, P = 5: x1 x3 x5
    auto read_x = [&](auto pc, int stage, int h, Raw& r) {
        constexpr int P = decltype(pc)::value;
        const float* a = xs + stage * XS + (wr * 32 + li) * 16 + ((2 * h + lh) ^ xr) * 4;
        constexpr int i0 = P == 0 ? 0 : 1, i1 = P == 0 ? 2 : P == 5 ? 3 : 2, i2 = P == 0 ? 4 : P == 5 ? 5 : 3;
        r.v[0] = *reinterpret_cast<const f32x4*>(a + i0 * XT);
        r.v[1] = *reinterpret_cast<const f32x4*>(a + i1 * XT);
        r.v[2] = *reinterpret_cast<const f32x4*>(a + i2 * XT);
        if constexpr (P >= 1 && P <= 4) r.v[3] = *reinterpret_cast<const f32x4*>(a + 4 * XT);
    };
    auto combine = [&](auto pc, const Raw& r) -> f32x4 {                    // the input transform (wino4_prepass_kernel)
        constexpr int P = decltype(pc)::value;
        if constexpr (P == 0) return 4.f * r.v[0] - 5.f * r.v[1] + r.v[2];
        else if constexpr (P == 1) return -4.f * (r.v[0] + r.v[1]) + r.v[2] + r.v[3];
        else if constexpr (P == 2) return 4.f * (r.v[0] - r.v[1]) - r.v[2] + r.v[3];
        else if constexpr (P == 3) return 2.f * (r.v[2] - r.v[0]) - r.v[1] + r.v[3];
        else if constexpr (P == 4) return 2.f * (r.v[0] - r.v[2]) - r.v[1] + r.v[3];
        else return 4.f * r.v[0] - 5.f * r.v[1] + r.v[2];
    };
    auto mfma2 = [&](auto pc, const Frag& f, int kk) {
        constexpr int P = decltype(pc)::value;
        acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[kk], f.b0[kk], acc[P][0], 0, 0, 0);
        acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[kk], f.b1[kk], acc[P][1], 0, 0, 0);
    };

    // ---- prologue: the six input tiles of group 0 and weight tiles 0 - 2.  Steady state: every step requests ONE input tile
    // of the next group (in the order the products need them: x0 x2 x4 | x1 x3 | x5) and the weight tile three steps ahead -- three
    // pieces per wave and step, so "all but the last three requests" = everything up to two steps ago = the next tile's operands.
    // (the prologue ends like a steady-state step: [one input tile, weight tile 2], so that step 0's "all but three" covers tile 1)
    static_for<5>([&](auto ic) { issue_x(ic, 0); });
    issue_b_taps(0, 0, 0, 0); issue_b_taps(1, 0, 0, 0);
    issue_b_taps(0, 1, 0, 1); issue_b_taps(1, 1, 0, 1);
    issue_x(std::integral_constant<int, 5>{}, 0);
    issue_b_taps(0, 2, 0, 2); issue_b_taps(1, 2, 0, 2);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
    Raw raw;
    read_x(std::integral_constant<int, 0>{}, 0, 0, raw);
    read_b(0, 0, f0);
    f0.a = combine(std::integral_constant<int, 0>{}, raw);

    // ---- tap part: 32 chunk groups x 6 products, rotated by half a step.  Step (grp, P) = [input fragments of the second half |
    // MFMAs of the first | wait, barrier | fragments of the next tile's first half | MFMAs of the second half with this step's
    // three requests].  The last group (its look-ahead reaches the conditioning tiles) is a second copy of the body, so that
    // neither copy branches.
    int buf = 0;                                                            // weight-ring slot of the current tile
    auto group_body = [&](int grp, auto lastc) {
        constexpr bool LASTG = decltype(lastc)::value;
        const int stage = grp & 1;
        static_for<6>([&](auto pc) {
            constexpr int P = decltype(pc)::value;
            constexpr int PN = (P + 1) % 6, P3 = (P + 3) % 6;
            constexpr int XI = P < 3 ? 2 * P : P == 3 ? 1 : P == 4 ? 3 : 5;  // the input tile of the next group requested in this step
            constexpr bool COND3 = LASTG && P >= 3;                         // the tile three steps ahead is a conditioning tile
            constexpr bool LAST = LASTG && P == 5;                          // the next tile is the first conditioning tile
            // half-step = [input fragments of the next half | 2 MFMA pairs | combine | weight fragments | 2 MFMA pairs]: the sixteen
            // input registers and the eight weight registers of the incoming half are never live together (register budget)
            read_x(pc, stage, 1, raw);
            mfma2(pc, f0, 0);
            mfma2(pc, f0, 1);
            __builtin_amdgcn_sched_barrier(0);
            f1.a = combine(pc, raw);
            read_b(buf, 1, f1);
            __builtin_amdgcn_sched_barrier(0);
            mfma2(pc, f0, 2);
            mfma2(pc, f0, 3);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int bufn = buf == NBUF - 1 ? 0 : buf + 1;
            if constexpr (!LAST) read_x(std::integral_constant<int, PN>{}, P == 5 ? stage ^ 1 : stage, 0, raw);
            if constexpr (COND3) cond_prepare();
            __builtin_amdgcn_sched_barrier(0);
            mfma2(pc, f1, 0);
            if constexpr (COND3) cond_issue(0, buf);
            else issue_x(std::integral_constant<int, XI>{}, grp + 1);       // (last group: fetches nothing, keeps the counts)
            __builtin_amdgcn_sched_barrier(0);
            mfma2(pc, f1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LAST) f0.a = *reinterpret_cast<const f32x4*>(xs + bufn * XT + (wr * 32 + li) * 16 + (lh ^ xr) * 4);
            else f0.a = combine(std::integral_constant<int, PN>{}, raw);
            read_b(bufn, 0, f0);
            __builtin_amdgcn_sched_barrier(0);
            mfma2(pc, f1, 2);
            if constexpr (COND3) cond_issue(1, buf);
            else issue_b_taps(0, P3, grp + (P >= 3 ? 1 : 0), buf);
            __builtin_amdgcn_sched_barrier(0);
            mfma2(pc, f1, 3);
            if constexpr (COND3) cond_issue(2, buf);
            else issue_b_taps(1, P3, grp + (P >= 3 ? 1 : 0), buf);
            __builtin_amdgcn_sched_barrier(0);
            buf = bufn;
        });
    };
    for (int grp = 0; grp < NG - 1; ++grp) group_body(grp, std::false_type{});
    group_body(NG - 1, std::true_type{});
    // ---- conditioning part (as wino4_fused_kernel): operand ring in the first input stage, three pieces per wave and tile
    static_for<6>([&](auto pc) {
        const int nsteps = nchunks(decltype(pc)::value);
        for (int s = 0; s < nsteps; ++s) {
            {
                const float* a = xs + buf * XT + (wr * 32 + li) * 16 + ((2 + lh) ^ xr) * 4;
                f1.a = *reinterpret_cast<const f32x4*>(a);
            }
            read_b(buf, 1, f1);
            cond_prepare();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) mfma2(pc, f0, kk);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int bufn = buf == NBUF - 1 ? 0 : buf + 1;
            {
                const float* a = xs + bufn * XT + (wr * 32 + li) * 16 + (lh ^ xr) * 4;
                f0.a = *reinterpret_cast<const f32x4*>(a);
            }
            read_b(bufn, 0, f0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                mfma2(pc, f1, kk);
                if (kk < 3) {
                    cond_issue(kk, buf);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            buf = bufn;
        }
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue (as wino4_fused_kernel)
    float* patch = smem + wave * (32 * 36);
    const int er = lane >> 3, ec4 = (lane & 7) * 4;
    const float bt = g.bias[n0 + wc * 64 + li], bs = g.bias[n0 + wc * 64 + 32 + li];
    const int ch0 = ((n0 + wc * 64) >> 6) * 32 + ec4;
    auto out_row = [&](int lr, int j) -> long long {
        const int gl = fr0 + lr;
        if (g.kind == 0) return (long long)(group_phase0(ph, g.d) + j * g.d) * g.PR + gl;
        int b, t0;
        if (g.kind == 1) {
            if (!frame_group(gl, sfr, g.BT, g.T, b, t0) || t0 + j * sfr >= g.T) return -1;
            return (long long)ph * g.PR + (long long)b * g.T + t0 + j * sfr;
        }
        if (!mixed_group(gl, g.BT, g.T, b, t0) || t0 + (j >> 1) >= g.T) return -1;
        return (long long)(ph + 16 * (j & 1)) * g.PR + (long long)b * g.T + t0 + (j >> 1);
    };
    static_for<4>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float tv, sv;
            if constexpr (j == 0) {
                tv = acc[0][0][r] + acc[1][0][r] + acc[2][0][r] + acc[3][0][r] + acc[4][0][r] + bt;
                sv = acc[0][1][r] + acc[1][1][r] + acc[2][1][r] + acc[3][1][r] + acc[4][1][r] + bs;
            } else if constexpr (j == 1) {
                tv = acc[1][0][r] - acc[2][0][r] + 2.f * (acc[3][0][r] - acc[4][0][r]) + bt;
                sv = acc[1][1][r] - acc[2][1][r] + 2.f * (acc[3][1][r] - acc[4][1][r]) + bs;
            } else if constexpr (j == 2) {
                tv = acc[1][0][r] + acc[2][0][r] + 4.f * (acc[3][0][r] + acc[4][0][r]) + bt;
                sv = acc[1][1][r] + acc[2][1][r] + 4.f * (acc[3][1][r] + acc[4][1][r]) + bs;
            } else {
                tv = acc[1][0][r] - acc[2][0][r] + 8.f * (acc[3][0][r] - acc[4][0][r]) + acc[5][0][r] + bt;
                sv = acc[1][1][r] - acc[2][1][r] + 8.f * (acc[3][1][r] - acc[4][1][r]) + acc[5][1][r] + bs;
            }
            patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = gate_tanh_sigmoid(tv, sv);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(patch + (er + 8 * qq) * 36 + ec4);
            const long long row = out_row(wr * 32 + er + 8 * qq, j);
            if (row >= 0) *reinterpret_cast<f32x4*>(g.acts + row * C + ch0) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    });
}

hipError_t launch_wino_fused2(const WinoFused2Args& a, hipStream_t st) {
    constexpr int BM = 64, BN = 128;
    const size_t lds = (size_t)(2 * 6 * BM * 16 + 3 * BN * 16) * sizeof(float);
    if (a.Mq % BM != 0 || a.phase_rows % BM != 0 || a.Mq % a.phase_rows != 0) return hipErrorInvalidValue;
    static PerDeviceOnce attr_set;
    if (hipError_t e = set_max_dyn_lds_once((const void*)wino4_fused2_kernel, lds, attr_set); e != hipSuccess) return e;
    const int numMt8 = (a.Mq / BM + 7) / 8 * 8;
    hipLaunchKernelGGL(wino4_fused2_kernel, dim3(numMt8 * (2 * C / BN)), dim3(256), lds, st, a);
    return hipGetLastError();
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }
}  // namespace

// group rows per block: frame groups (dilations >= 32) B x 4 ceil(T / 16) per phase, mixed groups (dilation 16) B x ceil(T / 2)
// per p0, both padded to the 128-row tile
// (padded to the row tile of the kernel that runs them: 64 rows for the fused kernels, 128 for the three-pass form's GEMM --
//  config 2's frame groups: 1 600 rows per phase instead of 1 664, i.e. 4 % less work in three of the seven layers)
static inline int group_row_tile(int form) { return form == 2 ? 128 : 64; }
static inline int frame_group_rows(int BT, int T, int form) {
    const int g = group_row_tile(form);
    return ((BT / T) * frame_groups_per_utt(T) + g - 1) / g * g;
}
static inline int mixed_group_rows(int BT, int T, int form) {
    const int g = group_row_tile(form);
    return ((BT / T) * mixed_groups_per_utt(T) + g - 1) / g * g;
}

// Per-layer operands (on the first call that takes this path): G for layers 1 .. 7 of every flow; V for the phase groups
// (dilations 2 - 8: weight combinations over the group's four phases) and the mixed groups (dilation 16).  The frame groups
// (dilations >= 32) need none in the fused form -- their products' conditioning weights are chunk ranges of cond_Bt itself --
// and six column-selected copies per phase ([32][6][1024][224], 176 MB per layer) in the three-pass form, built only when that
// form is asked for (`legacy_frames`).  3.6 GB in all (three-pass: + 6.3 GB).  A failed allocation frees what this call built.
int waveglow_build_wino(tts_hip_engine* e, bool legacy_frames) {
    WaveGlowDev& wg = e->wg;
    if (wg.wino_ready && (!legacy_frames || wg.wino_legacy_ready)) return TTS_HIP_OK;
    hipStream_t st = e->stream;
    std::vector<void*> fresh;                              // this call's allocations (moved to wg.allocs on success)
    auto fail = [&](int rc) {
        (void)hipStreamSynchronize(st);
        for (void* p : fresh) (void)hipFree(p);
        for (int k = 0; k < 12; ++k)
            for (int i = 1; i < 8; ++i) {
                WgLayerDev& ly = wg.flow[k].layer[i];
                if (!wg.wino_ready) ly.wino_G = ly.wino_V = nullptr;
                if (!wg.wino_legacy_ready) ly.wino_Vf = nullptr;
            }
        return rc;
    };
    for (int k = 0; k < 12; ++k)
        for (int i = 1; i < 8; ++i) {
            WgLayerDev& ly = wg.flow[k].layer[i];
            const int d = 1 << i;
            int rc;
            if (!wg.wino_ready) {
                if ((rc = dev_alloc(e, (size_t)6 * 2 * C * C, &ly.wino_G, fresh, false))) return fail(rc);
                hipLaunchKernelGGL(wino4_weights_kernel, dim3(blocks_for(2 * C * C)), dim3(256), 0, st, ly.in_Bt, ly.wino_G);
                if (d == 16) {                             // products 1 .. 4 carry the whole conditioning (K = 320)
                    if ((rc = dev_alloc(e, (size_t)16 * 4 * 2 * C * KMEL, &ly.wino_V, fresh, false))) return fail(rc);
                    hipLaunchKernelGGL(wino4_cond_weights_mixed_kernel, dim3(blocks_for((long long)16 * 2 * C * KMEL)), dim3(256), 0,
                                       st, ly.cond_Bt, ly.wino_V);
                } else if (d < NPH) {                      // eight group phases
                    if ((rc = dev_alloc(e, (size_t)(NPH / 4) * 6 * 2 * C * K4, &ly.wino_V, fresh, false))) return fail(rc);
                    hipLaunchKernelGGL(wino4_cond_weights_kernel, dim3(blocks_for((long long)(NPH / 4) * 6 * 2 * C * K4)), dim3(256),
                                       0, st, ly.cond_Bt, ly.wino_V, d);
                }
            }
            if (legacy_frames && !wg.wino_legacy_ready && d >= NPH) {
                if ((rc = dev_alloc(e, (size_t)NPH * 6 * 2 * C * K4, &ly.wino_Vf, fresh, false))) return fail(rc);
                hipLaunchKernelGGL(wino4_cond_weights_frames_kernel, dim3(blocks_for((long long)NPH * 6 * 2 * C * K4)), dim3(256), 0,
                                   st, ly.cond_Bt, ly.wino_Vf);
            }
            if (hipError_t herr = hipGetLastError(); herr != hipSuccess)
                return fail(set_err(e, TTS_HIP_EHIP, "waveglow_build_wino: %s", hipGetErrorString(herr)));
        }
    if (hipError_t herr = hipStreamSynchronize(st); herr != hipSuccess)
        return fail(set_err(e, TTS_HIP_EHIP, "waveglow_build_wino: %s", hipGetErrorString(herr)));
    wg.allocs.insert(wg.allocs.end(), fresh.begin(), fresh.end());
    wg.wino_ready = true;
    if (legacy_frames) wg.wino_legacy_ready = true;
    return TTS_HIP_OK;
}

// Layout of the per-call mel planes (floats): [phase groups: 6][PR][224] | 3 x [frame groups: 6][PRq][224] | [mixed: 4][PRm][320]
struct MelPlanes {
    size_t phases, frames[3], mixed, total;
    MelPlanes(int PR, int BT, int T, int form) {
        const size_t PRq = (size_t)frame_group_rows(BT, T, form), PRm = (size_t)mixed_group_rows(BT, T, form);
        phases = 0;
        frames[0] = (size_t)6 * PR * K4;
        frames[1] = frames[0] + 6 * PRq * K4;
        frames[2] = frames[1] + 6 * PRq * K4;
        mixed = frames[2] + 6 * PRq * K4;
        total = mixed + 4 * PRm * KMEL;
    }
};

// Workspace and the mel planes of one call (the mel does not change across layers and flows)
int waveglow_wino_begin(tts_hip_engine* e, const float* d_mel, int PR, int BT, int T, int form) {
    const bool three_pass = form == 2, need_U = form != 1;                 // form 1 transforms its inputs on the fly: no U planes
    WaveGlowDev& wg = e->wg;
    hipStream_t st = e->stream;
    const int PRq = frame_group_rows(BT, T, form), PRm = mixed_group_rows(BT, T, form);
    // U (and the three-pass form's P): six planes of 8 PR (phase groups), 32 PRq (frame groups) or 16 PRm (mixed groups) rows
    const size_t rows = 6 * (size_t)std::max(std::max((long long)(NPH / 4) * PR, (long long)NPH * PRq), (long long)16 * PRm);
    const MelPlanes mp(PR, BT, T, form);
    auto room = [&](DevBuf& b, size_t bytes) -> int {       // out of memory is its own status: the caller keeps the direct form
        const hipError_t err = b.ensure(bytes);
        if (err == hipSuccess) return TTS_HIP_OK;
        if (err == hipErrorOutOfMemory) (void)hipGetLastError();
        return set_err(e, err == hipErrorOutOfMemory ? TTS_HIP_ENOMEM : TTS_HIP_EHIP, "waveglow_wino_begin: hipMalloc(%zu bytes) -> %s",
                       bytes, hipGetErrorString(err));
    };
    int rc;
    if (need_U && (rc = room(wg.wino_U, rows * C * 4))) return rc;
    if (three_pass && (rc = room(wg.wino_P, rows * 2 * C * 4))) return rc;
    if ((rc = room(wg.wino_mel, mp.total * 4))) return rc;
    float* base = wg.wino_mel.f();
    hipLaunchKernelGGL(wino4_mel_planes_kernel, dim3(blocks_for((long long)PR * K4)), dim3(256), 0, st, d_mel, base + mp.phases, PR, BT, T);
    for (int si = 0; si < 3; ++si)
        hipLaunchKernelGGL(wino4_mel_planes_frames_kernel, dim3(blocks_for((long long)6 * PRq * K4)), dim3(256), 0, st, d_mel,
                           base + mp.frames[si], 1 << si, PRq, BT, T);
    hipLaunchKernelGGL(wino4_mel_planes_mixed_kernel, dim3(blocks_for((long long)PRm * KMEL)), dim3(256), 0, st, d_mel,
                       base + mp.mixed, PRm, BT, T);
    HIPCHK(e, hipGetLastError());
    return TTS_HIP_OK;
}

// One WN in-layer step (layer i >= 1 of a flow): acts_i = gate(conv_d(x) + cond + b) through the three passes of the header
int waveglow_wino_layer(tts_hip_engine* e, const WgLayerDev& ly, int i, const float* x, float* acts_i, int PR, int BT, int T) {
    WaveGlowDev& wg = e->wg;
    hipStream_t st = e->stream;
    const int d = 1 << i;
    float* U = wg.wino_U.f();
    float* P = wg.wino_P.f();
    const MelPlanes mp(PR, BT, T, wg.form_mode);
    const int PRq = frame_group_rows(BT, T, wg.form_mode), PRm = mixed_group_rows(BT, T, wg.form_mode);
    const bool phases = d <= 8, mixed = d == 16;
    const long long Mq = phases ? (long long)(NPH / 4) * PR : mixed ? (long long)16 * PRm : (long long)NPH * PRq;
    const bool no_prepass = wg.form_mode == 1;             // form 1 (default): input transform inside the GEMM's operand reads
    if (!no_prepass) hipLaunchKernelGGL(wino4_prepass_kernel, dim3(blocks_for(Mq * (C / 4))), dim3(256), 0, st, x, U, d, PR, BT, T, Mq);
    if (wg.form_mode != 2) {                               // fused GEMM + output transform + gate (form 2: the three passes)
        WinoFusedArgs a{};
        a.U = U;
        a.uplane = Mq * C;
        a.G = ly.wino_G;
        a.gplane = (long long)2 * C * C;
        a.bias = ly.in_bias;
        a.acts = acts_i;
        a.Mq = (int)Mq;
        a.phase_rows = phases ? PR : mixed ? PRm : PRq;
        a.kind = phases ? 0 : mixed ? 2 : 1;
        a.d = d;
        a.PR = PR;
        a.BT = BT;
        a.T = T;
        unsigned ccfg[6] = {0, 0, 0, 0, 0, 0};
        auto cc = [](unsigned n1, unsigned b1, unsigned n2 = 0, unsigned b2 = 0) { return n1 | b1 << 5 | n2 << 9 | b2 << 12; };
        if (phases) {                                      // V [8 group phases][6][1024][224]: [A | B | 0], [A | C], [B | C | 0]
            a.mel = wg.wino_mel.f() + mp.phases;
            a.mplane = (long long)PR * K4;
            a.ldm = K4;
            a.V = ly.wino_V;
            a.vplane = (long long)2 * C * K4;
            a.strideVp = 6 * a.vplane;
            a.ldv = K4;
            for (int p = 0; p < 6; ++p) ccfg[p] = cc(p == 1 || p == 2 ? 14 : 13, 0);
        } else if (mixed) {                                // products 1 .. 4: K = 320 against V [16][4][1024][320]
            a.mel = wg.wino_mel.f() + mp.mixed;
            a.mplane = (long long)PRm * KMEL;
            a.ldm = KMEL;
            a.V = ly.wino_V;
            a.vplane = (long long)2 * C * KMEL;
            a.strideVp = 4 * a.vplane;
            a.ldv = KMEL;
            a.pofs = 1;
            for (int p = 1; p < 5; ++p) ccfg[p] = cc(KMEL / 16, 0);
        } else {                                           // frame groups: chunk ranges of cond_Bt [32 phases][1024][320]
            a.mel = wg.wino_mel.f() + mp.frames[i - 5];
            a.mplane = (long long)PRq * K4;
            a.ldm = K4;
            a.V = ly.cond_Bt;
            a.vplane = 0;
            a.strideVp = (long long)2 * C * KMEL;
            a.ldv = KMEL;
            ccfg[0] = ccfg[5] = cc((SA + SB) / 16, 0);                         // [A | B]
            ccfg[1] = ccfg[2] = cc(SA / 16, 0, SC / 16, (SA + SB) / 16);       // [A | C]
            ccfg[3] = ccfg[4] = cc((SB + SC) / 16, SA / 16);                   // [B | C]
        }
        for (int p = 0; p < 6; ++p) (p < 4 ? a.cfg_lo : a.cfg_hi) |= (unsigned long long)ccfg[p] << (16 * (p & 3));
        if (no_prepass) {
            WinoFused2Args b{};
            b.x = x;
            b.G = a.G; b.gplane = a.gplane;
            b.mel = a.mel; b.mplane = a.mplane; b.ldm = a.ldm;
            b.V = a.V; b.vplane = a.vplane; b.strideVp = a.strideVp; b.ldv = a.ldv;
            b.pofs = a.pofs;
            b.cfg_lo = a.cfg_lo; b.cfg_hi = a.cfg_hi;
            b.bias = a.bias; b.acts = a.acts;
            b.Mq = a.Mq; b.phase_rows = a.phase_rows; b.kind = a.kind; b.d = a.d; b.PR = a.PR; b.BT = a.BT; b.T = a.T;
            timing_begin(e, 0);
            HIPCHK(e, launch_wino_fused2(b, st));
            timing_end(e);
            return TTS_HIP_OK;
        }
        timing_begin(e, 0);
        // form 3 (measurement): the fused kernel behind the pre-pass (64 x 128 tiles, two blocks per CU; measured at config 2
        // on one box: three passes 433 ms per step, this 415, without the pre-pass 409; 8-wave 128 x 128 blocks: 425)
        HIPCHK(e, (launch_wino_fused<2, 2, 3, 2>(a, st)));
        timing_end(e);
        return TTS_HIP_OK;
    }
    GemmArgs g{};
    g.M = (int)Mq;
    g.N = 2 * C;
    g.nphase = phases ? NPH / 4 : mixed ? 16 : NPH;
    g.phase_rows = phases ? PR : mixed ? PRm : PRq;
    g.frames = phases ? BT : mixed ? (BT / T) * mixed_groups_per_utt(T) : (BT / T) * frame_groups_per_utt(T);
    g.L = g.phase_rows;
    g.ldb = C;
    g.mode = EPI_LINEAR;
    g.act = ACT_NONE;
    g.split = 2 * C;
    g.ld0 = 2 * C;
    g.wide_epi = 1;
    const long long uplane = Mq * C, gplane = (long long)2 * C * C, pplane = Mq * 2 * C;
    if (!mixed) {                                          // six slices of K = 512 + 224
        g.nseg = 2;
        g.seg[0] = ASeg{U, C, 0, C, C, SEG_ROWS_Z, 0, Mq, 1};
        g.seg[1] = ASeg{wg.wino_mel.f() + (phases ? mp.phases : mp.frames[i - 5]), K4, 0, K4, K4, SEG_FRAME_Z, 0, (long long)g.phase_rows, 0};
        g.Bt = ly.wino_G;
        g.strideBz = gplane;
        g.Bt2 = phases ? ly.wino_V : ly.wino_Vf;
        g.ldb2 = K4;
        g.strideB2p = (long long)6 * 2 * C * K4;
        g.strideB2z = (long long)2 * C * K4;
        g.out0 = P;
        g.strideOutZ = pplane;
        timing_begin(e, 0);
        HIPCHK(e, phases && PR % 256 == 0 ? gemm_wn_wino(g, 6, st) : gemm_wn_wino_128(g, 6, st));
        timing_end(e);
    } else {
        // products 1 .. 4: K = 512 + 320 (they carry the whole conditioning) ...
        g.nseg = 2;
        g.seg[0] = ASeg{U + uplane, C, 0, C, C, SEG_ROWS_Z, 0, Mq, 1};
        g.seg[1] = ASeg{wg.wino_mel.f() + mp.mixed, KMEL, 0, KMEL, KMEL, SEG_FRAME_Z, 0, (long long)PRm, 0};
        g.Bt = ly.wino_G + gplane;
        g.strideBz = gplane;
        g.Bt2 = ly.wino_V;
        g.ldb2 = KMEL;
        g.strideB2p = (long long)4 * 2 * C * KMEL;
        g.strideB2z = (long long)2 * C * KMEL;
        g.out0 = P + pplane;
        g.strideOutZ = pplane;
        timing_begin(e, 0);
        HIPCHK(e, gemm_wn_wino_128(g, 4, st));
        timing_end(e);
        // ... products 0 and 5: K = 512, planes 0 and 5 (z stride of five planes)
        GemmArgs h = g;
        h.nseg = 1;
        h.seg[0] = ASeg{U, C, 0, C, C, SEG_ROWS_Z, 0, 5 * Mq, 0};
        h.Bt = ly.wino_G;
        h.strideBz = 5 * gplane;
        h.Bt2 = nullptr;
        h.out0 = P;
        h.strideOutZ = 5 * pplane;
        timing_begin(e, 0);
        HIPCHK(e, gemm_wn_wino_128(h, 2, st));
        timing_end(e);
    }
    hipLaunchKernelGGL(wino4_combine_kernel, dim3(blocks_for(Mq * (C / 4))), dim3(256), 0, st, P, ly.in_bias, acts_i, d, PR, BT, T, Mq);
    HIPCHK(e, hipGetLastError());
    return TTS_HIP_OK;
}
