// waveglow.hip -- WaveGlow flow inversion on gfx950.
//
// Replaces /root/reference/architectures/waveglow_arch.py:244-306 (WaveGlow.infer), :105-141 (WaveglowBlock.call) and
// architectures/layers/invertible_conv.py:41-51 (Invertible1x1Conv reverse).
//
// HBM layout (all float32, channels-last).  Positions are kept PHASE-MAJOR inside the engine: a position is a group of
// 8 samples, l = 32 * t + p (t = mel frame, p = phase 0..31); row m' = p * PR + f with f = b * T + t the global frame
// index and PR = B*T rounded up to the 256-row tile.  A conv tap l +- d then stays a constant row shift per tile
// (phase' = (p +- d) mod 32, frame carry = floor((p +- d) / 32)) and the phase is uniform per tile, which is what the
// low-rank conditioning below needs.  M' = 32 * PR rows:
//   x     [M'][512]   WN residual stream, updated in place by the residual GEMM epilogue
//   acts  [8][M'][512] gated activations tanh * sigmoid of the 8 layers of the current flow (in-layer GEMM epilogue)
//   a0p   [M'][16]    [audio_0 | 1 | 0..]: operand of the first layer of a flow (start conv composed into its taps)
//   audio [M'][8]     current flow state in the first n_rem columns; the last flow writes the caller's [B][L*8] directly
// Per flow: start (VALU) -> 8 x { in-layer implicit GEMM (K = 3 taps * 512 + 4 * 80 mel, N = 1024, gate epilogue),
// residual GEMM (K = 512, N = 512; not for the last layer) } -> folded skip/end + affine inverse + inverse 1x1 conv.
//
// Conditioning folding (exact algebra, load time): the reference upsamples the mel with a transposed conv
// (k 1024, stride 256), regroups 8 samples x 80 channels into 640 channels and applies a 640 -> 1024 1x1 conv per layer
// (waveglow_arch.py:245-253,125).  A group at phase p only sees mel frames t-3..t, so
//     cond_i[l] = V_{i,p} @ [mel[t], mel[t-1], mel[t-2], mel[t-3]] + const,   V_{i,p} = W_cond_i @ U_p  (1024 x 320),
// i.e. K = 320 instead of 640 (-14.7 % of the in-layer FLOPs), no upsampling pass and no [M][640] spectrogram in HBM.
// The price is 32 per-phase copies of the conditioning weights (42 MB per layer, 4 GB in all), streamed once per launch.
//
// Skip path folding (exact algebra, done once at load time): the reference sums the skip halves of the 8 res_skip convs
// and feeds the sum to the `end` 1x1 conv (waveglow_arch.py:129-141).  Both are linear, so
//     end(sum_i skip_i) = sum_i acts_i @ (W_skip_i @ W_end) + (sum_i b_skip_i) @ W_end + b_end .
// The 512 -> 512 skip GEMMs (and the whole res_skip conv of the last layer) disappear -- 9.6 % of the WN FLOPs and the
// read-modify-write of a [M][512] skip buffer per layer -- and are replaced by one [M, 8*512] x [8*512, 2h] product per
// flow (h <= 4), computed by the HBM-bound wn_end_fold_kernel straight from the stored activations.
#include "engine.h"
#include "gemm_f32.h"

#include <cmath>
#include <cstdlib>

using namespace ttsgemm;

namespace {

constexpr int C = 512;        // n_channels
constexpr int NCOND = 640;    // n_mel * n_group (reference layout of the conditioning input)
constexpr int KCONV = 3 * C;  // taps part of the in-layer K
constexpr int KCONV0 = 3 * 16;// first layer of a flow: taps act on [audio_0 | 1] (16-float rows)
constexpr int KMEL = 4 * 80;  // folded conditioning: 4 mel frames x 80 channels
constexpr int NPH = 32;       // phases (sample groups per mel frame)

// dst[n][koff + k] = src[k * src_ld + perm(n)]  for k < K   (Keras [K][N] kernel slice -> Bt rows)
// perm: 0 identity; 1 WN gate interleave (per group of 64 rows: 32 tanh channels then their 32 sigmoid partners, so a
//       wave's pair of adjacent 32-column MFMA tiles holds matching pre-activations for every N tile >= 64 columns)
// taps > 1: src is [taps][K/taps][N] and the K axis of dst is tap-interleaved in chunks of `bk`:
//   dst k = (c / bk) * taps * bk + tap * bk + c % bk     (matches gemm_f32_kernel's NI = taps tile order)
__global__ void pack_bt_kernel(const float* __restrict__ src, int K, int src_ld, float* __restrict__ dst, int N,
                               long long ldb, int koff, int perm, int taps, int bk) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx % K);
    int kd = k;
    if (taps > 1) {
        const int cpt = K / taps, tap = k / cpt, c = k % cpt;
        kd = (c / bk) * taps * bk + tap * bk + c % bk;
    }
    int sn = n;
    if (perm == 1) {
        const int grp = n >> 6, q = n & 63;
        sn = q < 32 ? grp * 32 + q : C + grp * 32 + (q - 32);
    }
    dst[(long long)n * ldb + koff + kd] = src[(long long)k * src_ld + sn];
}

__global__ void pack_bias_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dst,
                                 int N, int perm) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int sn = n;
    if (perm == 1) {
        const int grp = n >> 6, q = n & 63;
        sn = q < 32 ? grp * 32 + q : C + grp * 32 + (q - 32);
    }
    dst[n] = a[sn] + (b ? b[sn] : 0.f);
}

// Transposed-conv kernel [1024][80 out][80 in] -> UT[p][q*80 + j][c*8 + g] = W[(p*8 + g) + 256 q][c][j]
// (the "Bt" operand of V_{i,p} = WcT_i @ U_p: row = mel-window input (q, j), K = regrouped channel c*8 + g)
__global__ void pack_ut_kernel(const float* __restrict__ w, float* __restrict__ dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)NPH * KMEL * NCOND;
    if (idx >= total) return;
    const int k = (int)(idx % NCOND);
    const int r = (int)((idx / NCOND) % KMEL);
    const int p = (int)(idx / ((long long)NCOND * KMEL));
    const int c = k >> 3, gidx = k & 7, q = r / 80, j = r % 80;
    dst[idx] = w[((long long)(p * 8 + gidx + 256 * q) * 80 + c) * 80 + j];
}

// bias[n] += sum_k WcT[n][k] * b_up[k >> 3]     (upsampling bias pushed through the conditioning conv)
__global__ void cond_bias_kernel(const float* __restrict__ wct, const float* __restrict__ b_up, float* __restrict__ bias) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= 2 * C) return;
    float acc = 0.f;
    for (int k = 0; k < NCOND; ++k) acc = fmaf(wct[(long long)n * NCOND + k], b_up[k >> 3], acc);
    bias[n] += acc;
}

// ---- fp16 operand builders (run once, on first use of the fp16 path, from the packed fp32 device copies)
// `lo` (may be null) receives the second plane of the split-fp16 mode: lo = fp16(v - fp16(v))
__device__ __forceinline__ void put_split(_Float16* dst, _Float16* lo, long long i, float v) {
    const _Float16 hv = (_Float16)v;
    dst[i] = hv;
    if (lo) lo[i] = (_Float16)(v - (float)hv);
}
__global__ void cvt_half_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long long n,
                                _Float16* __restrict__ lo = nullptr) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) put_split(dst, lo, i, src[i]);
}
// in-layer taps: fp32 [1024][1536] tap-interleaved in chunks of 16 -> fp16 [1024][1536] tap-interleaved in chunks of 32
// (the fp16 kernel's K step is 32 halfs = 64-byte LDS rows, like 16 floats)
__global__ void cvt_taps_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, _Float16* __restrict__ lo = nullptr) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)2 * C * KCONV) return;
    const int n = (int)(i / KCONV), r = (int)(i % KCONV);
    const int tap = r / C, c = r % C;
    const int ks = (c / 16) * 48 + tap * 16 + c % 16, kd = (c / 32) * 96 + tap * 32 + c % 32;
    put_split(dst, lo, (long long)n * KCONV + kd, src[(long long)n * KCONV + ks]);
}
// first layer of a flow: fp32 [1024][3*16] -> fp16 [1024][3*32] (a0p rows are 32 halfs)
__global__ void cvt_taps0_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, _Float16* __restrict__ lo = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * C * 96) return;
    const int n = i / 96, r = i % 96, tap = r / 32, jj = r % 32;
    put_split(dst, lo, i, jj < 16 ? src[n * KCONV0 + tap * 16 + jj] : 0.f);
}

// fp16 modes: the conditioning operand of frame f is the contiguous window [mel_t | mel_{t-1} | mel_{t-2} | mel_{t-3}]
// (320 halfs = 10 K steps; four separate 80-wide segments would each be padded to 96 = 12 steps), zeros before the start
// of the utterance.  `lo` (may be null) receives the second plane of the split mode.
__global__ void mel_window_kernel(const float* __restrict__ mel, _Float16* __restrict__ dst, _Float16* __restrict__ lo,
                                  int BT, int T) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)BT * KMEL) return;
    const int f = (int)(i / KMEL), r = (int)(i % KMEL), q = r / 80, j = r % 80;
    const int t = f % T;
    put_split(dst, lo, i, t - q >= 0 ? mel[(long long)(f - q) * 80 + j] : 0.f);
}

// test hook (tts_hip_waveglow_probe_acts): phase-major rows m' = p * PR + b * T + t of one layer's gated activations ->
// natural order [B][T * 32][512] (position l = 32 t + p)
__global__ void probe_acts_kernel(const float* __restrict__ acts, float* __restrict__ out, int PR, int BT, int T) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // one float4 of one position
    if (idx >= (long long)BT * NPH * (C / 4)) return;
    const int c = (int)(idx % (C / 4)) * 4;
    const long long pos = idx / (C / 4);               // b * T * 32 + l
    const int p = (int)(pos % NPH);
    const long long f = pos / NPH;                     // b * T + t
    *reinterpret_cast<f32x4*>(out + pos * C + c) = *reinterpret_cast<const f32x4*>(acts + ((long long)p * PR + f) * C + c);
}

// audio[m'][0..3] = sigma * z[natural m][0..3]  (z null => zeros); m' = p * PR + f  <->  m = f * 32 + p
__global__ void init_audio_kernel(const float* __restrict__ z, float sigma, float* __restrict__ audio, int PR, int BT) {
    const long long mp = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (mp >= (long long)NPH * PR) return;
    const int p = (int)(mp / PR), f = (int)(mp % PR);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (z && f < BT) {
        v = *reinterpret_cast<const f32x4*>(z + ((long long)f * NPH + p) * 8);
        v *= sigma;
    }
    *reinterpret_cast<f32x4*>(audio + mp * 8) = v;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(audio + mp * 8 + 4) = zero;
}

// x[m][c] = sum_{j < h} audio[m][j] * w[j][c] + b[c]      (start 1x1 conv, waveglow_arch.py:108)
// Also writes a0p[m][..] = [audio_0 (h values) | 1 | 0 ...]: the operand of the first WN layer, whose dilated conv is
// composed with the start conv at load time (the constant 1 carries the start bias through the zero padding).
// HALF: x stays fp32 (master copy for the residual accumulation) and additionally gets an fp16 shadow x16 (the GEMM
// operand); a0p is written as 32 halfs per row.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <bool HALF>
__global__ void wn_start_kernel(const float* __restrict__ audio, const float* __restrict__ w,
                                const float* __restrict__ b, float* __restrict__ x, void* __restrict__ a0p_v,
                                _Float16* __restrict__ x16, long long M, int h, int split = 0) {
    // split != 0 (split-fp16 mode): the fp16 arrays are [2 planes][M][..]; plane 1 gets v - fp16(v)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one float4 of channels
    if (idx >= M * (C / 4)) return;
    const long long m = idx / (C / 4);
    const int c = (int)(idx % (C / 4)) * 4;
    if (c < (HALF ? 32 : 16)) {                         // the first threads of the row also write the a0p row
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = c + j;
            v[j] = col < h ? audio[m * 8 + col] : (col == h ? 1.f : 0.f);
        }
        if constexpr (HALF) {
            const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *reinterpret_cast<f16x4*>((_Float16*)a0p_v + m * 32 + c) = hv;
            if (split) {
                const f16x4 lv = {(_Float16)(v[0] - (float)hv[0]), (_Float16)(v[1] - (float)hv[1]),
                                  (_Float16)(v[2] - (float)hv[2]), (_Float16)(v[3] - (float)hv[3])};
                *reinterpret_cast<f16x4*>((_Float16*)a0p_v + M * 32 + m * 32 + c) = lv;
            }
        } else {
            const f32x4 fv = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>((float*)a0p_v + m * 16 + c) = fv;
        }
    }
    f32x4 acc = *reinterpret_cast<const f32x4*>(b + c);
    // the reference accumulates the dot product first and adds the bias last (Conv1D = conv + bias)
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < h; ++j) {
        const float a = audio[m * 8 + j];
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + j * C + c);
        s += a * wv;
    }
    acc += s;
    *reinterpret_cast<f32x4*>(x + m * C + c) = acc;
    if constexpr (HALF) {
        const f16x4 hv = {(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};
        *reinterpret_cast<f16x4*>(x16 + m * C + c) = hv;
        if (split) {
            const f16x4 lv = {(_Float16)(acc[0] - (float)hv[0]), (_Float16)(acc[1] - (float)hv[1]),
                              (_Float16)(acc[2] - (float)hv[2]), (_Float16)(acc[3] - (float)hv[3])};
            *reinterpret_cast<f16x4*>(x16 + M * C + m * C + c) = lv;
        }
    }
}

// Folded skip/end conv + affine inverse + inverse 1x1 conv (+ early-z prepend), RPW positions per wave.
//   out[m][o] = sum_layer sum_c acts[layer][m][c] * wfold[layer][o][c] + bfold[o]          (waveglow_arch.py:129-141)
//   audio_1 = (audio_1 - b) / exp(s); audio = [audio_0, audio_1] @ inv; prepend sigma * z_early   (:284-304)
// Lane l owns channels 4l..4l+3 and 256+4l..256+4l+3 (every wave-level load is one contiguous 1 KiB run); per layer
// the lane's 8x8 slice of wfold sits in registers and is reused for the RPW rows; the RPW*8 partial sums are reduced with the lane-halving exchange (63 shuffles for 64 values).
constexpr int RPW = 8;
template <bool HALF, bool SPLIT = false>
__global__ __launch_bounds__(256) void wn_end_fold_kernel(const void* __restrict__ acts_v, long long layer_stride,
                                                          const float* __restrict__ wfold,
                                                          const float* __restrict__ bfold,
                                                          const float* __restrict__ inv, float* __restrict__ audio_io,
                                                          float* __restrict__ audio_out, int out_natural,
                                                          const float* __restrict__ z, int zoff, int n_early,
                                                          float sigma, long long M, int h, int PR, int BT,
                                                          long long lo_plane = 0) {
    // lo_plane != 0 (split-fp16 mode): activation = hi + lo, lo at + lo_plane halfs
    // fp16 variants: the folded weights of a layer ([8 outputs][512]: 16 KB, the same for every wave) are staged in LDS once
    // per block and layer (double buffered; the next layer's rows are requested before this layer's arithmetic).  With
    // 8 consecutive channels per lane a lane's two weight float4 sit 32 B apart, and pulling those slices through the L1
    // per wave cost more than the halved activation stream saved (fp16 0.68 -> 0.42 ms, split 1.14 -> 0.90 ms).
    __shared__ __attribute__((aligned(16))) float wsm[HALF ? 2 : 1][HALF ? 8 * C : 4];
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long m0 = wave * RPW;
    const bool wave_valid = m0 < M;                       // (M is a multiple of 4 * RPW; with barriers in the fp16 variants
    if (!HALF && !wave_valid) return;                     //  nobody may leave early)
    const int cch = 2 * h;
    float acc[RPW * 8];                                   // index r * 8 + o
#pragma unroll
    for (int i = 0; i < RPW * 8; ++i) acc[i] = 0.f;
    if constexpr (HALF) {
        const f32x4* src = reinterpret_cast<const f32x4*>(wfold);
#pragma unroll
        for (int i = 0; i < 4; ++i) reinterpret_cast<f32x4*>(wsm[0])[threadIdx.x + i * 256] = src[threadIdx.x + i * 256];
        __syncthreads();
    }
    for (int layer = 0; layer < 8; ++layer) {
        f32x4 w0[8], w1[8], a0[RPW], a1[RPW], wnext[HALF ? 4 : 1];
        if constexpr (HALF) {
            if (layer + 1 < 8) {
                const f32x4* src = reinterpret_cast<const f32x4*>(wfold + (long long)(layer + 1) * 8 * C);
#pragma unroll
                for (int i = 0; i < 4; ++i) wnext[i] = src[threadIdx.x + i * 256];
            }
        }
        // fp32 activations: lane owns channels 4l..4l+3 and 256+4l..; fp16 activations: channels 8l..8l+7, so that one
        // 16-byte load per plane and row fetches them
        const float* wl = HALF ? wsm[layer & 1] + lane * 8 : wfold + ((long long)layer * 8) * C + lane * 4;
        constexpr int W1 = HALF ? 4 : C / 2;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            w0[o] = *reinterpret_cast<const f32x4*>(wl + o * C);
            w1[o] = *reinterpret_cast<const f32x4*>(wl + o * C + W1);
        }
        if constexpr (HALF) {
            // all loads of the layer first (a run-time test of `lo_plane` between them made the compiler wait for every
            // load before issuing the next: 1.9 ms instead of 0.6 ms), conversions afterwards
            typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
            f16x8v hv[RPW], lv[SPLIT ? RPW : 1];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const long long m = m0 + r < M ? m0 + r : M - 1;   // clamp: tail rows are computed but never stored
                const _Float16* al = (const _Float16*)acts_v + layer * layer_stride + m * C + lane * 8;
                hv[r] = *reinterpret_cast<const f16x8v*>(al);
                if constexpr (SPLIT) lv[r] = *reinterpret_cast<const f16x8v*>(al + lo_plane);
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                a0[r] = f32x4{(float)hv[r][0], (float)hv[r][1], (float)hv[r][2], (float)hv[r][3]};
                a1[r] = f32x4{(float)hv[r][4], (float)hv[r][5], (float)hv[r][6], (float)hv[r][7]};
                if constexpr (SPLIT) {
                    a0[r] += f32x4{(float)lv[r][0], (float)lv[r][1], (float)lv[r][2], (float)lv[r][3]};
                    a1[r] += f32x4{(float)lv[r][4], (float)lv[r][5], (float)lv[r][6], (float)lv[r][7]};
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const long long m = m0 + r < M ? m0 + r : M - 1;   // clamp: tail rows are computed but never stored
                const float* al = (const float*)acts_v + layer * layer_stride + lane * 4;
                a0[r] = *reinterpret_cast<const f32x4*>(al + m * C);
                a1[r] = *reinterpret_cast<const f32x4*>(al + m * C + C / 2);
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float p = acc[r * 8 + o];
#pragma unroll
                for (int j = 0; j < 4; ++j) p = fmaf(a0[r][j], w0[o][j], p);
#pragma unroll
                for (int j = 0; j < 4; ++j) p = fmaf(a1[r][j], w1[o][j], p);
                acc[r * 8 + o] = p;
            }
        if constexpr (HALF) {
            if (layer + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) reinterpret_cast<f32x4*>(wsm[(layer + 1) & 1])[threadIdx.x + i * 256] = wnext[i];
            }
            __syncthreads();
        }
    }
    if (!wave_valid) return;
    // 64 values over 64 lanes: after masks 32..1 lane l holds the full sum of index l = r * 8 + o
#pragma unroll
    for (int half = RPW * 4, msk = 32; half >= 1; half >>= 1, msk >>= 1) {
        const bool hi = (lane & msk) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            float a_lo = acc[i], a_hi = acc[i + half];
            // opaque copies: otherwise instcombine turns select(load, load) into a dynamically indexed load of the
            // register array, which lowers to a compare/select chain over every element (seen for <7, 8>: 3.3 k extra VALU)
            asm volatile("" : "+v"(a_lo), "+v"(a_hi));
            const float send = hi ? a_lo : a_hi;
            const float keep = hi ? a_hi : a_lo;
            acc[i] = keep + __shfl_xor(send, msk, 64);
        }
    }
    const float mine = acc[0] + bfold[lane & 7];
    // lanes 8r .. 8r+7 hold out[r][0..7]; lane 8r finishes position m0 + r
    float out[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) out[o] = __shfl(mine, (lane & ~7) + o, 64);
    const long long m = m0 + (lane >> 3);
    const int ph = (int)(m / PR), fr = (int)(m % PR);          // phase-major row -> (phase, frame)
    const long long mnat = (long long)fr * NPH + ph;           // natural position index b * L + t * 32 + p
    if ((lane & 7) == 0 && m < M && fr < BT) {
        float a[8], y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = j < cch ? audio_io[m * 8 + j] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < h) y[j] = a[j];
            else if (j < cch) y[j] = (a[j] - out[j - h]) / expf(out[j]);
            else y[j] = 0.f;
        }
        float res[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float t = 0.f;
            if (c < cch) {
                for (int j = 0; j < cch; ++j) t = fmaf(y[j], inv[j * cch + c], t);
            }
            res[c] = t;
        }
        float* dst = audio_out + (out_natural ? mnat : m) * 8;
        for (int j = 0; j < n_early; ++j) dst[j] = z ? sigma * z[mnat * 8 + zoff + j] : 0.f;
        for (int c = 0; c < cch; ++c) dst[n_early + c] = res[c];
    }
}

int pack_bt(tts_hip_engine* e, const float* d_src, int K, int src_ld, float* dst, int N, long long ldb, int koff,
            int perm, int taps = 1, int bk = 0) {
    const long long total = (long long)N * K;
    hipLaunchKernelGGL(pack_bt_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, d_src, K,
                       src_ld, dst, N, ldb, koff, perm, taps, bk);
    HIPCHK(e, hipGetLastError());
    return TTS_HIP_OK;
}

}  // namespace

void waveglow_free(tts_hip_engine* e) {
    for (void* p : e->wg.allocs) (void)hipFree(p);
    e->wg.allocs.clear();
    e->wg.x.release();
    e->wg.acts.release();
    e->wg.audio.release();
    e->wg.a0p.release();
    e->wg.x16.release();
    e->wg.acts16.release();
    e->wg.a0p16.release();
    e->wg.mel16.release();
    e->wg.wino_U.release();
    e->wg.wino_P.release();
    e->wg.wino_mel.release();
    e->wg.wino_ready = false;
    e->wg.wino_legacy_ready = false;
    e->wg.f16_ready = false;
    e->wg.x3_ready = false;
    e->wg.io_mel.release();
    e->wg.io_z.release();
    e->wg.io_out.release();
    e->wg.ready = false;
}

int waveglow_finalize(tts_hip_engine* e) {
    WaveGlowDev& wg = e->wg;
    waveglow_free(e);
    auto need = [&](const std::string& name, std::initializer_list<int64_t> dims, const HostTensor** out) -> int {
        const HostTensor* t = find_tensor(e, name);
        if (!t) return set_err(e, TTS_HIP_ENOTREADY, "missing tensor %s", name.c_str());
        if (t->dims != std::vector<int64_t>(dims))
            return set_err(e, TTS_HIP_EINVAL, "tensor %s has an unexpected shape", name.c_str());
        *out = t;
        return 0;
    };
    int rc;
    // staging buffers for raw Keras-layout kernels (largest: upsample 1024*80*80 = 6.55 M floats)
    DevBuf stage, stage2, ut, wct;
    HIPCHK(e, stage.ensure((size_t)1024 * 80 * 80 * 4));
    HIPCHK(e, stage2.ensure((size_t)1024 * 4 * 2 + 1024));
    HIPCHK(e, ut.ensure((size_t)NPH * KMEL * NCOND * 4));
    HIPCHK(e, wct.ensure((size_t)2 * C * NCOND * 4));
    auto put = [&](DevBuf& b, const HostTensor* t) -> int {
        HIPCHK(e, hipMemcpyAsync(b.p, t->data.data(), t->numel() * 4, hipMemcpyHostToDevice, e->stream));
        return 0;
    };
    auto done = [&]() {
        stage.release();
        stage2.release();
        ut.release();
        wct.release();
    };
#define WGCHK(x)          \
    if ((rc = (x))) {     \
        done();           \
        waveglow_free(e); \
        return rc;        \
    }
    const HostTensor *t, *t2;
    // ---- transposed-conv upsampling kernel -> per-phase operand UT (only used to fold the conditioning convs below)
    WGCHK(need("waveglow/upsample/kernel", {1024, 80, 80}, &t));
    WGCHK(put(stage, t));
    {
        const long long total = (long long)NPH * KMEL * NCOND;
        hipLaunchKernelGGL(pack_ut_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, stage.f(),
                           ut.f());
        HIPCHK(e, hipGetLastError());
    }
    WGCHK(need("waveglow/upsample/bias", {80}, &t));
    float* d_bup = stage2.f() + 4 * C;              // 80 floats behind the bias staging area
    HIPCHK(e, hipMemcpyAsync(d_bup, t->data.data(), 80 * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));

    // ---- flows
    int n_half = 4, n_rem = 8;
    for (int k = 0; k < 12; ++k) {
        if (k % 4 == 0 && k > 0) {
            n_half -= 1;
            n_rem -= 2;
        }
        WgFlowDev& fl = wg.flow[k];
        fl.n_rem = n_rem;
        fl.n_half = n_half;
        const std::string p = "waveglow/block-" + std::to_string(k);
        WGCHK(need(p + "/start_conv/kernel", {1, n_half, C}, &t));
        WGCHK(upload(e, t->data.data(), t->numel(), &fl.start_w, wg.allocs));
        WGCHK(need(p + "/start_conv/bias", {C}, &t));
        WGCHK(upload(e, t->data.data(), t->numel(), &fl.start_b, wg.allocs));
        for (int i = 0; i < 8; ++i) {
            WgLayerDev& ly = fl.layer[i];
            const std::string si = std::to_string(i);
            const int kconv = i == 0 ? KCONV0 : KCONV;
            WGCHK(dev_alloc(e, (size_t)2 * C * kconv, &ly.in_Bt, wg.allocs, false));
            WGCHK(need(p + "/in_conv-" + si + "/kernel", {3, C, 2 * C}, &t));
            if (i == 0) {
                // compose with the start conv (waveglow_arch.py:108): rows j < h: sum_c W_start[j][c] * W_in[tap][c][n];
                // row h: sum_c b_start[c] * W_in[tap][c][n]; rows > h: 0.  Laid out [3 taps][16][1024] for pack_bt.
                const HostTensor *ws, *bs;
                WGCHK(need(p + "/start_conv/kernel", {1, n_half, C}, &ws));
                WGCHK(need(p + "/start_conv/bias", {C}, &bs));
                std::vector<float> comp((size_t)3 * 16 * 2 * C, 0.f);
                std::vector<double> rowacc(2 * C);
                for (int tap = 0; tap < 3; ++tap)
                    for (int j = 0; j <= n_half; ++j) {
                        for (int n = 0; n < 2 * C; ++n) rowacc[n] = 0.0;
                        for (int c = 0; c < C; ++c) {
                            const double sv = j < n_half ? (double)ws->data[(size_t)j * C + c] : (double)bs->data[c];
                            const float* wr = t->data.data() + ((size_t)tap * C + c) * 2 * C;
                            for (int n = 0; n < 2 * C; ++n) rowacc[n] += sv * (double)wr[n];
                        }
                        float* dst = comp.data() + ((size_t)tap * 16 + j) * 2 * C;
                        for (int n = 0; n < 2 * C; ++n) dst[n] = (float)rowacc[n];
                    }
                HIPCHK(e, hipMemcpyAsync(stage.p, comp.data(), comp.size() * 4, hipMemcpyHostToDevice, e->stream));
                HIPCHK(e, hipStreamSynchronize(e->stream));
                WGCHK(pack_bt(e, stage.f(), KCONV0, 2 * C, ly.in_Bt, 2 * C, kconv, 0, 1));   // K order tap*16 + j
            } else {
                WGCHK(put(stage, t));
                WGCHK(pack_bt(e, stage.f(), 3 * C, 2 * C, ly.in_Bt, 2 * C, kconv, 0, 1, WN_TAPS, TTS_WN_BK));
            }
            HIPCHK(e, hipStreamSynchronize(e->stream));
            // conditioning conv: WcT[n'][k] (gate-permuted rows), then V_{i,p} = WcT @ U_p for the 32 phases
            WGCHK(need(p + "/cond_layer-" + si + "/kernel", {1, NCOND, 2 * C}, &t));
            WGCHK(put(stage, t));
            WGCHK(pack_bt(e, stage.f(), NCOND, 2 * C, wct.f(), 2 * C, NCOND, 0, 1));
            WGCHK(dev_alloc(e, (size_t)NPH * 2 * C * KMEL, &ly.cond_Bt, wg.allocs, false));
            {
                GemmArgs g{};
                g.M = 2 * C;
                g.N = KMEL;
                g.L = 2 * C;
                g.nseg = 1;
                g.seg[0] = ASeg{wct.f(), NCOND, 0, NCOND, NCOND};
                g.Bt = ut.f();
                g.ldb = NCOND;
                g.strideBz = (long long)KMEL * NCOND;
                g.mode = EPI_LINEAR;
                g.split = KMEL;
                g.out0 = ly.cond_Bt;
                g.ld0 = KMEL;
                g.strideOutZ = (long long)2 * C * KMEL;
                HIPCHK(e, gemm_small(g, NPH, e->stream));
            }
            WGCHK(need(p + "/in_conv-" + si + "/bias", {2 * C}, &t));
            WGCHK(need(p + "/cond_layer-" + si + "/bias", {2 * C}, &t2));
            WGCHK(dev_alloc(e, 2 * C, &ly.in_bias, wg.allocs, false));
            HIPCHK(e, hipMemcpyAsync(stage2.p, t->data.data(), 2 * C * 4, hipMemcpyHostToDevice, e->stream));
            HIPCHK(e, hipMemcpyAsync(stage2.f() + 2 * C, t2->data.data(), 2 * C * 4, hipMemcpyHostToDevice, e->stream));
            hipLaunchKernelGGL(pack_bias_kernel, dim3(4), dim3(256), 0, e->stream, stage2.f(), stage2.f() + 2 * C,
                               ly.in_bias, 2 * C, 1);
            hipLaunchKernelGGL(cond_bias_kernel, dim3(4), dim3(256), 0, e->stream, wct.f(), d_bup, ly.in_bias);
            HIPCHK(e, hipGetLastError());
            HIPCHK(e, hipStreamSynchronize(e->stream));
            // res_skip conv: keep only the residual half as a GEMM operand (layers 0..6); the skip half is folded below
            const int rs_full = i < 7 ? 2 * C : C;
            WGCHK(need(p + "/res_skip_conv-" + si + "/kernel", {1, C, rs_full}, &t));
            WGCHK(need(p + "/res_skip_conv-" + si + "/bias", {rs_full}, &t2));
            ly.rs_n = i < 7 ? C : 0;
            if (i < 7) {
                WGCHK(dev_alloc(e, (size_t)C * C, &ly.rs_Bt, wg.allocs, false));
                WGCHK(put(stage, t));
                WGCHK(pack_bt(e, stage.f(), C, rs_full, ly.rs_Bt, C, C, 0, 0));      // rows n < 512 = residual outputs
                HIPCHK(e, hipStreamSynchronize(e->stream));
                WGCHK(upload(e, t2->data.data(), C, &ly.rs_bias, wg.allocs));
            }
        }
        {
            // fold: wfold[i][o][c] = sum_s W_skip_i[c][s] * W_end[s][o];  bfold[o] = sum_i b_skip_i @ W_end + b_end
            const HostTensor *we, *be;
            WGCHK(need(p + "/end_conv/kernel", {1, C, 2 * n_half}, &we));
            WGCHK(need(p + "/end_conv/bias", {2 * n_half}, &be));
            const int no = 2 * n_half;
            std::vector<float> wf((size_t)8 * 8 * C, 0.f), bf(8, 0.f);
            std::vector<double> bsum(no, 0.0);
            for (int o = 0; o < no; ++o) bsum[o] = be->data[o];
            for (int i = 0; i < 8; ++i) {
                const HostTensor* wk = find_tensor(e, p + "/res_skip_conv-" + std::to_string(i) + "/kernel");
                const HostTensor* bk = find_tensor(e, p + "/res_skip_conv-" + std::to_string(i) + "/bias");
                const int rs_full = i < 7 ? 2 * C : C, soff = i < 7 ? C : 0;
                std::vector<double> row(no);
                for (int c = 0; c < C; ++c) {
                    for (int o = 0; o < no; ++o) row[o] = 0.0;
                    const float* ws = wk->data.data() + (size_t)c * rs_full + soff;
                    for (int sidx = 0; sidx < C; ++sidx) {
                        const double wv = ws[sidx];
                        const float* wend = we->data.data() + (size_t)sidx * no;
                        for (int o = 0; o < no; ++o) row[o] += wv * (double)wend[o];
                    }
                    for (int o = 0; o < no; ++o) wf[((size_t)i * 8 + o) * C + c] = (float)row[o];
                }
                for (int sidx = 0; sidx < C; ++sidx)
                    for (int o = 0; o < no; ++o)
                        bsum[o] += (double)bk->data[soff + sidx] * (double)we->data[(size_t)sidx * no + o];
            }
            for (int o = 0; o < no; ++o) bf[o] = (float)bsum[o];
            WGCHK(upload(e, wf.data(), wf.size(), &fl.end_w, wg.allocs));
            WGCHK(upload(e, bf.data(), bf.size(), &fl.end_b, wg.allocs));
        }
        // Invertible1x1Conv.build_inverse (invertible_conv.py:41-47): W = kernel[0]^T, W_inverse = inv(W)^T, and the
        // reverse conv (kernel layout [1][in][out]) computes out = audio @ W_inverse = audio @ inv(kernel[0]^T)^T.
        WGCHK(need("waveglow/invertible_conv-" + std::to_string(k) + "/conv/kernel", {1, n_rem, n_rem}, &t));
        {
            const int n = n_rem;
            std::vector<double> a((size_t)n * 2 * n, 0.0);      // [W | I], W[r][c] = kernel[c][r]
            for (int r = 0; r < n; ++r) {
                for (int c = 0; c < n; ++c) a[(size_t)r * 2 * n + c] = (double)t->data[(size_t)c * n + r];
                a[(size_t)r * 2 * n + n + r] = 1.0;
            }
            for (int col = 0; col < n; ++col) {                  // Gauss-Jordan with partial pivoting
                int piv = col;
                for (int r = col + 1; r < n; ++r)
                    if (std::fabs(a[(size_t)r * 2 * n + col]) > std::fabs(a[(size_t)piv * 2 * n + col])) piv = r;
                if (std::fabs(a[(size_t)piv * 2 * n + col]) < 1e-12) {
                    done();
                    waveglow_free(e);
                    return set_err(e, TTS_HIP_EINVAL, "invertible_conv-%d kernel is singular", k);
                }
                if (piv != col)
                    for (int c = 0; c < 2 * n; ++c) std::swap(a[(size_t)piv * 2 * n + c], a[(size_t)col * 2 * n + c]);
                const double d = a[(size_t)col * 2 * n + col];
                for (int c = 0; c < 2 * n; ++c) a[(size_t)col * 2 * n + c] /= d;
                for (int r = 0; r < n; ++r) {
                    if (r == col) continue;
                    const double f = a[(size_t)r * 2 * n + col];
                    if (f != 0.0)
                        for (int c = 0; c < 2 * n; ++c) a[(size_t)r * 2 * n + c] -= f * a[(size_t)col * 2 * n + c];
                }
            }
            std::vector<float> minv((size_t)n * n);              // M[j][c] = inv(W)[c][j]
            for (int j = 0; j < n; ++j)
                for (int c = 0; c < n; ++c) minv[(size_t)j * n + c] = (float)a[(size_t)c * 2 * n + n + j];
            WGCHK(upload(e, minv.data(), minv.size(), &fl.inv, wg.allocs));
        }
    }
#undef WGCHK
    HIPCHK(e, hipStreamSynchronize(e->stream));
    done();
    wg.ready = true;
    return TTS_HIP_OK;
}

// Builds the fp16 GEMM operands from the packed fp32 device copies (once).
static int waveglow_build_f16(tts_hip_engine* e) {
    WaveGlowDev& wg = e->wg;
    if (wg.f16_ready) return TTS_HIP_OK;
    hipStream_t st = e->stream;
    auto alloc_h = [&](size_t n, _Float16** out) -> int {
        void* p = nullptr;
        HIPCHK(e, hipMalloc(&p, n * sizeof(_Float16)));
        wg.allocs.push_back(p);
        *out = (_Float16*)p;
        return TTS_HIP_OK;
    };
    int rc;
    for (int k = 0; k < 12; ++k)
        for (int i = 0; i < 8; ++i) {
            WgLayerDev& ly = wg.flow[k].layer[i];
            _Float16 *a, *c, *r = nullptr;
            if (i == 0) {
                if ((rc = alloc_h((size_t)2 * C * 96, &a))) return rc;
                hipLaunchKernelGGL(cvt_taps0_kernel, dim3((2 * C * 96 + 255) / 256), dim3(256), 0, st, ly.in_Bt, a);
            } else {
                if ((rc = alloc_h((size_t)2 * C * KCONV, &a))) return rc;
                const long long n = (long long)2 * C * KCONV;
                hipLaunchKernelGGL(cvt_taps_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ly.in_Bt, a);
            }
            if ((rc = alloc_h((size_t)NPH * 2 * C * KMEL, &c))) return rc;
            {
                const long long n = (long long)NPH * 2 * C * KMEL;
                hipLaunchKernelGGL(cvt_half_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ly.cond_Bt, c, n,
                                   (_Float16*)nullptr);
            }
            if (ly.rs_n) {
                if ((rc = alloc_h((size_t)C * C, &r))) return rc;
                hipLaunchKernelGGL(cvt_half_kernel, dim3((C * C + 255) / 256), dim3(256), 0, st, ly.rs_Bt, r, (long long)C * C);
            }
            HIPCHK(e, hipGetLastError());
            ly.in_Bt16 = a;
            ly.cond_Bt16 = c;
            ly.rs_Bt16 = r;
        }
    HIPCHK(e, hipStreamSynchronize(st));
    wg.f16_ready = true;
    return TTS_HIP_OK;
}

// Split-fp16 operands ([2 planes] per matrix: hi = fp16(w), lo = fp16(w - hi)), built once from the packed fp32 copies.
static int waveglow_build_x3(tts_hip_engine* e) {
    WaveGlowDev& wg = e->wg;
    if (wg.x3_ready) return TTS_HIP_OK;
    hipStream_t st = e->stream;
    auto alloc_h = [&](size_t n, _Float16** out) -> int {
        void* p = nullptr;
        HIPCHK(e, hipMalloc(&p, n * sizeof(_Float16)));
        wg.allocs.push_back(p);
        *out = (_Float16*)p;
        return TTS_HIP_OK;
    };
    int rc;
    for (int k = 0; k < 12; ++k)
        for (int i = 0; i < 8; ++i) {
            WgLayerDev& ly = wg.flow[k].layer[i];
            _Float16 *a, *c, *r = nullptr;
            if (i == 0) {
                const size_t n = (size_t)2 * C * 96;
                if ((rc = alloc_h(2 * n, &a))) return rc;
                hipLaunchKernelGGL(cvt_taps0_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ly.in_Bt, a, a + n);
            } else {
                const size_t n = (size_t)2 * C * KCONV;
                if ((rc = alloc_h(2 * n, &a))) return rc;
                hipLaunchKernelGGL(cvt_taps_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ly.in_Bt, a, a + n);
            }
            {
                const size_t n = (size_t)NPH * 2 * C * KMEL;
                if ((rc = alloc_h(2 * n, &c))) return rc;
                hipLaunchKernelGGL(cvt_half_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ly.cond_Bt, c,
                                   (long long)n, c + n);
            }
            if (ly.rs_n) {
                const size_t n = (size_t)C * C;
                if ((rc = alloc_h(2 * n, &r))) return rc;
                hipLaunchKernelGGL(cvt_half_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ly.rs_Bt, r,
                                   (long long)n, r + n);
            }
            HIPCHK(e, hipGetLastError());
            ly.in_Bt_x3 = a;
            ly.cond_Bt_x3 = c;
            ly.rs_Bt_x3 = r;
        }
    HIPCHK(e, hipStreamSynchronize(st));
    wg.x3_ready = true;
    return TTS_HIP_OK;
}

// precision 0: exact fp32 MFMA path.  precision 1: fp16 operands (activations, mel and weights fp16 in HBM), fp32
// accumulation and fp32 epilogue math, fp32 master copy of the residual stream and of the flow state.  precision 2: split
// fp16 -- every GEMM operand is a pair of fp16 planes (hi, lo), three MFMAs per product (hi*hi + hi*lo + lo*hi), fp32
// accumulation: ~22 operand bits, i.e. fp32-class results at 3/16 of the fp32 MFMA cost.
int waveglow_run(tts_hip_engine* e, const float* d_mel, int B, int T, const float* d_z, float sigma, float* d_audio,
                 int precision) {
    WaveGlowDev& wg = e->wg;
    const bool half = precision == 1;
    const bool x3 = precision == 2;
    if (half) {
        int rc = waveglow_build_f16(e);
        if (rc) return rc;
    }
    if (x3) {
        int rc = waveglow_build_x3(e);
        if (rc) return rc;
    }
    const int BT = B * T;                                        // frames
    // rows per phase block, padded to the M tile: 256-row tiles unless 128-row tiles save at least 5 % of the rows
    const int pr256 = (BT + 255) / 256 * 256, pr128 = (BT + 127) / 128 * 128, pr64 = (BT + 63) / 64 * 64;
    bool tile128 = pr128 * 1.05 < pr256;
    // short utterances (a sentence at batch 1): 64-row tiles when they save padding
    const int pr_big = tile128 ? pr128 : pr256;
    bool row64 = x3 ? pr64 * 1.25 < pr256      // split fp16 has two tile shapes: 64 x 128 (about 25 % more time per row) and 256 x 256
                    : BT <= 512 &&
                      (half ? pr64 * 4 <= pr_big * 3 : pr64 < pr_big);   // fp16: the smaller tile only pays from -25 % rows
    // fp32: the Winograd form (wn_wino.hip) executes K ~1 090 per output instead of 1 856 in ONE kernel per layer on 64-row tiles.
    // It pays from about 150 frames per call (one sentence, measured on one box, Winograd / direct: 100 frames 16.9 / 15.0 ms,
    // 150: 18.3 / 20.9, 200: 19.4 / 26.0, 350: 32.2 / 37.6, 513: 48.5 / 61.2, 800: 64.0 / 84.9; the three-pass form of round 3
    // only paid from 384 frames: its two HBM-bound passes and six-slice launches cost 40 % at 100 frames)
#ifndef TTS_WINO_MIN_FRAMES
#define TTS_WINO_MIN_FRAMES 144
#endif
    const bool wino_size = precision == 0 && wg.form_mode >= 1 && BT >= TTS_WINO_MIN_FRAMES;
    // the three-pass form (measurement form 2) needs 128-row phase blocks; the fused kernels run on 64-row tiles
    if (wino_size && wg.form_mode == 2 && row64 && (double)pr128 * 1120.0 * 1.35 < (double)pr64 * 1856.0) {
        row64 = false;
        tile128 = pr128 * 1.05 < pr256;
    }
    const int PR = row64 ? pr64 : (tile128 && !x3) ? pr128 : pr256;
    const int NP = x3 ? 2 : 1;                                   // fp16 planes per operand
    const long long M = (long long)NPH * PR;                     // phase-major rows (incl. padding)
    // 128 x 128 tiles would leave block slots (3 per CU) empty -> 128 x 64 tiles, twice the blocks
    const bool tile64 = !row64 && tile128 && (M / 128) * 8 < 768;
    if ((double)M * C * 4.0 >= 2147483648.0 - 65536.0)
        return set_err(e, TTS_HIP_EINVAL, "waveglow_infer: B*T = %d frames exceeds one call's limit (~32000)", BT);
    HIPCHK(e, wg.x.ensure((size_t)M * C * 4));
    HIPCHK(e, wg.audio.ensure((size_t)M * 8 * 4));
    if (half || x3) {
        HIPCHK(e, wg.x16.ensure((size_t)NP * M * C * 2));
        HIPCHK(e, wg.acts16.ensure((size_t)8 * NP * M * C * 2));
        HIPCHK(e, wg.a0p16.ensure((size_t)NP * M * 32 * 2));
        HIPCHK(e, wg.mel16.ensure((size_t)NP * BT * KMEL * 2 + 256));
    } else {
        HIPCHK(e, wg.acts.ensure((size_t)8 * M * C * 4));      // activations of the 8 layers of one flow
        HIPCHK(e, wg.a0p.ensure((size_t)M * 16 * 4));
    }
    hipStream_t st = e->stream;
    // fp32 path, 128- / 256-row tiles: layers 1 .. 7 of a flow run in their Winograd form (wn_wino.hip)
    bool wino = wino_size && (!row64 || wg.form_mode != 2);               // (PR is a multiple of 64; form 2: of 128)
    if (wino) {
        // its operands (3.6 GB of weight planes on first use, 0.7 GB of workspace at config 2) are extra: when the device cannot
        // hold them -- and only then: any other error is the call's error -- this handle keeps the direct form from now on
        const bool three_pass = wg.form_mode == 2;
        size_t free_b = 0, total_b = 0;
        HIPCHK(e, hipMemGetInfo(&free_b, &total_b));
        const size_t need = wg.wino_ready && (!three_pass || wg.wino_legacy_ready) ? 0 : (size_t)(three_pass ? 11 : 4) << 30;
        int rc = free_b < need ? TTS_HIP_ENOMEM : waveglow_build_wino(e, three_pass);
        bool oom = rc == TTS_HIP_ENOMEM;
        if (!rc) {
            rc = waveglow_wino_begin(e, d_mel, PR, BT, T, wg.form_mode);
            oom = rc == TTS_HIP_ENOMEM;
        }
        if (rc && !oom) return rc;
        if (rc) {
            (void)hipGetLastError();                                      // (clears the sticky out-of-memory status)
            wg.form_mode = 0;
            wino = false;
        }
    }
    wg.last_form = wino ? 1 : 0;
    _Float16* x16 = (_Float16*)wg.x16.p;
    _Float16* acts16 = (_Float16*)wg.acts16.p;
    _Float16* mel16 = (_Float16*)wg.mel16.p;

    const unsigned mb = (unsigned)((M + 255) / 256);
    hipLaunchKernelGGL(init_audio_kernel, dim3(mb), dim3(256), 0, st, d_z, sigma, wg.audio.f(), PR, BT);
    if (half || x3) {
        const long long n = (long long)BT * KMEL;
        hipLaunchKernelGGL(mel_window_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_mel, mel16,
                           x3 ? mel16 + n : (_Float16*)nullptr, BT, T);
    }
    HIPCHK(e, hipGetLastError());

    int zoff = 4;
    for (int k = 11; k >= 0; --k) {
        const WgFlowDev& fl = wg.flow[k];
        const int h = fl.n_half;
        {
            const long long n4 = M * (C / 4);
            const dim3 grid((unsigned)((n4 + 255) / 256));
            if (half || x3)
                hipLaunchKernelGGL(wn_start_kernel<true>, grid, dim3(256), 0, st, wg.audio.f(), fl.start_w, fl.start_b,
                                   wg.x.f(), wg.a0p16.p, x16, M, h, x3 ? 1 : 0);
            else
                hipLaunchKernelGGL(wn_start_kernel<false>, grid, dim3(256), 0, st, wg.audio.f(), fl.start_w, fl.start_b,
                                   wg.x.f(), wg.a0p.p, (_Float16*)nullptr, M, h);
            HIPCHK(e, hipGetLastError());
        }
        for (int i = 0; i < 8; ++i) {
            const WgLayerDev& ly = fl.layer[i];
            const int d = 1 << i;
            GemmArgs g{};
            g.M = (int)M;
            g.N = 2 * C;
            g.L = T;                                   // sequence bounds are tested on the frame index inside a batch item
            g.phase_rows = PR;
            g.frames = BT;
            g.phase_step = d < 32 ? d : 1;             // taps at +-d groups: another phase block for d < 32, else +-d / 32 frames
            g.nseg = 7;
            g.bias = ly.in_bias;
            g.mode = EPI_GATE;
            g.split = 2 * C;
            g.wide_epi = 1;
            if (!half && !x3) {
                if (i == 0) {
                    // first layer: conv(start(a0)) composed at load time -> K = 3 taps x 16 (h + 1 used) instead of 3 x 512
                    for (int tap = 0; tap < 3; ++tap) g.seg[tap] = ASeg{wg.a0p.f(), 16, (tap - 1) * d, 16, 16, SEG_PHASE_TAP};
                    g.ldb = KCONV0;
                } else {
                    for (int tap = 0; tap < 3; ++tap) g.seg[tap] = ASeg{wg.x.f(), C, (tap - 1) * d, C, C, SEG_PHASE_TAP};
                    g.ldb = KCONV;
                }
                // folded conditioning: mel frames t, t-1, t-2, t-3 against the per-phase weights V_{i,p}
                for (int q = 0; q < 4; ++q) g.seg[3 + q] = ASeg{d_mel, 80, -q, 80, 80, SEG_FRAME};
                g.Bt = ly.in_Bt;
                g.Bt2 = ly.cond_Bt;
                g.ldb2 = KMEL;
                g.strideB2p = (long long)2 * C * KMEL;
                float* acts_i = wg.acts.f() + (size_t)i * M * C;
                g.out0 = acts_i;
                g.ld0 = C;
                if (wino && i > 0) {
                    const int rc = waveglow_wino_layer(e, ly, i, wg.x.f(), acts_i, PR, BT, T);
                    if (rc) return rc;
                } else {
                    timing_begin(e, i == 0 ? 3 : 0);
                    if (i == 0) HIPCHK(e, row64 ? gemm_wn_in0_r64(g, st) : tile64 ? gemm_wn_in0_64(g, st) : tile128 ? gemm_wn_in0_128(g, st) : gemm_wn_in0(g, st));
                    else HIPCHK(e, row64 ? gemm_wn_in_r64(g, st) : tile64 ? gemm_wn_in_64(g, st) : tile128 ? gemm_wn_in_128(g, st) : gemm_wn_in(g, st));
                    timing_end(e);
                }
                if (wg.probe_out && wg.probe_flow == k && wg.probe_layer == i) {        // test hook: stop here
                    const long long n4 = (long long)BT * NPH * (C / 4);
                    hipLaunchKernelGGL(probe_acts_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, acts_i,
                                       wg.probe_out, PR, BT, T);
                    HIPCHK(e, hipGetLastError());
                    return TTS_HIP_OK;
                }
                if (i < 7) {             // residual: x += acts_i @ W_res + b_res   (skip half folded into wn_end_fold)
                    GemmArgs r{};
                    r.M = (int)M;
                    r.N = C;
                    r.L = (int)M;
                    r.nseg = 1;
                    r.seg[0] = ASeg{acts_i, C, 0, C, C};
                    r.Bt = ly.rs_Bt;
                    r.ldb = C;
                    r.bias = ly.rs_bias;
                    r.mode = EPI_LINEAR;
                    r.act = ACT_NONE;
                    r.split = C;
                    r.out0 = wg.x.f();
                    r.ld0 = C;
                    r.acc0 = 1;
                    r.wide_epi = 1;
                    timing_begin(e, 1);
                    HIPCHK(e, row64 ? gemm_wn_res_r64(r, st) : tile64 ? gemm_wn_res_64(r, st) : gemm_wn_res_skip(r, st));
                    timing_end(e);
                }
            } else {
                // fp16 operands, described in float units (one unit = 2 halfs): ld / k / kpad / ldb are halved.  Split mode:
                // every operand has a second plane (the lo halves) at a fixed offset, loaded next to the first one.
                const long long plX = x3 ? (long long)M * C / 2 : 0, plA0 = x3 ? (long long)M * 16 : 0,
                                plMel = x3 ? (long long)BT * KMEL / 2 : 0;
                g.nseg = 4;                  // 3 taps + one contiguous 4-frame mel window
                if (i == 0) {
                    for (int tap = 0; tap < 3; ++tap)
                        g.seg[tap] = ASeg{(const float*)wg.a0p16.p, 16, (tap - 1) * d, 16, 16, SEG_PHASE_TAP, plA0};
                    g.ldb = 96 / 2;
                    g.planeB = x3 ? (long long)2 * C * 96 / 2 : 0;
                } else {
                    for (int tap = 0; tap < 3; ++tap)
                        g.seg[tap] = ASeg{(const float*)x16, C / 2, (tap - 1) * d, C / 2, C / 2, SEG_PHASE_TAP, plX};
                    g.ldb = KCONV / 2;
                    g.planeB = x3 ? (long long)2 * C * KCONV / 2 : 0;
                }
                g.seg[3] = ASeg{(const float*)mel16, KMEL / 2, 0, KMEL / 2, KMEL / 2, SEG_FRAME, plMel};
                g.Bt = (const float*)(x3 ? ly.in_Bt_x3 : ly.in_Bt16);
                g.Bt2 = (const float*)(x3 ? ly.cond_Bt_x3 : ly.cond_Bt16);
                g.ldb2 = KMEL / 2;
                g.strideB2p = (long long)2 * C * KMEL / 2;
                g.planeB2 = x3 ? (long long)NPH * 2 * C * KMEL / 2 : 0;
                _Float16* acts_i = acts16 + (size_t)i * NP * M * C;
                g.out0 = wg.x.f();           // unused by the gate epilogue (fp16 output below)
                g.ld0 = C;
                g.out0h = acts_i;
                g.ld0h = C;
                g.planeOut = (long long)M * C;
#ifdef TTS_DEBUG_HOOKS
                static const bool split_dil = getenv("TTS_TIME_SPLIT_DIL") != nullptr;    // measurement builds: time the
                timing_begin(e, i == 0 ? 3 : (split_dil && d >= 32) ? 2 : 0);             // layers with d >= 32 as kind 2
#else
                timing_begin(e, i == 0 ? 3 : 0);
#endif
                if (x3) HIPCHK(e, i == 0 ? gemm_wn_in0_x3(g, row64, st) : gemm_wn_in_x3(g, row64, st));
                else if (row64) HIPCHK(e, i == 0 ? gemm_wn_in0_r64h(g, st) : gemm_wn_in_r64h(g, st));
                else if (tile64) HIPCHK(e, i == 0 ? gemm_wn_in0_64h(g, st) : gemm_wn_in_64h(g, st));
                else HIPCHK(e, i == 0 ? gemm_wn_in0_h(g, tile128, st) : gemm_wn_in_h(g, tile128, st));
                timing_end(e);
                if (i < 7) {
                    GemmArgs r{};
                    r.M = (int)M;
                    r.N = C;
                    r.L = (int)M;
                    r.nseg = 1;
                    r.seg[0] = ASeg{(const float*)acts_i, C / 2, 0, C / 2, C / 2, SEG_ROWS, plX};
                    r.Bt = (const float*)(x3 ? ly.rs_Bt_x3 : ly.rs_Bt16);
                    r.ldb = C / 2;
                    r.planeB = x3 ? (long long)C * C / 2 : 0;
                    r.bias = ly.rs_bias;
                    r.mode = EPI_LINEAR;
                    r.act = ACT_NONE;
                    r.split = C;
                    r.out0 = wg.x.f();       // fp32 master of the residual stream (read-modify-write)
                    r.ld0 = C;
                    r.acc0 = 1;
                    r.out0h = x16;           // fp16 shadow = operand of the next layer's taps
                    r.ld0h = C;
                    r.planeOut = (long long)M * C;
                    r.wide_epi = 1;
                    timing_begin(e, 1);
                    if (x3) HIPCHK(e, gemm_wn_res_x3(r, row64, st));
                    else HIPCHK(e, row64 ? gemm_wn_res_r64h(r, st) : tile64 ? gemm_wn_res_64h(r, st) : gemm_wn_res_h(r, st));
                    timing_end(e);
                }
            }
        }
        const bool early = (k % 4 == 0) && k > 0;
        float* dst = (k == 0) ? d_audio : wg.audio.f();
        const long long waves = (M + RPW - 1) / RPW;
        const dim3 grid((unsigned)((waves + 3) / 4));
        if (x3)
            hipLaunchKernelGGL((wn_end_fold_kernel<true, true>), grid, dim3(256), 0, st, (const void*)acts16,
                               (long long)NP * M * C, fl.end_w, fl.end_b, fl.inv, wg.audio.f(), dst, k == 0 ? 1 : 0, d_z,
                               zoff, early ? 2 : 0, sigma, M, h, PR, BT, (long long)M * C);
        else if (half)
            hipLaunchKernelGGL((wn_end_fold_kernel<true, false>), grid, dim3(256), 0, st, (const void*)acts16,
                               (long long)M * C, fl.end_w, fl.end_b, fl.inv, wg.audio.f(), dst, k == 0 ? 1 : 0, d_z,
                               zoff, early ? 2 : 0, sigma, M, h, PR, BT, 0ll);
        else
            hipLaunchKernelGGL((wn_end_fold_kernel<false, false>), grid, dim3(256), 0, st, (const void*)wg.acts.p,
                               (long long)M * C, fl.end_w, fl.end_b, fl.inv, wg.audio.f(), dst, k == 0 ? 1 : 0, d_z,
                               zoff, early ? 2 : 0, sigma, M, h, PR, BT);
        HIPCHK(e, hipGetLastError());
        if (early) zoff += 2;
    }
    return TTS_HIP_OK;
}
