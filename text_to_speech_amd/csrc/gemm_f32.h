// gemm_f32.h -- exact-fp32 implicit-GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[m, n] = epilogue( sum_seg sum_k A_seg[m + shift_seg, k] * Bt[n, koff_seg + k] )
//
// One kernel serves every dense contraction of the path: the WaveGlow WN dilated convolution (three shifted
// taps of x plus the conditioning spectrogram are four "segments" of one K dimension), the WN res/skip 1x1, the
// transposed-conv upsampling (phase-batched over blockIdx.z), the Tacotron2 encoder/postnet k=5 convolutions, the
// attention memory projection and the mel-STFT DFT/mel products.
//
// Layout: A rows are positions (channels-last, K contiguous); Bt is the weight stored [N][K] (K contiguous), so both
// MFMA operands are fetched from LDS as one ds_read_b128 per lane covering 4 consecutive k.  A 32x32x2 MFMA takes
// A[i][k = lane>>5] / B[k = lane>>5][j]; lane (i, h) therefore feeds k = kb + 4h + kk for kk = 0..3 over four MFMAs.
// The result is a k-ordered fp32 fmaf chain (bit-exact fp32, no reduced precision).
//
// Tile: BM x BN x 32, 256 threads (4 waves, one per SIMD), register-staged double-buffered LDS with +4-float row
// padding (144-B rows: conflict-free ds_read_b128, MI355X_MICROARCH.md LDS table).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ttsgemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDSK = BK + 4;       // padded row length in floats (144 B)
constexpr int MAX_SEG = 5;

struct ASeg {
    const float* ptr;   // row m of the operand lives at ptr + (m + shift) * ld
    long long ld;       // row stride in floats
    int shift;          // row shift inside a sequence of L rows; rows shifted outside [0, L) read as zero
    int k;              // valid K extent of this segment (multiple of 4)
    int kpad;           // K extent in Bt (multiple of 32, zero padded)
};

enum { EPI_LINEAR = 0, EPI_GATE = 1 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

struct GemmArgs {
    int M, N;
    int L;                          // sequence length for shift bounds (M is a multiple of L, or L == M)
    int nseg;
    ASeg seg[MAX_SEG];
    long long strideAz;             // per-blockIdx.z offset added to every segment pointer
    const float* Bt;                // [N][ldb]
    long long ldb;
    long long strideBz;
    const float* bias;              // [N] or null
    long long strideBiasZ;
    // ---- epilogue
    int mode;                       // EPI_LINEAR / EPI_GATE
    int act;
    int split;                      // columns [0, split) -> out0, [split, N) -> out1 (col - split); split == N: single output
    float* out0; long long ld0; int acc0;     // accN: add to the existing value (read-modify-write)
    float* out1; long long ld1; int acc1;
    long long strideOutZ;
    const uint8_t* rowmask;         // optional [M]: rows with mask 0 produce act(altbias[n]) instead
    const float* altbias;
    int mask_out;                   // with rowmask: multiply the pre-activation by the mask and skip altbias
};

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ float sigmoid_exact(float v) { return 1.0f / (1.0f + expf(-v)); }

// WR x WC waves, each owning RT x CT MFMA tiles of 32x32.
template <int WR, int WC, int RT, int CT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
    constexpr int BM = WR * RT * 32;
    constexpr int BN = WC * CT * 32;
    constexpr int PA = BM / 32;     // float4 loads per thread for the A tile
    constexpr int PB = BN / 32;
    static_assert(WR * WC == 4, "4 waves");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][LDSK]
    float* Bs = smem + 2 * BM * LDSK;          // [2][BN][LDSK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give the 8 blocks that follow
    // each other on one XCD the same M tile and consecutive N tiles -> the A panel is fetched into that L2 once.
    const int numNt = (g.N + BN - 1) / BN;
    const int numMt = (g.M + BM - 1) / BM;
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int slot = bid >> 3;
    const int mt = (slot / numNt) * 8 + xcd;
    const int nt = slot % numNt;
    if (mt >= numMt) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const long long z = blockIdx.z;

    const int lrow = tid >> 3;          // 0..31: row inside a 32-row pass
    const int c4 = (tid & 7) * 4;       // k offset of this thread's float4

    // per-thread A rows: sequence position for shift bounds
    int a_l[PA];
    bool a_ok[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m0 + p * 32 + lrow;
        a_ok[p] = m < g.M;
        a_l[p] = m % g.L;
    }
    const float* Bt = g.Bt + z * g.strideBz;

    int nT = 0;
    for (int s = 0; s < g.nseg; ++s) nT += g.seg[s].kpad / BK;

    f32x4 ra[PA], rb[PB];
    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int s_cur = 0, kc_cur = 0, kglob = 0;   // tile iterator state (wave-uniform)

    auto load_tile = [&]() {
        const ASeg sg = g.seg[s_cur];
        const int kk = kc_cur * BK + c4;
        const bool kok = kk < sg.k;
        const float* base = sg.ptr + z * g.strideAz;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int m = m0 + p * 32 + lrow;
            const int l2 = a_l[p] + sg.shift;
            const bool ok = a_ok[p] && kok && l2 >= 0 && l2 < g.L;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(base + (long long)(m + sg.shift) * sg.ld + kk);
            ra[p] = v;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int n = n0 + p * 32 + lrow;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < g.N) v = *reinterpret_cast<const f32x4*>(Bt + (long long)n * g.ldb + kglob + c4);
            rb[p] = v;
        }
        // advance iterator
        kglob += BK;
        if (++kc_cur * BK >= sg.kpad) { kc_cur = 0; ++s_cur; }
    };
    auto store_tile = [&](int buf) {
        float* a = As + buf * BM * LDSK;
        float* b = Bs + buf * BN * LDSK;
#pragma unroll
        for (int p = 0; p < PA; ++p) *reinterpret_cast<f32x4*>(a + (p * 32 + lrow) * LDSK + c4) = ra[p];
#pragma unroll
        for (int p = 0; p < PB; ++p) *reinterpret_cast<f32x4*>(b + (p * 32 + lrow) * LDSK + c4) = rb[p];
    };

    const int li = lane & 31, lh = lane >> 5;
    auto compute = [&](int buf) {
        const float* a = As + buf * BM * LDSK + (wr * RT * 32 + li) * LDSK + lh * 4;
        const float* b = Bs + buf * BN * LDSK + (wc * CT * 32 + li) * LDSK + lh * 4;
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            f32x4 fa[RT], fb[CT];
#pragma unroll
            for (int i = 0; i < RT; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDSK + k8 * 8);
#pragma unroll
            for (int j = 0; j < CT; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDSK + k8 * 8);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
        }
    };

    load_tile();
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < nT; ++t) {
        const bool more = t + 1 < nT;
        if (more) load_tile();
        compute(t & 1);
        if (more) store_tile((t + 1) & 1);
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const float* bias = g.bias ? g.bias + z * g.strideBiasZ : nullptr;
    const int rbase = m0 + wr * RT * 32 + 4 * lh;
    const int cbase = n0 + wc * CT * 32 + li;
    if constexpr (WC == 1 && CT % 2 == 0) {
        if (g.mode == EPI_GATE) {
            // columns [n0, n0 + BN/2) hold the tanh pre-activations, [n0 + BN/2, n0 + BN) the matching sigmoid
            // pre-activations (weight rows are permuted at load time); output channel = nt * BN/2 + local column.
            constexpr int H = CT / 2;
            float* out = g.out0 + z * g.strideOutZ;
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    const int col_t = cbase + j * 32;
                    const int col_s = col_t + (BN / 2);
                    const float bt = bias ? bias[col_t] : 0.f;
                    const float bs = bias ? bias[col_s] : 0.f;
                    const int ch = nt * (BN / 2) + j * 32 + li;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
                        if (m < g.M) {
                            const float tv = tanhf(acc[i][j][r] + bt);
                            const float sv = sigmoid_exact(acc[i][j + H][r] + bs);
                            out[(long long)m * g.ld0 + ch] = tv * sv;
                        }
                    }
                }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int n = cbase + j * 32;
            if (n >= g.N) continue;
            const float bv = bias ? bias[n] : 0.f;
            const bool second = n >= g.split;
            float* out = (second ? g.out1 : g.out0) + z * g.strideOutZ;
            const long long ld = second ? g.ld1 : g.ld0;
            const int accf = second ? g.acc1 : g.acc0;
            const int nc = second ? n - g.split : n;
            const float ab = g.altbias ? g.altbias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
                if (m < g.M) {
                    float v = acc[i][j][r] + bv;
                    if (g.rowmask) {
                        const bool on = g.rowmask[m] != 0;
                        if (g.mask_out) v = on ? v : 0.f;
                        else v = on ? v : ab;
                    }
                    v = act_apply(v, g.act);
                    float* p = out + (long long)m * ld + nc;
                    if (accf) v += *p;
                    *p = v;
                }
            }
        }
}

template <int WR, int WC, int RT, int CT>
inline hipError_t launch_gemm(const GemmArgs& g, int batch_z, hipStream_t stream) {
    constexpr int BM = WR * RT * 32;
    constexpr int BN = WC * CT * 32;
    const int numNt = (g.N + BN - 1) / BN;
    const int numMt = (g.M + BM - 1) / BM;
    const int numMt8 = (numMt + 7) / 8 * 8;
    const size_t lds = (size_t)2 * (BM + BN) * LDSK * sizeof(float);
    auto kern = gemm_f32_kernel<WR, WC, RT, CT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(numMt8 * numNt, 1, batch_z);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, g);
    return hipGetLastError();
}

// Tile configurations: BIG = 256x128 (WN layers, upsampling); SMALL = 64x64 (Tacotron2-sized problems).
inline hipError_t gemm_big(const GemmArgs& g, int bz, hipStream_t s) { return launch_gemm<4, 1, 2, 4>(g, bz, s); }
inline hipError_t gemm_small(const GemmArgs& g, int bz, hipStream_t s) { return launch_gemm<2, 2, 1, 1>(g, bz, s); }

}  // namespace ttsgemm
