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

#include <atomic>
#include <stdint.h>
#include <utility>

#include "dev_util.h"

// Timing ablations of the fp16 loop (TTS_ABL = 1..7; results are garbage when set) only exist in a build made with
// -DTTS_DEBUG_HOOKS -DTTS_ABL=n; csrc/build.sh never passes either.
#ifndef TTS_DEBUG_HOOKS
#undef TTS_ABL
#endif
#ifndef TTS_ABL
#define TTS_ABL 0
#endif
// fp16 kernels issue v_mfma_f32_16x16x32_f16 (1) or v_mfma_f32_32x32x16_f16 (0): equal cycles per FLOP, but under load the
// chip holds a higher clock on the 16x16x32 shape (MI355X_MICROARCH.md, DVFS item 7; measured here: DESIGN.md 4.1b).
#ifndef TTS_H16
#define TTS_H16 1
#endif
// 8-wave fp16 kernels: only waves 0-3 (one per SIMD) issue the LDS-DMA of a K step; their SIMD partners 4-7 go straight to
// their operand reads and MFMAs.  An LDS-DMA instruction costs its wave 60-185 issue cycles (MI355X_MICROARCH.md, cycle
// constants), four per step and wave against 512 cycles of MFMA: with every wave loading, both waves of a SIMD sit in that
// head together after each barrier and the matrix pipe idles; with one loader per SIMD the partner's MFMAs cover it.
#ifndef TTS_LOADER4
#define TTS_LOADER4 1
#endif
// fp32 LDS-DMA kernels: K loop rotated by half a step (see "rotated loop" in the kernel).  0 = the plain loop.
#ifndef TTS_F32_ROT
#define TTS_F32_ROT 1
#endif

namespace ttsgemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int MAX_SEG = 8;

// f(integral_constant<int, 0>{}), ..., f(integral_constant<int, N - 1>{}): a loop whose index is a constant expression in the
// body (instruction immediates)
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// Row addressing of a segment.  With GemmArgs::phase_rows > 0 the M axis is "phase-major": row m = p * phase_rows + f
// (p = one of 32 sample-group phases inside a mel frame, f = frame row), which keeps every conv tap a constant row
// shift per block while making the phase block-uniform (needed for the per-phase conditioning weights).
enum { SEG_ROWS = 0,        // operand row = m + shift
       SEG_PHASE_TAP = 1,   // shift counts sample groups: phase' = (p + shift) mod 32, frame carry = floor((p + shift) / 32)
       SEG_FRAME = 2,       // operand row = f + shift (same operand for every phase, e.g. the mel frames)
       SEG_ROWS_Z = 3,      // operand row = m + blockIdx.z * zrows (one operand plane per z slice; no sequence bounds)
       SEG_FRAME_Z = 4 };   // operand row = f + blockIdx.z * zrows
struct ASeg {
    const float* ptr;   // row m of the operand lives at ptr + (m + shift) * ld
    long long ld;       // row stride in floats
    int shift;          // row shift inside a sequence of L rows; rows shifted outside [0, L) read as zero
    int k;              // valid K extent of this segment (multiple of 4)
    int kpad;           // K extent in Bt (multiple of BK, zero padded)
    int kind;           // SEG_ROWS / SEG_PHASE_TAP / SEG_FRAME
    long long plane;    // PL == 2 kernels: offset (in floats) of the operand's second fp16 plane (the "lo" halves)
    long long zrows;    // SEG_ROWS_Z / SEG_FRAME_Z: rows between the operand planes of consecutive z slices
    int b1;             // a sequential segment that multiplies the FIRST weight matrix (Bt) although Bt2 is set
};

enum { EPI_LINEAR = 0, EPI_GATE = 1 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

struct GemmArgs {
    int M, N;
    int L;                          // sequence length for shift bounds (M is a multiple of L, or L == M)
    int nseg;
    ASeg seg[MAX_SEG];
    long long strideAz;             // per-blockIdx.z offset added to every segment pointer
    int shift_z;                    // per-blockIdx.z row shift added to every SEG_ROWS segment (conv taps as z slices)
    const float* Bt;                // [N][ldb]
    long long ldb;
    long long strideBz;
    const float* bias;              // [N] or null
    long long strideBiasZ;
    // ---- phase-major mode (0 = off)
    int phase_rows;                 // rows per phase block (multiple of the M tile); M = 32 * phase_rows
    int frames;                     // real frame rows per phase block (rows f >= frames are padding)
    int phase_step;                 // > 0: tile order that keeps the three conv taps' phase blocks together on one XCD
                                    // (the tap offset in phases, a power of two <= 16; see the kernel's tile order)
    const float* Bt2;               // optional second weight matrix for the sequential segments [phase][N][ldb2]
    long long ldb2;
    long long strideB2p;            // per-phase stride of Bt2 (0: shared)
    long long strideB2z;            // per-blockIdx.z stride of Bt2
    int nphase;                     // phase blocks in M (0 = 32)
    // ---- epilogue
    int mode;                       // EPI_LINEAR / EPI_GATE
    int act;
    int split;                      // columns [0, split) -> out0, [split, N) -> out1 (col - split); split == N: single output
    float* out0; long long ld0; int acc0;     // accN: add to the existing value (read-modify-write)
    _Float16* out0h; long long ld0h;          // HALF kernels: fp16 output (gate) / fp16 shadow of out0 (linear); may be null
    long long planeB, planeB2;                // PL == 2 kernels: second-plane offsets (in floats) of Bt and Bt2
    long long planeOut;                       // PL == 2 kernels: second-plane offset (in halfs) of out0h
    int wide_epi;                             // LDS-DMA kernels, EPI_LINEAR, single output, no row mask, M % 32 == 0 and
                                              // N % 32 == 0: transpose each 32 x 32 accumulator tile through LDS and use
                                              // 16-byte loads / stores for out0 (incl. its read-modify-write) and out0h
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


// tanh(a) * sigmoid(b) = (E - 1) / ((E + 1) (1 + F)) with E = e^{2a}, F = e^{-b}: two v_exp_f32 and one v_rcp_f32 (1 ulp
// each) instead of the libm tanhf / expf / IEEE division sequences, which made the gate epilogue ALU-bound (~0.3 ms of a
// WN layer at config 2).  |a| is clamped at 15 (tanh(15) rounds to 1 in fp32) so E stays finite; b -> -inf gives
// F = inf -> rcp(inf) = 0, the correct limit.  Absolute error < 1e-7 (tests/test_waveglow_gpu.py: parity unchanged).
__device__ __forceinline__ float gate_tanh_sigmoid(float a, float b) {
    const float ac = fminf(fmaxf(a, -15.f), 15.f);
    const float E = __builtin_amdgcn_exp2f(ac * 2.885390081777927f);      // 2 log2(e)
    const float F = __builtin_amdgcn_exp2f(b * -1.4426950408889634f);
    return (E - 1.f) * __builtin_amdgcn_rcpf((E + 1.f) * (1.f + F));
}

// Raw buffer loads: out-of-range offsets (>= num_records = 2^31) return 0, which gives branch-free zero fill for
// rows outside the sequence / matrix (cdna_hip_programming.md T8).  Descriptors are built from wave-uniform values.
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, OOB, 0x00020000);
}
// Same, for a base pointer that is wave-uniform but selected at run time (e.g. one of two weight matrices): rebuilding
// the descriptor from readfirstlane'd halves keeps it in SGPRs; a select between two descriptors is lowered to a
// VGPR/scratch copy plus a waterfall loop around every buffer op (cdna_hip_programming.md T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_uniform(const float* base) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, OOB, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

// WR x WC waves, each owning RT x CT MFMA tiles of 32x32.  TAG only separates instantiations by name so that
// profiles list the WaveGlow in-layer GEMM, the res/skip GEMM and the generic uses as different kernels.
// BK = K extent of one LDS stage (16 or 32); OCC = blocks per CU the register/LDS budget is sized for.
// NI > 0: the first NI segments (same k / kpad) are interleaved chunk-wise in K -- tile order (kc, s) -- so that the
// shifted re-reads of one operand (the three conv taps of x) are one K step apart and hit L2 instead of HBM.
// PIPE_REG: tiles are fetched into registers (two stages) and written to a padded, double-buffered LDS image.
// PIPE_DMA: tiles are fetched straight into LDS (buffer_load ... lds, no VGPR round trip and no ds_write on the LDS
//           pipe); the LDS image is unpadded 64-B rows, XOR-swizzled through the per-lane SOURCE address (a wave
//           instruction writes 1 KiB linearly, cdna_hip_programming.md rule 21), triple buffered.
enum { PIPE_REG = 0, PIPE_DMA = 1 };
// HALF: the operands are fp16 (v_mfma_f32_32x32x16_f16, fp32 accumulate).  Everything outside the MFMA call and the
// epilogue stores works on 4-byte units, so an fp16 operand is described to the kernel in "float units": ld, k, kpad, ldb
// are HALF the element counts and one 16-byte LDS fragment (4 float units) carries 8 halfs = the K = 16 slice of one
// MFMA -- the DMA pipeline, the swizzle and the segment addressing are shared with the fp32 kernel.
// PL = 2 (fp16 only): every operand is a pair of fp16 planes (hi, lo) with v ~= hi + lo (22 significant bits), and a
// product is the three MFMAs hi*hi + hi*lo + lo*hi accumulated in fp32 -- fp32-class accuracy at 3/16 of the fp32 MFMA cost.
template <int WR, int WC, int RT, int CT, int BK, int OCC, int TAG, int NI = 0, int PIPE = PIPE_REG, bool HALF = false,
          int NBD = 3, int PL = 1>
__global__ __launch_bounds__(WR * WC * 64, OCC) void gemm_f32_kernel(const GemmArgs g) {
    static_assert(!HALF || PIPE == PIPE_DMA, "fp16 operands use the LDS-DMA pipeline");
    static_assert(PL == 1 || (PL == 2 && HALF), "two planes are the split-fp16 mode");
    constexpr int BM = WR * RT * 32;
    constexpr int BN = WC * CT * 32;
    constexpr bool DMA = PIPE == PIPE_DMA;
    static_assert(!DMA || BK == 16, "the LDS-DMA image is laid out for 64-byte rows");
    constexpr int LDSK = DMA ? BK : BK + 4; // LDS row in floats: padded (144 B / 80 B) or swizzled 64 B
    constexpr int NBUF = DMA ? NBD : 2;     // DMA: NBD - 1 tiles in flight
    constexpr int TPR = BK / 4;             // threads (float4) per tile row
    constexpr int LW = (HALF && WR * WC == 8 && TTS_LOADER4) ? 4 : WR * WC;     // waves that stage tiles
    constexpr int NT = LW * 64;             // staging threads: all 4 or 8 waves, or the first 4 of 8 (TTS_LOADER4)
    constexpr int RPP = NT / TPR;           // rows staged per pass of the staging threads
    constexpr int PA = BM / RPP;            // float4 loads per thread for the A tile
    constexpr int PB = BN / RPP;
    // fp32 LDS-DMA kernels run the rotated K loop (below).  There a wave stages CONTIGUOUS rows -- piece p of wave w covers rows
    // (w * PA + p) * 16 .. + 15 of the tile instead of p * 64 + w * 16 .. -- so that all pieces of one operand side share ONE
    // LDS base (one M0 write) and differ in the instruction's immediate offset (p KiB), which the hardware adds to the LDS
    // address AND to the buffer offset: descriptors are based ROT_SH bytes low and the per-lane offsets carry ROT_SH - p KiB.
    constexpr bool ROT = DMA && !HALF && PL == 1 && NBD == 3 && BK == 16 && TTS_F32_ROT && (!(TTS_ABL) || TTS_ABL >= 9);
    constexpr int ROT_SH = ROT ? 3072 : 0;
    static_assert(!ROT || (PA <= 4 && PB <= 4), "immediate offsets are 12 bits");
    static_assert(WR * WC == 4 || WR * WC == 8, "4 or 8 waves");
    static_assert(BK == 16 || BK == 32, "BK");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [NBUF][PL][BM][LDSK]
    float* Bs = smem + NBUF * PL * BM * LDSK;  // [NBUF][PL][BN][LDSK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give the 8 blocks that follow
    // each other on one XCD the same M tile and consecutive N tiles -> the A panel is fetched into that L2 once.
    const int numNt = (g.N + BN - 1) / BN;
    const int numMt = (g.M + BM - 1) / BM;
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int slot = bid >> 3;
    int mt = (slot / numNt) * 8 + xcd;
    const int nt = slot % numNt;
    if (g.phase_rows > 0 && g.phase_step > 0) {
        // Conv taps of a tile (phase p, frame range F) read the activation rows of the tiles (p - s, F) and (p + s, F),
        // s = phase_step: with M tiles dealt round-robin to the XCDs every activation row was fetched from HBM three times
        // (once per reading tile, each into a different L2).  Here every XCD walks its own contiguous share of the sequence
        // "F outer, then the phases along the orbits of +s": p = r, r + s, r + 2 s, ... for r = 0 .. s - 1, so that the tiles
        // in flight on an XCD at any time (~8) read a sliding window of activation rows that its L2 (4 MB = 8 row tiles) holds.
        const int tpp = g.phase_rows / BM;                         // frame tiles per phase block
        const int seq = xcd * ((numMt + 7) >> 3) + slot / numNt;   // this XCD's chunk of the sequence
        if (seq >= numMt) return;
        const int ft = seq >> 5, i = seq & 31;                     // 32 phases per frame range
        const int per = 32 / g.phase_step;                         // orbit length
        const int ph = (i % per) * g.phase_step + i / per;
        mt = ph * tpp + ft;
        if (ft >= tpp) return;
    }
    if (mt >= numMt) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const long long z = blockIdx.z;

    const int lrow = (tid & (NT - 1)) / TPR;   // row inside a staging pass (waves >= LW never stage)
    // k offset of this thread's float4.  DMA: LDS slot (row, c') receives global chunk c' ^ ((row >> 2) & 3)
    //          (H16: c' ^ (-(row >> 2) & 3), the permutation that makes the 16-row x 4-chunk operand reads conflict-free)
    const int c4 = DMA ? (((tid & 3) ^ ((TTS_H16 && HALF ? -(tid >> 4) : (tid >> 4)) & 3)) * 4) : (tid % TPR) * 4;
    // tile row that piece p of this thread stages (A / B side), and what its per-lane byte offset carries besides the address
    auto rowA = [&](int p) -> int { return ROT ? (wave * PA + p) * 16 + (lane >> 2) : p * RPP + lrow; };
    auto rowB = [&](int p) -> int { return ROT ? (wave * PB + p) * 16 + (lane >> 2) : p * RPP + lrow; };
    auto adj = [&](int p) -> unsigned { return (unsigned)(ROT_SH - (ROT ? p * 1024 : 0)); };
    const int li = lane & 31, lh = lane >> 5;
    // Accumulator register r of a 32 x 32 tile holds (row trow(r), column tcol(r)) of the tile.  32x32 MFMA: column =
    // lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  H16 (four 16x16x32 sub-tiles, registers 4 s .. 4 s + 3 for
    // s = 2 ri + cj): row = 16 ri + 4 (lane >> 4) + (r & 3), column = 16 cj + (lane & 15).
    constexpr bool H16 = HALF && TTS_H16;
    auto trow = [&](int r) -> int { return H16 ? 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3) : (r & 3) + 8 * (r >> 2) + 4 * lh; };
    auto tcol = [&](int r) -> int { return H16 ? 16 * ((r >> 2) & 1) + (lane & 15) : li; };

    // ---- accumulators start from bias (+ the previous output value for read-modify-write epilogues), so the
    //      epilogue is a pure store and the old values are fetched under the first tile's load latency
    const float* bias = g.bias ? g.bias + z * g.strideBiasZ : nullptr;
    const int rbase = m0 + wr * RT * 32;             // + trow(r): first row of the wave's tiles
    const int cbase = n0 + wc * CT * 32;             // + tcol(r): first column of the wave's tiles
    // the output side (out0 / out1) is uniform per block: `split` is a multiple of BN or >= N (checked at launch)
    const bool second = n0 >= g.split;
    float* const outp = (second ? g.out1 : g.out0) + z * g.strideOutZ;
    const long long ldo = second ? g.ld1 : g.ld0;
    const int ncol0 = second ? cbase - g.split : cbase;     // the wave's first output column on that side
    // 32 x 32 accumulator tiles: one f32x16 per tile (32x32 MFMA), or four f32x4 sub-tiles (H16) -- separate values, so that
    // the compiler never has to carry a 16-register tuple through control flow for a 4-register update
    f32x16 acc[H16 ? 1 : RT][H16 ? 1 : CT];
    f32x4 accq[H16 ? RT : 1][H16 ? CT : 1][4];
    auto A = [&](int i, int j, int r) -> float {
        if constexpr (H16) return accq[i][j][r >> 2][r & 3];
        else return acc[i][j][r];
    };
    auto setA = [&](int i, int j, int r, float v) {
        if constexpr (H16) accq[i][j][r >> 2][r & 3] = v;
        else acc[i][j][r] = v;
    };
    {
        float bv[CT][2];                             // per column half (the 32x32 layout has one column per lane)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = cbase + j * 32 + tcol(4 * h);
                bv[j][h] = (bias && n < g.N) ? bias[n] : 0.f;
            }
        const bool rmw = g.mode == EPI_LINEAR && (second ? g.acc1 : g.acc0) && !(DMA && g.wide_epi);
        if (rmw) {
            // branch-free batch of dword buffer loads (rows >= M and columns >= N read as 0), one wait for all
            const __amdgpu_buffer_rsrc_t rsO = make_rsrc(outp + (long long)(m0 + wr * RT * 32) * ldo);
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = i * 32 + trow(r);                               // row inside the wave's rows
                        const bool ok = (rbase + rl < g.M) && (cbase + j * 32 + tcol(r) < g.N);
                        const unsigned off = ok ? (unsigned)((rl * (int)ldo + ncol0 + j * 32 + tcol(r)) * 4) : OOB;
                        setA(i, j, r, bv[j][(r >> 2) & 1] + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsO, off, 0, 0)));
                    }
        } else {
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) setA(i, j, r, bv[j][(r >> 2) & 1]);
        }
    }

    // ---- operand addressing
    const bool PH = g.phase_rows > 0;
    const int ph = PH ? m0 / g.phase_rows : 0;          // block-uniform phase
    const int f0 = PH ? m0 - ph * g.phase_rows : m0;    // first frame row of the tile
    int a_l[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int fr = f0 + rowA(p);
        const bool ok = PH ? fr < g.frames : fr < g.M;
        a_l[p] = ok ? fr % g.L : -0x40000000;           // padding rows never pass the [0, L) test below
    }
    // (absolute operand row of the tile's first row, shift used for the sequence-bounds test) of a segment
    auto seg_row = [&](const ASeg& sg, long long& abs_row, int& vshift) {
        if (sg.kind == SEG_PHASE_TAP) {
            const int ps = ph + sg.shift;
            vshift = ps >> 5;                            // frame carry (arithmetic shift: floor)
            abs_row = (long long)(ps & 31) * g.phase_rows + f0 + vshift;
        } else if (sg.kind == SEG_FRAME) {
            vshift = sg.shift;
            abs_row = (long long)f0 + sg.shift;
        } else if (sg.kind == SEG_ROWS_Z) {
            vshift = 0;
            abs_row = (long long)m0 + z * sg.zrows;
        } else if (sg.kind == SEG_FRAME_Z) {
            vshift = 0;
            abs_row = (long long)f0 + z * sg.zrows;
        } else {
            vshift = sg.shift + (int)z * g.shift_z;
            abs_row = (long long)m0 + vshift;
        }
    };
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(g.Bt + z * g.strideBz + (long long)n0 * g.ldb);
    unsigned b_off[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p)
        b_off[p] = (n0 + rowB(p) < g.N) ? (unsigned)((rowB(p) * (int)g.ldb + c4) * 4) + adj(p) : OOB;

    // tile iterator (wave-uniform) over the sequential segments [NI, nseg): current segment parameters live in
    // registers, refreshed only at a crossing
    int s_cur = NI, kc_cur = 0, kglob = 0, kglob2 = 0;
    int seg_k = 0, seg_kpad = 0;
    unsigned seg_plane = 0;                          // byte offset of the current sequential segment's second plane
    bool seg_b1 = false;                             // the current sequential segment multiplies Bt (not Bt2)
    __amdgpu_buffer_rsrc_t rsA;
    const float* ptrSeqA = nullptr;                  // base of rsA (the rotated loop rebuilds descriptors from pointers)
    unsigned a_off[PA];
    auto enter_segment = [&]() {
        const ASeg sg = g.seg[s_cur];
        seg_k = sg.k;
        seg_kpad = sg.kpad;
        seg_plane = (unsigned)(sg.plane * 4);
        seg_b1 = sg.b1 != 0;
        long long abs_row;
        int vshift;
        seg_row(sg, abs_row, vshift);
        ptrSeqA = sg.ptr + z * g.strideAz + abs_row * sg.ld;
        rsA = make_rsrc(ptrSeqA);
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int l2 = a_l[p] + vshift;
            a_off[p] = (l2 >= 0 && l2 < g.L) ? (unsigned)((rowA(p) * (int)sg.ld + c4) * 4) + adj(p) : OOB;
        }
    };
    // second weight matrix (per-phase) for the sequential segments
    const bool B2 = g.Bt2 != nullptr;
    const float* const bbase1 = g.Bt + z * g.strideBz + (long long)n0 * g.ldb;
    const float* const bbase2 = (B2 ? g.Bt2 + z * g.strideB2z : g.Bt) + (long long)ph * g.strideB2p + (long long)n0 * g.ldb2;
    unsigned b_off2[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p)
        b_off2[p] = (n0 + rowB(p) < g.N) ? (unsigned)((rowB(p) * (int)g.ldb2 + c4) * 4) + adj(p) : OOB;

    // ---- tile fetch: the same address generation feeds either register stages (PIPE_REG) or LDS-DMA (PIPE_DMA);
    // `emit(isA, p, rsrc, voffset)` is the sink.
    int nSeq = 0;
    for (int s = NI; s < g.nseg; ++s) nSeq += g.seg[s].kpad / BK;

    // interleaved group state (NI > 0): the NI segments are shifted views of ONE operand (same pointer and row stride,
    // checked at launch), so a single descriptor based at the smallest shift serves all of them; per tile only a scalar
    // byte delta and one validity bit per staged row change.
    static_assert(NI <= 4, "NI");
    const ASeg sg0 = g.seg[0];
    // per interleaved segment: byte offset of its tile base from the smallest one (32-bit: the operand spans < 2 GiB,
    // checked at launch) and the validity bits of the staged rows.  Only these survive into the K loop.
    unsigned dl0 = 0, dl1 = 0, dl2 = 0, dl3 = 0;
    __amdgpu_buffer_rsrc_t rsI = rsB;
    const float* ptrIbase = g.Bt;                    // base of rsI
    unsigned baseI[PA], vmaskI[PA];
    {
        long long ab0 = 0, ab1 = 0, ab2 = 0, ab3 = 0;
        int vs0 = 0, vs1 = 0, vs2 = 0, vs3 = 0;
        if (NI > 0) seg_row(g.seg[0], ab0, vs0);
        if (NI > 1) seg_row(g.seg[NI > 1 ? 1 : 0], ab1, vs1); else ab1 = ab0;
        if (NI > 2) seg_row(g.seg[NI > 2 ? 2 : 0], ab2, vs2); else ab2 = ab0;
        if (NI > 3) seg_row(g.seg[NI > 3 ? 3 : 0], ab3, vs3); else ab3 = ab0;
        long long row0 = ab0;
        row0 = ab1 < row0 ? ab1 : row0;
        row0 = ab2 < row0 ? ab2 : row0;
        row0 = ab3 < row0 ? ab3 : row0;
        if (NI > 0) {
            ptrIbase = sg0.ptr + z * g.strideAz + row0 * sg0.ld;
            rsI = make_rsrc(ptrIbase);
        }
        const int ld4 = (int)sg0.ld * 4;
        dl0 = (unsigned)((int)(ab0 - row0) * ld4);
        dl1 = (unsigned)((int)(ab1 - row0) * ld4);
        dl2 = (unsigned)((int)(ab2 - row0) * ld4);
        dl3 = (unsigned)((int)(ab3 - row0) * ld4);
        auto vs_of = [&](int s2) { return s2 == 0 ? vs0 : s2 == 1 ? vs1 : s2 == 2 ? vs2 : vs3; };
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            baseI[p] = (unsigned)((rowA(p) * (int)sg0.ld + c4) * 4) + adj(p);
            unsigned vm = 0;
#pragma unroll
            for (int s = 0; s < NI; ++s) {
                const int l2 = a_l[p] + vs_of(s);
                vm |= (l2 >= 0 && l2 < g.L) ? (1u << s) : 0u;
            }
            vmaskI[p] = vm;
        }
    }
    auto delta_of = [&](int s) { return s == 0 ? dl0 : s == 1 ? dl1 : s == 2 ? dl2 : dl3; };
    const int kI = sg0.k;
    const int nI = NI > 0 ? NI * (sg0.kpad / BK) : 0;
    int si = 0, kci = 0;                             // (segment, chunk) of the next interleaved tile to load
    const int nAll = nI + nSeq;
    int t_load = 0;                                  // index of the next tile to fetch

    auto fetch_next = [&](auto&& emit) {             // issues the loads of tile `t_load` (if any) and advances
        if (t_load < nAll) {
            bool cur_b1 = false;
            if (NI > 0 && t_load < nI) {
                const unsigned delta = delta_of(si) + (unsigned)(kci * BK * 4);
                const bool kok = kci * BK + c4 < kI;
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    bool ok = kok && ((vmaskI[p] >> si) & 1u);
#if TTS_ABL
                    // ablation 7: the taps of a layer with dilation >= 32 sit in the tile's own phase block -- price what ONE
                    // shared activation tile per K chunk would save by making the outer taps' stagings fetch nothing
                    if (HALF && TTS_ABL == 7 && g.phase_step == 1 && si != 1) ok = false;
#endif
                    emit(true, p, rsI, ok ? baseI[p] + delta : OOB, (unsigned)(sg0.plane * 4));
                }
                if (++si == NI) {
                    si = 0;
                    ++kci;
                }
            } else {
                const unsigned kb = (unsigned)(kc_cur * BK * 4);
                const bool kok = kc_cur * BK + c4 < seg_k;
                const unsigned pl_a = seg_plane;     // (read before a segment crossing below may change it)
                cur_b1 = seg_b1;
#pragma unroll
                for (int p = 0; p < PA; ++p) emit(true, p, rsA, kok ? a_off[p] + kb : OOB, pl_a);
                if (++kc_cur * BK >= seg_kpad) {
                    kc_cur = 0;
                    if (++s_cur < g.nseg) enter_segment();
                }
            }
            // weight tile: the sequential segments may use their own (per-phase) matrix Bt2
            const bool use2 = B2 && !(NI > 0 && t_load < nI) && !cur_b1;
            const __amdgpu_buffer_rsrc_t rsW = make_rsrc_uniform(use2 ? bbase2 : bbase1);
            const unsigned kg = (unsigned)((use2 ? kglob2 : kglob) * 4);
            const unsigned pl_b = (unsigned)((use2 ? g.planeB2 : g.planeB) * 4);
#pragma unroll
            for (int p = 0; p < PB; ++p) emit(false, p, rsW, (use2 ? b_off2[p] : b_off[p]) + kg, pl_b);
            if (use2) kglob2 += BK;
            else kglob += BK;
        }
        ++t_load;
    };

    // ---- MFMA on one LDS buffer
    auto mfma_h = [](f16x8 x, f16x8 y, f32x16 c) -> f32x16 { return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0); };
    // 16x16x32: sub-tile s = 2 * ri + cj of a 32 x 32 accumulator tile
    auto mfma_q = [](f16x8 x, f16x8 y, f32x4& c) { c = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0); };
    const int xr = (li >> 2) & 3;                    // DMA image: chunk XOR of this lane's operand rows
    auto compute_chunk = [&](int buf, int k8) {
        if constexpr (H16) {
            // one call covers the whole 64-byte stage (K = 32 halfs): lane (row q = lane & 15, chunk c = lane >> 4) of a
            // 16-row operand block reads logical chunk c of its row, stored at slot c ^ (-(q >> 2) & 3)
            if (k8 != 0) return;
            const int q16 = lane & 15;
            const int ko = (((lane >> 4) ^ (-(q16 >> 2))) & 3) * 4;
            const float* a = As + buf * PL * BM * LDSK + (wr * RT * 32 + q16) * LDSK + ko;
            const float* b = Bs + buf * PL * BN * LDSK + (wc * CT * 32 + q16) * LDSK + ko;
            f32x4 fa[RT][2], fb[CT][2];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) fa[i][h] = *reinterpret_cast<const f32x4*>(a + (i * 32 + h * 16) * LDSK);
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) fb[j][h] = *reinterpret_cast<const f32x4*>(b + (j * 32 + h * 16) * LDSK);
            if constexpr (PL == 2) {
                f32x4 fal[RT][2], fbl[CT][2];
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) fal[i][h] = *reinterpret_cast<const f32x4*>(a + (BM + i * 32 + h * 16) * LDSK);
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h) fbl[j][h] = *reinterpret_cast<const f32x4*>(b + (BN + j * 32 + h * 16) * LDSK);
                // hi*lo and lo*hi first, hi*hi last: the small terms are added before the large one lands in the fp32 sum
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j)
#pragma unroll
                        for (int sidx = 0; sidx < 4; ++sidx) {
                            const int ri = sidx >> 1, cj = sidx & 1;
                            mfma_q(__builtin_bit_cast(f16x8, fal[i][ri]), __builtin_bit_cast(f16x8, fb[j][cj]), accq[i][j][sidx]);
                            mfma_q(__builtin_bit_cast(f16x8, fa[i][ri]), __builtin_bit_cast(f16x8, fbl[j][cj]), accq[i][j][sidx]);
                            mfma_q(__builtin_bit_cast(f16x8, fa[i][ri]), __builtin_bit_cast(f16x8, fb[j][cj]), accq[i][j][sidx]);
                        }
                return;
            }
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx)
                        mfma_q(__builtin_bit_cast(f16x8, fa[i][sidx >> 1]), __builtin_bit_cast(f16x8, fb[j][sidx & 1]), accq[i][j][sidx]);
            return;
        }
        const int koff = DMA ? (((2 * k8 + lh) ^ xr) * 4) : (lh * 4 + k8 * 8);
        const float* a = As + buf * PL * BM * LDSK + (wr * RT * 32 + li) * LDSK + koff;
        const float* b = Bs + buf * PL * BN * LDSK + (wc * CT * 32 + li) * LDSK + koff;
        if constexpr (PL == 2) {
            f32x4 fa[RT], fal[RT], fb[CT], fbl[CT];
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDSK);
                fal[i] = *reinterpret_cast<const f32x4*>(a + (BM + i * 32) * LDSK);
            }
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDSK);
                fbl[j] = *reinterpret_cast<const f32x4*>(b + (BN + j * 32) * LDSK);
            }
            // hi*lo and lo*hi first, hi*hi last: the small terms are added before the large one lands in the fp32 sum
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    acc[i][j] = mfma_h(__builtin_bit_cast(f16x8, fal[i]), __builtin_bit_cast(f16x8, fb[j]), acc[i][j]);
                    acc[i][j] = mfma_h(__builtin_bit_cast(f16x8, fa[i]), __builtin_bit_cast(f16x8, fbl[j]), acc[i][j]);
                    acc[i][j] = mfma_h(__builtin_bit_cast(f16x8, fa[i]), __builtin_bit_cast(f16x8, fb[j]), acc[i][j]);
                }
            return;
        }
        f32x4 fa[RT], fb[CT];
#if TTS_ABL == 8 && defined(TTS_ABL_F32)
        if (!HALF) {                                 // ablation 8: the MFMAs of a K step without their operand reads
#pragma unroll
            for (int i = 0; i < RT; ++i) fa[i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + i);
#pragma unroll
            for (int j = 0; j < CT; ++j) fb[j] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane - j);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < RT; ++i) {
#if TTS_ABL
            if (HALF && (TTS_ABL == 5 || TTS_ABL == 6)) { fa[i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(i + k8); continue; }   // ablation: no A-side LDS reads
#endif
            fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDSK);
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDSK);
        if constexpr (HALF) {
            // lane (row, h) holds k = 8h .. 8h+7 of this 16-wide K slice: exactly the 32x32x16 f16 operand layout
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j)
                    acc[i][j] = mfma_h(__builtin_bit_cast(f16x8, fa[i]), __builtin_bit_cast(f16x8, fb[j]), acc[i][j]);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
        }
    };

    if (nSeq > 0) enter_segment();
    if constexpr (ROT) {
        // ---- rotated loop (fp32 LDS-DMA kernels).  Measured on the plain loop below (config 2, 5.69 ms per launch, debug
        // builds with -DTTS_ABL_F32): without the DMA issue 5.19 ms, without the operand reads 5.54, without wait + barrier 5.60,
        // DMA + barrier alone 1.22 -- a wave issued its 6 DMA pieces, then its operand reads, and only then had MFMAs to offer.
        // Here the step boundary sits in the MIDDLE of a tile's MFMAs: a step = [reads of chunk 1 | MFMAs of chunk 0 | wait +
        // barrier | reads of the NEXT tile's chunk 0 | MFMAs of chunk 1 with the DMA pieces of tile t + 3 issued one per MFMA row
        // group].  Every operand read has 32 MFMAs (2 048 cycles) to land, a wave leaves the barrier with 32 MFMAs ready, and the
        // pieces of one operand side share one M0 value.  All three LDS buffers are requested up front; tile t + 3 goes into the
        // buffer of tile t once every wave has its fragments of tile t in registers (the barrier of step t).  Effect: 5.67 ->
        // 5.60 ms; what remains of the DMA cost (the same loop without its pieces: 5.23 ms) is ~50 cycles of matrix-pipe time
        // per piece wherever the piece is placed -- a per-instruction price, so only fewer bytes per MFMA would lower it.
        constexpr int LT = PA + PB;
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        unsigned vo[LT];                                 // per-lane source offsets of the prepared tile's pieces
        __amdgpu_buffer_rsrc_t rsTa = rsB, rsTb = rsB;   // its descriptors (A side, B side)
        auto prepare_next = [&]() {                      // address math of tile `t_load`.  Past the last tile the pieces fetch
            if (t_load >= nAll) {                        // nothing (all lanes out of range: zeros into a buffer nobody reads
#pragma unroll                                           // again), so that the loop body and its counted waits are uniform
                for (int q = 0; q < LT; ++q) vo[q] = OOB;
                return;
            }
            const bool inter = NI > 0 && t_load < nI;
            bool cur_b1 = false;
            if (inter) {
                const unsigned delta = delta_of(si) + (unsigned)(kci * BK * 4);
                const bool kok = kci * BK + c4 < kI;
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    const bool ok = kok && ((vmaskI[p] >> si) & 1u);
                    vo[p] = ok ? baseI[p] + delta : OOB;
                }
                if (++si == NI) {
                    si = 0;
                    ++kci;
                }
                rsTa = make_rsrc_uniform(ptrIbase - ROT_SH / 4);
            } else {
                const unsigned kb = (unsigned)(kc_cur * BK * 4);
                const bool kok = kc_cur * BK + c4 < seg_k;
#pragma unroll
                for (int p = 0; p < PA; ++p) vo[p] = kok ? a_off[p] + kb : OOB;
                rsTa = make_rsrc_uniform(ptrSeqA - ROT_SH / 4);
                cur_b1 = seg_b1;
                if (++kc_cur * BK >= seg_kpad) {
                    kc_cur = 0;
                    if (++s_cur < g.nseg) enter_segment();      // (rsA changes for the NEXT tile; rsTa keeps this tile's copy)
                }
            }
            const bool use2 = B2 && !inter && !cur_b1;
            rsTb = make_rsrc_uniform((use2 ? bbase2 : bbase1) - ROT_SH / 4);
            const unsigned kg = (unsigned)((use2 ? kglob2 : kglob) * 4);
#pragma unroll
            for (int p = 0; p < PB; ++p) vo[PA + p] = (use2 ? b_off2[p] : b_off[p]) + kg;
            if (use2) kglob2 += BK;
            else kglob += BK;
            ++t_load;
        };
        auto issue_piece = [&](auto qc, int bufd) {      // piece q of the prepared tile -> LDS buffer bufd
            constexpr int q = decltype(qc)::value;
            const unsigned voff = vo[q];                 // (a by-value copy: with `vo[q]` as the builtin's operand hipcc's host
                                                         //  pass silently drops the kernel's stub -- undefined symbol at load)
            if constexpr (q < PA) {
                float* dst = As + bufd * BM * LDSK + wave * PA * 16 * LDSK;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsTa, (lds_ptr_t)dst, 16, voff, 0, q * 1024, 0);
            } else {
                float* dst = Bs + bufd * BN * LDSK + wave * PB * 16 * LDSK;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsTb, (lds_ptr_t)dst, 16, voff, 0, (q - PA) * 1024, 0);
            }
        };
        auto read_ops = [&](int bufr, int k8, f32x4 (&fa)[RT], f32x4 (&fb)[CT]) {
            const int koff = ((2 * k8 + lh) ^ xr) * 4;
            const float* a = As + bufr * BM * LDSK + (wr * RT * 32 + li) * LDSK + koff;
            const float* b = Bs + bufr * BN * LDSK + (wc * CT * 32 + li) * LDSK + koff;
#pragma unroll
            for (int i = 0; i < RT; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDSK);
#pragma unroll
            for (int j = 0; j < CT; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDSK);
        };
        auto mfma_row = [&](const f32x4 (&fa)[RT], const f32x4 (&fb)[CT], int kk, int i) {
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
        };
        // prologue: all three buffers requested, tile 0 landed, its first chunk in registers
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            prepare_next();
            static_for<LT>([&](auto qc) { issue_piece(qc, b); });
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LT) : "memory");
        __builtin_amdgcn_s_barrier();
        f32x4 fa0[RT], fb0[CT], fa1[RT], fb1[CT];
        read_ops(0, 0, fa0, fb0);
        int buf = 0;
        for (int t = 0; t < nAll; ++t) {
            read_ops(buf, 1, fa1, fb1);
            prepare_next();                              // tile t + 3 (address math only)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < RT; ++i) mfma_row(fa0, fb0, kk, i);
            // every fragment of tile t is in registers; tile t + 1 has landed (tile t + 2 may stay in flight)
            __builtin_amdgcn_sched_barrier(0);           // (keeps the MFMAs above: hipcc would sink them below the barrier)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LT) : "memory");
            __builtin_amdgcn_s_barrier();
            const int bufn1 = buf == 2 ? 0 : buf + 1;
            read_ops(bufn1, 0, fa0, fb0);                // (past the last tile: stale bytes that nothing uses -- an unconditional
                                                         // read keeps hipcc's own lgkmcnt bookkeeping off the MFMAs below)
            __builtin_amdgcn_sched_barrier(0);           // tile t + 3 -> the buffer of tile t, one piece per MFMA row group
            static_for<4 * RT>([&](auto gc) {
                constexpr int grp = decltype(gc)::value;
                mfma_row(fa1, fb1, grp / RT, grp % RT);
                if constexpr (grp < LT) {
                    if (TTS_ABL != 10) issue_piece(gc, buf);     // (ablation 10: the rotated loop without its DMA issue)
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            static_assert(LT <= 4 * RT, "one DMA piece per MFMA row group");
            buf = bufn1;
        }
        // the last three steps issued fetch-nothing pieces: they must have written their zeros before the epilogue reuses LDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (DMA) {
        // NBUF (3 or 4) LDS buffers.  Step t: issue tile t+NBUF-1's DMA into the buffer read at step t-1 (every wave has
        // passed the barrier that ended that step), run tile t's MFMAs, then wait until all but the newest NBUF-2 tiles'
        // DMA have landed (counted vmcnt: tile t+1 is in LDS) and barrier.  No VGPR staging, no ds_write.
#if TTS_ABL
        constexpr int LT = PL * ((HALF && (TTS_ABL == 4 || TTS_ABL == 6)) ? PB : PA + PB);
#else
        constexpr int LT = PL * (PA + PB);           // DMA instructions per tile per wave
#endif
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        auto dma_tile = [&](int buf) {
            fetch_next([&](bool isA, int p, const __amdgpu_buffer_rsrc_t& rs, unsigned voff, unsigned pofs) {
                // wave-uniform LDS base of this instruction's 16 rows; lane l lands at base + 16 * l
                float* dst = (isA ? As + buf * PL * BM * LDSK : Bs + buf * PL * BN * LDSK) + (p * RPP + wave * 16) * LDSK;
#if TTS_ABL
                if (HALF && (TTS_ABL == 4 || TTS_ABL == 6) && isA) return;       // ablation: no A-side DMA
#endif
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
                if constexpr (PL == 2) {             // the "lo" plane: same rows, second LDS image
                    float* dst2 = dst + (isA ? BM : BN) * LDSK;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst2, 16, voff + pofs, 0, 0, 0);
                }
            });
        };
        // prologue: NBUF - 1 tiles in flight (dma_tile is a no-op past the last tile, but still counts no instructions,
        // so the counted waits below are only used while enough real tiles remain)
        const bool loader = wave < LW;               // wave-uniform
        if (loader) {
#pragma unroll
            for (int b = 0; b < NBUF - 1; ++b) dma_tile(b);
        }
        // wait for tile 0: at most min(nAll, NBUF - 1) - 1 newer tiles may stay in flight
        if (nAll >= NBUF - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * LT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int buf = 0, bufn = NBUF - 1;                // buffer of tile t, buffer for tile t + NBUF - 1
#ifdef TTS_ABL_F32
        constexpr int ABL = TTS_ABL;                 // measurement builds: the same ablations on the fp32 loop
#else
        constexpr int ABL = HALF ? TTS_ABL : 0;      // timing ablations of the fp16 loop (results are garbage when != 0)
#endif
        for (int t = 0; t < nAll; ++t) {
            if (ABL != 2 && loader) dma_tile(bufn);
            if (ABL != 3) {
#pragma unroll
                for (int k8 = 0; k8 < BK / 8; ++k8) compute_chunk(buf, k8);
            }
            if (ABL != 1) {
                // tile t + 1 must have landed: tiles t + 2 .. t + NBUF - 1 may stay in flight while they all exist
                if (t + NBUF - 1 < nAll) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * LT) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            buf = buf == NBUF - 1 ? 0 : buf + 1;
            bufn = bufn == NBUF - 1 ? 0 : bufn + 1;
        }
    } else {
        // Two register stages: tile t+1 waits in one while tile t+2 is being fetched into the other (prefetch distance
        // of two K steps).  Step t on LDS buffer `buf`: fetch tile t+2 into the free stage, run all but the last
        // quarter of the MFMAs of tile t, write tile t+1 to the other LDS buffer, finish the MFMAs, barrier.
        f32x4 ra0[PA], rb0[PB], ra1[PA], rb1[PB];
        auto load_next = [&](f32x4 (&ra)[PA], f32x4 (&rb)[PB]) {
            fetch_next([&](bool isA, int p, const __amdgpu_buffer_rsrc_t& rs, unsigned voff, unsigned) {
                if (isA) ra[p] = buf_load4(rs, voff);
                else rb[p] = buf_load4(rs, voff);
            });
        };
        auto store_tile = [&](int buf, const f32x4 (&ra)[PA], const f32x4 (&rb)[PB]) {
            float* a = As + buf * BM * LDSK;
            float* b = Bs + buf * BN * LDSK;
#pragma unroll
            for (int p = 0; p < PA; ++p) *reinterpret_cast<f32x4*>(a + (p * RPP + lrow) * LDSK + c4) = ra[p];
#pragma unroll
            for (int p = 0; p < PB; ++p) *reinterpret_cast<f32x4*>(b + (p * RPP + lrow) * LDSK + c4) = rb[p];
        };
        auto k_step = [&](int buf, bool has_next, f32x4 (&ra_ld)[PA], f32x4 (&rb_ld)[PB], const f32x4 (&ra_st)[PA],
                          const f32x4 (&rb_st)[PB]) {
            load_next(ra_ld, rb_ld);
#pragma unroll
            for (int k8 = 0; k8 < BK / 8 - 1; ++k8) compute_chunk(buf, k8);
            if (has_next) store_tile(buf ^ 1, ra_st, rb_st);
            compute_chunk(buf, BK / 8 - 1);
            __syncthreads();
        };
        load_next(ra0, rb0);                             // tile 0
        store_tile(0, ra0, rb0);
        load_next(ra1, rb1);                             // tile 1 (kept in stage 1)
        __syncthreads();
        for (int t = 0; t < nAll; t += 2) {
            k_step(0, t + 1 < nAll, ra0, rb0, ra1, rb1);                 // fetch t+2 -> stage 0, write t+1 from stage 1
            if (t + 1 < nAll) k_step(1, t + 2 < nAll, ra1, rb1, ra0, rb0);   // fetch t+3 -> stage 1, write t+2 from stage 0
        }
    }

    // ---------------- epilogue ----------------
    // C/D map of a 32 x 32 tile: trow(r) / tcol(r) (defined with the accumulators above)
    if constexpr (CT % 2 == 0) {
        if (g.mode == EPI_GATE) {
            // weight rows are permuted at load time in groups of 64: 32 tanh pre-activations followed by the 32 matching
            // sigmoid pre-activations, so MFMA column tiles (2q, 2q + 1) of a wave pair up; output channel =
            // (first column of the pair / 64) * 32 + lane column.
            float* out = outp;
            if constexpr (DMA) {
                if (g.wide_epi) {
                    // same LDS transpose as the linear epilogue below: the 32 x 32 gated tile of a (tanh, sigmoid) pair
                    // leaves as 16-byte (fp32) / 8-byte (fp16 planes) stores, 8 lanes per row segment
                    __builtin_amdgcn_s_barrier();
                    float* patch = smem + wave * (32 * 36);
                    const int prow = lane >> 3, pc4 = (lane & 7) * 4;
#pragma unroll
                    for (int i = 0; i < RT; ++i)
#pragma unroll
                        for (int q = 0; q < CT / 2; ++q) {
                            const int ch4 = ((n0 + wc * CT * 32) / 64 + q) * 32 + pc4;
                            const int mrow0 = m0 + wr * RT * 32 + i * 32;
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                patch[trow(r) * 36 + tcol(r)] = gate_tanh_sigmoid(A(i, 2 * q, r), A(i, 2 * q + 1, r));
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                            for (int qq = 0; qq < 4; ++qq) {
                                const f32x4 v = *reinterpret_cast<const f32x4*>(patch + (prow + 8 * qq) * 36 + pc4);
                                const long long mrow = mrow0 + prow + 8 * qq;
                                if constexpr (HALF) {
                                    typedef _Float16 f16x4e __attribute__((ext_vector_type(4)));
                                    const f16x4e hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                                    *reinterpret_cast<f16x4e*>(g.out0h + mrow * g.ld0h + ch4) = hv;
                                    if constexpr (PL == 2) {
                                        const f16x4e lv = {(_Float16)(v[0] - (float)hv[0]), (_Float16)(v[1] - (float)hv[1]),
                                                           (_Float16)(v[2] - (float)hv[2]), (_Float16)(v[3] - (float)hv[3])};
                                        *reinterpret_cast<f16x4e*>(g.out0h + g.planeOut + mrow * g.ld0h + ch4) = lv;
                                    }
                                } else {
                                    *reinterpret_cast<f32x4*>(out + mrow * g.ld0 + ch4) = v;
                                }
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
                    return;
                }
            }
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int q = 0; q < CT / 2; ++q) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ch = ((n0 + wc * CT * 32) / 64 + q) * 32 + tcol(r);
                        const int m = rbase + i * 32 + trow(r);
                        if (m < g.M) {
                            const float gv = gate_tanh_sigmoid(A(i, 2 * q, r), A(i, 2 * q + 1, r));
                            if constexpr (HALF) {
                                const _Float16 hv = (_Float16)gv;
                                g.out0h[(long long)m * g.ld0h + ch] = hv;
                                if constexpr (PL == 2) g.out0h[g.planeOut + (long long)m * g.ld0h + ch] = (_Float16)(gv - (float)hv);
                            } else {
                                out[(long long)m * g.ld0 + ch] = gv;
                            }
                        }
                    }
                }
            return;
        }
    }
    if constexpr (DMA) {
        if (g.wide_epi) {
            // The C/D layout of the 32 x 32 MFMA gives a lane one column and 16 scattered rows: stored directly that is
            // 16 dword stores (+ 16 loads for a read-modify-write, + 2-byte stores for fp16 shadows) per tile.  The pipeline
            // buffers are free now, so each wave transposes its tiles one at a time through a private 32 x 36 float patch of
            // LDS: afterwards lane l holds 4 consecutive columns of row (l / 8) + 8 pass, i.e. 16-byte accesses, 8 lanes per
            // 128-byte row segment.  (Residual GEMM of the split-fp16 mode: 0.61 -> see DESIGN.)
            float* patch = smem + wave * (32 * 36);
            const int prow = lane >> 3, pc4 = (lane & 7) * 4;
            bool accum = second ? g.acc1 : g.acc0;
#if TTS_ABL == 11 || TTS_ABL == 12
            // ablations 11 / 12 (timing only): the fp16 residual launch without the fp32 master of x -- 11: no read, no fp32 store
            // (what an fp16-only residual stream would move); 12: the master as an fp16 (hi, lo) pair (4 B in, 4 B out, no shadow)
            constexpr bool kNoMaster = HALF && TTS_ABL == 11, kPairMaster = HALF && TTS_ABL == 12;
            if (kNoMaster) accum = false;
#else
            constexpr bool kNoMaster = false, kPairMaster = false;
#endif
            // read-modify-write: in the HBM-bound fp16 residual launch the old values of ALL this wave's tiles are requested up
            // front (tile by tile they are RT x CT dependent HBM latencies per block): 290 -> 263 - 273 us.  Not in the fp32 and
            // split-fp16 kernels, which are MFMA-bound and lose 1.5 % / gain nothing with 64 more live registers in the epilogue.
#ifndef TTS_EPI_PREFETCH
#define TTS_EPI_PREFETCH 1
#endif
            constexpr bool kPrefetch = TTS_EPI_PREFETCH && HALF && PL == 1 && RT * CT <= 4;
            f32x4 oldall[kPrefetch ? RT * CT : 1][4];
            if (kPrefetch && accum) {
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            oldall[kPrefetch ? i * CT + j : 0][q] = *reinterpret_cast<const f32x4*>(
                                outp + (long long)(m0 + wr * RT * 32 + i * 32 + prow + 8 * q) * ldo + ncol0 + j * 32 + pc4);
            }
            __builtin_amdgcn_s_barrier();                         // every wave is done reading the last tile
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const int mrow0 = m0 + wr * RT * 32 + i * 32;
                    const int ncol = ncol0 + j * 32 + pc4;       // first of this lane's 4 output columns
                    f32x4 oldv[4];
                    if (accum) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            oldv[q] = kPrefetch ? oldall[kPrefetch ? i * CT + j : 0][q]
                                                : *reinterpret_cast<const f32x4*>(outp + (long long)(mrow0 + prow + 8 * q) * ldo + ncol);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) patch[trow(r) * 36 + tcol(r)] = A(i, j, r);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private patch: no barrier needed
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(patch + (prow + 8 * q) * 36 + pc4);
                        if (accum) v += oldv[q];
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = act_apply(v[k], g.act);
                        const long long mrow = mrow0 + prow + 8 * q;
                        if (!kNoMaster && !kPairMaster) *reinterpret_cast<f32x4*>(outp + mrow * ldo + ncol) = v;
                        if constexpr (HALF) {
                            if (kPairMaster) {                   // (hi, lo) pair written into the fp32 master's bytes: 4 B per element
                                typedef _Float16 f16x4p __attribute__((ext_vector_type(4)));
                                const f16x4p hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                                const f16x4p lv = {(_Float16)(v[0] - (float)hv[0]), (_Float16)(v[1] - (float)hv[1]),
                                                   (_Float16)(v[2] - (float)hv[2]), (_Float16)(v[3] - (float)hv[3])};
                                _Float16* mp = reinterpret_cast<_Float16*>(outp + mrow * ldo + ncol);
                                *reinterpret_cast<f16x4p*>(mp) = hv;
                                *reinterpret_cast<f16x4p*>(mp + 4) = lv;
                                continue;
                            }
                            if (g.out0h && !second) {
                                typedef _Float16 f16x4e __attribute__((ext_vector_type(4)));
                                const f16x4e hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                                *reinterpret_cast<f16x4e*>(g.out0h + mrow * g.ld0h + ncol) = hv;
                                if constexpr (PL == 2) {
                                    const f16x4e lv = {(_Float16)(v[0] - (float)hv[0]), (_Float16)(v[1] - (float)hv[1]),
                                                       (_Float16)(v[2] - (float)hv[2]), (_Float16)(v[3] - (float)hv[3])};
                                    *reinterpret_cast<f16x4e*>(g.out0h + g.planeOut + mrow * g.ld0h + ncol) = lv;
                                }
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // patch reads done before the next tile overwrites it
                }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = cbase + j * 32 + tcol(r);
                if (n >= g.N) continue;
                const int nc = ncol0 + j * 32 + tcol(r);
                const float ab = g.altbias ? g.altbias[n] : 0.f;
                const int m = rbase + i * 32 + trow(r);
                if (m < g.M) {
                    float v = A(i, j, r);
                    if (g.rowmask) {
                        const bool on = g.rowmask[m] != 0;
                        v = on ? v : (g.mask_out ? 0.f : ab);
                    }
                    const float ov = act_apply(v, g.act);
                    outp[(long long)m * ldo + nc] = ov;
                    if constexpr (HALF) {
                        if (g.out0h && !second) {
                            const _Float16 hv = (_Float16)ov;
                            g.out0h[(long long)m * g.ld0h + nc] = hv;
                            if constexpr (PL == 2) g.out0h[g.planeOut + (long long)m * g.ld0h + nc] = (_Float16)(ov - (float)hv);
                        }
                    }
                }
            }
        }
}

template <int WR, int WC, int RT, int CT, int BK, int OCC, int TAG, int NI = 0, int PIPE = PIPE_REG, bool HALF = false,
          int NBD = 3, int PL = 1>
inline hipError_t launch_gemm(const GemmArgs& g, int batch_z, hipStream_t stream) {
    constexpr int BM = WR * RT * 32;
    constexpr int BN = WC * CT * 32;
    constexpr int LDSK = PIPE == PIPE_DMA ? BK : BK + 4;
    constexpr int NBUF = PIPE == PIPE_DMA ? NBD : 2;
    const int numNt = (g.N + BN - 1) / BN;
    const int numMt = (g.M + BM - 1) / BM;
    const int numMt8 = (numMt + 7) / 8 * 8;
    const size_t lds = (size_t)NBUF * PL * (BM + BN) * LDSK * sizeof(float);
    if (g.split < g.N && g.split % BN != 0) return hipErrorInvalidValue;     // output side must be uniform per block
    if (g.phase_rows > 0) {
        // tiles must not straddle phases; the interleaved-tap descriptor spans at most the whole operand (31-bit offsets)
        const int nph = g.nphase > 0 ? g.nphase : 32;
        if (g.phase_rows % BM != 0 || g.M != nph * g.phase_rows) return hipErrorInvalidValue;
        if ((double)nph * g.phase_rows * (double)g.seg[0].ld * 4.0 >= 2147483648.0 - 65536.0) return hipErrorInvalidValue;
    }
    if (g.wide_epi) {
        if (PIPE != PIPE_DMA || g.rowmask || g.M % BM != 0 || g.N % BN != 0 || g.ld0 % 4 != 0 ||
            (g.split < g.N && g.ld1 % 4 != 0) || (size_t)(WR * WC) * 32 * 36 * sizeof(float) > lds)
            return hipErrorInvalidValue;
    }
    if (NI > 0) {
        if (g.nseg < NI) return hipErrorInvalidValue;
        for (int i = 1; i < NI; ++i)
            if (g.seg[i].k != g.seg[0].k || g.seg[i].kpad != g.seg[0].kpad || g.seg[i].ptr != g.seg[0].ptr ||
                g.seg[i].ld != g.seg[0].ld || g.seg[i].plane != g.seg[0].plane)
                return hipErrorInvalidValue;
    }
    auto kern = gemm_f32_kernel<WR, WC, RT, CT, BK, OCC, TAG, NI, PIPE, HALF, NBD, PL>;
    // two engine handles may launch from two host threads (stream(overlap=True)): the flag is atomic, and setting the
    // attribute twice is harmless
    static PerDeviceOnce attr_set;
    if (hipError_t e = set_max_dyn_lds_once((const void*)kern, lds, attr_set); e != hipSuccess) return e;
    dim3 grid(numMt8 * numNt, 1, batch_z);
    hipLaunchKernelGGL(kern, grid, dim3(WR * WC * 64), lds, stream, g);
    return hipGetLastError();
}

// Tile configurations: BIG = 256x128 (WN layers, upsampling); SMALL = 64x64 (Tacotron2-sized problems).
enum { TAG_GENERIC = 0, TAG_WN_IN = 1, TAG_WN_RES_SKIP = 2, TAG_WN_IN0 = 3, TAG_WN_WINO = 4 };
#ifndef TTS_WN_BK
#define TTS_WN_BK 16
#define TTS_WN_OCC 2
#endif
#ifndef TTS_WN_RT
#define TTS_WN_WR 4      // 4 x 1 waves, each RT x 4 tiles of 32x32: block tile (128 * RT) x 128
#define TTS_WN_RT 2
#endif
#ifndef TTS_WN_PIPE
#define TTS_WN_PIPE PIPE_DMA
#endif
inline hipError_t gemm_big(const GemmArgs& g, int bz, hipStream_t s) { return launch_gemm<4, 1, 2, 4, 32, 1, TAG_GENERIC>(g, bz, s); }
// WN in-layer GEMM: the three conv taps are interleaved in K (weights packed to match, see pack_bt_kernel)
constexpr int WN_TAPS = 3;
inline hipError_t gemm_wn_in(const GemmArgs& g, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, TTS_WN_RT, 4, TTS_WN_BK, TTS_WN_OCC, TAG_WN_IN, WN_TAPS, TTS_WN_PIPE>(g, 1, s); }
// 128-row-tile variants: used when padding the phase blocks to 256 rows would waste more work (e.g. batch 1)
inline hipError_t gemm_wn_in_128(const GemmArgs& g, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, 1, 4, TTS_WN_BK, 3, TAG_WN_IN, WN_TAPS, TTS_WN_PIPE>(g, 1, s); }
inline hipError_t gemm_wn_in0_128(const GemmArgs& g, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, 1, 4, TTS_WN_BK, 3, TAG_WN_IN0, WN_TAPS, TTS_WN_PIPE>(g, 1, s); }
// 128 x 64 tiles for short utterances (a few hundred frames at batch 1): twice the blocks, so that every CU gets work
inline hipError_t gemm_wn_in_64(const GemmArgs& g, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, 1, 2, TTS_WN_BK, 4, TAG_WN_IN, WN_TAPS, TTS_WN_PIPE>(g, 1, s); }
inline hipError_t gemm_wn_in0_64(const GemmArgs& g, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, 1, 2, TTS_WN_BK, 4, TAG_WN_IN0, WN_TAPS, TTS_WN_PIPE>(g, 1, s); }
inline hipError_t gemm_wn_res_64(const GemmArgs& g, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, 1, 2, TTS_WN_BK, 4, TAG_WN_RES_SKIP, 0, TTS_WN_PIPE>(g, 1, s); }
inline hipError_t gemm_wn_in_64h(const GemmArgs& g, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, 1, 2, TTS_WN_BK, 4, TAG_WN_IN, WN_TAPS, PIPE_DMA, true>(g, 1, s); }
inline hipError_t gemm_wn_in0_64h(const GemmArgs& g, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, 1, 2, TTS_WN_BK, 4, TAG_WN_IN0, WN_TAPS, PIPE_DMA, true>(g, 1, s); }
inline hipError_t gemm_wn_res_64h(const GemmArgs& g, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, 1, 2, TTS_WN_BK, 4, TAG_WN_RES_SKIP, 0, PIPE_DMA, true>(g, 1, s); }
// 64 x 128 tiles (2 x 2 waves): phase blocks are padded to the M tile, so short utterances waste up to BM - 1 frames per
// phase; 64-row tiles halve that padding (e.g. 170 frames: 192 rows per phase instead of 256)
inline hipError_t gemm_wn_in_r64(const GemmArgs& g, hipStream_t s) { return launch_gemm<2, 2, 1, 2, TTS_WN_BK, 4, TAG_WN_IN, WN_TAPS, TTS_WN_PIPE>(g, 1, s); }
inline hipError_t gemm_wn_in0_r64(const GemmArgs& g, hipStream_t s) { return launch_gemm<2, 2, 1, 2, TTS_WN_BK, 4, TAG_WN_IN0, WN_TAPS, TTS_WN_PIPE>(g, 1, s); }
inline hipError_t gemm_wn_res_r64(const GemmArgs& g, hipStream_t s) { return launch_gemm<2, 2, 1, 2, TTS_WN_BK, 4, TAG_WN_RES_SKIP, 0, TTS_WN_PIPE>(g, 1, s); }
inline hipError_t gemm_wn_in_r64h(const GemmArgs& g, hipStream_t s) { return launch_gemm<2, 2, 1, 2, TTS_WN_BK, 4, TAG_WN_IN, WN_TAPS, PIPE_DMA, true>(g, 1, s); }
inline hipError_t gemm_wn_in0_r64h(const GemmArgs& g, hipStream_t s) { return launch_gemm<2, 2, 1, 2, TTS_WN_BK, 4, TAG_WN_IN0, WN_TAPS, PIPE_DMA, true>(g, 1, s); }
inline hipError_t gemm_wn_res_r64h(const GemmArgs& g, hipStream_t s) { return launch_gemm<2, 2, 1, 2, TTS_WN_BK, 4, TAG_WN_RES_SKIP, 0, PIPE_DMA, true>(g, 1, s); }
// split-fp16 variants (two fp16 planes per operand, three MFMAs per product), 8 waves.  In-layer GEMM: 256 x 256 block tile
// with TWO LDS buffers (131 KB; a K step carries 3x the MFMA work of the fp16 kernel, so one tile of prefetch covers the
// latency: 2 142 us vs 2 429 us for 256 x 128 tiles with three buffers); 64 x 128 tiles for short utterances.
inline hipError_t gemm_wn_in_x3(const GemmArgs& g, bool small, hipStream_t s) {
    return small ? launch_gemm<2, 2, 1, 2, TTS_WN_BK, 2, TAG_WN_IN, WN_TAPS, PIPE_DMA, true, 3, 2>(g, 1, s)
                 : launch_gemm<4, 2, 2, 4, TTS_WN_BK, 1, TAG_WN_IN, WN_TAPS, PIPE_DMA, true, 2, 2>(g, 1, s);
}
#ifndef TTS_X3_VAR
#define TTS_X3_VAR 1   // bit 0: 256 x 256 tiles for the first layer of a flow (613 vs 690 us)
#endif
inline hipError_t gemm_wn_in0_x3(const GemmArgs& g, bool small, hipStream_t s) {
    if (!small && (TTS_X3_VAR & 1)) return launch_gemm<4, 2, 2, 4, TTS_WN_BK, 1, TAG_WN_IN0, WN_TAPS, PIPE_DMA, true, 2, 2>(g, 1, s);
    return small ? launch_gemm<2, 2, 1, 2, TTS_WN_BK, 2, TAG_WN_IN0, WN_TAPS, PIPE_DMA, true, 3, 2>(g, 1, s)
                 : launch_gemm<4, 2, 2, 2, TTS_WN_BK, 1, TAG_WN_IN0, WN_TAPS, PIPE_DMA, true, 3, 2>(g, 1, s);
}
// residual GEMM (K = 512: short main loop, heavy read-modify-write epilogue): 128 x 128 tiles, 4 waves, two LDS buffers =
// 64 KB, so two blocks share a CU and one block's epilogue overlaps the other's loop (494 us; 540-570 us with one
// 256 x 128 block per CU, 584 us with 64 x 128 tiles)
inline hipError_t gemm_wn_res_x3(const GemmArgs& g, bool small, hipStream_t s) {
    return small ? launch_gemm<2, 2, 1, 2, TTS_WN_BK, 2, TAG_WN_RES_SKIP, 0, PIPE_DMA, true, 3, 2>(g, 1, s)
                 : launch_gemm<2, 2, 2, 2, TTS_WN_BK, 2, TAG_WN_RES_SKIP, 0, PIPE_DMA, true, 2, 2>(g, 1, s);
}
// fp16-operand variants (activations and weights fp16 in HBM, fp32 accumulate): same tiles and pipeline
#ifndef TTS_H_NBUF
#define TTS_H_NBUF 3   // LDS buffers of that kernel (4 = three tiles in flight was measured no faster: 898 vs 855-885 us)
#endif
#ifndef TTS_H_WC
#define TTS_H_WC 2     // fp16 in-layer GEMM: 8 waves, 256 x 256 block tile (the fp16 loop is bound by L2 -> LDS traffic)
#endif
inline hipError_t gemm_wn_in_h(const GemmArgs& g, bool t128, hipStream_t s) {
    if (!t128 && TTS_H_WC == 2)
        return launch_gemm<TTS_WN_WR, 2, TTS_WN_RT, 4, TTS_WN_BK, 1, TAG_WN_IN, WN_TAPS, PIPE_DMA, true, TTS_H_NBUF>(g, 1, s);
    return t128 ? launch_gemm<TTS_WN_WR, 1, 1, 4, TTS_WN_BK, 3, TAG_WN_IN, WN_TAPS, PIPE_DMA, true>(g, 1, s)
                : launch_gemm<TTS_WN_WR, 1, TTS_WN_RT, 4, TTS_WN_BK, TTS_WN_OCC, TAG_WN_IN, WN_TAPS, PIPE_DMA, true>(g, 1, s);
}
inline hipError_t gemm_wn_in0_h(const GemmArgs& g, bool t128, hipStream_t s) {
    return t128 ? launch_gemm<TTS_WN_WR, 1, 1, 4, TTS_WN_BK, 3, TAG_WN_IN0, WN_TAPS, PIPE_DMA, true>(g, 1, s)
                : launch_gemm<TTS_WN_WR, 1, TTS_WN_RT, 4, TTS_WN_BK, TTS_WN_OCC, TAG_WN_IN0, WN_TAPS, PIPE_DMA, true>(g, 1, s);
}
#ifndef TTS_HRES_WC
#define TTS_HRES_WC 1
#define TTS_HRES_RT 1
#define TTS_HRES_OCC 3
#endif
inline hipError_t gemm_wn_res_h(const GemmArgs& g, hipStream_t s) {
    return launch_gemm<TTS_WN_WR, TTS_HRES_WC, TTS_HRES_RT, 4, TTS_WN_BK, TTS_HRES_OCC, TAG_WN_RES_SKIP, 0, PIPE_DMA, true>(g, 1, s);
}
// first layer of a flow (start conv composed into the taps: K = 48 + 320); own TAG so profiles list it separately
inline hipError_t gemm_wn_in0(const GemmArgs& g, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, TTS_WN_RT, 4, TTS_WN_BK, TTS_WN_OCC, TAG_WN_IN0, WN_TAPS, TTS_WN_PIPE>(g, 1, s); }
#ifndef TTS_WN_RES_RT
#define TTS_WN_RES_RT 1   // residual GEMM (N = 512): 128-row tiles -> 6400 blocks, fills the 512 block slots more evenly
#define TTS_WN_RES_OCC 3
#endif
inline hipError_t gemm_wn_res_skip(const GemmArgs& g, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, TTS_WN_RES_RT, 4, TTS_WN_BK, TTS_WN_RES_OCC, TAG_WN_RES_SKIP, 0, TTS_WN_PIPE>(g, 1, s); }
// Winograd form of the in-layer GEMM (wn_wino.hip): four z slices (one per product) on pair rows, K = 512 + 160 each
inline hipError_t gemm_wn_wino(const GemmArgs& g, int nz, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, TTS_WN_RT, 4, TTS_WN_BK, TTS_WN_OCC, TAG_WN_WINO, 0, TTS_WN_PIPE>(g, nz, s); }
inline hipError_t gemm_wn_wino_128(const GemmArgs& g, int nz, hipStream_t s) { return launch_gemm<TTS_WN_WR, 1, 1, 4, TTS_WN_BK, 3, TAG_WN_WINO, 0, TTS_WN_PIPE>(g, nz, s); }
inline hipError_t gemm_small(const GemmArgs& g, int bz, hipStream_t s) { return launch_gemm<2, 2, 1, 1, 32, 1, TAG_GENERIC>(g, bz, s); }

}  // namespace ttsgemm
