// Microbenchmark: what the fp32 matrix pipe sustains chip-wide on v_mfma_f32_32x32x2_f32 with the register traffic of the
// WN in-layer GEMM (8 independent 32x32 accumulators per wave, operands re-used from registers), and the shader clock it
// holds meanwhile (s_memtime ticks per 100-MHz s_memrealtime tick), for all-zero and for N(0,1)-like operands.
//   usage: mfma_f32_rate [waves_per_simd (1|2)] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, unsigned long long* clk,
                                                    int iters) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = in[(threadIdx.x * 16 + i) & 4095];
        b[i] = in[(threadIdx.x * 16 + 8 + i + blockIdx.x) & 4095];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + i) & 7], b[(k + 3 * i) & 7], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    float *in, *out;
    unsigned long long* clk;
    hipMalloc(&in, 4096 * 4);
    hipMalloc(&out, (size_t)ncu * 2 * 256 * 4);
    hipMalloc(&clk, 16);
    std::vector<float> h(4096);
    for (int mode = 0; mode < 3; ++mode) {
        srand(1);
        for (auto& v : h) {
            float u = 0;
            for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX;
            v = mode == 0 ? 0.f : mode == 1 ? u - 6.f : (u - 6.f) * 1e-3f;
        }
        hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<8>, dim3(ncu * wps), dim3(256), 0, 0, in, out, clk, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c[2];
            hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            const double flop = (double)ncu * wps * 4 * iters * 64.0 * 4096.0;
            printf("%s operands, %d wave(s) per SIMD: %.2f ms, %.1f TFLOP/s, shader clock %.3f GHz (%llu / %llu ticks)\n",
                   mode == 0 ? "zero" : mode == 1 ? "N(0,1)" : "N(0,1e-3)", wps, ms, flop / ms / 1e9, c[0] / (c[1] * 10.0), c[0], c[1]);
        }
    }
    return 0;
}
