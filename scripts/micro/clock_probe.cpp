// Microbenchmark: shader clock seen by tiny dependent kernels (s_memtime ticks per s_memrealtime 100 MHz tick).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void probe(unsigned long long* out, float* sink, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (v == 12345.f) *sink = v;
}
__global__ void heavy(float* sink, int iters) {
    float v = threadIdx.x, w = blockIdx.x;
    for (int i = 0; i < iters; ++i) { v = fmaf(v, 1.0001f, w); w = fmaf(w, 0.9999f, v); }
    if (v == 12345.f) *sink = v + w;
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long* d; hipMalloc(&d, 16); float* sink; hipMalloc(&sink, 4);
    unsigned long long h[2];
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 1) { hipLaunchKernelGGL(heavy, dim3(2048), dim3(256), 0, s, sink, 2000000); }   // ~ tens of ms of load first
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(probe, dim3(64), dim3(256), 0, s, d, sink, 2000);
            hipStreamSynchronize(s);
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("mode %d rep %d: %llu shader ticks / %llu realtime ticks -> %.2f GHz\n", mode, rep, h[0], h[1], h[0] / (h[1] * 10.0) );
        }
    }
    return 0;
}
