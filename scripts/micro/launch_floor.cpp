// Microbenchmark: dependent kernel chain inside a hipGraph (what does one tiny kernel cost between two others?)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void empty_k(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void touch_k(const float* __restrict__ in, float* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.f;
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    float *a, *b; hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20); hipMemset(a, 0, 1 << 20);
    for (int mode = 0; mode < 3; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        const int per = 7 * 32;
        for (int i = 0; i < per; ++i) {
            if (mode == 0) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s, nullptr);
            else if (mode == 1) hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, s, nullptr);
            else hipLaunchKernelGGL(touch_k, dim3(64), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 64 * 256);
        }
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        const int reps = 25;
        for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("mode %d: %.2f us per kernel (%d kernels/graph)\n", mode, us / (reps * per), per);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
