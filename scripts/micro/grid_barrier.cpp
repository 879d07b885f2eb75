// Microbenchmark: cost of a grid-wide barrier + all-to-all exchange of a small vector between the 256 CUs of an MI355X
// (what a persistent Tacotron2 decoder kernel would pay per dependency edge of a decoder step).
//   mode 0: barrier only (one atomic counter, monotonic target)
//   mode 1: barrier + every block publishes 4*B floats and then reads the whole 1024*B vector (device-scope accesses)
//   mode 2: like 1, hierarchical barrier (one counter per XCD, then a global one)
//   mode 3: like 1, flag barrier: every block stores the step to its own flag, all 256 threads poll the 256 flags with
//           relaxed device-scope loads (no read-modify-write atomics, no acquire per poll), one fence at the end
//   mode 4: like 3 but the counter barrier with relaxed polling
// Every spin loop is bounded (SPIN_LIMIT) so the grid always drains.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr long long SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ unsigned ld_acquire(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long long spins = 0;
        while (ld_acquire(counter) < target) {
            if (++spins > SPIN_LIMIT) { ok = false; *err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counters, float* vec, float* sink, int steps, int mode,
                                                      int B, int* err) {
    const int nb = gridDim.x;
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        float* v = vec + (size_t)(s & 1) * 1024 * 8;
        if (mode >= 1) {
            // publish this block's slice (4 floats per batch row), device scope so that other XCDs see it
            if (tid < 4 * B)
                __hip_atomic_store(v + (tid / 4) * 1024 + blockIdx.x * 4 + (tid & 3), acc + (float)s, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (mode == 2) {
            const int xcd = blockIdx.x & 7;
            __syncthreads();
            if (tid == 0) {
                const unsigned old = __hip_atomic_fetch_add(counters + 64 + xcd * 64, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if ((old + 1) % (nb / 8) == 0)      // last arrival of this XCD in this round
                    __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                long long spins = 0;
                while (ld_acquire(counters) < (unsigned)(8 * (s + 1))) {
                    if (++spins > SPIN_LIMIT) { *err = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        } else if (mode == 3) {
            __syncthreads();                       // this block's published data is issued
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (tid == 0) __hip_atomic_store(counters + 1024 + blockIdx.x, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long spins = 0;
            while (true) {
                const unsigned f = tid < nb ? __hip_atomic_load(counters + 1024 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                if (__syncthreads_and(f >= (unsigned)(s + 1))) break;
                if (++spins > SPIN_LIMIT) { *err = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else if (mode == 4) {
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                long long spins = 0;
                while (__hip_atomic_load(counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(nb * (s + 1))) {
                    if (++spins > SPIN_LIMIT) { *err = 1; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        } else {
            if (!grid_barrier(counters, (unsigned)(nb * (s + 1)), err)) break;
        }
        if (*err) break;
        if (mode >= 1) {
            for (int i = tid; i < 1024 * B; i += 256)
                acc += __hip_atomic_load(v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 12345.f) sink[0] = acc;
}

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int nb = prop.multiProcessorCount;
    unsigned* counters;
    float *vec, *sink;
    int* err;
    CHECK(hipMalloc(&counters, 4096 * 4));
    CHECK(hipMalloc(&vec, 2 * 1024 * 8 * 4));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&err, 4));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    for (int mode = 0; mode < 5; ++mode)
        for (int B : {1, 8}) {
            if (mode == 0 && B == 8) continue;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipMemsetAsync(counters, 0, 4096 * 4, st));
                CHECK(hipMemsetAsync(err, 0, 4, st));
                int steps = 2000;
                void* args[] = {&counters, &vec, &sink, &steps, &mode, &B, &err};
                CHECK(hipStreamSynchronize(st));
                auto t0 = std::chrono::steady_clock::now();
                CHECK(hipLaunchCooperativeKernel((const void*)barrier_kernel, dim3(nb), dim3(256), args, 0, st));
                CHECK(hipStreamSynchronize(st));
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                int herr = 0;
                CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                if (rep == 1) printf("mode %d B %d blocks %d: %.2f us per barrier%s\n", mode, B, nb, us / steps, herr ? "  (SPIN LIMIT HIT)" : "");
            }
        }
    return 0;
}
