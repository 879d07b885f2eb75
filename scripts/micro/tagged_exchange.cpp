// Microbenchmark: tagged (step, value) exchange between the CUs of an MI355X WITHOUT a barrier -- the mechanism the BiLSTM
// kernel uses between 4 blocks (csrc/tacotron2.hip), here at the scale a persistent decoder step would need it.
// Every value is published as ONE 8-byte (tag, fp32) agent-scope store into a parity double buffer; a consumer polls the
// words it needs until the tag equals the step.  A hop = publish -> visible to every consumer.
//   pattern 0: all -> all     every block publishes V values, every block reads all nb*V   (LSTM h: V = 4 B)
//   pattern 1: P -> all       the first P blocks publish V values, every block reads P*V   (prenet / query / energies)
//   pattern 2: P -> P         the first P blocks publish and read; the others idle in a poll of one word (chain inside a
//                             small group of CUs)
// poll kinds: 0 = 8-byte atomic loads (one tagged value per lane and load), 1 = 16-byte sc1 loads (two tagged values).
// Each step's published value depends on the polled values of the previous step, so hops cannot overlap.
// Every spin loop is bounded (SPIN_LIMIT) so the grid always drains.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr long long SPIN_LIMIT = 1 << 20;
typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u64 ld8(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32x4 ld16(const u64* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int POLL>
__global__ __launch_bounds__(256) void exchange_kernel(u64* buf, float* sink, int steps, int pattern, int P, int V,
                                                       int sleep, int* err) {
    const int nb = gridDim.x, tid = threadIdx.x, blk = blockIdx.x;
    const int producers = pattern == 0 ? nb : P;
    const bool produce = blk < producers;
    const bool consume = pattern != 2 || blk < P;
    const int total = producers * V;                       // tagged values per hop
    __shared__ float red[256];
    __shared__ int stop_s;
    if (tid == 0) stop_s = 0;
    __syncthreads();
    float carry = 0.f;
    for (int s = 0; s < steps; ++s) {
        u64* v = buf + (size_t)(s & 1) * 8192;
        const unsigned tag = (unsigned)(s + 1);
        if (produce && tid < V) {
            const float val = carry * 0.5f + (float)(blk + tid);
            __hip_atomic_store(v + blk * V + tid, ((u64)tag << 32) | __builtin_bit_cast(unsigned, val), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        float acc = 0.f;
        if (consume) {
            if (POLL == 0) {
                for (int i = tid; i < total; i += 256) {
                    long long spins = 0;
                    u64 w = ld8(v + i);
                    while ((unsigned)(w >> 32) != tag) {
                        if (++spins > SPIN_LIMIT) { *err = 1; stop_s = 1; break; }
                        if (sleep) __builtin_amdgcn_s_sleep(1);
                        w = ld8(v + i);
                    }
                    acc += __builtin_bit_cast(float, (unsigned)w);
                }
            } else {
                for (int i = tid * 2; i < total; i += 512) {
                    long long spins = 0;
                    u32x4 w = ld16(v + i);
                    while (w[1] != tag || w[3] != tag) {
                        if (++spins > SPIN_LIMIT) { *err = 1; stop_s = 1; break; }
                        if (sleep) __builtin_amdgcn_s_sleep(1);
                        w = ld16(v + i);
                    }
                    acc += __builtin_bit_cast(float, w[0]) + __builtin_bit_cast(float, w[2]);
                }
            }
        } else {
            // idle blocks follow the chain through one word so that they leave the loop with everyone else
            long long spins = 0;
            u64 w = ld8(v);
            while ((unsigned)(w >> 32) != tag) {
                if (++spins > SPIN_LIMIT) { *err = 1; stop_s = 1; break; }
                __builtin_amdgcn_s_sleep(8);
                w = ld8(v);
            }
        }
        // block reduction (what a step kernel does with the gathered vector) -> next published value depends on it
        red[tid] = acc;
        __syncthreads();
        if (tid < 64) {
            float a = red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192];
            for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
            if (tid == 0) red[0] = a;
        }
        __syncthreads();
        carry = red[0] * 1e-6f;
        if (stop_s) break;
        __syncthreads();
    }
    if (carry == 12345.f) sink[0] = carry;
}

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int nb = prop.multiProcessorCount;
    u64* buf;
    float* sink;
    int* err;
    CHECK(hipMalloc(&buf, 2 * 8192 * 8));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&err, 4));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    struct Cfg { int pattern, P, V, poll, sleep; const char* what; };
    const Cfg cfgs[] = {
        {0, 0, 4, 0, 1, "all->all   V=4  (LSTM h, B=1)        ld8  sleep"},
        {0, 0, 4, 0, 0, "all->all   V=4  (LSTM h, B=1)        ld8  spin"},
        {0, 0, 4, 1, 0, "all->all   V=4  (LSTM h, B=1)        ld16 spin"},
        {0, 0, 32, 0, 0, "all->all   V=32 (LSTM h, B=8)        ld8  spin"},
        {0, 0, 32, 1, 0, "all->all   V=32 (LSTM h, B=8)        ld16 spin"},
        {1, 64, 4, 0, 0, "64->all    V=4  (prenet 256 values)  ld8  spin"},
        {1, 32, 4, 0, 0, "32->all    V=4  (128 values)         ld8  spin"},
        {1, 32, 32, 1, 0, "32->all    V=32 (1024 values)        ld16 spin"},
        {2, 32, 4, 0, 0, "32->32     V=4  (128 values)         ld8  spin"},
        {2, 8, 16, 0, 0, "8->8       V=16 (128 values)         ld8  spin"},
        {2, 2, 64, 0, 0, "2->2       V=64 (128 values)         ld8  spin"},
    };
    for (const Cfg& c : cfgs) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemsetAsync(buf, 0, 2 * 8192 * 8, st));
            CHECK(hipMemsetAsync(err, 0, 4, st));
            int steps = 4000;
            int pattern = c.pattern, P = c.P, V = c.V, sleep = c.sleep;
            void* args[] = {&buf, &sink, &steps, &pattern, &P, &V, &sleep, &err};
            CHECK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            const void* fn = c.poll ? (const void*)exchange_kernel<1> : (const void*)exchange_kernel<0>;
            CHECK(hipLaunchCooperativeKernel(fn, dim3(nb), dim3(256), args, 0, st));
            CHECK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            int herr = 0;
            CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            if (rep == 1) printf("%-52s blocks %d: %.2f us per hop%s\n", c.what, nb, us / steps, herr ? "  (SPIN LIMIT HIT)" : "");
        }
    }
    return 0;
}
