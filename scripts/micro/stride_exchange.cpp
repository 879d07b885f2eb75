// Microbenchmark: all -> all tagged exchange of 1024 values (the LSTM h hop at batch 1) with the 16-byte (2 tagged values)
// units of the exchange area placed `stride` bytes apart.  Dense (stride 16) puts the whole 8 KB that 256 CUs poll into one
// or two memory channels; larger strides spread the polls over channels.  Also: sentinel polling -- one lane per block waits
// (with s_sleep) for the LAST unit to carry the tag before the block polls everything once.
// Every spin loop is bounded.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr long long SPIN_LIMIT = 1 << 18;

__global__ __launch_bounds__(256) void k(char* buf, float* sink, int steps, unsigned stride, unsigned half_bytes, int mode, int* err) {
    const int blk = blockIdx.x, tid = threadIdx.x;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 0x80000000u, 0x00020000);
    __shared__ float red[256];
    __shared__ int stop_s;
    if (tid == 0) stop_s = 0;
    __syncthreads();
    float carry = 0.f;
    for (int s = 0; s < steps; ++s) {
        const unsigned tag = s + 1;
        const unsigned base = (s & 1) * half_bytes;
        if (tid < 4) {          // value index blk * 4 + tid -> unit (blk * 4 + tid) / 2, slot & 1
            const unsigned vi = blk * 4 + tid;
            u64* p = (u64*)(buf + base + (size_t)(vi >> 1) * stride + (vi & 1) * 8);
            __hip_atomic_store(p, ((u64)tag << 32) | __builtin_bit_cast(unsigned, carry + vi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (mode == 1) {        // sentinel: wait for the last block's unit first (one lane, sleeping)
            if (tid == 0) {
                long long spins = 0;
                const unsigned off = base + 511u * stride;
                while (true) {
                    asm volatile("" ::: "memory");
                    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
                    if (w[1] == tag && w[3] == tag) break;
                    if (++spins > SPIN_LIMIT) { *err = 1; stop_s = 1; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
        }
        float acc = 0.f;
        {
            const unsigned o0 = base + (unsigned)tid * stride, o1 = base + (unsigned)(tid + 256) * stride;
            long long spins = 0;
            u32x4 a, b;
            while (true) {
                asm volatile("" ::: "memory");
                a = __builtin_amdgcn_raw_buffer_load_b128(rs, o0, 0, 16);
                b = __builtin_amdgcn_raw_buffer_load_b128(rs, o1, 0, 16);
                if (a[1] == tag && a[3] == tag && b[1] == tag && b[3] == tag) break;
                if (++spins > SPIN_LIMIT) { *err = 1; stop_s = 1; break; }
            }
            acc = __builtin_bit_cast(float, a[0]) + __builtin_bit_cast(float, a[2]) + __builtin_bit_cast(float, b[0]) + __builtin_bit_cast(float, b[2]);
        }
        red[tid] = acc;
        __syncthreads();
        if (tid < 64) {
            float v = red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192];
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (tid == 0) red[0] = v;
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
    char* buf; float* sink; int* err;
    const size_t cap = 2 * (512 * 8192 + 4096);
    CHECK(hipMalloc(&buf, cap));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&err, 4));
    for (int mode = 0; mode < 2; ++mode)
        for (unsigned stride : {16u, 64u, 128u, 256u, 1024u, 4096u, 4096u + 256u, 8192u}) {
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipMemset(buf, 0, cap));
                CHECK(hipMemset(err, 0, 4));
                int steps = 4000;
                unsigned half = 512 * stride + 4096;
                void* args[] = {&buf, &sink, &steps, &stride, &half, &mode, &err};
                auto t0 = std::chrono::steady_clock::now();
                CHECK(hipLaunchCooperativeKernel((const void*)k, dim3(256), dim3(256), args, 0, 0));
                CHECK(hipDeviceSynchronize());
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                int herr = 0;
                CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                if (rep == 1) printf("%s all->all 1024 values, unit stride %5u B: %.2f us per hop%s\n", mode ? "sentinel" : "direct  ", stride, us / steps,
                                     herr ? "  (SPIN LIMIT HIT)" : "");
            }
        }
    return 0;
}
