// Microbenchmark: is a tagged exchange between CUs of the SAME XCD (through that XCD's L2: sc0 loads, plain stores) faster
// than the agent-scope one (sc1, through the memory side)?  Also prints the block -> XCC_ID mapping of a 256-block grid.
//   mode 0: 32 blocks with blockIdx % 8 == 0 exchange 128 values, agent scope (sc1)
//   mode 1: same blocks, L2 scope (sc0 loads / sc0 stores)
//   mode 2: 32 blocks blockIdx < 32 (spread over all XCDs), agent scope
//   mode 3: same as 1 but blocks < 32 (NOT on one XCD): must FAIL to converge quickly or be slow -- shows sc0 is XCD local
// Every spin loop is bounded.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr long long SPIN_LIMIT = 1 << 18;

template <int AUX>
__device__ __forceinline__ u32x4 ld16(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX);
}

template <int SCOPE>   // 0: agent (sc1), 1: L2 / workgroup (sc0)
__global__ __launch_bounds__(256) void k(u64* buf, int* xcc, float* sink, int steps, int sel, int* err) {
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blk] = (int)(id & 0xf);
    }
    const bool member = sel == 0 ? (blk % 8 == 0) : (blk < 32);
    const int rank = sel == 0 ? blk / 8 : blk;
    if (!member) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 0x80000000u, 0x00020000);
    __shared__ float red[4];
    __shared__ int stop_s;
    if (tid == 0) stop_s = 0;
    __syncthreads();
    float carry = 0.f;
    for (int s = 0; s < steps; ++s) {
        const unsigned tag = s + 1;
        u64* v = buf + (size_t)(s & 1) * 1024;
        if (tid < 4) {
            const u64 w = ((u64)tag << 32) | __builtin_bit_cast(unsigned, carry + rank + tid);
            if (SCOPE == 0) __hip_atomic_store(v + rank * 4 + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(v + rank * 4 + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        float acc = 0.f;
        if (tid < 64) {                      // one wave polls the 128 values (64 pairs)
            const unsigned off = ((s & 1) * 1024 + tid * 2) * 8;
            long long spins = 0;
            u32x4 w;
            while (true) {
                asm volatile("" ::: "memory");
                w = ld16<SCOPE == 0 ? 16 : 1>(rs, off);
                if (__all(w[1] == tag && w[3] == tag)) break;
                if (++spins > SPIN_LIMIT) { *err = 1; stop_s = 1; break; }
            }
            acc = __builtin_bit_cast(float, w[0]) + __builtin_bit_cast(float, w[2]);
            for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
            if (tid == 0) red[0] = acc;
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
    u64* buf; int* xcc; float* sink; int* err;
    CHECK(hipMalloc(&buf, 2 * 1024 * 8));
    CHECK(hipMalloc(&xcc, 256 * 4));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&err, 4));
    for (int mode = 0; mode < 4; ++mode) {
        const int scope = (mode == 1 || mode == 3) ? 1 : 0, sel = mode < 2 ? 0 : 1;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemset(buf, 0, 2 * 1024 * 8));
            CHECK(hipMemset(err, 0, 4));
            int steps = 4000;
            void* args[] = {&buf, &xcc, &sink, &steps, (void*)&sel, &err};
            auto t0 = std::chrono::steady_clock::now();
            CHECK(hipLaunchCooperativeKernel(scope ? (const void*)k<1> : (const void*)k<0>, dim3(256), dim3(256), args, 0, 0));
            CHECK(hipDeviceSynchronize());
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            int herr = 0;
            CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            if (rep == 1) printf("mode %d (%s, %s): %.2f us per hop%s\n", mode, scope ? "sc0 / L2 scope" : "sc1 / agent scope",
                                 sel == 0 ? "blocks 0,8,16,..." : "blocks 0..31", us / steps, herr ? "  (SPIN LIMIT HIT)" : "");
        }
    }
    int h[256];
    CHECK(hipMemcpy(h, xcc, sizeof h, hipMemcpyDeviceToHost));
    printf("XCC_ID of blocks 0..31:");
    for (int i = 0; i < 32; ++i) printf(" %d", h[i]);
    int ok = 1;
    for (int i = 0; i < 256; ++i) ok &= h[i] == h[i % 8];
    printf("\nblock b runs on the XCC of block b %% 8 for all 256 blocks: %s\n", ok ? "yes" : "NO");
    return 0;
}
