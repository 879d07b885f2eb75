// Microbenchmark: LDS-DMA (buffer_load_dwordx4 ... lds) streaming of K-slices of a row-major matrix, the access pattern of
// the fp16 GEMM's operand tiles.  A piece (one wave instruction, 1 KiB) covers either 16 rows x 64 B (the kernel's K = 32
// halfs per step: half of every 128-byte line per piece, the other half one step later) or 8 rows x 128 B (whole lines).
// 256 blocks x 256 threads (4 loader waves), every block streams 256 rows x `kbytes` bytes per row, tile by tile, into a
// ring of three LDS buffers with a counted vmcnt wait per tile; nothing is computed.  Prints useful TB/s per variant, for rows
// shared by all blocks (L2-resident, like the weight tiles) and rows private to each block (256 MB in all).
// build: hipcc --offload-arch=gfx950 -O3 -o bin/dma_rowbytes dma_rowbytes.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int ROWS = 256;

template <int RB>
__device__ __forceinline__ void stream_body(const char* base, long long row_stride, long long block_stride, int ktiles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int TILE = ROWS * RB;                      // bytes per tile
    constexpr int PPW = TILE / 1024 / 4;                 // wave instructions per tile and wave
    constexpr int LPR = RB / 16;                         // lanes per row
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long long)blockIdx.x * block_stride), 0,
                                                                        0x80000000u, 0x00020000);
    unsigned off[PPW];
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
        const int row = (p * 4 + wave) * (64 / LPR) + lane / LPR;
        off[p] = (unsigned)(row * row_stride + (lane % LPR) * 16);
    }
    auto issue = [&](int t, int buf) {
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
            char* dst = lds + buf * TILE + (p * 4 + wave) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, off[p] + (unsigned)(t * RB), 0, 0, 0);
        }
    };
    issue(0, 0);
    issue(1, 1);
    int buf = 2;
    for (int t = 0; t < ktiles; ++t) {
        if (t + 2 < ktiles) {
            issue(t + 2, buf);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        buf = buf == 2 ? 0 : buf + 1;
    }
    if (sink && tid == 0) sink[blockIdx.x] = ((float*)lds)[5];
}
__global__ __launch_bounds__(256) void stream64(const char* b, long long rs, long long bs, int kt, float* s) { stream_body<64>(b, rs, bs, kt, s); }
__global__ __launch_bounds__(256) void stream128(const char* b, long long rs, long long bs, int kt, float* s) { stream_body<128>(b, rs, bs, kt, s); }

void run(int RB, const char* d, long long row_stride, long long block_stride, int kbytes, const char* what) {
    const int ktiles = kbytes / RB;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    void (*k)(const char*, long long, long long, int, float*) = RB == 64 ? stream64 : stream128;
    const size_t lds = (size_t)3 * ROWS * RB;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), lds, 0, d, row_stride, block_stride, ktiles, (float*)nullptr);
    CHECK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), lds, 0, d, row_stride, block_stride, ktiles, (float*)nullptr);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double bytes = 256.0 * ROWS * kbytes;
    printf("%-28s row piece %3d B: %7.1f us per launch, %6.2f TB/s useful, %5.1f B/clk/CU at 2.1 GHz\n", what, RB, 1e3 * ms / reps,
           bytes / (ms / reps * 1e-3) / 1e12, bytes / 256 / (ms / reps * 1e-3) / 2.1e9);
}

int main() {
    const long long row_stride = 4096;                               // bytes: 2048 halfs per row (K = 1856 rounded up)
    const int kbytes = 3712;                                         // K = 1856 halfs
    char* d;
    CHECK(hipMalloc(&d, ROWS * row_stride + (1 << 20)));
    CHECK(hipMemset(d, 1, ROWS * row_stride + (1 << 20)));
    run(64, d, row_stride, 0, kbytes, "shared rows (L2-resident)");
    run(128, d, row_stride, 0, kbytes, "shared rows (L2-resident)");
    CHECK(hipFree(d));
    const long long bs = (long long)ROWS * row_stride;
    CHECK(hipMalloc(&d, 256 * bs + (1 << 20)));
    CHECK(hipMemset(d, 1, 256 * bs + (1 << 20)));
    run(64, d, row_stride, bs, kbytes, "private rows (HBM)");
    run(128, d, row_stride, bs, kbytes, "private rows (HBM)");
    CHECK(hipFree(d));
    return 0;
}
