// dev_util.h -- small helpers shared by the kernel files (no kernel templates in here: cheap to include).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace ttsgemm {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only, and one process may own engines on
// several GPUs: one flag per (kernel instantiation, device).  Two host threads may race here (stream(overlap=True) drives
// two handles): the flags are atomic and setting the attribute twice is harmless.
struct PerDeviceOnce {
    static constexpr int kMaxDevices = 64;
    std::atomic<bool> set[kMaxDevices];
};
inline hipError_t set_max_dyn_lds_once(const void* kern, size_t lds, PerDeviceOnce& once) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < PerDeviceOnce::kMaxDevices;
    if (tracked && once.set[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && tracked) once.set[dev].store(true, std::memory_order_release);
    return e;
}

__device__ __forceinline__ float sigmoid_exact(float v) { return 1.0f / (1.0f + expf(-v)); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

}  // namespace ttsgemm
