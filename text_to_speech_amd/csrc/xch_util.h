// xch_util.h -- device helpers shared by the decoder kernels that exchange small vectors between CUs inside one launch
// (taco_persist.hip, taco_fused.hip): tagged 8-byte publishes, DPP cross-lane moves and wave reductions, fast gate functions.
#pragma once
#include <hip/hip_runtime.h>

#include "dev_util.h"

namespace ttsxch {
using namespace ttsgemm;

typedef unsigned long long u64;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float bitsf(unsigned v) { return __builtin_bit_cast(float, v); }

__device__ __forceinline__ void publish(u64* p, unsigned tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | fbits(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Cross-lane moves on the DPP path (a few cycles) instead of ds_bpermute (an LDS round trip, ~100+ cycles each): the
// step's critical path holds ~50 dependent reductions steps, which cost more than the arithmetic.
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;          // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int DPP_REV4 = 0x1B;                           // quad_perm [3,2,1,0]: lane ^ 3
constexpr int DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR4 = 0x124, DPP_ROR8 = 0x128;   // i -> 7 - i, i -> 15 - i, rotations
__device__ __forceinline__ float lane_bcast(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// sum / max over the 64 lanes, result in every lane: four DPP steps inside each row of 16, then the four row results
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp<DPP_XOR1>(v);
    v += dpp<DPP_XOR2>(v);
    v += dpp<DPP_HALF_MIRROR>(v);
    v += dpp<DPP_MIRROR>(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp<DPP_XOR1>(v));
    v = fmaxf(v, dpp<DPP_XOR2>(v));
    v = fmaxf(v, dpp<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp<DPP_MIRROR>(v));
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}
// partner exchange lane ^ (1 << S) inside a row of 16 lanes
template <int S>
__device__ __forceinline__ float row_xor(float v, int lane) {
    if constexpr (S == 0) return dpp<DPP_XOR1>(v);
    else if constexpr (S == 1) return dpp<DPP_XOR2>(v);
    else if constexpr (S == 2) return dpp<DPP_HALF_MIRROR>(dpp<DPP_REV4>(v));    // (i ^ 7) ^ 3 = i ^ 4: two symmetric moves
    else return dpp<DPP_ROR8>(v);
}

// LSTM gate non-linearities on the transcendental unit: sigmoid(x) = rcp(1 + 2^(-x log2 e)), tanh(x) = 2 sigmoid(2 x) - 1
// (v_exp_f32 / v_rcp_f32, 1 ulp each; absolute error < 2e-7) instead of libm expf / tanhf and IEEE divisions, which were a
// third of the compute on the step's critical path.
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float tanh_fast(float x) {
    const float xc = fminf(fmaxf(x, -15.f), 15.f);
    return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(xc * -2.885390081777927f)) - 1.f;
}

}  // namespace ttsxch
