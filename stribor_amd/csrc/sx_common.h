// Shared host/device helpers for libstribor_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/stribor_hip.h"

#define SX_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));

void sx_set_error(const char *fmt, ...);

#define SX_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) {                                          \
            sx_set_error(__VA_ARGS__);                          \
            return SX_E_BADARG;                                 \
        }                                                       \
    } while (0)

#define SX_LAUNCH_CHECK()                                                        \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            sx_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return (int)e_;                                                      \
        }                                                                        \
    } while (0)

static inline hipStream_t sx_stream(void *s) { return (hipStream_t)s; }

// ---- bf16 <-> f32 (storage only; all arithmetic is fp32) --------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;   // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(uint16_t, h);
}

// ---- wave-level sums ----------------------------------------------------------------------------
// sum over the `width` consecutive lanes a lane belongs to (width = power of two <= 64)
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float group_sum_rt(float v, int width) {
    for (int o = width / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- fast fp32 transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32 are 1 ulp) --------------------
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// tanh(x) = 1 - 2/(exp(2x)+1): saturates correctly at +-inf, |abs err| <~ 2e-7
__device__ __forceinline__ float fast_tanh(float x) {
    float e = __builtin_amdgcn_exp2f(x * 2.88539008177792682f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
