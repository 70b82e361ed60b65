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
// experiment knobs (environment, by declared name only): csrc/sx_build_id.cpp
extern "C" int sx_debug_knob(const char *name, int dflt);

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
// sum over a segment of `len` (any value <= 64) consecutive lanes starting at a multiple of `len`: after the loop the
// segment's FIRST lane (pos == 0) holds the total; fixed order, no atomics (pos = lane's index inside its segment)
__device__ __forceinline__ float segment_sum_rt(float v, int pos, int len) {
    for (int o = 1; o < len; o <<= 1) {
        const float t = __shfl_down(v, o, 64);
        if (pos + o < len) v += t;
    }
    return v;
}
// Work units of the one-lane-per-(row, live column) kernels.  Dense: 64 consecutive elements per unit (rows may
// straddle units: fine when no per-row sum is wanted, or when n_live is a power of two <= 64).  Row-aligned (per-row
// log-det sums for any other n_live, deterministic -- no float atomics): n_live <= 64: a unit holds 64 / n_live WHOLE
// rows (the remaining lanes idle); n_live > 64: a unit is one row, walked in ceil(n_live / 64) chunks by one wave.
struct sx_units {
    int rows_per_unit;      // 0 = dense
    int chunks;             // chunks per unit (1 unless a row is wider than a wave)
    int64_t n_units;
};
__host__ __device__ static inline sx_units sx_make_units(int64_t n_rows, int n_live, bool row_aligned) {
    sx_units u;
    if (!row_aligned) { u.rows_per_unit = 0; u.chunks = 1; u.n_units = (n_rows * n_live + 63) >> 6; }
    else if (n_live <= 64) { u.rows_per_unit = 64 / n_live; u.chunks = 1; u.n_units = (n_rows + u.rows_per_unit - 1) / u.rows_per_unit; }
    else { u.rows_per_unit = 1; u.chunks = (n_live + 63) >> 6; u.n_units = n_rows; }
    return u;
}
// first element and element count of chunk `c` of unit `g`
__device__ __forceinline__ void sx_unit_span(const sx_units &u, int64_t g, int c, int64_t n_rows, int n_live, int64_t *e0, int *n_here) {
    if (u.rows_per_unit == 0) {
        const int64_t n_elem = n_rows * n_live;
        *e0 = g << 6;
        *n_here = (int)((n_elem - *e0) < 64 ? (n_elem - *e0) : 64);
    } else if (u.chunks == 1) {
        const int64_t row0 = g * u.rows_per_unit;
        const int64_t rows = (n_rows - row0) < u.rows_per_unit ? (n_rows - row0) : u.rows_per_unit;
        *e0 = row0 * n_live;
        *n_here = (int)rows * n_live;
    } else {
        *e0 = g * n_live + 64 * c;
        *n_here = (n_live - 64 * c) < 64 ? (n_live - 64 * c) : 64;
    }
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
