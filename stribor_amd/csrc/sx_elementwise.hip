// HBM-bound element-wise kernels of the coupling-flow path (parameters already in HBM).
// One pass over [n_rows, dim]: coalesced 8/16-byte accesses, per-row log-det sums by wave shuffles.
#include "sx_common.h"
#include <stdarg.h>
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------
// error string
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void sx_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *sx_last_error(void) { return g_err; }
extern "C" int sx_abi_version(void) { return SX_ABI_VERSION; }

static int grid_for(int64_t work_items, int block, int max_blocks = 256 * 8) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// ------------------------------------------------------------------------------------------------
// vector row access: 4 consecutive columns per thread
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__device__ __forceinline__ f32x4 load4(const void *base, int64_t elem_off) {
    if constexpr (BF16) {
        u16x4 v = *reinterpret_cast<const u16x4 *>(reinterpret_cast<const uint16_t *>(base) + elem_off);
        return f32x4{bf16_to_f32(v.x), bf16_to_f32(v.y), bf16_to_f32(v.z), bf16_to_f32(v.w)};
    } else {
        return *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(base) + elem_off);
    }
}
template <bool BF16>
__device__ __forceinline__ void store4(void *base, int64_t elem_off, f32x4 v) {
    if constexpr (BF16) {
        u16x4 o{f32_to_bf16(v.x), f32_to_bf16(v.y), f32_to_bf16(v.z), f32_to_bf16(v.w)};
        *reinterpret_cast<u16x4 *>(reinterpret_cast<uint16_t *>(base) + elem_off) = o;
    } else {
        *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(base) + elem_off) = v;
    }
}
template <bool BF16>
__device__ __forceinline__ float load1(const void *base, int64_t elem_off) {
    if constexpr (BF16) return bf16_to_f32(reinterpret_cast<const uint16_t *>(base)[elem_off]);
    else return reinterpret_cast<const float *>(base)[elem_off];
}
template <bool BF16>
__device__ __forceinline__ void store1(void *base, int64_t elem_off, float v) {
    if constexpr (BF16) reinterpret_cast<uint16_t *>(base)[elem_off] = f32_to_bf16(v);
    else reinterpret_cast<float *>(base)[elem_off] = v;
}

// ------------------------------------------------------------------------------------------------
// K3: affine coupling, element-wise part            (stribor/flows/affine.py:104-109, coupling.py:78,95)
// Fast path: every access is 16 B per lane (8-B accesses run at 0.54-0.70x the 16-B rate on this chip): a thread owns
// CPT = 8 (bf16 storage) or 4 (fp32) consecutive columns of one row; TPR = dim / CPT threads per row is a power of two
// <= 64, live columns are the contiguous CPT-aligned range [l0, l0 + n_live).
// ------------------------------------------------------------------------------------------------
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
// NT: streaming (non-temporal) loads of x / params and stores of y -- every byte is touched once; UNR: rows in flight
// per thread (independent iterations issued together: more loads outstanding per lane).
template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, typename T>
__device__ __forceinline__ void st_stream(T *p, T v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool BF16, bool REVERSE, int UNR, bool NT>
__global__ __launch_bounds__(256) void affine_coupling_vec_kernel(
    const void *__restrict__ x, void *__restrict__ y, float *__restrict__ ldj,
    const float *__restrict__ params, int64_t pstride, int l0, int n_live, int64_t n_rows, int dim,
    int tpr_log2, int ldj_acc, float ldj_scale) {
    constexpr int CPT = BF16 ? 8 : 4, NQ = CPT / 4;
    const int tpr = 1 << tpr_log2;
    const int64_t n_vec = n_rows << tpr_log2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v0 < n_vec; v0 += stride * UNR) {
        f32x4 xv[UNR][NQ], ls[UNR][NQ], sh[UNR][NQ];
        bool live[UNR], ok[UNR];
        int64_t row[UNR];
        int c[UNR];
        // all loads of the UNR independent rows first
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t v = v0 + u * stride;
            ok[u] = v < n_vec;
            row[u] = ok[u] ? v >> tpr_log2 : 0;
            c[u] = ((int)(v & (tpr - 1))) * CPT;
            live[u] = ok[u] && c[u] >= l0 && c[u] < l0 + n_live;
            if (ok[u]) {
                if constexpr (BF16) {
                    const u16x8 q = ld_stream<NT>(reinterpret_cast<const u16x8 *>(reinterpret_cast<const uint16_t *>(x) + row[u] * dim + c[u]));
                    xv[u][0] = f32x4{bf16_to_f32(q[0]), bf16_to_f32(q[1]), bf16_to_f32(q[2]), bf16_to_f32(q[3])};
                    xv[u][1] = f32x4{bf16_to_f32(q[4]), bf16_to_f32(q[5]), bf16_to_f32(q[6]), bf16_to_f32(q[7])};
                } else {
                    xv[u][0] = ld_stream<NT>(reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(x) + row[u] * dim + c[u]));
                }
            }
            if (live[u]) {
                const float *p = params + row[u] * pstride + (c[u] - l0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    ls[u][q] = ld_stream<NT>(reinterpret_cast<const f32x4 *>(p + 4 * q));
                    sh[u][q] = ld_stream<NT>(reinterpret_cast<const f32x4 *>(p + n_live + 4 * q));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float s = 0.f;
            if (live[u]) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    f32x4 &X = xv[u][q];
                    const f32x4 L = ls[u][q], S = sh[u][q];
                    if constexpr (REVERSE) {
                        X.x = (X.x - S.x) * fast_exp(-L.x);
                        X.y = (X.y - S.y) * fast_exp(-L.y);
                        X.z = (X.z - S.z) * fast_exp(-L.z);
                        X.w = (X.w - S.w) * fast_exp(-L.w);
                    } else {
                        X.x = X.x * fast_exp(L.x) + S.x;
                        X.y = X.y * fast_exp(L.y) + S.y;
                        X.z = X.z * fast_exp(L.z) + S.z;
                        X.w = X.w * fast_exp(L.w) + S.w;
                    }
                    s += (L.x + L.y) + (L.z + L.w);
                }
            }
            if (ok[u]) {
                if constexpr (BF16) {
                    u16x8 o;
                    o[0] = f32_to_bf16(xv[u][0].x); o[1] = f32_to_bf16(xv[u][0].y); o[2] = f32_to_bf16(xv[u][0].z); o[3] = f32_to_bf16(xv[u][0].w);
                    o[4] = f32_to_bf16(xv[u][1].x); o[5] = f32_to_bf16(xv[u][1].y); o[6] = f32_to_bf16(xv[u][1].z); o[7] = f32_to_bf16(xv[u][1].w);
                    st_stream<NT>(reinterpret_cast<u16x8 *>(reinterpret_cast<uint16_t *>(y) + row[u] * dim + c[u]), o);
                } else {
                    st_stream<NT>(reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(y) + row[u] * dim + c[u]), xv[u][0]);
                }
            }
            if (ldj != nullptr) {   // wave-uniform (the whole wave shares ok[u] except in the last, partial wave: tpr divides 64)
                s = group_sum_rt(s, tpr);
                if (ok[u] && ((v0 + u * stride) & (tpr - 1)) == 0) ldj[row[u]] = (ldj_acc ? ldj[row[u]] : 0.f) + ldj_scale * s;
            }
        }
    }
}

// Generic path: one wave per row, lanes stride over columns; any dim / live set.
template <bool BF16, bool REVERSE>
__global__ __launch_bounds__(256) void affine_coupling_generic_kernel(
    const void *__restrict__ x, void *__restrict__ y, float *__restrict__ ldj,
    const float *__restrict__ params, int64_t pstride, const int32_t *__restrict__ live_idx, int l0,
    int n_live, int64_t n_rows, int dim, int ldj_acc, float ldj_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *pos = reinterpret_cast<int *>(smem);   // column -> live slot or -1
    for (int c = threadIdx.x; c < dim; c += blockDim.x) pos[c] = -1;
    __syncthreads();
    for (int i = threadIdx.x; i < n_live; i += blockDim.x) pos[live_idx ? live_idx[i] : l0 + i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t row = wave; row < n_rows; row += n_waves) {
        float s = 0.f;
        for (int c = lane; c < dim; c += 64) {
            float xv = load1<BF16>(x, row * dim + c);
            int i = pos[c];
            if (i >= 0) {
                float ls = params[row * pstride + i];
                float sh = params[row * pstride + n_live + i];
                xv = REVERSE ? (xv - sh) * fast_exp(-ls) : xv * fast_exp(ls) + sh;
                s += ls;
            }
            store1<BF16>(y, row * dim + c, xv);
        }
        if (ldj != nullptr) {
            s = group_sum<64>(s);
            if (lane == 0) ldj[row] = (ldj_acc ? ldj[row] : 0.f) + ldj_scale * s;
        }
    }
}

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

extern "C" int sx_affine_coupling(const void *x, void *y, float *ldj, const float *params,
                                  int64_t params_stride, const int32_t *live_idx, int32_t live_start,
                                  int32_t n_live, int64_t n_rows, int32_t dim, int32_t dtype,
                                  int32_t reverse, int32_t ldj_accumulate, float ldj_scale, void *stream) {
    SX_REQUIRE(x && y && params, "sx_affine_coupling: null pointer");
    SX_REQUIRE(dim > 0 && n_live >= 0 && n_live <= dim && n_rows >= 0, "sx_affine_coupling: bad sizes");
    SX_REQUIRE(dtype == SX_F32 || dtype == SX_BF16, "sx_affine_coupling: bad dtype");
    if (n_rows == 0) return SX_OK;
    hipStream_t st = sx_stream(stream);
    const int cpt = dtype == SX_BF16 ? 8 : 4;
    const bool fast = live_idx == nullptr && dim % cpt == 0 && pow2(dim / cpt) && dim / cpt <= 64 &&
                      live_start % cpt == 0 && n_live % cpt == 0 && params_stride % 4 == 0 &&
                      live_start + n_live <= dim && (((uintptr_t)params) & 15) == 0 &&
                      (((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 15) == 0;
    if (fast) {
        const int tl = ilog2(dim / cpt);
        // Streaming (non-temporal) accesses and, for bf16 storage, two rows in flight per thread: measured on 2^20 x 64
        // (tools/sweep_affine.py; fraction of 8 TB/s): plain loads 0.64-0.69, nt 0.72-0.73, nt + 2 rows + 32 workgroups
        // per CU 0.75 (bf16) / 0.70 (fp32).  SX_AFFINE_VARIANT = 10 * rows-in-flight + nt, SX_AFFINE_GRID = workgroups per
        // CU override the choice (experiments; read once).
        static const int variant_env = sx_debug_knob("SX_AFFINE_VARIANT", 0);
        static const int grid_env = sx_debug_knob("SX_AFFINE_GRID", 0);
        const int variant = variant_env ? variant_env : (dtype == SX_BF16 ? 21 : 11);
        const int grid_mul = grid_env ? grid_env : (dtype == SX_BF16 ? 32 : 16);
        const int unr = variant / 10 < 1 ? 1 : variant / 10;
        const int grid = grid_for((n_rows << tl) / unr + 1, 256, 256 * grid_mul);
#define SX_AC3(BF, RV, U, N)                                                                  \
    hipLaunchKernelGGL((affine_coupling_vec_kernel<BF, RV, U, N>), dim3(grid), dim3(256), 0, st, x, y, ldj, \
                       params, params_stride, live_start, n_live, n_rows, dim, tl, ldj_accumulate, ldj_scale)
#define SX_AC(BF, RV)                                                                         \
    do {                                                                                      \
        switch (variant) {                                                                    \
            case 11: SX_AC3(BF, RV, 1, true); break;                                          \
            case 20: SX_AC3(BF, RV, 2, false); break;                                         \
            case 21: SX_AC3(BF, RV, 2, true); break;                                          \
            case 40: SX_AC3(BF, RV, 4, false); break;                                         \
            case 41: SX_AC3(BF, RV, 4, true); break;                                          \
            default: SX_AC3(BF, RV, 1, false); break;                                         \
        }                                                                                     \
    } while (0)
        if (dtype == SX_BF16) { if (reverse) SX_AC(true, true); else SX_AC(true, false); }
        else { if (reverse) SX_AC(false, true); else SX_AC(false, false); }
#undef SX_AC
#undef SX_AC3
    } else {
        const int grid = grid_for(n_rows * 64, 256);
        const size_t lds = (size_t)dim * sizeof(int);
#define SX_AG(BF, RV)                                                                               \
    hipLaunchKernelGGL((affine_coupling_generic_kernel<BF, RV>), dim3(grid), dim3(256), lds, st, x, y, ldj, \
                       params, params_stride, live_idx, live_start, n_live, n_rows, dim, ldj_accumulate, ldj_scale)
        if (dtype == SX_BF16) { if (reverse) SX_AG(true, true); else SX_AG(true, false); }
        else { if (reverse) SX_AG(false, true); else SX_AG(false, false); }
#undef SX_AG
    }
    SX_LAUNCH_CHECK();
    return SX_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward of sx_affine_coupling (training, layer-wise path): one lane per (row, live column) element.
//   reverse:  y = (x - sh) e^{-ls}:  dx = gy e^{-ls},  dsh = -dx,  dls = -gy y + gldj * ldj_scale
//   forward:  y = x e^{ls} + sh:     dx = gy e^{ls},   dsh = gy,   dls = gy x e^{ls} + gldj * ldj_scale
// gx's pass-through columns are the caller's (a copy of gy); gparams has the layout of params (packed rows).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_coupling_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                  const float *__restrict__ gldj,
                                                                  const float *__restrict__ params, int64_t pstride,
                                                                  float *__restrict__ gx, float *__restrict__ gparams,
                                                                  const int32_t *__restrict__ live_idx, int l0, int n_live,
                                                                  int64_t n_rows, int dim, int reverse, float ldj_scale) {
    const int64_t total = n_rows * n_live;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t row = e / n_live;
        const int i = (int)(e - row * n_live);
        const int col = live_idx ? live_idx[i] : l0 + i;
        const float ls = params[row * pstride + i], sh = params[row * pstride + n_live + i];
        const float xv = x[row * dim + col], g = gy[row * dim + col], gl = gldj[row] * ldj_scale;
        float dx, dsh, dls;
        if (reverse) {
            const float e_ = expf(-ls);
            dx = g * e_;
            dsh = -dx;
            dls = -g * ((xv - sh) * e_) + gl;
        } else {
            const float e_ = expf(ls);
            dx = g * e_;
            dsh = g;
            dls = g * (xv * e_) + gl;
        }
        gx[row * dim + col] = dx;
        gparams[row * 2 * n_live + i] = dls;
        gparams[row * 2 * n_live + n_live + i] = dsh;
    }
}

extern "C" int sx_affine_coupling_bwd(const float *x, const float *gy, const float *gldj, const float *params,
                                      int64_t params_stride, float *gx, float *gparams, const int32_t *live_idx,
                                      int32_t live_start, int32_t n_live, int64_t n_rows, int32_t dim, int32_t reverse,
                                      float ldj_scale, void *stream) {
    SX_REQUIRE(x && gy && gldj && params && gx && gparams, "sx_affine_coupling_bwd: null pointer");
    SX_REQUIRE(dim > 0 && n_live > 0 && n_live <= dim && n_rows >= 0, "sx_affine_coupling_bwd: bad sizes");
    if (n_rows == 0) return SX_OK;
    const int grid = grid_for(n_rows * n_live, 256);
    hipLaunchKernelGGL(affine_coupling_bwd_kernel, dim3(grid), dim3(256), 0, sx_stream(stream), x, gy, gldj, params,
                       params_stride, gx, gparams, live_idx, live_start, n_live, n_rows, dim, reverse, ldj_scale);
    SX_LAUNCH_CHECK();
    return SX_OK;
}

// ------------------------------------------------------------------------------------------------
// Time-conditioned affine coupling (ContinuousAffineCoupling, stribor/flows/coupling.py:98-213): the conditioner's
// (log_scale, shift) are multiplied by a time embedding of the row's t before the affine map, so that the layer is
// the identity at t = 0:   ls' = ls * phi(a_i t),  sh' = sh * phi(b_i t)   with phi = the time net
// (net/time_net.py: TimeIdentity t, TimeLinear s t, TimeTanh tanh(s t), TimeLog log(e^s t + 1)).
// One lane per (row, column); pass-through columns are copied; the row log-det is a shuffle sum for power-of-two
// widths <= 64 and float atomics otherwise.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float time_embed(int kind, float s, float t) {
    switch (kind) {
        case SX_TIME_IDENTITY: return t;
        case SX_TIME_LINEAR: return s * t;
        case SX_TIME_TANH: return tanhf(s * t);
        default: return logf(expf(s) * t + 1.f);                       // SX_TIME_LOG
    }
}

template <bool BF16, bool ALIGNED>
__global__ __launch_bounds__(256) void time_affine_coupling_kernel(const void *__restrict__ x, void *__restrict__ y,
                                                                   float *__restrict__ ldj, const float *__restrict__ params,
                                                                   int64_t pstride, const float *__restrict__ t,
                                                                   const float *__restrict__ tscale, int time_kind,
                                                                   const int32_t *__restrict__ live_idx, int l0, int n_live,
                                                                   int64_t n_rows, int dim, int reverse, int ldj_mode,
                                                                   int ldj_acc, float ldj_scale) {
    extern __shared__ __attribute__((aligned(16))) char tc_smem[];
    int *slot = reinterpret_cast<int *>(tc_smem);                      // column -> index among the live columns, or -1
    for (int c = threadIdx.x; c < dim; c += blockDim.x) slot[c] = -1;
    __syncthreads();
    for (int i = threadIdx.x; i < n_live; i += blockDim.x) slot[live_idx ? live_idx[i] : l0 + i] = i;
    __syncthreads();
    const int64_t total = n_rows * dim;
    const int64_t total_up = (total + 63) & ~(int64_t)63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // one lane = one element e of [n_rows, dim]: returns its log-det part
    auto body = [&](int64_t e, int64_t row) -> float {
        float ld = 0.f;
        const int col = (int)(e - row * dim);
        const int i = slot[col];
        float xv = BF16 ? bf16_to_f32(reinterpret_cast<const uint16_t *>(x)[e]) : reinterpret_cast<const float *>(x)[e];
        float out = xv;
        if (i >= 0) {
            const float tv = t[row];
            const float ls = params[row * pstride + i] * time_embed(time_kind, tscale ? tscale[i] : 0.f, tv);
            const float sh = params[row * pstride + n_live + i] * time_embed(time_kind, tscale ? tscale[n_live + i] : 0.f, tv);
            out = reverse ? (xv - sh) * expf(-ls) : xv * expf(ls) + sh;   // coupling.py:196-199
            ld = ls;                                                       // :201
        }
        if (BF16) reinterpret_cast<uint16_t *>(y)[e] = f32_to_bf16(out);
        else reinterpret_cast<float *>(y)[e] = out;
        return ld;
    };
    if constexpr (ALIGNED) {    // any row length, no atomics: row-aligned units (sx_common.h), fixed-order sums
        const int lane = threadIdx.x & 63;
        const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = stride >> 6;
        const sx_units units = sx_make_units(n_rows, dim, true);
        for (int64_t u = wave; u < units.n_units; u += n_waves) {
            float row_acc = 0.f;
            for (int chunk = 0; chunk < units.chunks; ++chunk) {
                int64_t e0;
                int n_here;
                sx_unit_span(units, u, chunk, n_rows, dim, &e0, &n_here);
                const bool valid = lane < n_here;
                const int64_t e = e0 + lane, row = valid ? e / dim : 0;
                const float s0 = valid ? body(e, row) : 0.f;
                if (units.chunks == 1) {
                    const int pos = lane % dim;
                    const float s = segment_sum_rt(s0, pos, dim);
                    if (valid && pos == 0) ldj[row] = (ldj_acc ? ldj[row] : 0.f) + ldj_scale * s;
                } else {
                    row_acc += s0;
                }
            }
            if (units.chunks > 1) {
                const float s = group_sum<64>(row_acc);
                if (lane == 0) ldj[u] = (ldj_acc ? ldj[u] : 0.f) + ldj_scale * s;
            }
        }
        return;
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total_up; e += stride) {
        const bool valid = e < total;
        const int64_t row = valid ? e / dim : 0;
        const float ld = valid ? body(e, row) : 0.f;
        if (ldj_mode == 1) {
            const float s = group_sum_rt(ld, dim);
            if (valid && (e & (dim - 1)) == 0) ldj[row] = (ldj_acc ? ldj[row] : 0.f) + ldj_scale * s;
        }
    }
}

extern "C" int sx_time_affine_coupling(const void *x, void *y, float *ldj, const float *params, int64_t params_stride,
                                       const float *t, const float *tscale, int32_t time_kind, const int32_t *live_idx,
                                       int32_t live_start, int32_t n_live, int64_t n_rows, int32_t dim, int32_t dtype,
                                       int32_t reverse, int32_t ldj_accumulate, float ldj_scale, void *stream) {
    SX_REQUIRE(x && y && params && t, "sx_time_affine_coupling: null pointer");
    SX_REQUIRE(dim > 0 && n_live >= 0 && n_live <= dim && n_rows >= 0, "sx_time_affine_coupling: bad sizes");
    SX_REQUIRE(dtype == SX_F32 || dtype == SX_BF16, "sx_time_affine_coupling: bad dtype");
    SX_REQUIRE(time_kind >= SX_TIME_IDENTITY && time_kind <= SX_TIME_LOG, "sx_time_affine_coupling: unknown time net %d", time_kind);
    SX_REQUIRE(time_kind == SX_TIME_IDENTITY || tscale != nullptr, "sx_time_affine_coupling: this time net needs its scale vector");
    if (n_rows == 0) return SX_OK;
    hipStream_t st = sx_stream(stream);
    int ldj_mode = 0;
    if (ldj) {
        ldj_mode = (pow2(dim) && dim <= 64) ? 1 : 2;      // 2: row-aligned units, deterministic sums (no atomics)
    }
    const int grid = ldj_mode == 2 ? grid_for(sx_make_units(n_rows, dim, true).n_units * 64, 256) : grid_for(n_rows * dim, 256);
    const size_t lds = (size_t)dim * sizeof(int);
#define SX_TA(BF, AL)                                                                                             \
    hipLaunchKernelGGL((time_affine_coupling_kernel<BF, AL>), dim3(grid), dim3(256), lds, st, x, y, ldj, params, params_stride, t, \
                       tscale, time_kind, live_idx, live_start, n_live, n_rows, dim, reverse, ldj_mode, ldj_accumulate, ldj_scale)
    if (dtype == SX_BF16) { if (ldj_mode == 2) SX_TA(true, true); else SX_TA(true, false); }
    else { if (ldj_mode == 2) SX_TA(false, true); else SX_TA(false, false); }
#undef SX_TA
    SX_LAUNCH_CHECK();
    return SX_OK;
}

// ------------------------------------------------------------------------------------------------
// K7: Permute / Flip — bit-exact column gather                 (stribor/flows/permute.py:35,38,71,75)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void permute_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                      const int32_t *__restrict__ idx, int64_t n_rows, int dim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *sidx = reinterpret_cast<int *>(smem);
    for (int c = threadIdx.x; c < dim; c += blockDim.x) sidx[c] = idx[c];
    __syncthreads();
    const int64_t total = n_rows * dim;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t row = i / dim;
        const int j = (int)(i - row * dim);
        y[i] = x[row * dim + sidx[j]];
    }
}

// Rows staged through LDS: every global access is a coalesced 16-byte chunk; the gather happens in LDS.
// One 256-thread workgroup moves 4 KiB = 4096 / row_bytes whole rows per pass.
template <typename T>
__global__ __launch_bounds__(256) void permute_lds_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                          const int32_t *__restrict__ idx, int64_t n_rows, int dim) {
    __shared__ __attribute__((aligned(16))) char tile[4096];
    __shared__ int sidx[2048];
    constexpr int EPC = 16 / sizeof(T);                       // elements per 16-byte chunk
    const int row_bytes = dim * (int)sizeof(T);
    const int rows_per_pass = 4096 / row_bytes;
    const int chunks_per_row = row_bytes / 16;
    for (int c = threadIdx.x; c < dim; c += 256) sidx[c] = idx[c];
    const int64_t n_pass = (n_rows + rows_per_pass - 1) / rows_per_pass;
    for (int64_t pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
        const int64_t row0 = pass * rows_per_pass;
        const int r = threadIdx.x / chunks_per_row, ch = threadIdx.x - r * chunks_per_row;
        const bool ok = row0 + r < n_rows;
        __syncthreads();                                      // previous pass's reads done (and sidx ready)
        if (ok)
        {
            // every byte is touched once: non-temporal accesses for 4-byte elements (measured: fp32 0.66 -> 0.73 of the HBM peak;
            // 2-byte elements 0.72 -> 0.67, so they keep the plain form)
            const f32x4 *src = reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(x) + (row0 + r) * row_bytes + ch * 16);
            *reinterpret_cast<f32x4 *>(tile + threadIdx.x * 16) = sizeof(T) == 4 ? __builtin_nontemporal_load(src) : *src;
        }
        __syncthreads();
        if (ok) {
            const T *trow = reinterpret_cast<const T *>(tile + r * row_bytes);
            T o[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) o[e] = trow[sidx[ch * EPC + e]];
            f32x4 *dstp = reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(y) + (row0 + r) * row_bytes + ch * 16);
            if (sizeof(T) == 4) __builtin_nontemporal_store(*reinterpret_cast<const f32x4 *>(o), dstp);
            else *dstp = *reinterpret_cast<const f32x4 *>(o);
        }
    }
}

extern "C" int sx_permute(const void *x, void *y, const int32_t *idx, int64_t n_rows, int32_t dim,
                          int32_t elem_bytes, void *stream) {
    SX_REQUIRE(x && y && idx, "sx_permute: null pointer");
    SX_REQUIRE(x != y, "sx_permute: in-place permutation is not supported");
    SX_REQUIRE(dim > 0 && n_rows >= 0, "sx_permute: bad sizes");
    SX_REQUIRE(elem_bytes == 2 || elem_bytes == 4, "sx_permute: elem_bytes must be 2 or 4");
    if (n_rows == 0) return SX_OK;
    const int row_bytes = dim * elem_bytes;
    if (row_bytes % 16 == 0 && row_bytes <= 4096 && 4096 % row_bytes == 0 && dim <= 2048 &&
        (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        const int64_t n_pass = (n_rows + 4096 / row_bytes - 1) / (4096 / row_bytes);
        const int g = (int)(n_pass < 256 * 8 ? n_pass : 256 * 8);
        if (elem_bytes == 2)
            hipLaunchKernelGGL(permute_lds_kernel<uint16_t>, dim3(g), dim3(256), 0, sx_stream(stream),
                               (const uint16_t *)x, (uint16_t *)y, idx, n_rows, dim);
        else
            hipLaunchKernelGGL(permute_lds_kernel<uint32_t>, dim3(g), dim3(256), 0, sx_stream(stream),
                               (const uint32_t *)x, (uint32_t *)y, idx, n_rows, dim);
        SX_LAUNCH_CHECK();
        return SX_OK;
    }
    const int grid = grid_for(n_rows * dim, 256);
    const size_t lds = (size_t)dim * sizeof(int);
    if (elem_bytes == 2)
        hipLaunchKernelGGL(permute_kernel<uint16_t>, dim3(grid), dim3(256), lds, sx_stream(stream),
                           (const uint16_t *)x, (uint16_t *)y, idx, n_rows, dim);
    else
        hipLaunchKernelGGL(permute_kernel<uint32_t>, dim3(grid), dim3(256), lds, sx_stream(stream),
                           (const uint32_t *)x, (uint32_t *)y, idx, n_rows, dim);
    SX_LAUNCH_CHECK();
    return SX_OK;
}

// ------------------------------------------------------------------------------------------------
// K8: UnitNormal.log_prob + ldj accumulator              (stribor/dist/normal.py:37,52-54; flow.py:129)
// ------------------------------------------------------------------------------------------------
#define SX_HALF_LOG_2PI 0.91893853320467274178f

// bf16 rows with 16-byte accesses: 8 columns per thread
__global__ __launch_bounds__(256) void unit_normal_bf16x8_kernel(const void *__restrict__ x, const float *__restrict__ ldj,
                                                                 float *__restrict__ out, int64_t n_rows, int dim,
                                                                 int tpr_log2) {
    typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
    const int tpr = 1 << tpr_log2;
    const int64_t n_vec = n_rows << tpr_log2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vec; v += stride) {
        const int64_t row = v >> tpr_log2;
        const int c = ((int)(v & (tpr - 1))) << 3;
        const u16x8 u = *reinterpret_cast<const u16x8 *>(reinterpret_cast<const uint16_t *>(x) + row * dim + c);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = bf16_to_f32(u[e]);
            s += f * f;
        }
        s = group_sum_rt(s, tpr);
        if ((v & (tpr - 1)) == 0) out[row] = -0.5f * s - (float)dim * SX_HALF_LOG_2PI + (ldj ? ldj[row] : 0.f);
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void unit_normal_vec4_kernel(const void *__restrict__ x,
                                                               const float *__restrict__ ldj,
                                                               float *__restrict__ out, int64_t n_rows,
                                                               int dim, int tpr_log2) {
    const int tpr = 1 << tpr_log2;
    const int64_t n_vec = n_rows << tpr_log2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vec; v += stride) {
        const int64_t row = v >> tpr_log2;
        const int c = ((int)(v & (tpr - 1))) << 2;
        f32x4 xv = load4<BF16>(x, row * dim + c);
        float s = (xv.x * xv.x + xv.y * xv.y) + (xv.z * xv.z + xv.w * xv.w);
        s = group_sum_rt(s, tpr);
        if ((v & (tpr - 1)) == 0)
            out[row] = -0.5f * s - (float)dim * SX_HALF_LOG_2PI + (ldj ? ldj[row] : 0.f);
    }
}
template <bool BF16>
__global__ __launch_bounds__(256) void unit_normal_generic_kernel(const void *__restrict__ x,
                                                                  const float *__restrict__ ldj,
                                                                  float *__restrict__ out, int64_t n_rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t row = wave; row < n_rows; row += n_waves) {
        float s = 0.f;
        for (int c = lane; c < dim; c += 64) {
            float v = load1<BF16>(x, row * dim + c);
            s += v * v;
        }
        s = group_sum<64>(s);
        if (lane == 0) out[row] = -0.5f * s - (float)dim * SX_HALF_LOG_2PI + (ldj ? ldj[row] : 0.f);
    }
}

extern "C" int sx_unit_normal_logprob(const void *x, const float *ldj, float *out, int64_t n_rows,
                                      int32_t dim, int32_t dtype, void *stream) {
    SX_REQUIRE(x && out, "sx_unit_normal_logprob: null pointer");
    SX_REQUIRE(dim > 0 && n_rows >= 0, "sx_unit_normal_logprob: bad sizes");
    SX_REQUIRE(dtype == SX_F32 || dtype == SX_BF16, "sx_unit_normal_logprob: bad dtype");
    if (n_rows == 0) return SX_OK;
    hipStream_t st = sx_stream(stream);
    const bool fast = dim % 4 == 0 && pow2(dim / 4) && dim / 4 <= 64 && (((uintptr_t)x) & 15) == 0;
    if (dtype == SX_BF16 && dim % 8 == 0 && pow2(dim / 8) && dim / 8 <= 64 && (((uintptr_t)x) & 15) == 0) {
        const int tl = ilog2(dim / 8);
        hipLaunchKernelGGL(unit_normal_bf16x8_kernel, dim3(grid_for(n_rows << tl, 256)), dim3(256), 0, st, x, ldj, out,
                           n_rows, dim, tl);
    } else if (fast) {
        const int tl = ilog2(dim / 4);
        const int grid = grid_for(n_rows << tl, 256);
        if (dtype == SX_BF16)
            hipLaunchKernelGGL(unit_normal_vec4_kernel<true>, dim3(grid), dim3(256), 0, st, x, ldj, out, n_rows, dim, tl);
        else
            hipLaunchKernelGGL(unit_normal_vec4_kernel<false>, dim3(grid), dim3(256), 0, st, x, ldj, out, n_rows, dim, tl);
    } else {
        const int grid = grid_for(n_rows * 64, 256);
        if (dtype == SX_BF16)
            hipLaunchKernelGGL(unit_normal_generic_kernel<true>, dim3(grid), dim3(256), 0, st, x, ldj, out, n_rows, dim);
        else
            hipLaunchKernelGGL(unit_normal_generic_kernel<false>, dim3(grid), dim3(256), 0, st, x, ldj, out, n_rows, dim);
    }
    SX_LAUNCH_CHECK();
    return SX_OK;
}

// ------------------------------------------------------------------------------------------------
// K9: sum of fp32 values into one fp64 (per-thread fp64 partials, one atomic per block)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_f64_kernel(const float *__restrict__ v, int64_t n, double *out) {
    __shared__ double part[4];
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(v) & 15) == 0) {          // 16-byte loads over the aligned body, scalars for the tail
        const int64_t n4 = n >> 2;
        const f32x4 *v4 = reinterpret_cast<const f32x4 *>(v);
        // four independent 16-byte loads in flight per lane (one load per trip left the memory pipe idle for most of its
        // latency); the partial sums are added in a fixed order, so the result does not depend on timing
        int64_t i = tid;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const f32x4 t0 = __builtin_nontemporal_load(v4 + i), t1 = __builtin_nontemporal_load(v4 + i + stride);
            const f32x4 t2 = __builtin_nontemporal_load(v4 + i + 2 * stride), t3 = __builtin_nontemporal_load(v4 + i + 3 * stride);
            a0 += ((double)t0.x + (double)t0.y) + ((double)t0.z + (double)t0.w);
            a1 += ((double)t1.x + (double)t1.y) + ((double)t1.z + (double)t1.w);
            a2 += ((double)t2.x + (double)t2.y) + ((double)t2.z + (double)t2.w);
            a3 += ((double)t3.x + (double)t3.y) + ((double)t3.z + (double)t3.w);
        }
        for (; i < n4; i += stride) {
            const f32x4 t = __builtin_nontemporal_load(v4 + i);
            a0 += ((double)t.x + (double)t.y) + ((double)t.z + (double)t.w);
        }
        acc = (a0 + a1) + (a2 + a3);
        for (int64_t i = (n4 << 2) + tid; i < n; i += stride) acc += (double)v[i];
    } else {
        for (int64_t i = tid; i < n; i += stride) acc += (double)v[i];
    }
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

extern "C" int sx_sum_f64(const float *v, int64_t n, double *out, void *stream) {
    SX_REQUIRE(v && out, "sx_sum_f64: null pointer");
    SX_REQUIRE(n >= 0, "sx_sum_f64: bad size");
    if (n == 0) return SX_OK;
    const int grid = grid_for((n + 3) / 4, 256, 2048);
    hipLaunchKernelGGL(sum_f64_kernel, dim3(grid), dim3(256), 0, sx_stream(stream), v, n, out);
    SX_LAUNCH_CHECK();
    return SX_OK;
}


// ------------------------------------------------------------------------------------------------
// out[0] = max(out[0], max |a|, max |b|) over two fp32 arrays (b may be NULL): the magnitude of a backward pass's incoming
// adjoints, from which the slab kernels derive their power-of-two normalisation -- ONE launch over ALL elements (the torch form
// was two strided-sample reductions + a maximum).  Non-negative floats order like their bit patterns: one atomicMax per block;
// NaNs are skipped.  `out` must hold a non-negative value (zero) on entry.
__global__ __launch_bounds__(256) void absmax2_kernel(const float *__restrict__ a, int64_t na, const float *__restrict__ b,
                                                      int64_t nb, float *__restrict__ out) {
    float m = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < na; i += stride) {
        if (i + 3 < na && ((reinterpret_cast<uintptr_t>(a) & 15) == 0)) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(a + i);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        } else {
            for (int64_t j = i; j < na && j < i + 4; ++j) m = fmaxf(m, fabsf(a[j]));
        }
    }
    if (b != nullptr)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(b[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(reinterpret_cast<unsigned int *>(out), __float_as_uint(m));
    }
}
extern "C" int sx_absmax2(const float *a, int64_t na, const float *b, int64_t nb, float *out, void *stream) {
    SX_REQUIRE(a && out && na >= 0 && nb >= 0, "sx_absmax2: bad arguments");
    if (na == 0 && (b == nullptr || nb == 0)) return SX_OK;
    int64_t g = (na / 4 + 255) / 256;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(absmax2_kernel, dim3((int)g), dim3(256), 0, sx_stream(stream), a, na, b, b ? nb : 0, out);
    SX_LAUNCH_CHECK();
    return SX_OK;
}
