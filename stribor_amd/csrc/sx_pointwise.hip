// Parameter-free element-wise flows of the reference: Sigmoid / Logit (stribor/flows/sigmoid.py:9-56), ELU and
// LeakyReLU (flows/activations.py:11-101), Cumsum / Diff over the last axis (flows/cumsum.py:9-92).
//
// One pass over HBM: read x, write y and -- when asked -- the per-element log-derivative and / or its row sum.
// For the inverse-direction kinds the log-derivative is the one Transform.inverse_and_log_det_jacobian returns
// (flow.py:42-47): MINUS the forward log-derivative evaluated at the value just produced, in the same pass.
// Element kinds: one lane per element, consecutive lanes = consecutive addresses; the row sum is a shuffle sum when a
// row is a power-of-two group inside a wave, float atomics otherwise.  Cumsum is a sequential scan per row (one lane
// per row, running sum in double) so that it reproduces torch.cumsum bit for bit (test_cumsum.py asserts equality).
#include "sx_common.h"
#include <stdlib.h>

__device__ __forceinline__ float pw_softplus(float v) { return v > 20.f ? v : log1pf(expf(v)); }   // F.softplus

template <bool BF16>
__device__ __forceinline__ float pw_load(const void *p, int64_t off) {
    if constexpr (BF16) return bf16_to_f32(reinterpret_cast<const uint16_t *>(p)[off]);
    else return reinterpret_cast<const float *>(p)[off];
}
template <bool BF16>
__device__ __forceinline__ void pw_store(void *p, int64_t off, float v) {
    if constexpr (BF16) reinterpret_cast<uint16_t *>(p)[off] = f32_to_bf16(v);
    else reinterpret_cast<float *>(p)[off] = v;
}

#include "sx_pointwise_core.h"

// VEC = 4: one lane = 4 consecutive elements of one row (dim % 4 == 0): 16-byte (fp32) / 8-byte (bf16) accesses
template <bool BF16, int VEC, bool ALIGNED>
__global__ __launch_bounds__(256) void pointwise_kernel(const void *__restrict__ x, void *__restrict__ y,
                                                        float *__restrict__ ldj, float *__restrict__ ldiag,
                                                        int64_t n_rows, int dim, int kind, float param, float log_slope,
                                                        int ldj_mode /*0 none, 1 group, 2 row-aligned units*/, int ldj_acc) {
    const int gdim = dim / VEC;                         // lanes per row
    const int64_t total = n_rows * gdim;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int row_shift = 31 - __builtin_clz(gdim);     // used when gdim is a power of two
    // one lane = VEC consecutive elements at (vector) index i: load, evaluate, store; returns the lane's log-det part
    auto body = [&](int64_t i) -> float {
        float ld_sum = 0.f;
        float xv[VEC], out[VEC], ld[VEC];
        if constexpr (VEC == 4) {
            if constexpr (BF16) {
                const u16x4 u = __builtin_nontemporal_load(reinterpret_cast<const u16x4 *>(x) + i);     // every byte is touched once
                xv[0] = bf16_to_f32(u.x); xv[1] = bf16_to_f32(u.y); xv[2] = bf16_to_f32(u.z); xv[3] = bf16_to_f32(u.w);
            } else {
                const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(x) + i);
                xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w;
            }
        } else xv[0] = pw_load<BF16>(x, i);
#pragma unroll
        for (int c = 0; c < VEC; ++c) { pw_eval(kind, param, log_slope, xv[c], out[c], ld[c]); ld_sum += ld[c]; }
        if (y) {
            if constexpr (VEC == 4) {
                if constexpr (BF16) __builtin_nontemporal_store(u16x4{f32_to_bf16(out[0]), f32_to_bf16(out[1]), f32_to_bf16(out[2]), f32_to_bf16(out[3])}, reinterpret_cast<u16x4 *>(y) + i);
                else __builtin_nontemporal_store(f32x4{out[0], out[1], out[2], out[3]}, reinterpret_cast<f32x4 *>(y) + i);
            } else pw_store<BF16>(y, i, out[0]);
        }
        if (ldiag) {
            if constexpr (VEC == 4) reinterpret_cast<f32x4 *>(ldiag)[i] = f32x4{ld[0], ld[1], ld[2], ld[3]};
            else ldiag[i] = ld[0];
        }
        return ld_sum;
    };
    if constexpr (ALIGNED) {
        // row sums for any row length without atomics: row-aligned units (sx_common.h), fixed-order sums
        const int lane = threadIdx.x & 63;
        const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = stride >> 6;
        const sx_units units = sx_make_units(n_rows, gdim, true);
        for (int64_t u = wave; u < units.n_units; u += n_waves) {
            float row_acc = 0.f;
            for (int chunk = 0; chunk < units.chunks; ++chunk) {
                int64_t e0;
                int n_here;
                sx_unit_span(units, u, chunk, n_rows, gdim, &e0, &n_here);
                const bool valid = lane < n_here;
                const float s0 = valid ? body(e0 + lane) : 0.f;
                if (units.chunks == 1) {
                    const int pos = lane % gdim;
                    const float s = segment_sum_rt(s0, pos, gdim);
                    if (valid && pos == 0) { const int64_t r = (e0 + lane) / gdim; ldj[r] = (ldj_acc ? ldj[r] : 0.f) + s; }
                } else {
                    row_acc += s0;
                }
            }
            if (units.chunks > 1) {
                const float s = group_sum<64>(row_acc);
                if (lane == 0) ldj[u] = (ldj_acc ? ldj[u] : 0.f) + s;
            }
        }
        return;
    }
    // the loop bound is rounded up to whole waves so that every lane of a wave reaches the shuffle sum
    const int64_t total_up = (total + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_up; i += stride) {
        const bool valid = i < total;
        const float ld_sum = valid ? body(i) : 0.f;
        if (ldj_mode == 1) {            // gdim is a power of two <= 64: rows are aligned lane groups
            const float s = group_sum_rt(ld_sum, gdim);
            if (valid && (i & (gdim - 1)) == 0) { const int64_t r = i >> row_shift; ldj[r] = (ldj_acc ? ldj[r] : 0.f) + s; }
        }
    }
}

// Cumsum / Diff: a wave owns 64 consecutive rows = one contiguous span of 64*dim elements.  The span is copied into
// the wave's LDS slice with coalesced loads (row stride dim+1 dwords: conflict-free column walks), each lane scans
// its own row there, and the span goes back out coalesced.  Rows longer than the slice allows take the direct loop.
extern __shared__ __attribute__((aligned(16))) float pw_smem[];
template <bool BF16>
__global__ __launch_bounds__(256) void cumsum_kernel(const void *__restrict__ x, void *__restrict__ y, int64_t n_rows,
                                                     int dim, int diff, int staged) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_groups = (n_rows + 63) >> 6;
    const int RS = dim + 1;
    float *sp = pw_smem + (size_t)wave * 64 * RS;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < n_groups; grp += (int64_t)gridDim.x * 4) {
        const int64_t r0 = grp << 6;
        const int rows = (int)((n_rows - r0) < 64 ? (n_rows - r0) : 64);
        const int total = rows * dim;
        if (staged) {
            for (int i = lane; i < total; i += 64) {
                const int r = i / dim, c = i - r * dim;
                sp[r * RS + c] = pw_load<BF16>(x, r0 * dim + i);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (lane < rows) {
            // cumsum.py:62: torch.cumsum on CPU keeps the running sum in double (acc_type<float>) and rounds each
            // output; diff (:33): x - pad(x)[..., :-1]
            double acc = 0.0;
            float prev = 0.f;
            float *row = sp + lane * RS;
            const int64_t g0 = (r0 + lane) * dim;
            for (int c = 0; c < dim; ++c) {
                const float v = staged ? row[c] : pw_load<BF16>(x, g0 + c);
                float o;
                if (diff) { o = v - prev; prev = v; }
                else { acc += (double)v; o = (float)acc; }
                if (staged) row[c] = o; else pw_store<BF16>(y, g0 + c, o);
            }
        }
        if (staged) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int i = lane; i < total; i += 64) {
                const int r = i / dim, c = i - r * dim;
                pw_store<BF16>(y, r0 * dim + i, sp[r * RS + c]);
            }
        }
    }
}

// Vectorised form (fp32, dim % 4 == 0, 16-byte aligned): every global access and every LDS access moves 16 B.  The
// wave's 64 rows are one contiguous span; row stride in LDS = dim + 4 dwords: 16-byte aligned, and the 64 lanes of a
// ds_read_b128 / ds_write_b128 column walk (lane = row) cover all 32 banks 8 times -- the minimum for 1 KiB.
__global__ __launch_bounds__(256) void cumsum_vec_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n_rows,
                                                         int dim, int diff) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_groups = (n_rows + 63) >> 6;
    const int RS = dim + 4, Q = dim >> 2;                 // Q float4 per row
    float *sp = pw_smem + (size_t)wave * 64 * RS;
    const float inv_q = 1.0f / (float)Q;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < n_groups; grp += (int64_t)gridDim.x * 4) {
        const int64_t r0 = grp << 6;
        const int rows = (int)((n_rows - r0) < 64 ? (n_rows - r0) : 64);
        const int total4 = rows * Q;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(x + r0 * dim);
        for (int i = lane; i < total4; i += 64) {
            const int r = (int)(((float)i + 0.5f) * inv_q), c4 = i - r * Q;      // i / Q, exact for i < 2^22
            *reinterpret_cast<f32x4 *>(sp + r * RS + 4 * c4) = src[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < rows) {
            // cumsum.py:62: torch.cumsum on CPU keeps the running sum in double (acc_type<float>) and rounds each
            // output; diff (:33): x - pad(x)[..., :-1]
            double acc = 0.0;
            float prev = 0.f;
            f32x4 *row = reinterpret_cast<f32x4 *>(sp + lane * RS);
            for (int c4 = 0; c4 < Q; ++c4) {
                const f32x4 v = row[c4];
                f32x4 o;
                if (diff) {
                    o.x = v.x - prev; o.y = v.y - v.x; o.z = v.z - v.y; o.w = v.w - v.z;
                    prev = v.w;
                } else {
                    acc += (double)v.x; o.x = (float)acc;
                    acc += (double)v.y; o.y = (float)acc;
                    acc += (double)v.z; o.z = (float)acc;
                    acc += (double)v.w; o.w = (float)acc;
                }
                row[c4] = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f32x4 *dst = reinterpret_cast<f32x4 *>(y + r0 * dim);
        for (int i = lane; i < total4; i += 64) {
            const int r = (int)(((float)i + 0.5f) * inv_q), c4 = i - r * Q;
            dst[i] = *reinterpret_cast<const f32x4 *>(sp + r * RS + 4 * c4);
        }
    }
}

// The same with the NEXT group's rows in flight while the current group is scanned (dim <= 64: the wave's 64 x dim span
// is at most 16 float4 per lane).  A wave of cumsum_vec_kernel alternates load / scan / store phases, so only a third of the
// resident waves have loads outstanding at any time; here every wave always has one group (16 KB at dim = 64) in flight.
__global__ __launch_bounds__(256) void cumsum_vec_pipe_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n_rows,
                                                              int dim, int diff) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_groups = (n_rows + 63) >> 6;
    const int RS = dim + 4, Q = dim >> 2;                 // Q <= 16 float4 per row
    float *sp = pw_smem + (size_t)wave * 64 * RS;
    const float inv_q = 1.0f / (float)Q;
    const int64_t step = (int64_t)gridDim.x * 4;
    f32x4 nxt[16];
    auto fetch = [&](int64_t grp) {
        if (grp >= n_groups) return;
        const int64_t r0 = grp << 6;
        const int rows = (int)((n_rows - r0) < 64 ? (n_rows - r0) : 64);
        const int total4 = rows * Q;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(x + r0 * dim);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int i = lane + 64 * kk;
            if (kk < Q && i < total4) nxt[kk] = __builtin_nontemporal_load(src + i);     // every byte is touched once
        }
    };
    int64_t grp = (int64_t)blockIdx.x * 4 + wave;
    fetch(grp);
    for (; grp < n_groups; grp += step) {
        const int64_t r0 = grp << 6;
        const int rows = (int)((n_rows - r0) < 64 ? (n_rows - r0) : 64);
        const int total4 = rows * Q;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int i = lane + 64 * kk;
            if (kk < Q && i < total4) {
                const int r = (int)(((float)i + 0.5f) * inv_q), c4 = i - r * Q;      // i / Q, exact for i < 2^22
                *reinterpret_cast<f32x4 *>(sp + r * RS + 4 * c4) = nxt[kk];
            }
        }
        fetch(grp + step);                                   // lands during the scan and the stores below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < rows) {
            double acc = 0.0;                                // cumsum.py:62 (see cumsum_vec_kernel)
            float prev = 0.f;
            f32x4 *row = reinterpret_cast<f32x4 *>(sp + lane * RS);
            for (int c4 = 0; c4 < Q; ++c4) {
                const f32x4 v = row[c4];
                f32x4 o;
                if (diff) {
                    o.x = v.x - prev; o.y = v.y - v.x; o.z = v.z - v.y; o.w = v.w - v.z;
                    prev = v.w;
                } else {
                    acc += (double)v.x; o.x = (float)acc;
                    acc += (double)v.y; o.y = (float)acc;
                    acc += (double)v.z; o.z = (float)acc;
                    acc += (double)v.w; o.w = (float)acc;
                }
                row[c4] = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f32x4 *dst = reinterpret_cast<f32x4 *>(y + r0 * dim);
        for (int i = lane; i < total4; i += 64) {
            const int r = (int)(((float)i + 0.5f) * inv_q), c4 = i - r * Q;
            __builtin_nontemporal_store(*reinterpret_cast<const f32x4 *>(sp + r * RS + 4 * c4), dst + i);
        }
    }
}

// ---- backward (training, layer-wise path): dL/dx = gy * d(out)/dx + gldj[row] * d(ld)/dx, per element ----------------
__device__ __forceinline__ float pw_grad(int kind, float param, float x, float gy, float gl) {
    switch (kind) {
        case SX_PW_SIGMOID: {
            const float sg = 1.f / (1.f + expf(-x));
            return gy * sg * (1.f - sg) + gl * (1.f - 2.f * sg);
        }
        case SX_PW_LOGIT: {
            const float y = fminf(fmaxf(x, PW_TINY), PW_ONE_MINUS_EPS);
            const float inv = 1.f / (y * (1.f - y));
            return (x == y) ? (gy + gl * (2.f * y - 1.f)) * inv : 0.f;                 // clamped ends: constant
        }
        case SX_PW_ELU: return x > 0.f ? gy : gy * expf(x) + gl;
        case SX_PW_ELU_INV: {
            const float d = x > 0.f ? 1.f : 1.f / (1.f + x);
            return gy * d + (x < 0.f ? -gl * d : 0.f);                                 // ld = max(-out, 0)
        }
        default: return gy * (x >= 0.f ? 1.f : param);                                  // LeakyReLU / its inverse
    }
}

__global__ __launch_bounds__(256) void pointwise_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                            const float *__restrict__ gldj, const float *__restrict__ gldiag,
                                                            float *__restrict__ gx, int64_t n_rows, int dim, int kind, float param) {
    const int64_t total = n_rows * dim;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
        gx[i] = pw_grad(kind, param, x[i], gy[i], (gldj ? gldj[i / dim] : 0.f) + (gldiag ? gldiag[i] : 0.f));
}

// cumsum's adjoint is the reversed cumsum, diff's the reversed diff: one lane per row, from the last column back
__global__ __launch_bounds__(256) void cumsum_bwd_kernel(const float *__restrict__ gy, float *__restrict__ gx, int64_t n_rows,
                                                         int dim, int diff) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
        double acc = 0.0;
        float prev = 0.f;
        for (int c = dim - 1; c >= 0; --c) {
            const float v = gy[r * dim + c];
            if (diff) { gx[r * dim + c] = v - prev; prev = v; }
            else { acc += (double)v; gx[r * dim + c] = (float)acc; }
        }
    }
}

extern "C" int sx_pointwise_bwd(const float *x, const float *gy, const float *gldj, const float *gldiag, float *gx, int64_t n_rows, int32_t dim,
                                int32_t kind, float param, void *stream) {
    SX_REQUIRE(x && gy && gx, "sx_pointwise_bwd: null pointer");
    SX_REQUIRE(dim > 0 && n_rows >= 0, "sx_pointwise_bwd: bad sizes");
    SX_REQUIRE(kind >= SX_PW_SIGMOID && kind <= SX_PW_DIFF, "sx_pointwise_bwd: unknown kind %d", kind);
    if (n_rows == 0) return SX_OK;
    hipStream_t st = sx_stream(stream);
    if (kind == SX_PW_CUMSUM || kind == SX_PW_DIFF) {
        int64_t g = (n_rows + 255) / 256;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(cumsum_bwd_kernel, dim3((int)g), dim3(256), 0, st, gy, gx, n_rows, dim, kind == SX_PW_DIFF);
    } else {
        int64_t g = (n_rows * dim + 255) / 256;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(pointwise_bwd_kernel, dim3((int)g), dim3(256), 0, st, x, gy, gldj, gldiag, gx, n_rows, dim, kind, param);
    }
    SX_LAUNCH_CHECK();
    return SX_OK;
}

extern "C" int sx_pointwise(const void *x, void *y, float *ldj, float *ldiag, int64_t n_rows, int32_t dim,
                            int32_t dtype, int32_t kind, float param, int32_t ldj_accumulate, void *stream) {
    SX_REQUIRE(x != nullptr, "sx_pointwise: null x");
    SX_REQUIRE(dim > 0 && n_rows >= 0, "sx_pointwise: bad sizes");
    SX_REQUIRE(dtype == SX_F32 || dtype == SX_BF16, "sx_pointwise: bad dtype");
    SX_REQUIRE(kind >= SX_PW_SIGMOID && kind <= SX_PW_DIFF, "sx_pointwise: unknown kind %d", kind);
    if (n_rows == 0) return SX_OK;
    hipStream_t st = sx_stream(stream);
    if (kind == SX_PW_CUMSUM || kind == SX_PW_DIFF) {
        SX_REQUIRE(x != y, "sx_pointwise: cumsum / diff need a separate output");
        if (y != nullptr) {                 // y == NULL: only the (zero) log-determinants are wanted
            int64_t g = (n_rows + 255) / 256;
            if (g > 2048) g = 2048;
            const size_t lds4 = (size_t)4 * 64 * (dim + 4) * sizeof(float);
            if (dtype == SX_F32 && dim % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && lds4 <= 150 * 1024 && dim <= 16384) {
                // the attribute is set to this launch's need (a blanket 160 KiB cost the spline slab backward a workgroup
                // per CU), and only when it changes
                static size_t set_to[64];
                int dev = 0;
                (void)hipGetDevice(&dev);
                if (lds4 > 48 * 1024 && set_to[dev & 63] != lds4) {
                    (void)hipFuncSetAttribute((const void *)cumsum_vec_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
                    set_to[dev & 63] = lds4;
                }
                static const bool no_pipe = sx_debug_knob("SX_CUMSUM_NO_PIPE", 0) != 0;          // experiments, read once
                if (dim <= 64 && !no_pipe) {
                    static size_t set_p[64];
                    if (lds4 > 48 * 1024 && set_p[dev & 63] != lds4) {
                        (void)hipFuncSetAttribute((const void *)cumsum_vec_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
                        set_p[dev & 63] = lds4;
                    }
                    // persistent grid: as many workgroups as are resident at once, each streaming its share of the groups
                    int64_t per_cu = (160 * 1024) / (int64_t)lds4;
                    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
                    const int64_t gp = g < 256 * per_cu ? g : 256 * per_cu;
                    hipLaunchKernelGGL(cumsum_vec_pipe_kernel, dim3((int)gp), dim3(256), lds4, st, (const float *)x, (float *)y, n_rows,
                                       dim, kind == SX_PW_DIFF);
                } else
                hipLaunchKernelGGL(cumsum_vec_kernel, dim3((int)g), dim3(256), lds4, st, (const float *)x, (float *)y, n_rows, dim,
                                   kind == SX_PW_DIFF);
                SX_LAUNCH_CHECK();
            } else {
            const size_t lds = (size_t)4 * 64 * (dim + 1) * sizeof(float);
            const int staged = lds <= 150 * 1024;
            const size_t dyn = staged ? lds : 0;
            if (dyn > 48 * 1024) {
                (void)hipFuncSetAttribute((const void *)cumsum_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
                (void)hipFuncSetAttribute((const void *)cumsum_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            }
            if (dtype == SX_BF16) hipLaunchKernelGGL(cumsum_kernel<true>, dim3((int)g), dim3(256), dyn, st, x, y, n_rows, dim, kind == SX_PW_DIFF, staged);
            else hipLaunchKernelGGL(cumsum_kernel<false>, dim3((int)g), dim3(256), dyn, st, x, y, n_rows, dim, kind == SX_PW_DIFF, staged);
            SX_LAUNCH_CHECK();
            }
        }
        if (ldj && !ldj_accumulate) {
            hipError_t e = hipMemsetAsync(ldj, 0, n_rows * sizeof(float), st);
            if (e != hipSuccess) { sx_set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
        }
        if (ldiag) {
            hipError_t e = hipMemsetAsync(ldiag, 0, n_rows * dim * sizeof(float), st);
            if (e != hipSuccess) { sx_set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
        }
        return SX_OK;
    }
    const size_t esz = dtype == SX_BF16 ? 2 : 4;
    const bool vec4 = (dim % 4 == 0) && (((uintptr_t)x % (4 * esz)) == 0) && (y == nullptr || ((uintptr_t)y % (4 * esz)) == 0) &&
                      (ldiag == nullptr || ((uintptr_t)ldiag % 16) == 0);
    const int gdim = vec4 ? dim / 4 : dim;
    int ldj_mode = 0;
    if (ldj) {
        ldj_mode = ((gdim & (gdim - 1)) == 0 && gdim <= 64) ? 1 : 2;      // 2: row-aligned units, deterministic sums
    }
    // log(negative_slope) in double like math.log (activations.py:99); the inverse kind carries 1 / slope
    float log_slope = 0.f;
    if (kind == SX_PW_LEAKY_RELU || kind == SX_PW_LEAKY_RELU_INV) {
        SX_REQUIRE(param > 0.f, "sx_pointwise: LeakyReLU slope must be positive");
        log_slope = (float)(kind == SX_PW_LEAKY_RELU ? log((double)param) : -log((double)param));
    }
    int64_t g = ldj_mode == 2 ? (sx_make_units(n_rows, gdim, true).n_units + 3) / 4 : (n_rows * gdim + 255) / 256;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
#define SX_PWL2(BF, V, AL)                                                                                        \
    hipLaunchKernelGGL((pointwise_kernel<BF, V, AL>), dim3((int)g), dim3(256), 0, st, x, y, ldj, ldiag, n_rows, dim, kind, \
                       param, log_slope, ldj_mode, ldj_accumulate)
#define SX_PWL(BF, V) do { if (ldj_mode == 2) SX_PWL2(BF, V, true); else SX_PWL2(BF, V, false); } while (0)
    if (dtype == SX_BF16) { if (vec4) SX_PWL(true, 4); else SX_PWL(true, 1); }
    else { if (vec4) SX_PWL(false, 4); else SX_PWL(false, 1); }
#undef SX_PWL
#undef SX_PWL2
    SX_LAUNCH_CHECK();
    return SX_OK;
}
