// Weight re-layout for the MFMA path: one nn.Linear -> v_mfma_f32_32x32x2_f32 A-operand fragments.
//
// v_mfma_f32_32x32x2_f32 computes D(32x32) = A(32x2) . B(2x32) + C with (cdna_hip_programming.md §3)
//   A: lane l holds A[i = l&31][k = l>>5]        B: lane l holds B[k = l>>5][j = l&31]
//   C/D: lane l, register r holds C[row = kmap(r, l>>5)][col = l&31],  kmap(r,h) = (r&3) + 8*(r>>2) + 4*h
// The fused kernels keep samples on the MFMA column (lane&31) and features on the C rows, so a C tile
// is directly the B operand of the next GEMM provided k-step s of lane-half h is matched with the
// weight column kmap(s,h).  That permutation is baked in here, once per parameter update.
#include "sx_common.h"

__host__ __device__ static inline int sx_kmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// the library's default GEMM arithmetic (both are built in; sx_pack_linear / sx_flow_run take `precision`)
extern "C" int sx_fragment_mode(void) { return SX_GEMM_F16X3; }

extern "C" size_t sx_packed_linear_floats(int32_t m_tiles, int32_t k_tiles) {
    return (size_t)m_tiles * k_tiles * 1024 + (size_t)m_tiles * 32;
}

// one pack: the floats o = first, first + stride, ... of the blob (workgroups of one launch, or of one job's row of a batched launch)
__device__ __forceinline__ void pack_linear_body(const float *__restrict__ W, const float *__restrict__ b,
                                                 int out_dim, int in_dim,
                                                 const int32_t *__restrict__ row_idx,
                                                 const int32_t *__restrict__ col_idx, int m_tiles,
                                                 int k_tiles, const float *__restrict__ row_scale,
                                                 const float *__restrict__ bias_scale, float fold_ones,
                                                 int frag_mode, int transpose, uint32_t *__restrict__ err_flag,
                                                 float *__restrict__ dst, uint32_t *__restrict__ bound_out, int first, int stride) {
    // transpose: the operand is W^T (row slots index W's columns, column slots index W's rows)
    auto Wat = [&](int r, int c) -> float { return transpose ? W[(int64_t)c * in_dim + r] : W[(int64_t)r * in_dim + c]; };
    const int n_a = m_tiles * k_tiles * 1024;
    const int total = n_a + m_tiles * 32;
    for (int o = first; o < total; o += stride) {
        float v = 0.f;
        if (o < n_a) {
            // o = (((m*k_tiles + kt)*4 + g)*64 + lane)*4 + e
            const int e = o & 3, lane = (o >> 2) & 63, g = (o >> 8) & 3;
            const int mk = o >> 10;
            const int kt = mk % k_tiles, m = mk / k_tiles;
            const int row = row_idx[32 * m + (lane & 31)];
            const int col = col_idx[32 * kt + sx_kmap(4 * g + e, lane >> 5)];
            if (frag_mode == 0) {
                if (row >= 0 && col >= 0) v = Wat(row, col) * (row_scale ? row_scale[32 * m + (lane & 31)] : 1.f);
            } else {
                // fp16 x 3 split fragments for v_mfma_f32_32x32x16_f16: [s(2)][hi,lo][lane][8 halfs];
                // this float holds halfs j = 2q, 2q+1 of k16-step s (k slot = kmap(8s + j, lane>>5)).
                const int q = e, part = g & 1, s16 = g >> 1;
                uint32_t bits = 0;
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * q + jj;
                    const int col2 = col_idx[32 * kt + sx_kmap(8 * s16 + j, lane >> 5)];
                    float w = 0.f;
                    if (row >= 0 && col2 >= 0) w = Wat(row, col2) * (row_scale ? row_scale[32 * m + (lane & 31)] : 1.f);
                    // |w| > 65504 rounds to inf here (and inf - inf below): every product it enters is non-finite, and
                    // the flag tells the caller why
                    if (fabsf(w) > 65504.f && err_flag != nullptr)
                        __hip_atomic_fetch_or(err_flag, SX_FLAG_F16_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const _Float16 hi = (_Float16)w;
                    const _Float16 lo = (_Float16)(w - (float)hi);
                    const _Float16 pick = part ? lo : hi;
                    bits |= (uint32_t)__builtin_bit_cast(uint16_t, pick) << (16 * jj);
                }
                v = __uint_as_float(bits);
            }
        } else {
            // bias[m][h][r]
            const int q = o - n_a;
            const int r = q & 15, h = (q >> 4) & 1, m = q >> 5;
            const int slot = 32 * m + sx_kmap(r, h);
            const int row = row_idx[slot];
            if (row >= 0) {
                double acc = b != nullptr ? (double)b[row] : 0.0;
                if (fold_ones != 0.f || bound_out != nullptr) {          // b' = b + fold * sum over the live input slots of W[row][.]
                    double rs = 0.0, pos = 0.0, neg = 0.0;
                    const double rsc = row_scale ? (double)row_scale[slot] : 1.0;
                    for (int c = 0; c < 32 * k_tiles; ++c) {
                        const int col = col_idx[c];
                        if (col >= 0) {
                            const double wv = (double)Wat(row, col);
                            rs += wv;
                            if (wv * rsc > 0.0) pos += wv * rsc; else neg += wv * rsc;
                        }
                    }
                    acc += (double)fold_ones * rs;
                    v = (float)(acc * (bias_scale ? (double)bias_scale[slot] : 1.0));
                    if (bound_out != nullptr) {
                        // the largest |output| of this packed row over inputs in [0, 1]^k (the folded tanh r = (1 - tanh) / 2 that
                        // feeds the spline phases): bias' + the positive (negative) packed weights; kept as the running maximum of
                        // every row packed into this slot (non-negative floats order like their bit patterns; NaN -> inf)
                        const float bd = (float)fmax(fabs((double)v + pos), fabs((double)v + neg));
                        atomicMax(bound_out, __float_as_uint(bd == bd ? bd : __builtin_inff()));
                    }
                } else
                v = (float)(acc * (bias_scale ? (double)bias_scale[slot] : 1.0));
            }
        }
        dst[o] = v;
    }
}
__global__ __launch_bounds__(256) void pack_linear_kernel(const float *__restrict__ W, const float *__restrict__ b,
                                                          int out_dim, int in_dim,
                                                          const int32_t *__restrict__ row_idx,
                                                          const int32_t *__restrict__ col_idx, int m_tiles,
                                                          int k_tiles, const float *__restrict__ row_scale,
                                                          const float *__restrict__ bias_scale, float fold_ones,
                                                          int frag_mode, int transpose, uint32_t *__restrict__ err_flag,
                                                          float *__restrict__ dst, uint32_t *__restrict__ bound_out) {
    pack_linear_body(W, b, out_dim, in_dim, row_idx, col_idx, m_tiles, k_tiles, row_scale, bias_scale, fold_ones, frag_mode, transpose,
                     err_flag, dst, bound_out, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
// a table of packs in ONE launch: blockIdx.y = job (a training step re-packs every weight of a flow: 100+ launches of 3 - 9 us before)
__global__ __launch_bounds__(256) void pack_linear_batch_kernel(const sx_pack_job *__restrict__ jobs, int frag_mode,
                                                                uint32_t *__restrict__ err_flag) {
    const sx_pack_job j = jobs[blockIdx.y];
    pack_linear_body(j.W, j.b, j.out_dim, j.in_dim, j.row_idx, j.col_idx, j.m_tiles, j.k_tiles, j.row_scale, j.bias_scale, j.fold_ones,
                     frag_mode, j.transpose, err_flag, j.dst, reinterpret_cast<uint32_t *>(j.bound_out),
                     blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

static int pack_linear_impl(const float *W, const float *b, int32_t out_dim, int32_t in_dim,
                              const int32_t *row_idx, const int32_t *col_idx, int32_t m_tiles, int32_t k_tiles,
                              const float *row_scale, const float *bias_scale, float fold_ones, int32_t transpose,
                              int32_t precision, uint32_t *err_flag, float *dst, float *bound_out, void *stream) {
    SX_REQUIRE(precision == SX_GEMM_F32 || precision == SX_GEMM_F16X3, "sx_pack_linear: unknown precision %d", precision);
    const int frag_mode = precision;
    SX_REQUIRE(W && row_idx && col_idx && dst, "sx_pack_linear: null pointer");
    SX_REQUIRE(m_tiles > 0 && k_tiles > 0 && out_dim > 0 && in_dim > 0, "sx_pack_linear: bad sizes");
    const int total = (int)sx_packed_linear_floats(m_tiles, k_tiles);
    const int grid = (total + 255) / 256;
    hipLaunchKernelGGL(pack_linear_kernel, dim3(grid > 1024 ? 1024 : grid), dim3(256), 0, sx_stream(stream), W, b,
                       out_dim, in_dim, row_idx, col_idx, m_tiles, k_tiles, row_scale, bias_scale, fold_ones, frag_mode, transpose, err_flag, dst,
                       reinterpret_cast<uint32_t *>(bound_out));
    SX_LAUNCH_CHECK();
    return SX_OK;
}

extern "C" int sx_pack_linear(const float *W, const float *b, int32_t out_dim, int32_t in_dim,
                              const int32_t *row_idx, const int32_t *col_idx, int32_t m_tiles, int32_t k_tiles,
                              const float *row_scale, const float *bias_scale, float fold_ones, int32_t transpose,
                              int32_t precision, uint32_t *err_flag, float *dst, void *stream) {
    return pack_linear_impl(W, b, out_dim, in_dim, row_idx, col_idx, m_tiles, k_tiles, row_scale, bias_scale, fold_ones, transpose,
                            precision, err_flag, dst, nullptr, stream);
}

extern "C" int sx_pack_linear_bound(const float *W, const float *b, int32_t out_dim, int32_t in_dim,
                                    const int32_t *row_idx, const int32_t *col_idx, int32_t m_tiles, int32_t k_tiles,
                                    const float *row_scale, const float *bias_scale, float fold_ones, int32_t transpose,
                                    int32_t precision, uint32_t *err_flag, float *dst, float *bound_out, void *stream) {
    SX_REQUIRE(bound_out != nullptr, "sx_pack_linear_bound: null bound_out");
    return pack_linear_impl(W, b, out_dim, in_dim, row_idx, col_idx, m_tiles, k_tiles, row_scale, bias_scale, fold_ones, transpose,
                            precision, err_flag, dst, bound_out, stream);
}

extern "C" int sx_pack_linear_batch(const sx_pack_job *jobs, int32_t n_jobs, int32_t max_floats, int32_t precision,
                                    uint32_t *err_flag, void *stream) {
    SX_REQUIRE(precision == SX_GEMM_F32 || precision == SX_GEMM_F16X3, "sx_pack_linear_batch: unknown precision %d", precision);
    SX_REQUIRE(jobs != nullptr && n_jobs > 0 && n_jobs <= 65535, "sx_pack_linear_batch: 1 .. 65535 jobs (got %d)", n_jobs);
    SX_REQUIRE(max_floats > 0, "sx_pack_linear_batch: max_floats = the largest sx_packed_linear_floats of the table");
    const int gx = (max_floats + 255) / 256;
    hipLaunchKernelGGL(pack_linear_batch_kernel, dim3(gx > 64 ? 64 : gx, n_jobs), dim3(256), 0, sx_stream(stream), jobs, (int)precision,
                       err_flag);
    SX_LAUNCH_CHECK();
    return SX_OK;
}
