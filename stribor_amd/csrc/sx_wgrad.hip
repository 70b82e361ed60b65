// Weight-gradient contraction of the training backward pass:  dW[i][j] += sum_n A[n][i] * B[n][j],
// db[i] += sum_n A[n][i], with A = [n_rows, M] and B = [n_rows, Nc] row-major fp32 slices of the per-row
// factors the backward flow kernel leaves in HBM (M, Nc <= 128; n_rows ~ 1e6).
//
// A tall-skinny "A^T B" whose reduction axis is the batch: library GEMMs run it on a handful of workgroups
// (measured 1.5 ms per 64x64x2^20 product = 9 % of HBM rate).  Here the batch is split over the whole chip and
// the row-major operands ARE the MFMA fragments: v_mfma_f32_32x32x2_f32 wants A^T[i = lane&31][k = lane>>5] and
// B[k = lane>>5][j = lane&31] with k = the row, i.e. each lane loads one float of a row (lanes of a half read
// 128 contiguous bytes), no transposes.  Each wave keeps the (M/32) x (Nc/32) output tiles in registers over its
// row slice and the workgroup adds them to dW with float atomics once (rows of 128 B per lane half).
#include "sx_common.h"

template <int MT, int NT>
__global__ __launch_bounds__(256) void wgrad_kernel(const float *__restrict__ A, int64_t lda,
                                                    const float *__restrict__ B, int64_t ldb, int64_t n_rows,
                                                    float *__restrict__ dW, int64_t ldw, float *__restrict__ db,
                                                    int m_valid, int n_valid) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    float bsum[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) bsum[m] = 0.f;

    // rows are dealt to waves in blocks of 16 (8 MFMA k-steps of 2 rows), grid-strided
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int64_t w_id = (int64_t)blockIdx.x * 4 + wave;
    for (int64_t r0 = w_id * 16; r0 < n_rows; r0 += n_waves * 16) {
        float a[8][MT], b[8][NT];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int64_t row = r0 + 2 * s + kk;
            const bool ok = row < n_rows;
#pragma unroll
            for (int m = 0; m < MT; ++m) a[s][m] = (ok && 32 * m + i < m_valid) ? A[row * lda + 32 * m + i] : 0.f;
#pragma unroll
            for (int n = 0; n < NT; ++n) b[s][n] = (ok && 32 * n + i < n_valid) ? B[row * ldb + 32 * n + i] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                bsum[m] += a[s][m];
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][m], b[s][n], acc[m][n], 0, 0, 0);
            }
        }
    }
    // C layout: lane (col j = lane&31, half) holds rows kmap(r, half)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * kk, col = 32 * n + i;
                if (row < m_valid && col < n_valid) atomicAdd(&dW[(int64_t)row * ldw + col], acc[m][n][r]);
            }
    if (db != nullptr) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float t = bsum[m] + __shfl_xor(bsum[m], 32, 64);
            if (kk == 0 && 32 * m + i < m_valid) atomicAdd(&db[32 * m + i], t);
        }
    }
}

extern "C" int sx_wgrad(const float *A, int64_t lda, int32_t M, const float *B, int64_t ldb, int32_t Nc,
                        int64_t n_rows, float *dW, int64_t ldw, float *db, void *stream) {
    SX_REQUIRE(A && B && dW, "sx_wgrad: null pointer");
    SX_REQUIRE(M >= 1 && M <= 128 && Nc >= 1 && Nc <= 128 && n_rows >= 0, "sx_wgrad: M, Nc must be in 1..128");
    if (n_rows == 0) return SX_OK;
    const int mt = (M + 31) / 32, nt = (Nc + 31) / 32;
    int64_t g = (n_rows + 16 * 4 - 1) / (16 * 4);
    if (g > 1024) g = 1024;
    hipStream_t st = sx_stream(stream);
#define SX_WG(MT_, NT_)                                                                                            \
    if (mt == MT_ && nt == NT_) {                                                                                  \
        hipLaunchKernelGGL((wgrad_kernel<MT_, NT_>), dim3((int)g), dim3(256), 0, st, A, lda, B, ldb, n_rows, dW, ldw, db, \
                           M, Nc);                                                                                 \
        SX_LAUNCH_CHECK();                                                                                         \
        return SX_OK;                                                                                              \
    }
    SX_WG(1, 1) SX_WG(1, 2) SX_WG(2, 1) SX_WG(2, 2) SX_WG(2, 4) SX_WG(4, 2) SX_WG(4, 1) SX_WG(1, 4) SX_WG(4, 4)
    SX_WG(3, 1) SX_WG(3, 2) SX_WG(3, 3) SX_WG(3, 4) SX_WG(1, 3) SX_WG(2, 3) SX_WG(4, 3)
#undef SX_WG
    sx_set_error("sx_wgrad: unsupported tile shape %d x %d", mt, nt);
    return SX_E_UNSUPPORTED;
}
