// Inverses of small triangular fp64 matrices (D <= 128), batched: the parameter preprocessing of AffineLU /
// MatrixExponential in training -- (L U)^-1 = U^-1 L^-1 re-derived every step.  Library triangular solves against the
// identity take 57 us per 128 x 128 matrix and run one matrix at a time; here one workgroup inverts one matrix and
// the batch runs side by side.  Forward substitution down the columns of X = T^-1 (an upper-triangular matrix is the
// lower-triangular one of the reversed index order, so both cases walk the same loop) is a chain of D dependent steps:
// what matters is the length of one step.  EIGHT lanes share a column (a wave owns eight columns): a step's dot product
// sum_k T[i][k] X[k][j] is split over them (k = k0 + 8 m + p), summed with three cross-lane exchanges, and lane p = 0
// writes X[i][j].  T and X sit in LDS in packed triangular form (2 x 64.5 KB at D = 128); a wave only ever reads the X
// entries it wrote itself and DS operations of a wave execute in order, so the loop has no barriers.  (Round 2: one
// lane per column, T through the scalar cache -- 320 us per launch at D = 128, every step a serial walk of up to 127
// products; this form: DESIGN.md 4.3.2.)
#include "sx_common.h"

extern __shared__ __attribute__((aligned(16))) double tri_lds[];      // Tp[D (D + 1) / 2] ++ Xp[D (D + 1) / 2], packed lower, by rows

__global__ __launch_bounds__(1024) void tri_inverse_kernel(const double *__restrict__ T, double *__restrict__ X, int D,
                                                           int lower, int unit) {
    const double *Tm = T + (int64_t)blockIdx.x * D * D;
    double *Xm = X + (int64_t)blockIdx.x * D * D;
    const int np = D * (D + 1) / 2;
    double *Tp = tri_lds, *Xp = tri_lds + np;
    const int tid = threadIdx.x;
    // stage T (rows coalesced); in the reversed order for an upper-triangular matrix
    for (int e = tid; e < D * D; e += 1024) {
        const int ii = e / D, kk = e - ii * D;
        const int i = lower ? ii : D - 1 - ii, k = lower ? kk : D - 1 - kk;
        if (k <= i) Tp[i * (i + 1) / 2 + k] = Tm[e];
    }
    __syncthreads();
    const int j = tid >> 3, p = tid & 7;
    const int k0 = (tid >> 6) << 3;                 // the wave's first column: entries of column j above row j are zeros
    if (k0 < D) {                                   // wave-uniform
        const int jj = j < D ? j : D - 1;           // (D not a multiple of 8: the spare columns of the last wave compute, never write)
        for (int i = k0; i < D; ++i) {
            const double *trow = Tp + i * (i + 1) / 2;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int k = k0 + p;
            for (; k + 24 < i; k += 32) {
                const double t0 = trow[k], t1 = trow[k + 8], t2 = trow[k + 16], t3 = trow[k + 24];
                const double x0 = Xp[k * (k + 1) / 2 + (jj <= k ? jj : 0)], x1 = Xp[(k + 8) * (k + 9) / 2 + (jj <= k + 8 ? jj : 0)];
                const double x2 = Xp[(k + 16) * (k + 17) / 2 + (jj <= k + 16 ? jj : 0)], x3 = Xp[(k + 24) * (k + 25) / 2 + (jj <= k + 24 ? jj : 0)];
                s0 += t0 * (jj <= k ? x0 : 0.0);
                s1 += t1 * (jj <= k + 8 ? x1 : 0.0);
                s2 += t2 * (jj <= k + 16 ? x2 : 0.0);
                s3 += t3 * (jj <= k + 24 ? x3 : 0.0);
            }
            for (; k < i; k += 8) {
                const double xv = Xp[k * (k + 1) / 2 + (jj <= k ? jj : 0)];
                s0 += trow[k] * (jj <= k ? xv : 0.0);
            }
            double sum = (s0 + s1) + (s2 + s3);
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 4, 64);
            double x = (i == jj ? 1.0 : 0.0) - sum;
            if (!unit) x /= trow[i];
            if (p == 0 && j <= i && j < D) Xp[i * (i + 1) / 2 + j] = x;
        }
    }
    __syncthreads();
    // X out, whole matrix (zeros outside the triangle), rows coalesced
    for (int e = tid; e < D * D; e += 1024) {
        const int ii = e / D, kk = e - ii * D;
        const int i = lower ? ii : D - 1 - ii, k = lower ? kk : D - 1 - kk;
        Xm[e] = k <= i ? Xp[i * (i + 1) / 2 + k] : 0.0;
    }
}

extern "C" int sx_tri_inverse_f64(const double *T, double *X, int32_t batch, int32_t D, int32_t lower, int32_t unit,
                                  void *stream) {
    SX_REQUIRE(T && X, "sx_tri_inverse_f64: null pointer");
    SX_REQUIRE(batch >= 0 && D >= 1 && D <= 128, "sx_tri_inverse_f64: D must be in 1..128");
    if (batch == 0) return SX_OK;
    const size_t lds = (size_t)D * (D + 1) * sizeof(double);      // two packed triangles
    static bool raised_on[64];                        // once per device: 129 KiB of LDS for D = 128
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool &raised = raised_on[dev & 63];
    if (lds > 48 * 1024 && !raised) {
        hipError_t e = hipFuncSetAttribute((const void *)tri_inverse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { sx_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        raised = true;
    }
    hipLaunchKernelGGL(tri_inverse_kernel, dim3(batch), dim3(1024), lds, sx_stream(stream), T, X, (int)D, (int)lower, (int)unit);
    SX_LAUNCH_CHECK();
    return SX_OK;
}
