// Inverses of small triangular fp64 matrices (D <= 128), batched: the parameter preprocessing of AffineLU /
// MatrixExponential in training -- (L U)^-1 = U^-1 L^-1 re-derived every step.  Library triangular solves against the
// identity take 57 us per 128 x 128 matrix and run one matrix at a time; here one workgroup inverts one matrix and
// the batch runs side by side.  Thread j owns column j of X = T^-1 (forward / back substitution); every lane walks the
// same (i, k) loop -- X is triangular too, so the entries a shorter column would skip are zeros -- which makes the
// T[i][k] loads wave-uniform (scalar cache) and leaves the column in LDS as the lane's private, dynamically indexed storage
// (no barriers: a lane only ever reads what it wrote).
#include "sx_common.h"

extern __shared__ __attribute__((aligned(16))) double tri_cols[];      // [D][D]: tri_cols[k * D + j] = X[k][j]

__global__ __launch_bounds__(128) void tri_inverse_kernel(const double *__restrict__ T, double *__restrict__ X, int D,
                                                          int lower, int unit) {
    const double *Tm = T + (int64_t)blockIdx.x * D * D;
    double *Xm = X + (int64_t)blockIdx.x * D * D;
    const int j = threadIdx.x;
    if (j >= D) return;
    for (int step = 0; step < D; ++step) {
        const int i = lower ? step : D - 1 - step;
        const int k0 = lower ? 0 : i + 1, k1 = lower ? i : D;           // the already-solved entries of this column
        const double *trow = Tm + (int64_t)i * D;
        // eight independent accumulators: the LDS reads and the (wave-uniform) T loads of a group are issued together
        double s[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] = 0.0;
        int k = k0;
        for (; k + 7 < k1; k += 8) {
            double t[8], c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { t[u] = trow[k + u]; c[u] = tri_cols[(k + u) * D + j]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += t[u] * c[u];
        }
        for (; k < k1; ++k) s[0] += trow[k] * tri_cols[k * D + j];
        double x = (i == j ? 1.0 : 0.0) - (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])));
        if (!unit) x /= trow[i];
        tri_cols[i * D + j] = x;
        Xm[(int64_t)i * D + j] = x;
    }
}

extern "C" int sx_tri_inverse_f64(const double *T, double *X, int32_t batch, int32_t D, int32_t lower, int32_t unit,
                                  void *stream) {
    SX_REQUIRE(T && X, "sx_tri_inverse_f64: null pointer");
    SX_REQUIRE(batch >= 0 && D >= 1 && D <= 128, "sx_tri_inverse_f64: D must be in 1..128");
    if (batch == 0) return SX_OK;
    const size_t lds = (size_t)D * D * sizeof(double);
    static bool raised_on[64];                        // once per device: 128 KiB of LDS for D = 128
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool &raised = raised_on[dev & 63];
    if (lds > 48 * 1024 && !raised) {
        hipError_t e = hipFuncSetAttribute((const void *)tri_inverse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { sx_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        raised = true;
    }
    hipLaunchKernelGGL(tri_inverse_kernel, dim3(batch), dim3(128), lds, sx_stream(stream), T, X, (int)D, (int)lower, (int)unit);
    SX_LAUNCH_CHECK();
    return SX_OK;
}
