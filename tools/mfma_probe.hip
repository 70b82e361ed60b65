// Microbenchmark: what does one wave per SIMD sustain on v_mfma_f32_32x32x2_f32 when (a) chains are
// dependent, (b) A operands come from LDS by ds_read_b128, (c) transcendental VALU work is slotted between
// the MFMAs?  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) float smem[];

template <int CHAINS, int LDS, int VALU>
__global__ __launch_bounds__(256) void probe(float *out, int iters, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) smem[i] = 0.001f * (i & 127);
    __syncthreads();
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    f32x16 b;
    for (int r = 0; r < 16; ++r) b[r] = 0.01f * (lane + r);
    f32x16 side;
    for (int r = 0; r < 16; ++r) side[r] = 0.1f * r + lane * 0.001f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 a;
            if (LDS) a = reinterpret_cast<const f32x4 *>(smem)[((it & 7) * 4 + g) * 64 + lane];
            else a = f32x4{b[g], b[g + 4], b[g + 8], b[g + 12]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[4 * g + e], acc[c], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VALU; ++v) {   // one tanh per v: mul, exp, add, rcp, fma
                    float x = side[(4 * g + e + v) & 15];
                    float ex = __builtin_amdgcn_exp2f(x * 2.885390f);
                    side[(4 * g + e + v) & 15] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(ex + 1.0f);
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int r = 0; r < 16; ++r) s += side[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS, int LDS, int VALU>
void run(const char *name, int blocks_per_cu) {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<CHAINS, LDS, VALU><<<256 * blocks_per_cu, 256, 32768>>>(out, 10, cyc);
    hipEventRecord(e0);
    probe<CHAINS, LDS, VALU><<<256 * blocks_per_cu, 256, 32768>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma = (double)iters * 16 * CHAINS;
    const double tflops = mfma * 4096.0 * 4 * 256 * blocks_per_cu / (ms * 1e-3) / 1e12;
    printf("%-34s blocks/CU=%d  %.1f cycles per MFMA (s_memtime, per wave)  %.1f TFLOP/s  %.3f ms\n", name, blocks_per_cu,
           (double)h / mfma, tflops, ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int b = 1; b <= 2; ++b) {
        run<1, 0, 0>("1 chain, regs", b);
        run<2, 0, 0>("2 chains, regs", b);
        run<1, 1, 0>("1 chain, A from LDS", b);
        run<2, 1, 0>("2 chains, A from LDS", b);
        run<2, 1, 1>("2 chains, LDS, 1 tanh per k-step", b);
        run<2, 1, 2>("2 chains, LDS, 2 tanh per k-step", b);
        run<1, 1, 1>("1 chain, LDS, 1 tanh per k-step", b);
        run<2, 0, 2>("2 chains, regs, 2 tanh per k-step", b);
    }
    return 0;
}
