// Microbenchmark for the fp16 x 3 split GEMM of the fused flow kernel: what does a SIMD sustain on
// v_mfma_f32_32x32x16_f16 accumulation chains (6 per 32-deep tile) when (a) 1, 2 or 3 waves share the SIMD,
// (b) the A fragments come from LDS by ds_read_b128, (c) the kernel's VALU work (exp2, rcp, add, the fp16
// hi/lo split) is issued beside the MFMAs, (d) the next tile's B operand is made from this tile's result?
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f16_probe.hip -o tools/mfma_f16_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

__device__ __forceinline__ uint32_t pk_rtz(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ uint32_t pk_residual(uint32_t ph, float v0, float v1) {
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(ph), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(ph), "v"(v1));
    return l;
}

// one "tile" = 6 MFMAs on one accumulator (2 k16-steps x {lo.hi, hi.lo, hi.hi}).
// LDS: A from LDS;  TRANS: exp2+add+rcp on TRANS registers of a side tile per tile;  SPLIT: fp16 split of SPLIT pairs
// per tile;  DEP: the B operand of the next tile is the split of this tile's accumulator (serial dependence).
template <int LDS, int TRANS, int SPLIT, int DEP>
__global__ __launch_bounds__(256) void probe(float *out, int iters, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) smem[i] = 0x3c003c00u;   // fp16 1.0 pairs
    __syncthreads();
    f32x16 acc, side;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; side[r] = 0.1f * r + lane * 0.001f; }
    u32x4 bhi[2], blo[2];
    for (int s = 0; s < 2; ++s)
        for (int q = 0; q < 4; ++q) { bhi[s][q] = 0x38003800u + lane; blo[s][q] = 0x10001000u + q; }
    const u32x4 *lp = reinterpret_cast<const u32x4 *>(smem) + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 ahu, alu;
            if (LDS) { ahu = lp[((it & 3) * 4 + 2 * s) * 64]; alu = lp[((it & 3) * 4 + 2 * s + 1) * 64]; }
            else { ahu = bhi[s] + 1u; alu = blo[s] + 1u; }
            const h8 ah = __builtin_bit_cast(h8, ahu), al = __builtin_bit_cast(h8, alu);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(h8, bhi[s]), acc, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < TRANS / 2; ++v) {
                const int r = (s * (TRANS / 2) + v) & 15;
                side[r] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(side[r]) + 1.0f);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(h8, blo[s]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(h8, bhi[s]), acc, 0, 0, 0);
        }
        if (SPLIT) {
            const f32x16 &src = DEP ? acc : side;
#pragma unroll
            for (int p = 0; p < SPLIT; ++p) {
                const uint32_t ph = pk_rtz(src[(2 * p) & 15], src[(2 * p + 1) & 15]);
                bhi[(p >> 2) & 1][p & 3] = ph;
                blo[(p >> 2) & 1][p & 3] = pk_residual(ph, src[(2 * p) & 15], src[(2 * p + 1) & 15]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float sm = 0.f;
    for (int r = 0; r < 16; ++r) sm += acc[r] + side[r];
    out[blockIdx.x * 256 + threadIdx.x] = sm + (float)(bhi[0][0] + blo[1][3]);
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

template <int LDS, int TRANS, int SPLIT, int DEP>
void run(const char *name, int blocks_per_cu) {
    float *out; unsigned long long *cyc, h[2];
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&cyc, 16);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int lds = 50 * 1024;     // 3 workgroups per CU at most, like the flow kernel
    hipFuncSetAttribute((const void *)probe<LDS, TRANS, SPLIT, DEP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    probe<LDS, TRANS, SPLIT, DEP><<<256 * blocks_per_cu, 256, lds>>>(out, 10, cyc);
    hipEventRecord(e0);
    probe<LDS, TRANS, SPLIT, DEP><<<256 * blocks_per_cu, 256, lds>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const double tiles_per_simd = (double)iters * blocks_per_cu;       // one wave per SIMD per block
    const double tflops = (double)iters * 6 * 32768.0 * 4 * 256 * blocks_per_cu / (ms * 1e-3) / 1e12;
    printf("%-44s waves/SIMD=%d  wave: %.0f cyc/tile  SIMD: %.1f ns/tile  %.0f TFLOP/s executed  clock %.2f GHz\n", name, blocks_per_cu,
           (double)h[0] / iters, ms * 1e6 / tiles_per_simd, tflops, (double)h[0] / (h[1] * 10.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int b = 1; b <= 3; ++b) {
        run<0, 0, 0, 0>("MFMA chain, regs", b);
        run<1, 0, 0, 0>("MFMA chain, A from LDS", b);
        run<1, 8, 0, 0>("+ 8 (exp2,add,rcp) per tile", b);
        run<1, 16, 0, 0>("+ 16 (exp2,add,rcp) per tile", b);
        run<1, 0, 8, 0>("+ split of 8 pairs (independent)", b);
        run<1, 0, 8, 1>("+ split of 8 pairs of the accumulator (dep)", b);
        run<1, 8, 8, 1>("+ 8 trans + dependent split", b);
        run<1, 16, 8, 1>("+ 16 trans + dependent split", b);
    }
    return 0;
}
