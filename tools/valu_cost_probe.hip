// Issue cost of the VALU instructions the fused flow kernel uses, alone and beside v_mfma_f32_32x32x16_f16
// (one wave per SIMD, independent instructions, inline asm so the compiler cannot re-select them).
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_cost_probe.hip -o tools/valu_cost_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define REP8(X) X X X X X X X X
// FILL(i): one filler instruction on registers that nothing else depends on
#define DEF_KERNEL(NAME, ASM, NFILL)                                                                               \
    template <int MFMA>                                                                                            \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, unsigned long long *cyc) {                  \
        float a0 = threadIdx.x * 0.001f + 0.5f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;                       \
        float d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;                                      \
        f32x16 acc;                                                                                                \
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;                                                                 \
        h8 ha, hb;                                                                                                 \
        for (int r = 0; r < 8; ++r) { ha[r] = (_Float16)(0.01f * r); hb[r] = (_Float16)(0.02f * r + threadIdx.x * 0.001f); } \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                      \
        for (int it = 0; it < iters; ++it) {                                                                       \
            if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);                          \
            if (NFILL >= 1) asm volatile(ASM : "+v"(d0) : "v"(a0), "v"(a1));                                       \
            if (NFILL >= 2) asm volatile(ASM : "+v"(d1) : "v"(a1), "v"(a2));                                       \
            if (NFILL >= 3) asm volatile(ASM : "+v"(d2) : "v"(a2), "v"(a3));                                       \
            if (NFILL >= 4) asm volatile(ASM : "+v"(d3) : "v"(a3), "v"(a0));                                       \
            if (NFILL >= 5) asm volatile(ASM : "+v"(d4) : "v"(a0), "v"(a2));                                       \
            if (NFILL >= 6) asm volatile(ASM : "+v"(d5) : "v"(a1), "v"(a3));                                       \
            if (NFILL >= 7) asm volatile(ASM : "+v"(d6) : "v"(a2), "v"(a0));                                       \
            if (NFILL >= 8) asm volatile(ASM : "+v"(d7) : "v"(a3), "v"(a1));                                       \
        }                                                                                                          \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                      \
        float s = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;                                                           \
        for (int r = 0; r < 16; ++r) s += acc[r];                                                                  \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                                   \
        if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;                                                   \
    }

#define K(NAME, ASM) DEF_KERNEL(NAME##_4, ASM, 4) DEF_KERNEL(NAME##_8, ASM, 8)
K(k_add, "v_add_f32 %0, %1, %2")
K(k_exp, "v_exp_f32 %0, %1")
K(k_rcp, "v_rcp_f32 %0, %1")
K(k_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %1, %2")
K(k_mixlo, "v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]")
K(k_mixhi, "v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
K(k_mix32, "v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]")
K(k_cvt16, "v_cvt_f32_f16 %0, %1")
K(k_cvt16s, "v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
K(k_and, "v_and_b32 %0, %1, %2")
K(k_fma, "v_fma_f32 %0, %1, %2, %0")
K(k_perm, "v_perm_b32 %0, %1, %2, %1")
K(k_cvtf16, "v_cvt_f16_f32 %0, %1")
K(k_pack, "v_pack_b32_f16 %0, %1, %2")
DEF_KERNEL(k_none_0, "", 0)

template <class KT>
double run(KT kern, int iters) {
    float *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    (void)hipMalloc(&cyc, 8);
    kern<<<256, 256>>>(out, 10, cyc);
    kern<<<256, 256>>>(out, iters, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    (void)hipFree(out); (void)hipFree(cyc);
    return (double)h / iters;
}

#define REPORT(NAME)                                                                                               \
    {                                                                                                              \
        const double a4 = run(NAME##_4<0>, N), a8 = run(NAME##_8<0>, N), m4 = run(NAME##_4<1>, N), m8 = run(NAME##_8<1>, N); \
        printf("%-10s alone: %.1f cyc/instr   beside one MFMA per gap: 4 fillers %.1f cyc/gap, 8 fillers %.1f cyc/gap\n", #NAME, \
               (a8 - a4) / 4.0, m4, m8);                                                                           \
    }

int main() {
    const int N = 20000;
    printf("bare MFMA gap: %.1f cyc\n", run(k_none_0<1>, N));
    REPORT(k_add) REPORT(k_fma) REPORT(k_and) REPORT(k_exp) REPORT(k_rcp) REPORT(k_pkrtz) REPORT(k_mixlo) REPORT(k_mixhi)
    REPORT(k_mix32) REPORT(k_cvt16) REPORT(k_cvt16s) REPORT(k_perm) REPORT(k_cvtf16) REPORT(k_pack)
    return 0;
}
