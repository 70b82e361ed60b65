// Microbenchmark: how do TWO waves of one SIMD share the vector and the matrix pipe when each wave's stream is phases of pure VALU
// work (the fp16 split of the cfg-4 kernel: cvt_pkrtz + v_fma_mix + max3) followed by phases of dependent MFMAs
// (v_mfma_f32_32x32x16_f16, one accumulation chain), with a workgroup barrier per step -- the structure of flow_fused_kernel<1,4,2,7>?
//   variant 0  lockstep: every wave [VALU block][MFMA block] barrier
//   variant 1  staggered: waves 4..7 run [MFMA block][VALU block] between the same barriers (their phases are the other half's complement)
//   variant 2  variant 1 + s_setprio 1 around the MFMA block
//   variant 3  lockstep, finely interleaved stream: the VALU block's instructions spread between the MFMAs
//   variant 4  variant 0 without barriers        variant 5  variant 1 without barriers
//   variant 6  variant 3 staggered by half a block  variant 7  lockstep + s_setprio 1 around the MFMA block
//   variant 8  ONE wave per SIMD (4-wave workgroups), twice the work per wave, interleaved stream of two independent chains
// NV = VALU instructions per step and wave (split pairs x 5), NM = MFMAs per step and wave.
// Build: hipcc -O3 --offload-arch=gfx950 tools/coexec_probe.hip -o tools/coexec_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pk_rtz(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ uint32_t pk_residual(uint32_t ph, float v0, float v1) {
    float l0, l1;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(ph), "v"(v0));
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(ph), "v"(v1));
    return pk_rtz(l0, l1);
}
// one split "pair": 5 vector instructions (cvt_pkrtz, 2 fma_mix, cvt_pkrtz, max3)
__device__ __forceinline__ void split_pair(float &v0, float &v1, uint32_t &hi, uint32_t &lo, float &mx) {
    asm volatile("" : "+v"(v0), "+v"(v1));
    hi = pk_rtz(v0, v1);
    lo = pk_residual(hi, v0, v1);
    mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1)));
    asm volatile("" : "+v"(hi), "+v"(lo), "+v"(mx));
}

template <int VARIANT, int PAIRS, int NM>
__global__ __launch_bounds__(512) void probe(float *out, int iters, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr bool ONE = VARIANT == 8;
    const bool late = !ONE && wave >= 4;
    constexpr bool BARRIER = VARIANT != 4 && VARIANT != 5;
    constexpr bool STAG = VARIANT == 1 || VARIANT == 2 || VARIANT == 5 || VARIANT == 6;
    constexpr bool PRIO = VARIANT == 2 || VARIANT == 7;
    constexpr bool FINE = VARIANT == 3 || VARIANT == 6 || VARIANT == 8;
    f32x16 acc, acc2;
    float src[32];
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    for (int r = 0; r < 32; ++r) src[r] = 0.37f * r + lane * 0.011f;
    u32x4 bhi = {0x38003800u + lane, 0x38013801u, 0x38023802u, 0x38033803u}, blo = {0x10001000u, 0x10011001u + lane, 0x10021002u, 0x10031003u};
    u32x4 ahi = {0x3c003c00u, 0x3c013c01u + lane, 0x3c023c02u, 0x3c033c03u}, alo = {0x0c000c00u + lane, 0x0c010c01u, 0x0c020c02u, 0x0c030c03u};
    float mx = 0.f;
    uint32_t hsum = 0;
    auto valu_block = [&](int n_pairs) {
#pragma unroll
        for (int p = 0; p < n_pairs; ++p) {
            uint32_t hi, lo;
            split_pair(src[(2 * p) & 31], src[(2 * p + 1) & 31], hi, lo, mx);
            hsum ^= hi + lo;
        }
    };
    auto mfma_block = [&](int n, f32x16 &a) {
#pragma unroll
        for (int m = 0; m < n; ++m)
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (m & 1) ? ahi : alo), __builtin_bit_cast(h8, (m & 2) ? bhi : blo), a, 0, 0, 0);
    };
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (FINE) {
            // PAIRS pairs spread over NM MFMAs (ONE: twice of both, two independent accumulation chains)
            constexpr int REP = ONE ? 2 : 1;
            if (STAG && late) { mfma_block(NM / 2, acc); }
#pragma unroll
            for (int m = 0; m < NM * REP; ++m) {
                if (ONE && (m & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, ahi), __builtin_bit_cast(h8, blo), acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (m & 1) ? ahi : alo), __builtin_bit_cast(h8, (m & 2) ? bhi : blo), acc, 0, 0, 0);
                const int p0 = m * PAIRS / NM, p1 = (m + 1) * PAIRS / NM;
#pragma unroll
                for (int p = p0; p < p1; ++p) {
                    uint32_t hi, lo;
                    split_pair(src[(2 * p) & 31], src[(2 * p + 1) & 31], hi, lo, mx);
                    hsum ^= hi + lo;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (STAG && late) {
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            mfma_block(NM, acc);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            valu_block(PAIRS);
        } else {
            valu_block(PAIRS);
            __builtin_amdgcn_sched_barrier(0);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            mfma_block(NM, acc);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float sm = mx + (float)hsum;
    for (int r = 0; r < 16; ++r) sm += acc[r] + acc2[r];
    for (int r = 0; r < 32; ++r) sm += src[r];
    out[blockIdx.x * 512 + threadIdx.x] = sm;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) { cyc[2 * (threadIdx.x >> 8)] = t1 - t0; cyc[2 * (threadIdx.x >> 8) + 1] = w1 - w0; }
}

template <int VARIANT, int PAIRS, int NM>
void run(const char *name) {
    float *out; unsigned long long *cyc, h[4] = {0, 0, 0, 0};
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, 32);
    hipMemset(cyc, 0, 32);
    const int iters = 2000;
    const int threads = VARIANT == 8 ? 256 : 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<VARIANT, PAIRS, NM><<<256, threads>>>(out, 10, cyc);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        probe<VARIANT, PAIRS, NM><<<256, threads>>>(out, iters, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    const double per_it = (double)h[0] / iters;
    const int work = VARIANT == 8 ? 2 : 2;       // MFMA blocks per SIMD and iteration (two waves, or one wave with twice the work)
    printf("%-58s PAIRS %3d NM %3d: %7.0f cycles / iteration (wave 0; wave 4: %7.0f)  MFMA floor %5d  VALU-issue floor ~%5d   %.3f ms  %.2f GHz\n", name, PAIRS, NM,
           per_it, (double)h[2] / iters, work * NM * 32, work * (PAIRS * 5 * 4 + NM * 8), best, h[1] ? (double)h[0] / (h[1] * 10.0) : 0.0);
    hipFree(out); hipFree(cyc);
}

int main() {
#define ALL(P, M)                                                                      \
    run<0, P, M>("0 lockstep [VALU][MFMA] barrier");                                   \
    run<1, P, M>("1 staggered: waves 4-7 [MFMA][VALU]");                               \
    run<2, P, M>("2 staggered + setprio 1 around MFMA");                               \
    run<7, P, M>("7 lockstep + setprio 1 around MFMA");                                \
    run<3, P, M>("3 lockstep, VALU spread between the MFMAs");                         \
    run<6, P, M>("6 spread + waves 4-7 half a block behind");                          \
    run<4, P, M>("4 as 0, no barrier");                                                \
    run<5, P, M>("5 as 1, no barrier");                                                \
    run<8, P, M>("8 ONE wave per SIMD, two chains, spread");
    ALL(32, 48)      // a dense half-layer: 4 tiles x 8 pairs, 48 MFMAs
    ALL(64, 72)      // a coupling: ~350 vector instructions, 72 MFMAs
    ALL(32, 24)
    return 0;
}
