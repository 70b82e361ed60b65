// does the 16x16x32 f16 MFMA shape keep its bare-loop lead (+17 .. 20 %: tools/mfma_shape_probe.hip) once the
// instruction stream looks like the cfg-4 kernel's -- ~5 vector instructions per 32x32x16 MFMA riding between the MFMAs (fp16 split:
// v_cvt_pk_f16_f32 / v_fma_mix_f32 / v_max3_f32; sigmoid: v_exp_f32 / v_rcp_f32 / v_add_f32), the A fragments of every product
// re-read from LDS by ds_read_b128 one k-step ahead, two waves per SIMD?  The same arithmetic per iteration in both shapes: one
// 32 x 32 output tile += A[32][32] B[32][32] in three fp16 products = 6 v_mfma_f32_32x32x16_f16 (one accumulator) or
// 12 v_mfma_f32_16x16x32_f16 (four 16 x 16 accumulators), 30 riders and 4 fragment loads either way, RANDOM operands.
// Proceed with a 16-column fragment layout for MODE 7 / 8 only if the 16x16x32 variant is >= 8 % faster per iteration (wall time).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_riders_probe.hip -o tools/mfma_shape_riders_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t rnd_h2(uint32_t s) {
    const uint32_t r = hash(s);
    auto one = [](uint32_t v) { return ((v & 1u) << 15) | ((12u + ((v >> 1) & 3u)) << 10) | ((v >> 3) & 0x3ffu); };
    return one(r) | (one(r >> 13) << 16);
}
// five riders behind one 32x32x16 MFMA (a split pair = cvt_pk + 2 fma_mix + cvt_pk + max3 every other gap, a sigmoid exp/add/rcp in
// the others): KIND alternates so that an iteration carries 12 cvt_pk, 12 fma_mix, 6 max3 -- and 3 exp, 3 add, 3 rcp -- = 30 + 9
#define SPLIT5(v0, v1, ph, pl, l0, l1, mx)                                                         \
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(v0), "v"(v1));                      \
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(ph), "v"(v0)); \
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(ph), "v"(v1)); \
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pl) : "v"(l0), "v"(l1));                      \
    asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(v0), "v"(v1));
#define SIG3(u, e)                                                         \
    asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(u));                   \
    asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e));                       \
    asm volatile("v_rcp_f32 %0, %1" : "=v"(u) : "v"(e));

extern __shared__ uint32_t lds[];

template <int SHAPE, int RIDERS>      // SHAPE 0: 32x32x16, 1: 16x16x32; RIDERS 0: bare, 1: the cfg-4 mix
__global__ __launch_bounds__(512) void probe(float *out, int iters, unsigned long long *cyc) {
    const uint32_t lane = threadIdx.x & 63, gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = threadIdx.x; i < 4 * 4 * 256; i += blockDim.x) lds[i] = rnd_h2(i * 7 + 3);      // four A fragments x two sets, 4 KB each
    u32x4 b[2][2];
    for (int s = 0; s < 2; ++s)
        for (int p = 0; p < 2; ++p)
            for (int q = 0; q < 4; ++q) b[s][p][q] = rnd_h2(gid * 64 + 32 + s * 16 + p * 8 + q);
    f32x16 acc; f32x4 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    c0 = c1 = c2 = c3 = f32x4{0.f, 0.f, 0.f, 0.f};
    float v0 = 0.37f + lane * 0.01f, v1 = -1.21f + lane * 0.02f, l0, l1, mx = 0.f, u = 0.3f, e;
    uint32_t ph, pl;
    __syncthreads();
    const u32x4 *fr = reinterpret_cast<const u32x4 *>(lds) + lane;
    u32x4 ah = fr[0], al = fr[64];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // the NEXT k-step's A fragments (hi, lo) are requested before this step's MFMAs go out
            const u32x4 nh = fr[(2 * (1 - s)) * 64 + 128 * (it & 1)], nl = fr[(2 * (1 - s) + 1) * 64 + 128 * (it & 1)];
            const h8 hah = __builtin_bit_cast(h8, ah), hal = __builtin_bit_cast(h8, al);
            const h8 bh = __builtin_bit_cast(h8, b[s][0]), bl = __builtin_bit_cast(h8, b[s][1]);
            if (SHAPE == 0) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hal, bh, acc, 0, 0, 0);
                if (RIDERS) { SPLIT5(v0, v1, ph, pl, l0, l1, mx) }
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hah, bl, acc, 0, 0, 0);
                if (RIDERS) { SPLIT5(v1, v0, ph, pl, l0, l1, mx) }
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hah, bh, acc, 0, 0, 0);
                if (RIDERS) { SIG3(u, e) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(u), "v"(v0)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v0) : "v"(u), "v"(v1)); }
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // six instructions of half the flops each per k-half: the riders of one 32x32x16 gap spread over two 16x16x32 gaps
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hal, bh, c0, 0, 0, 0);
                if (RIDERS) { asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(v0), "v"(v1)); asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(ph), "v"(v0)); asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(ph), "v"(v1)); }
                __builtin_amdgcn_sched_barrier(0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hal, bh, c1, 0, 0, 0);
                if (RIDERS) { asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pl) : "v"(l0), "v"(l1)); asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(v0), "v"(v1)); }
                __builtin_amdgcn_sched_barrier(0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hah, bl, c0, 0, 0, 0);
                if (RIDERS) { asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(v1), "v"(v0)); asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(ph), "v"(v1)); asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(ph), "v"(v0)); }
                __builtin_amdgcn_sched_barrier(0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hah, bl, c1, 0, 0, 0);
                if (RIDERS) { asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pl) : "v"(l0), "v"(l1)); asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(v1), "v"(v0)); }
                __builtin_amdgcn_sched_barrier(0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hah, bh, c2, 0, 0, 0);
                if (RIDERS) { SIG3(u, e) }
                __builtin_amdgcn_sched_barrier(0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hah, bh, c3, 0, 0, 0);
                if (RIDERS) { asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(u), "v"(v0)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v0) : "v"(u), "v"(v1)); }
                __builtin_amdgcn_sched_barrier(0);
            }
            ah = nh; al = nl;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float sm = mx + v0 + v1 + u + __uint_as_float(ph ^ pl);
    for (int r = 0; r < 16; ++r) sm += acc[r];
    for (int r = 0; r < 4; ++r) sm += c0[r] + c1[r] + c2[r] + c3[r];
    out[gid] = sm;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

template <int SHAPE, int RIDERS>
void run(const char *name, int threads) {
    float *out; unsigned long long *cyc, h[2];
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&cyc, 16);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<SHAPE, RIDERS><<<256, threads, 16384>>>(out, 100, cyc);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        probe<SHAPE, RIDERS><<<256, threads, 16384>>>(out, iters, cyc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const double flops = (double)iters * 196608.0 * (threads / 64) * 256;
    printf("%-28s %-18s waves/SIMD %d: %6.1f cycles / iteration  %7.3f ms  %6.0f TFLOP/s executed  clock %.2f GHz\n", name, RIDERS ? "cfg-4 riders + LDS" : "LDS reads only",
           threads / 256, (double)h[0] / iters, best, flops / (best * 1e-3) / 1e12, h[1] ? (double)h[0] / (h[1] * 10.0) : 0.0);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    for (int rep = 0; rep < 2; ++rep)
        for (int threads = 256; threads <= 512; threads += 256) {
            run<0, 0>("32x32x16_f16 x 6", threads);
            run<1, 0>("16x16x32_f16 x 12", threads);
            run<0, 1>("32x32x16_f16 x 6", threads);
            run<1, 1>("16x16x32_f16 x 12", threads);
        }
    return 0;
}
