// VERDICT r4 #9 (experiment): would the 16x16x32 f16 MFMA shape pay for the fp16 x 3 split GEMMs of the fused kernels?
// MI355X_MICROARCH.md ('DVFS give-back' item 7): in bare bf16 loops on random data the 16x16x32 shape delivered ~1.15x the FLOP/s of
// 32x32x16 at about equal cycles per FLOP (the chip holds a higher clock on it).  Here: the same product -- one 32 x 32 output tile
// (x 32 samples... per wave: C[32][32] += A[32][32] B[32][32], three fp16 products: lo.hi, hi.lo, hi.hi) -- as 6 v_mfma_f32_32x32x16_f16
// on one accumulator or as 12 v_mfma_f32_16x16x32_f16 on four 16 x 16 accumulators, RANDOM operands (two operand sets alternating, so
// the inputs toggle between consecutive instructions), operands in registers, 1 and 2 waves per SIMD, every CU busy.
// Adopt a new fragment layout only if this shows >= 10 % at the same arithmetic; otherwise record the number.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o tools/mfma_shape_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// a random fp16 pair with exponents in [2^-3, 2^1): bit pattern sign | exp 12..15 | mantissa
__device__ __forceinline__ uint32_t rnd_h2(uint32_t s) {
    const uint32_t r = hash(s);
    auto one = [](uint32_t v) { return ((v & 1u) << 15) | ((12u + ((v >> 1) & 3u)) << 10) | ((v >> 3) & 0x3ffu); };
    return one(r) | (one(r >> 13) << 16);
}

template <int SHAPE>      // 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(512) void probe(float *out, int iters, unsigned long long *cyc) {
    const uint32_t lane = threadIdx.x & 63, gid = blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 a[2][2], b[2][2];       // [operand set][hi / lo]
    for (int s = 0; s < 2; ++s)
        for (int p = 0; p < 2; ++p)
            for (int q = 0; q < 4; ++q) { a[s][p][q] = rnd_h2(gid * 64 + s * 16 + p * 8 + q); b[s][p][q] = rnd_h2(gid * 64 + 32 + s * 16 + p * 8 + q); }
    f32x16 acc; f32x4 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    c0 = c1 = c2 = c3 = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {           // two k16 (k32) halves of a 32-deep tile = the two operand sets
            const h8 ah = __builtin_bit_cast(h8, a[s][0]), al = __builtin_bit_cast(h8, a[s][1]);
            const h8 bh = __builtin_bit_cast(h8, b[s][0]), bl = __builtin_bit_cast(h8, b[s][1]);
            if (SHAPE == 0) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            } else {
                // the same flops: a 32 x 32 x 16 product = four 16 x 16 blocks x (k = 16) ... as 16x16x32 instructions the k-halves of
                // two such steps merge: per `s` two blocks' worth -- six instructions of half the flops each
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c3, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float sm = 0.f;
    for (int r = 0; r < 16; ++r) sm += acc[r];
    for (int r = 0; r < 4; ++r) sm += c0[r] + c1[r] + c2[r] + c3[r];
    out[gid] = sm;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

template <int SHAPE>
void run(const char *name, int threads) {
    float *out; unsigned long long *cyc, h[2];
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&cyc, 16);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<SHAPE><<<256, threads>>>(out, 100, cyc);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        probe<SHAPE><<<256, threads>>>(out, iters, cyc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    // flops per iteration and wave: SHAPE 0: 6 x (2 x 32 x 32 x 16); SHAPE 1: 12 x (2 x 16 x 16 x 32) -- both 196,608
    const double flops = (double)iters * (SHAPE == 0 ? 6.0 * 2 * 32 * 32 * 16 : 12.0 * 2 * 16 * 16 * 32) * (threads / 64) * 256;
    printf("%-34s waves/SIMD %d: %6.1f cycles / iteration  %7.3f ms  %6.0f TFLOP/s executed  clock %.2f GHz\n", name, threads / 256, (double)h[0] / iters, best,
           flops / (best * 1e-3) / 1e12, h[1] ? (double)h[0] / (h[1] * 10.0) : 0.0);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    for (int rep = 0; rep < 2; ++rep)
        for (int threads = 256; threads <= 512; threads += 256) {
            run<0>("v_mfma_f32_32x32x16_f16 x 6", threads);
            run<1>("v_mfma_f32_16x16x32_f16 x 12", threads);
        }
    return 0;
}
