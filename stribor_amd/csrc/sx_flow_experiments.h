// Timing experiments of the fused flow kernel -- NOT part of the product.  Included by sx_flow_kernel.h only under -DSX_EXPERIMENTS
// (tools/knob_sweep.sh, tools/experiments/cfg4_stamps.sh, the variant builds of tools/experiments/README.md); the shipped library is
// built without it and every hook below is an empty macro there.  Inside namespace SX_PREC_NS, like the rest of the kernel header.
//
//   -DSX_X=<bits>        compile-time ablations that keep the code straight-line (results WRONG, only the time matters):
//                          4 no hidden transcendentals   8 no scale exp2   16 no MFMA (gemm_tile_f)   32 no weight ds_read
//                          64 no fp16 split   128 no per-step wait + barrier   256 MODE 11: the state's loads / stores folded onto 4 MB
//                          512 MODE 11: no weight-gradient contraction (turn + contract)
//   -DSX_DEBUG_KNOBS     run-time ablation bits (environment SX_DBG: 1 no weight re-staging, 2 no per-step wait + barrier, 16 no MFMA) and
//                        the in-kernel phase stamps (SX_PROF=1 prints them for one wave: -DSX_PROF_THREAD=<thread of workgroup 3>)
#pragma once
#ifndef SX_X
#define SX_X 0
#endif
#ifdef SX_DEBUG_KNOBS
__device__ int g_sx_dbg;   // set by the host before launch (hipMemcpyToSymbol)
__device__ __forceinline__ int smem_dbg() { return __builtin_amdgcn_readfirstlane(g_sx_dbg); }
#define SX_DBG(bit) (smem_dbg() & (bit))
// In-kernel phase stamps (MI355X guide, 'In-kernel stamps')
__device__ unsigned long long g_sx_prof[16];
__device__ unsigned long long g_sx_span[1024][2];
struct prof_t {
    unsigned long long acc[16];
    unsigned long long last;
    unsigned long long t0, w0;
};
__device__ __forceinline__ void sx_stamp(prof_t &p, int id) {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    p.acc[id] += t - p.last;
    p.last = t;
}
#define SX_STAMP(p, id) sx_stamp(p, id)
#ifndef SX_PROF_THREAD
#define SX_PROF_THREAD 64       // (wave 1; -DSX_PROF_THREAD=320 stamps its SIMD partner in an 8-wave workgroup)
#endif
#define SX_EXP_KERNEL_BEGIN(pf)                                                                    \
    do {                                                                                           \
        for (int i_ = 0; i_ < 16; ++i_) (pf).acc[i_] = 0;                                          \
        (pf).last = __builtin_amdgcn_s_memtime();                                                  \
        (pf).t0 = (pf).last; (pf).w0 = wall_clock64();   /* wall_clock64: 100 MHz constant clock */ \
    } while (0)
#define SX_EXP_KERNEL_END(pf)                                                                      \
    do {                                                                                           \
        SX_STAMP(pf, 7);             /* epilogue of the last chunk */                              \
        if (blockIdx.x == 3 && threadIdx.x == SX_PROF_THREAD) {                                    \
            for (int i_ = 0; i_ < 8; ++i_) g_sx_prof[i_] = (pf).acc[i_];                           \
            g_sx_prof[8] = __builtin_amdgcn_s_memtime() - (pf).t0;                                 \
            g_sx_prof[9] = wall_clock64() - (pf).w0;                                               \
        }                                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < 1024) {      /* start / end time of every workgroup (100 MHz ticks) */ \
            g_sx_span[blockIdx.x][0] = (pf).w0;                                                    \
            g_sx_span[blockIdx.x][1] = wall_clock64();                                             \
        }                                                                                          \
    } while (0)
static inline void sx_exp_before_launch() {
    const char *e = getenv("SX_DBG");
    int v = e ? atoi(e) : 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sx_dbg), &v, sizeof(int));
    static unsigned long long zero[1024][2];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sx_span), zero, sizeof(zero));
}
template <class ARGS>
static inline void sx_exp_after_launch(const ARGS &a) {
    if (!getenv("SX_PROF")) return;
    (void)hipStreamSynchronize(a.stream);
    unsigned long long p[16];
    (void)hipMemcpyFromSymbol(p, HIP_SYMBOL(g_sx_prof), sizeof(p));
    static const char *names[8] = {"chunk-prologue", "wait+barrier", "desc+dma-issue", "gemm1 / spline block: first tile", "gemm2 / spline block: 3 pairs", "affine / spline block: last element", "step-tail", "epilogue"};
    unsigned long long tot = 0;
    for (int i = 0; i < 8; ++i) tot += p[i];
    fprintf(stderr, "[sx prof] thread %d of block 3, cycles:", (int)SX_PROF_THREAD);
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %s=%llu (%.1f%%)", names[i], p[i], 100.0 * p[i] / (tot ? tot : 1));
    fprintf(stderr, " total=%llu; wave lifetime %.1f us at %.2f GHz (s_memtime / 100 MHz wall clock)\n", tot, p[9] * 0.01,
            p[9] ? (double)p[8] / (p[9] * 10.0) : 0.0);
    static unsigned long long span[1024][2];
    (void)hipMemcpyFromSymbol(span, HIP_SYMBOL(g_sx_span), sizeof(span));
    unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
    int nb = 0;
    for (int b = 0; b < 1024 && b < a.grid; ++b) {
        if (!span[b][1]) continue;
        ++nb;
        s0 = span[b][0] < s0 ? span[b][0] : s0; s1 = span[b][0] > s1 ? span[b][0] : s1;
        e0 = span[b][1] < e0 ? span[b][1] : e0; e1 = span[b][1] > e1 ? span[b][1] : e1;
    }
    fprintf(stderr, "[sx prof] %d workgroups: starts spread %.1f us; first end %.1f us, last end %.1f us after the first start\n",
            nb, (s1 - s0) * 0.01, (e0 - s0) * 0.01, (e1 - s0) * 0.01);
    if (getenv("SX_PROF_DUMP")) {
        for (int b = 0; b < 1024 && b < a.grid; ++b)
            fprintf(stderr, "[sx span] %d %.2f %.2f\n", b, (span[b][0] - s0) * 0.01, (span[b][1] - s0) * 0.01);
    }
}
#define SX_EXP_BEFORE_LAUNCH() sx_exp_before_launch()
#define SX_EXP_AFTER_LAUNCH(a) sx_exp_after_launch(a)
#else   // compile-time ablations only
#define SX_DBG(bit) 0
struct prof_t {};
#define SX_STAMP(p, id) ((void)0)
#define SX_EXP_KERNEL_BEGIN(pf) ((void)0)
#define SX_EXP_KERNEL_END(pf) ((void)0)
#define SX_EXP_BEFORE_LAUNCH() ((void)0)
#define SX_EXP_AFTER_LAUNCH(a) ((void)0)
#endif
