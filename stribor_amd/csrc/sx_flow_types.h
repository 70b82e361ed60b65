// Types shared by the fused-flow host dispatcher and its per-(tiles, hidden-tiles) kernel objects.
#pragma once
#include "sx_common.h"

// compact device-side copy of sx_step / sx_program (kernarg)
struct dstep {
    uint8_t kind, c0, ct, t0, tt, reverse, act, pad;
    uint32_t blob_off, blob_floats;
    float ldj_scale, ldj_const;
    uint32_t mask;      // RQS phases: live slots of the transformed tile
};
struct dprog {
    int32_t n_steps, dim, latent_dim, x_tiles, identity_cols, pad;
    dstep steps[SX_MAX_STEPS];
};


struct sx_flow_args {
    dprog prog;
    const float *blobs; const void *x; const float *latent; const int32_t *in_col; const int32_t *out_col;
    void *y; float *ldj_out; float *logp_out; double *sum_out; float *mlp_out; int64_t mlp_out_stride;
    int mlp_out_dim; int64_t n_rows; int buf_floats; int bf16; int mlp_mode; int grid; int lds; hipStream_t stream;
    const float *row_t;
    float *side;
    int side_width;
    uint32_t *work;
    uint32_t *flags;
    const float *frag_in;
    float *frag_out;
    float *acc_out;
    uint32_t *redo;         // sx_flow_run2: the redo list (sx_flow_kernel.h, flow_kargs), or NULL
    int redo_pass;          // 1: this launch is the exact pass over the listed groups
};


// Layers per launch of the training backward with in-kernel weight-gradient contraction (MODE 11): every layer keeps
// 6 accumulator tiles (96 registers) alive over the whole launch.  With 2 the kernel needs 553 registers (984 B of
// scratch per lane on top of the 512-register file); with 1 it fits (457, no scratch).
#ifndef SX_BWD_SLOTS
#define SX_BWD_SLOTS 1
#endif

// sample tiles (of 32 rows) per wave: 2 while the state fits the register file twice over, else 1
// (measured on cfg 2, MI355X: 2 sample tiles per wave buy nothing over 1 at 2 workgroups per CU -- 1.87e9 rows/s
// either way -- and cost registers, so 1 is the default; -DSX_NS_OVERRIDE=2 rebuilds the 64-rows-per-wave form)
#ifdef SX_NS_OVERRIDE
#define SX_NS_FOR(TX) (SX_NS_OVERRIDE)
#else
#define SX_NS_FOR(TX) 1
#endif

// waves per SIMD the kernel is compiled for (= workgroups per CU): the D = 128 state (+ a second copy for the
// dense linear layers) needs the whole 512-register file; D <= 64 coupling flows run 3 per SIMD (168 VGPRs; measured
// +7 % over 2 on cfg 2: the third wave fills issue slots the other two leave at s_waitcnt / barriers)
#ifndef SX_RQS_WAVES
#define SX_RQS_WAVES 2
#endif
// MODE 18 / 19: MODE 3 / 12 with the training forward's side outputs (same occupancy)
#define SX_MODE_BASE(MODE) ((MODE) == 18 ? 3 : ((MODE) == 19 ? 12 : (MODE)))
// waves per SIMD of the 128-column linear / split-coupling kernels (MODE 7 / 8 on four tiles): one workgroup of 4 x this per CU
#ifndef SX_M7_WAVES
#define SX_M7_WAVES 2
#endif
// MODE 7 / 8 on four tiles: who issues a step's weight DMA, and when.  0: every wave, behind the step's barrier (as everywhere else);
// 1: waves 0..3 only, behind the barrier; 2: waves 0..3 only, at the END of their step -- the older half of an 8-wave workgroup wins the
// SIMD's issue arbitration, finishes a step at ~65 % of its duration and waits at the next barrier: the refill then costs the younger
// half (the critical path) nothing
#ifndef SX_M7_DMA_WHO
#define SX_M7_DMA_WHO 2
#endif
#ifndef SX_WAVES_FOR
#define SX_WAVES_FOR(TX, MODE) SX_WAVES_FOR_(TX, SX_MODE_BASE(MODE))
#define SX_WAVES_FOR_(TX, MODE) ((MODE) == 11 || (MODE) == 14 ? 1 : (MODE) == 20 ? ((TX) >= 4 ? 1 : 2) : ((MODE) == 15 || (MODE) == 16 || (MODE) == 17) ? ((TX) >= 4 ? 1 : 2) : ((MODE) == 10 || (MODE) == 12 || (MODE) == 13) ? ((TX) >= 4 ? 1 : SX_RQS_WAVES) : (MODE) == 9 ? ((TX) >= 4 ? 1 : 2) : (MODE) >= 7 ? ((TX) >= 4 ? SX_M7_WAVES : 2) : (MODE) >= 5 ? ((TX) >= 4 ? 2 : 4) : ((TX) >= 4 ? 1 : ((MODE) == 3 ? SX_RQS_WAVES : ((MODE) == 0 ? 3 : 2))))
#endif

// waves per workgroup: the pure split-coupling kernels (MODE 5 / 6) run 8-wave workgroups -- D <= 64 (128 VGPRs): two
// per CU = 4 waves per SIMD sharing two weight rings; D = 128 (216 VGPRs): one per CU = 2 waves per SIMD on one ring
#ifndef SX_BLOCK_WAVES
#define SX_BLOCK_WAVES(TX, MODE) ((((MODE) == 7 || (MODE) == 8) && (TX) >= 4) ? 4 * SX_M7_WAVES : ((MODE) == 5 || (MODE) == 6) ? 8 : 4)
#endif
// workgroups per CU the kernel is compiled for
#define SX_BLOCKS_FOR(TX, MODE) (SX_WAVES_FOR(TX, MODE) * 4 / SX_BLOCK_WAVES(TX, MODE))

// Kernel-MODE families, one object per (tiles, hidden tiles, arithmetic, family) -- VERDICT r4 #8c: every object used to instantiate
// all twenty MODEs of its pair, so a one-line spline edit rebuilt everything:
//   0  affine / dense / MLP / time-conditioned inference programs (MODE 0, 1, 2, 5 .. 9, 15, 20)
//   1  spline, cubic-spline and mixed programs (MODE 3, 10, 12, 13, 14, 16 .. 19): the only objects that depend on sx_flow_spline.h
//      and sx_cubic_core.h (round 6, ADVICE r5: MODE 10 -- rational-quadratic couplings behind deep conditioners -- sat in family 0,
//      whose prerequisite list has neither header: a spline edit left stale MODE 10 kernels behind an up-to-date build id.  The
//      lists are checked, not trusted: `make depcheck` recompiles one family-0 and one family-2 object with a changed constant in
//      every header their lists omit and compares the ISA)
//   2  training backward programs (MODE 4, 11): the only objects that depend on sx_flow_bwd.h
#define SX_MODE_FAMILY(MODE) ((MODE) == 4 || (MODE) == 11 ? 2 : ((MODE) == 3 || (MODE) == 10 || (MODE) == 12 || (MODE) == 13 || (MODE) == 14 || ((MODE) >= 16 && (MODE) <= 19)) ? 1 : 0)
// `make depcheck` (Makefile): -DSX_DEPCHECK=<family> plants an instruction in every entry point of the headers that family's FASTDEPS
// prerequisite list omits -- the kernels call into sx_flow_spline.h (which is where sx_cubic_core.h is used) and sx_flow_bwd.h only
// through rqs_phase / rqs_triple / cubic_phase / cubic_triple and coupling_affine_bwd* / linear_bwd_half / wacc_zero
#if defined(SX_DEPCHECK) && (SX_DEPCHECK == 0 || SX_DEPCHECK == 2)
#define SX_DEP_MARK_SPLINE asm volatile("s_nop 3")
#else
#define SX_DEP_MARK_SPLINE ((void)0)
#endif
#if defined(SX_DEPCHECK) && (SX_DEPCHECK == 0 || SX_DEPCHECK == 1)
#define SX_DEP_MARK_BWD asm volatile("s_nop 3")
#else
#define SX_DEP_MARK_BWD ((void)0)
#endif
#define SX_DECL_FLOW_F(T, H, F) int sx_flow_launch_f16x3_t##T##h##H##_f##F(const sx_flow_args &a); int sx_flow_launch_f32x_t##T##h##H##_f##F(const sx_flow_args &a);
#define SX_DECL_FLOW(T, H) SX_DECL_FLOW_F(T, H, 0) SX_DECL_FLOW_F(T, H, 1) SX_DECL_FLOW_F(T, H, 2)
SX_DECL_FLOW(1, 1) SX_DECL_FLOW(1, 2) SX_DECL_FLOW(1, 4)
SX_DECL_FLOW(2, 1) SX_DECL_FLOW(2, 2) SX_DECL_FLOW(2, 4)
SX_DECL_FLOW(4, 1) SX_DECL_FLOW(4, 2) SX_DECL_FLOW(4, 4)
SX_DECL_FLOW(8, 1) SX_DECL_FLOW(8, 2) SX_DECL_FLOW(8, 4)      // 4 data + 4 adjoint tiles: backward programs of 128-column flows (MODE 4); 8 data tiles: MODE 20
#undef SX_DECL_FLOW
#undef SX_DECL_FLOW_F
