// Host-side sanitizer driver (test infrastructure; built by `make -C stribor_amd/csrc asan`, run by tests/test_host_cpu.py).
//
// The library's HOST code -- the launchers and argument validators of every sx_* entry point -- compiled for the CPU only
// (hipcc --offload-host-only) with -fsanitize=address,undefined, driven WITHOUT a GPU: every call below must come back with a status
// (never crash, never trip a sanitizer report).  Three parts:
//   1. the plain argument checks of the element-wise / packing / weight-gradient entry points (null pointers, bad sizes, bad enums,
//      misalignment): each must return non-zero and leave a message in sx_last_error();
//   2. sx_flow_launch_info / sx_flow_run / sx_flow_bwd_partials / sx_flow_bwd_run over VALID programs (cfg 2-, cfg 3-, cfg 4-like, the
//      128-column backward program) with n_rows = 0 -- the validators run, nothing is launched -- and over every single-field mutation
//      of those programs with the out-of-range values the round-3 fuzz found (negative / 255 / 256 / 2^30 tile and step fields);
//   3. a seeded random fuzz of whole sx_program structs through the validators (no launch: n_rows = 0 for sx_flow_run).
// GPU AddressSanitizer does not exist on this pool: this job covers the host side only (sanitizer row).
#include "../include/stribor_hip.h"
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// No device code exists in this build (--offload-host-only): the objects' static constructors would hand the HIP runtime fat binaries
// that are not there.  The registration entry points are taken over here (the executable's definitions win over libamdhip64's), the
// fat-binary symbols themselves are defined by the generated asan/fatbin_stubs.c; nothing below ever launches a kernel.
extern "C" {
void **__hipRegisterFatBinary(const void *) { static void *handle; return &handle; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipRegisterManagedVar(void *, void **, void *, const char *, size_t, unsigned) {}
}

static int g_fail = 0;
static long g_calls = 0;
#define EXPECT_BAD(call)                                                                         \
    do {                                                                                         \
        ++g_calls;                                                                               \
        int rc_ = (call);                                                                        \
        if (rc_ == 0) { fprintf(stderr, "FAIL %s:%d: accepted: %s\n", __FILE__, __LINE__, #call); ++g_fail; } \
        else if (!sx_last_error() || !sx_last_error()[0]) { fprintf(stderr, "FAIL %s:%d: no message: %s\n", __FILE__, __LINE__, #call); ++g_fail; } \
    } while (0)
#define EXPECT_OK(call)                                                                          \
    do {                                                                                         \
        ++g_calls;                                                                               \
        int rc_ = (call);                                                                        \
        if (rc_ != 0) { fprintf(stderr, "FAIL %s:%d: rc %d (%s): %s\n", __FILE__, __LINE__, rc_, sx_last_error(), #call); ++g_fail; } \
    } while (0)

static uint64_t g_rng = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return (uint32_t)(g_rng >> 16); }

static uint32_t g_blob_cursor;
static sx_step step(int kind, int c0, int ct, int t0, int tt, int rev, int act, size_t floats, int pad = 0) {
    sx_step s;
    memset(&s, 0, sizeof(s));
    s.kind = kind; s.c0 = c0; s.ct = ct; s.t0 = t0; s.tt = tt; s.reverse = rev; s.act = act; s.pad_ = pad;
    const uint32_t n = (uint32_t)((floats + 255) / 256 * 256);
    s.blob_off = g_blob_cursor; s.blob_floats = n; g_blob_cursor += n;
    s.ldj_scale = -1.f;
    return s;
}
static sx_program header(int dim, int x_tiles, int tiles, int h_tiles) {
    sx_program p;
    memset(&p, 0, sizeof(p));
    p.dim = dim; p.x_tiles = x_tiles; p.tiles = tiles; p.h_tiles = h_tiles; p.identity_cols = 1;
    g_blob_cursor = 256;
    return p;
}
static size_t plf(int m, int k) { return sx_packed_linear_floats(m, k); }

// cfg 2: eight split tanh couplings on two tiles
static sx_program prog_cfg2() {
    sx_program p = header(64, 2, 2, 2);
    for (int i = 0; i < 8; ++i)
        p.steps[p.n_steps++] = step(SX_STEP_COUPLING_AFFINE, i % 2 ? 1 : 0, 1, i % 2 ? 0 : 1, 1, 1, SX_ACT_TANH_FOLDED, plf(2, 1) + plf(2, 2));
    return p;
}
// cfg 4: (dense layer, split coupling) x 4 on four tiles
static sx_program prog_cfg4() {
    sx_program p = header(128, 4, 4, 2);
    for (int i = 0; i < 4; ++i) {
        p.steps[p.n_steps++] = step(SX_STEP_LINEAR_TILE, 0, 4, 0, 1, 0, 4, plf(4, 4) + 1);
        p.steps[p.n_steps++] = step(SX_STEP_COUPLING_AFFINE, i % 2 ? 2 : 0, 2, i % 2 ? 0 : 2, 2, 1, SX_ACT_TANH_FOLDED, plf(2, 2) + plf(4, 2));
    }
    return p;
}
// whole-layer dense steps on two tiles + a general coupling (MODE 2)
static sx_program prog_dense64() {
    sx_program p = header(64, 2, 2, 1);
    p.steps[p.n_steps++] = step(SX_STEP_LINEAR_TILE, 0, 2, 0, 1, 0, 2, plf(2, 2) + 1);
    p.steps[p.n_steps++] = step(SX_STEP_COUPLING_AFFINE, 0, 2, 0, 2, 0, SX_ACT_RELU, plf(1, 2) + plf(4, 1));
    p.steps[p.n_steps++] = step(SX_STEP_ROW_SCALE_EXP, 0, 0, 0, 2, 0, 0, 64);
    return p;
}
// cfg 3: one spline coupling = hidden step + per 8-column group a triple of phase steps
static sx_program prog_cfg3() {
    sx_program p = header(64, 2, 2, 2);
    p.steps[p.n_steps++] = step(SX_STEP_RQS_HIDDEN, 0, 1, 0, 0, 1, SX_ACT_TANH_FOLDED, plf(2, 1));
    for (int g = 0; g < 4; ++g)
        for (int ph = 0; ph < 3; ++ph)
            p.steps[p.n_steps++] = step(SX_STEP_RQS_PHASE, g, ph, 1, 16, 1, 0, plf(4, 2) + 4, -1);
    return p;
}
// 128-column backward program: 4 data + 4 adjoint tiles
static sx_program prog_bwd128() {
    sx_program p = header(128, 4, 8, 2);
    p.steps[p.n_steps++] = step(SX_STEP_COUPLING_AFFINE_BWD_A, 0, 2, 2, 0, 0, SX_ACT_TANH_FOLDED, plf(2, 2) + plf(4, 2));
    p.steps[p.n_steps++] = step(SX_STEP_COUPLING_AFFINE_BWD_B, 0, 2, 2, 0, 0, SX_ACT_TANH_FOLDED, plf(2, 4) + plf(2, 2));
    p.steps[p.n_steps++] = step(SX_STEP_LINEAR_BWD, 0, 4, 0, 1, 0, 0, plf(4, 4));
    p.steps[p.n_steps++] = step(SX_STEP_LINEAR_BWD, 4, 4, 4, 1, 1, 0, plf(4, 4));
    return p;
}
// 160 columns on eight data tiles: hidden step + one step per transformed tile
static sx_program prog_wide() {
    sx_program p = header(160, 8, 8, 2);
    p.steps[p.n_steps++] = step(SX_STEP_WIDE_HIDDEN, 0, 4, 4, 4, 1, SX_ACT_TANH_FOLDED, plf(2, 4));
    for (int t = 4; t < 8; ++t) p.steps[p.n_steps++] = step(SX_STEP_WIDE_AFFINE_TILE, 0, 0, t, 1, 1, SX_ACT_TANH_FOLDED, plf(2, 2));
    return p;
}
// conditioner MLP program
static sx_program prog_mlp() {
    sx_program p = header(64, 2, 2, 2);
    p.steps[p.n_steps++] = step(SX_STEP_MLP_HIDDEN, 0, 2, 0, 0, 0, SX_ACT_TANH, plf(2, 2));
    p.steps[p.n_steps++] = step(SX_STEP_MLP_OUT_TILE, 0, 0, 0, 1, 0, 0, plf(1, 2));
    return p;
}

static void touch_validators(const sx_program &p, int64_t rows_info) {
    int32_t g = 0, b = 0, l = 0, np = 0;
    int64_t pf = 0;
    float dummy[4] = {0, 0, 0, 0};
    ++g_calls;
    (void)sx_flow_launch_info(&p, rows_info, &g, &b, &l);
    (void)sx_flow_launch_info(&p, rows_info, nullptr, nullptr, nullptr);
    // n_rows = 0: every check runs, nothing is launched (x / blobs only need to be non-null)
    (void)sx_flow_run(&p, dummy, dummy, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0,
                      SX_F32, SX_GEMM_F16X3, nullptr, nullptr, nullptr);
    (void)sx_flow_run(&p, dummy, dummy, dummy, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, dummy, 4, 4, dummy, dummy, 0,
                      SX_BF16, SX_GEMM_F32, nullptr, nullptr, nullptr);
    (void)sx_flow_bwd_partials(&p, rows_info, &np, &pf);
    (void)sx_flow_bwd_run(&p, dummy, dummy, dummy, nullptr, nullptr, dummy, dummy, 0, nullptr, nullptr, nullptr);
}

int main() {
    float buf[64];
    double dbuf[8];
    int32_t ibuf[64];
    memset(buf, 0, sizeof(buf)); memset(dbuf, 0, sizeof(dbuf)); memset(ibuf, 0, sizeof(ibuf));
    float *f = buf;
    int32_t *ix = ibuf;
    printf("abi %d, build %s, default arithmetic %d\n", sx_abi_version(), sx_build_id(), sx_fragment_mode());

    // ---- 1. plain argument checks ------------------------------------------------------------------------------------------
    EXPECT_BAD(sx_permute(nullptr, f, ix, 4, 4, 4, nullptr));
    EXPECT_BAD(sx_permute(f, f, ix, 4, 4, 4, nullptr));                          // in place
    EXPECT_BAD(sx_permute(f, f + 16, ix, 4, 0, 4, nullptr));
    EXPECT_BAD(sx_permute(f, f + 16, ix, -1, 4, 4, nullptr));
    EXPECT_BAD(sx_permute(f, f + 16, ix, 4, 4, 3, nullptr));
    EXPECT_BAD(sx_affine_coupling(nullptr, f, f, f, 0, nullptr, 0, 2, 4, 4, SX_F32, 0, 0, 1.f, nullptr));
    EXPECT_BAD(sx_affine_coupling(f, f, f, f, 0, nullptr, 0, 5, 4, 4, SX_F32, 0, 0, 1.f, nullptr));      // n_live > dim
    EXPECT_BAD(sx_affine_coupling(f, f, f, f, 0, nullptr, 0, 2, 4, 4, 7, 0, 0, 1.f, nullptr));            // dtype
    EXPECT_BAD(sx_affine_coupling(f, f, f, f, 0, nullptr, 0, 2, -4, 4, SX_F32, 0, 0, 1.f, nullptr));
    EXPECT_BAD(sx_affine_coupling_bwd(f, f, f, f, 0, nullptr, f, nullptr, 0, 2, 4, 4, 0, 1.f, nullptr));
    EXPECT_BAD(sx_affine_coupling_bwd(f, f, f, f, 0, f, f, nullptr, 0, 0, 4, 4, 0, 1.f, nullptr));
    EXPECT_BAD(sx_time_affine_coupling(f, f, f, f, 0, nullptr, f, SX_TIME_TANH, nullptr, 0, 2, 4, 4, SX_F32, 0, 0, 1.f, nullptr));
    EXPECT_BAD(sx_time_affine_coupling(f, f, f, f, 0, f, f, 9, nullptr, 0, 2, 4, 4, SX_F32, 0, 0, 1.f, nullptr));
    EXPECT_BAD(sx_time_affine_coupling(f, f, f, f, 0, f, nullptr, SX_TIME_LOG, nullptr, 0, 2, 4, 4, SX_F32, 0, 0, 1.f, nullptr));
    EXPECT_BAD(sx_rqs_coupling(f, f, f, f, nullptr, 0, nullptr, 0, 2, 4, -3.f, 3.f, -3.f, 3.f, 4, 4, SX_F32, 0, 0, 1.f, nullptr, nullptr));
    EXPECT_BAD(sx_rqs_coupling(f, f, f, f, f, 0, nullptr, 0, 2, 0, -3.f, 3.f, -3.f, 3.f, 4, 4, SX_F32, 0, 0, 1.f, nullptr, nullptr));       // bins
    EXPECT_BAD(sx_rqs_coupling(f, f, f, f, f, 0, nullptr, 0, 2, 4, 3.f, -3.f, -3.f, 3.f, 4, 4, SX_F32, 0, 0, 1.f, nullptr, nullptr));       // empty domain
    EXPECT_BAD(sx_rqs_coupling(f, f, f, f, f, 0, nullptr, 0, 2, 2000, -3.f, 3.f, -3.f, 3.f, 4, 4, SX_F32, 0, 0, 1.f, nullptr, nullptr));    // min bin width
    EXPECT_BAD(sx_cubic_coupling(f, f, f, f, f, 0, nullptr, 0, 2, 200, -3.f, 3.f, 4, 4, SX_F32, 0, 0, 1.f, nullptr));
    EXPECT_BAD(sx_cubic_coupling(f, f, f, f, f, 0, nullptr, 0, 2, 4, 3.f, 3.f, 4, 4, SX_F32, 0, 0, 1.f, nullptr));
    EXPECT_BAD(sx_pointwise(nullptr, f, f, f, 4, 4, SX_F32, SX_PW_SIGMOID, 0.f, 0, nullptr));
    EXPECT_BAD(sx_pointwise(f, f, f, f, 4, 4, SX_F32, 99, 0.f, 0, nullptr));
    EXPECT_BAD(sx_pointwise(f, f, f, f, 4, 4, SX_F32, SX_PW_CUMSUM, 0.f, 0, nullptr));                    // cumsum in place
    EXPECT_BAD(sx_pointwise(f, f + 16, f, f, 4, 4, SX_F32, SX_PW_LEAKY_RELU, -1.f, 0, nullptr));
    EXPECT_BAD(sx_pointwise_bwd(f, nullptr, f, f, f, 4, 4, SX_PW_SIGMOID, 0.f, nullptr));
    EXPECT_BAD(sx_pointwise_bwd(f, f, f, f, f, 4, 0, SX_PW_SIGMOID, 0.f, nullptr));
    EXPECT_BAD(sx_absmax2(nullptr, 4, nullptr, 0, f, nullptr));
    EXPECT_BAD(sx_absmax2(f, -1, nullptr, 0, f, nullptr));
    EXPECT_BAD(sx_unit_normal_logprob(f, nullptr, nullptr, 4, 4, SX_F32, nullptr));
    EXPECT_BAD(sx_unit_normal_logprob(f, nullptr, f, 4, 4, 5, nullptr));
    EXPECT_BAD(sx_sum_f64(nullptr, 4, dbuf, nullptr));
    EXPECT_BAD(sx_sum_f64(f, -4, dbuf, nullptr));
    EXPECT_BAD(sx_pack_linear(nullptr, f, 4, 4, ix, ix, 1, 1, nullptr, nullptr, 0.f, 0, SX_GEMM_F16X3, nullptr, f, nullptr));
    EXPECT_BAD(sx_pack_linear(f, f, 4, 4, ix, ix, 0, 1, nullptr, nullptr, 0.f, 0, SX_GEMM_F16X3, nullptr, f, nullptr));
    EXPECT_BAD(sx_pack_linear(f, f, 4, 4, ix, ix, 1, 1, nullptr, nullptr, 0.f, 0, 42, nullptr, f, nullptr));
    EXPECT_BAD(sx_pack_linear_bound(f, f, 4, 4, ix, ix, 1, 1, nullptr, nullptr, 0.f, 0, SX_GEMM_F16X3, nullptr, f, nullptr, nullptr));
    {
        sx_pack_job pj{};
        EXPECT_BAD(sx_pack_linear_batch(nullptr, 1, 1056, SX_GEMM_F16X3, nullptr, nullptr));
        EXPECT_BAD(sx_pack_linear_batch(&pj, 0, 1056, SX_GEMM_F16X3, nullptr, nullptr));
        EXPECT_BAD(sx_pack_linear_batch(&pj, 1, 0, SX_GEMM_F16X3, nullptr, nullptr));
        EXPECT_BAD(sx_pack_linear_batch(&pj, 1, 1056, 42, nullptr, nullptr));
        sx_reduce_job rj{};
        EXPECT_BAD(sx_wgrad_reduce_batch(nullptr, f, &rj, 1, 1, 1056, nullptr));
        EXPECT_BAD(sx_wgrad_reduce_batch(f, f, nullptr, 1, 1, 1056, nullptr));
        EXPECT_BAD(sx_wgrad_reduce_batch(f, f, &rj, 0, 1, 1056, nullptr));
        EXPECT_BAD(sx_wgrad_reduce_batch(f, f, &rj, 1, 0, 1056, nullptr));
    }
    EXPECT_BAD(sx_wgrad_reduce(nullptr, 1, 32, 32, f, 32, f, 32, 32, nullptr, nullptr, nullptr));
    EXPECT_BAD(sx_wgrad_reduce(f, 0, 32, 32, f, 32, f, 32, 32, nullptr, nullptr, nullptr));
    EXPECT_BAD(sx_wgrad_reduce(f, 1, 129, 32, f, 32, f, 32, 32, nullptr, nullptr, nullptr));
    EXPECT_BAD(sx_wgrad(nullptr, 4, 4, f, 4, 4, 4, SX_WGRAD_ROW_MAJOR, f, 4, f, nullptr, nullptr, f, nullptr));
    EXPECT_BAD(sx_wgrad(f, 4, 4, f, 4, 4, 4, 2, f, 4, f, nullptr, nullptr, f, nullptr));                   // layout
    EXPECT_BAD(sx_wgrad(f, 4, 4, f, 4, 129, 4, SX_WGRAD_ROW_MAJOR, f, 4, f, nullptr, nullptr, f, nullptr));
    EXPECT_BAD(sx_wgrad(f + 1, 4, 4, f, 4, 4, 4, SX_WGRAD_ROW_GROUPS, f, 4, f, nullptr, nullptr, f, nullptr));     // misaligned groups
    EXPECT_BAD(sx_wgrad_layer(nullptr, 4096, 4, 1, 2, 1, 64, f, 64, f, nullptr, f, 32, f, nullptr, f, nullptr));
    EXPECT_BAD(sx_wgrad_layer(f, 100, 4, 1, 2, 1, 64, f, 64, f, nullptr, f, 32, f, nullptr, f, nullptr));
    EXPECT_BAD(sx_colsum(f, 2, 4, 4, f, f, nullptr));                                                      // lda < M
    EXPECT_BAD(sx_tri_inverse_f64(nullptr, dbuf, 1, 2, 1, 0, nullptr));
    EXPECT_BAD(sx_tri_inverse_f64(dbuf, dbuf, 1, 129, 1, 0, nullptr));
    // size queries take anything
    (void)sx_packed_linear_floats(0, 0); (void)sx_packed_linear_floats(-1, 7); (void)sx_packed_linear_floats(INT_MAX, INT_MAX);
    (void)sx_wgrad_scratch_floats(0, 0, 0); (void)sx_wgrad_scratch_floats(2048, 128, 3); (void)sx_wgrad_scratch_floats(-5, -5, 9);
    (void)sx_wgrad_layer_scratch_floats(1, 2, 1); (void)sx_wgrad_layer_scratch_floats(-1, 99, 0);
    (void)sx_rqs_slab_slots(0); (void)sx_rqs_slab_slots(32); (void)sx_rqs_slab_slots(-3); (void)sx_rqs_slab_slots(INT_MAX);
    (void)sx_rqs_slab_scratch_floats(0, 32, 64); (void)sx_rqs_slab_scratch_floats((int64_t)1 << 40, 32, 64); (void)sx_rqs_slab_scratch_floats(-1, -1, -1);
    (void)sx_rqs_slab_l1_scratch_floats(64, 64); (void)sx_rqs_slab_l1_scratch_floats(-1, 1 << 30);
    (void)sx_rqs_slab_fwd_scratch_floats(0, 32); (void)sx_rqs_slab_fwd_scratch_floats((int64_t)1 << 40, 32); (void)sx_rqs_slab_fwd_scratch_floats(-1, INT_MAX);
    // the forward slab pass: every argument rule, then an empty batch (no launch)
    EXPECT_BAD(sx_rqs_slab_fwd(nullptr, f, 64, 64, f, f, f, nullptr, 32, 32, nullptr, 0, 16, -3.f, 3.f, -3.f, 3.f, 4, 64, 1, 1.f, 0, 0, 0, f, nullptr, nullptr));
    EXPECT_BAD(sx_rqs_slab_fwd(f, f, 64, 64, f, f, f, nullptr, 32, 32, nullptr, 0, 16, -3.f, 3.f, -3.f, 3.f, 4, 64, 1, 1.f, 0, 0, 0, nullptr, nullptr, nullptr));   // ldj without scratch
    EXPECT_BAD(sx_rqs_slab_fwd(f, f, 64, 64, f, f, nullptr, nullptr, 32, 32, nullptr, 0, 17, -3.f, 3.f, -3.f, 3.f, 4, 64, 1, 1.f, 0, 0, 0, nullptr, nullptr, nullptr));   // bins
    EXPECT_BAD(sx_rqs_slab_fwd(f, f, 320, 257, f, f, nullptr, nullptr, 32, 32, nullptr, 0, 16, -3.f, 3.f, -3.f, 3.f, 4, 64, 1, 1.f, 0, 0, 0, nullptr, nullptr, nullptr)); // hidden
    EXPECT_BAD(sx_rqs_slab_fwd(f, f, 32, 64, f, f, nullptr, nullptr, 32, 32, nullptr, 0, 16, -3.f, 3.f, -3.f, 3.f, 4, 64, 1, 1.f, 0, 0, 0, nullptr, nullptr, nullptr));   // ld_h < hidden
    EXPECT_BAD(sx_rqs_slab_fwd(f, f, 64, 64, f, f, nullptr, nullptr, 32, 65, nullptr, 0, 16, -3.f, 3.f, -3.f, 3.f, 4, 64, 1, 1.f, 0, 0, 0, nullptr, nullptr, nullptr));   // n_live > dim
    EXPECT_BAD(sx_rqs_slab_fwd(f, f, 64, 64, f, f, nullptr, nullptr, 32, 32, nullptr, 0, 16, 3.f, -3.f, -3.f, 3.f, 4, 64, 1, 1.f, 0, 0, 0, nullptr, nullptr, nullptr));   // empty domain
    EXPECT_BAD(sx_rqs_slab_fwd(f, f, 64, 64, f, f, nullptr, nullptr, 32, 32, nullptr, 0, 16, -3.f, 3.f, -3.f, 3.f, -1, 64, 1, 1.f, 0, 0, 0, nullptr, nullptr, nullptr));  // n_rows < 0
    EXPECT_BAD(sx_rqs_slab_fwd(f, f, 64, 64, f, f, nullptr, nullptr, 32, 32, nullptr, 0, 16, -3.f, 3.f, -3.f, 3.f, 4, 64, 2, 1.f, 0, 0, 0, nullptr, nullptr, nullptr));   // reverse = 2 is the cubic splines' reference mode
    EXPECT_BAD(sx_rqs_slab_fwd(f, f, 64, 64, f, f, nullptr, nullptr, 32, 32, nullptr, 0, 16, -3.f, 3.f, -2.f, 3.f, 4, 64, 1, 1.f, 0, 0, 1, nullptr, nullptr, nullptr));   // cubic: one domain
    EXPECT_BAD(sx_rqs_slab_fwd(f, f, 64, 64, f, f, nullptr, nullptr, 32, 32, nullptr, 5, 16, -3.f, 3.f, -3.f, 3.f, 4, 64, 1, 1.f, 0, 0, 0, nullptr, nullptr, nullptr));   // pass-through columns without their list
    EXPECT_BAD(sx_rqs_slab_fwd(f, f, 64, 64, f, f, nullptr, nullptr, 32, 32, (const int32_t *)f, 33, 16, -3.f, 3.f, -3.f, 3.f, 4, 64, 1, 1.f, 0, 0, 0, nullptr, nullptr, nullptr));   // more than dim - n_live
    EXPECT_OK(sx_rqs_slab_fwd(f, f, 64, 64, f, f, nullptr, nullptr, 32, 32, nullptr, 0, 16, -3.f, 3.f, -3.f, 3.f, 0, 64, 1, 1.f, 0, 0, 0, nullptr, nullptr, nullptr));
    EXPECT_OK(sx_rqs_slab_fwd(f, f, 64, 64, f, f, nullptr, nullptr, 32, 32, nullptr, 0, 16, -3.f, 3.f, -3.f, 3.f, 0, 64, 2, 1.f, 0, 0, 1, nullptr, nullptr, nullptr));
    // sx_flow_run2 (round 6): the exact redo pass's arguments come together, and only with the fp16 x 3 arithmetic
    (void)sx_flow_redo_words(0); (void)sx_flow_redo_words(-1); (void)sx_flow_redo_words((int64_t)1 << 40);
    if (sx_flow_redo_words(64) != 2 + 2 * 2) { fprintf(stderr, "sx_flow_redo_words(64)\n"); return 1; }
    (void)sx_rqs_slab_hidden_floats(0, 64); (void)sx_rqs_slab_hidden_floats((int64_t)1 << 40, 64); (void)sx_rqs_slab_hidden_floats(-1, INT_MAX);
    EXPECT_BAD(sx_rqs_slab_hidden(nullptr, nullptr, f, nullptr, f, 4, 64, 0, 160, 1, nullptr, nullptr));
    EXPECT_BAD(sx_rqs_slab_hidden(f, nullptr, f, nullptr, f, 4, 64, 8, 160, 1, nullptr, nullptr));        // latent_dim without latent
    EXPECT_BAD(sx_rqs_slab_hidden(f, f, f, nullptr, f, 4, 100, 29, 160, 1, nullptr, nullptr));            // more than 128 input columns
    EXPECT_BAD(sx_rqs_slab_hidden(f, nullptr, f, nullptr, f, 4, 64, 0, 257, 1, nullptr, nullptr));        // hidden
    EXPECT_BAD(sx_rqs_slab_hidden(f, nullptr, f, nullptr, f, 4, 64, 0, 160, 77, nullptr, nullptr));       // activation
    EXPECT_BAD(sx_rqs_slab_hidden(f, nullptr, f, nullptr, f, -4, 64, 0, 160, 1, nullptr, nullptr));
    EXPECT_OK(sx_rqs_slab_hidden(f, nullptr, f, nullptr, f, 0, 64, 0, 160, 1, nullptr, nullptr));
    { uint32_t cm[4] = {0xffffu, 0, 0, 0}; EXPECT_OK(sx_rqs_slab_hidden(f, nullptr, f, cm, f, 0, 64, 0, 160, 1, nullptr, nullptr)); }
    (void)sx_flow_bwd_max_steps();

    // ---- 2. valid programs: accepted; every single-field mutation: a status, no crash ---------------------------------------------
    sx_program progs[] = {prog_cfg2(), prog_cfg4(), prog_dense64(), prog_cfg3(), prog_bwd128(), prog_wide(), prog_mlp()};
    const char *names[] = {"cfg2", "cfg4", "dense64", "cfg3", "bwd128", "wide160", "mlp"};
    const int n_progs = (int)(sizeof(progs) / sizeof(progs[0]));
    for (int i = 0; i < n_progs; ++i) {
        int32_t g = 0, b = 0, l = 0;
        ++g_calls;
        const int rc = sx_flow_launch_info(&progs[i], 1 << 20, &g, &b, &l);
        if (rc != 0) { fprintf(stderr, "FAIL: valid program %s rejected: %s\n", names[i], sx_last_error()); ++g_fail; }
        else printf("  %-26s grid %5d block %4d lds %6d B\n", names[i], g, b, l);
        if (rc == 0 && (g < 1 || b < 64 || b > 1024 || l < 16 || l > 160 * 1024)) { fprintf(stderr, "FAIL: %s: launch shape\n", names[i]); ++g_fail; }
    }
    EXPECT_BAD(sx_flow_launch_info(nullptr, 4, nullptr, nullptr, nullptr));
    {
        sx_program p = progs[0];
        float d[4] = {0, 0, 0, 0};
        EXPECT_OK(sx_flow_run(&p, d, d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, SX_F32,
                              SX_GEMM_F16X3, nullptr, nullptr, nullptr));
        EXPECT_BAD(sx_flow_run(&p, d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, SX_F32,
                               SX_GEMM_F16X3, nullptr, nullptr, nullptr));                                   // null x
        EXPECT_BAD(sx_flow_run(&p, d, d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, -1, SX_F32,
                               SX_GEMM_F16X3, nullptr, nullptr, nullptr));                                   // n_rows < 0
        EXPECT_BAD(sx_flow_run(&p, d, d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, 3,
                               SX_GEMM_F16X3, nullptr, nullptr, nullptr));                                   // dtype
        EXPECT_BAD(sx_flow_run(&p, d, d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, SX_F32,
                               5, nullptr, nullptr, nullptr));                                               // precision
        EXPECT_BAD(sx_flow_run(&p, nullptr, d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, SX_F32,
                               SX_GEMM_F16X3, nullptr, nullptr, nullptr));                                   // null blobs
        p.identity_cols = 0;
        EXPECT_BAD(sx_flow_run(&p, d, d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, SX_F32,
                               SX_GEMM_F16X3, nullptr, nullptr, nullptr));                                   // needs in_col
        // a dense layer that does not cover all output slabs, a blob smaller than the step needs
        sx_program q = progs[1];
        q.steps[0].act = 2;
        EXPECT_BAD(sx_flow_launch_info(&q, 4, nullptr, nullptr, nullptr));
        q = progs[1]; q.steps[0].blob_floats = 256;
        EXPECT_BAD(sx_flow_launch_info(&q, 4, nullptr, nullptr, nullptr));
    }
    static const int32_t bad_i[] = {-1, 0, 1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 24, 25, 31, 32, 33, 127, 128, 129, 255, 256, 257, 65535, 1 << 30, INT_MAX, INT_MIN};
    const int n_bad = (int)(sizeof(bad_i) / sizeof(bad_i[0]));
    for (int i = 0; i < n_progs; ++i) {
        // header fields
        for (int fld = 0; fld < 8; ++fld)
            for (int v = 0; v < n_bad; ++v) {
                sx_program p = progs[i];
                reinterpret_cast<int32_t *>(&p)[fld] = bad_i[v];
                if (fld == 0 && (bad_i[v] < 0 || bad_i[v] > SX_MAX_STEPS)) { touch_validators(p, 1000); continue; }
                touch_validators(p, 1000);
            }
        // every field of every step
        for (int s = 0; s < progs[i].n_steps; ++s)
            for (int fld = 0; fld < 12; ++fld)
                for (int v = 0; v < n_bad; ++v) {
                    sx_program p = progs[i];
                    reinterpret_cast<int32_t *>(&p.steps[s])[fld] = bad_i[v];
                    touch_validators(p, (int64_t)1 << (v % 40));
                }
    }

    // ---- 3. random programs -------------------------------------------------------------------------------------------------------
    static const int kinds[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 77, -1};
    long accepted = 0;
    for (int it = 0; it < 60000; ++it) {
        sx_program p;
        memset(&p, 0, sizeof(p));
        static const int tl[] = {1, 2, 4, 8, 3, 0};
        p.tiles = tl[rnd() % 6];
        p.x_tiles = (rnd() % 4 == 0) ? (int)(rnd() % 9) : p.tiles;
        p.h_tiles = tl[rnd() % 4];
        p.dim = (rnd() % 8 == 0) ? (int)(rnd() % 300) - 10 : 32 * p.x_tiles - (int)(rnd() % 3) * 4;
        p.latent_dim = (rnd() % 6 == 0) ? (int)(rnd() % 70) - 2 : 0;
        p.identity_cols = rnd() % 2;
        p.pad_ = (rnd() % 16 == 0) ? (int)(rnd() % 400) : 0;
        p.n_steps = (rnd() % 64 == 0) ? (int)(rnd() % 300) - 20 : 1 + (int)(rnd() % 12);
        uint32_t cursor = 256;
        const int ns = p.n_steps < 0 ? 0 : (p.n_steps > SX_MAX_STEPS ? SX_MAX_STEPS : p.n_steps);
        for (int s = 0; s < ns; ++s) {
            sx_step &st = p.steps[s];
            st.kind = kinds[rnd() % (sizeof(kinds) / sizeof(kinds[0]))];
            if (s > 0 && rnd() % 3 == 0) st.kind = p.steps[s - 1].kind;            // runs of one kind (pairs, triples)
            st.c0 = (int)(rnd() % 6) - (rnd() % 16 == 0); st.ct = (int)(rnd() % 6);
            st.t0 = (int)(rnd() % 9) - (rnd() % 16 == 0); st.tt = (rnd() % 4 == 0) ? (int)(rnd() % 40) : (int)(rnd() % 5);
            if (s > 0 && rnd() % 2 == 0) { st.c0 = p.steps[s - 1].c0; st.t0 = p.steps[s - 1].t0; st.tt = p.steps[s - 1].tt; st.ct = p.steps[s - 1].ct + (int)(rnd() % 2); }
            st.reverse = rnd() % 3; st.act = (rnd() % 8 == 0) ? (int)(rnd() % 300) : (int)(rnd() % 10);
            st.pad_ = (rnd() % 4 == 0) ? (int)rnd() : (int)(rnd() % 4);
            st.blob_floats = (rnd() % 16 == 0) ? rnd() % 100000 : 256u * (1 + rnd() % 200);
            st.blob_off = (rnd() % 16 == 0) ? rnd() % 5000 : cursor;
            cursor += st.blob_floats / 256 * 256 + 256;
        }
        int32_t g = 0, b = 0, l = 0;
        ++g_calls;
        if (sx_flow_launch_info(&p, 1 + (int64_t)(rnd() % 100000), &g, &b, &l) == 0) {
            ++accepted;
            if (g < 1 || b < 64 || b > 1024 || l < 16 || l > 160 * 1024) { fprintf(stderr, "FAIL: accepted random program with grid %d block %d lds %d\n", g, b, l); ++g_fail; }
        }
        if (it % 8 == 0) touch_validators(p, 12345);
    }
    printf("%ld calls, %ld random programs accepted, %d failures\n", g_calls, accepted, g_fail);
    return g_fail ? 1 : 0;
}
