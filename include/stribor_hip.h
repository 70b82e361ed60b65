/*
 * stribor_hip.h — C ABI of libstribor_hip.so: the MI355X (gfx950) kernels behind stribor's
 * coupling-flow hot path (NormalizingFlow.log_prob / forward / inverse over Coupling(Affine|Spline),
 * Affine, AffineLU, MatrixExponential, Permute/Flip, UnitNormal).
 *
 * The reference (mbilos/stribor, /root/reference) is pure Python and has NO FFI; its plugin
 * boundary is the `Transform` class protocol (stribor/flow.py:8-69).  The entry points below are
 * therefore what a ctypes binding inside stribor's own classes would call — each one cites the
 * reference method whose torch-op chain it replaces.  INTEGRATION.md shows that binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - the library allocates nothing, keeps no per-stream or per-device state and never synchronises:
 *     all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream) of
 *     the CURRENT device (callers hipSetDevice / torch.cuda.device first); scratch and counters are
 *     caller-owned (`work`, `scratch` arguments below);
 *   - tensors are row-major, rows = samples; `x`/`y` are [n_rows, dim] with element type
 *     `dtype` (SX_F32 or SX_BF16 storage); all arithmetic is fp32; ldj / log-prob are fp32;
 *   - every function returns 0 on success, a negative SX_E* code for a bad argument, or a
 *     positive hipError_t; sx_last_error() gives a thread-local message;
 *   - data-dependent error conditions of the reference (ValueError / assert inside
 *     rational_quadratic_spline.py:175-178,223) are reported through a caller-owned device
 *     flag word (`err_flag`, bit mask SX_FLAG_*), never by aborting.
 */
#ifndef STRIBOR_HIP_H
#define STRIBOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SX_ABI_VERSION 3

/* storage dtypes of x / y */
#define SX_F32  0
#define SX_BF16 1

/* error codes */
#define SX_OK            0
#define SX_E_BADARG     -1
#define SX_E_UNSUPPORTED -2

/* device error-flag bits */
#define SX_FLAG_RQS_NEG_DISCRIMINANT 1u   /* rational_quadratic_spline.py:223 assert */
#define SX_FLAG_NONFINITE            2u
#define SX_FLAG_F16_RANGE            4u   /* an operand of the fp16 x 3 GEMMs exceeded 65504 (weights at pack time, flow state /
                                             activations at run time); the affected rows were returned as NaN              */

/* GEMM arithmetic of the MFMA path (both are in the library; chosen per call):
 *   SX_GEMM_F16X3: both operands split hi + lo in fp16, three products per 16-deep step on v_mfma_f32_32x32x16_f16 with
 *                  fp32 accumulation (~2^-22 relative per product, fp32-grade) -- on the matrix pipe, beside the VALU.
 *                  Operands must stay within fp16's range (|v| <= 65504): see SX_FLAG_F16_RANGE.
 *   SX_GEMM_F32:   v_mfma_f32_32x32x2_f32, an exact fp32 fma chain with no range limit (3x slower on cfg 2). */
#define SX_GEMM_F32   0
#define SX_GEMM_F16X3 1

/* hidden activations of the conditioner (torch.nn names, stribor/net/mlp.py:38-39) */
#define SX_ACT_IDENTITY  0
#define SX_ACT_TANH      1
#define SX_ACT_RELU      2
#define SX_ACT_SIGMOID   3
#define SX_ACT_ELU       4
#define SX_ACT_SOFTPLUS  5
#define SX_ACT_LEAKYRELU 6
#define SX_ACT_SILU      7
#define SX_ACT_GELU      8
/* coupling steps only: weights packed with the tanh / exp constants folded in (see sx_pack_linear):
 * hidden = 1/(exp2(z') + 1), scale = exp2(ls'); ldj_scale carries the 1/(+-log2 e) factor */
#define SX_ACT_TANH_FOLDED 9

int         sx_abi_version(void);
/* The library's default GEMM arithmetic (SX_GEMM_F16X3). */
int         sx_fragment_mode(void);
const char *sx_last_error(void);
/* 16 hex digits identifying the sources this library was built from (sha256 over csrc/ + this header): measurement
 * files under profiles/ record it so that counter-derived figures are only quoted for the build they were taken on. */
const char *sx_build_id(void);

/* ------------------------------------------------------------------------------------------
 * Elementwise / HBM-bound kernels (params already in HBM)
 * ------------------------------------------------------------------------------------------ */

/* Permute / Flip: y[n, j] = x[n, idx[j]]   (stribor/flows/permute.py:35,38,71,75).
 * Bit-exact byte move; elem_bytes is 2 or 4. */
int sx_permute(const void *x, void *y, const int32_t *idx, int64_t n_rows, int32_t dim,
               int32_t elem_bytes, void *stream);

/* Affine coupling, element-wise part (stribor/flows/affine.py:104-109 + coupling.py:78,95).
 *   params[n, 0:n_live]        = log_scale of the live (transformed, mask==0) dims
 *   params[n, n_live:2*n_live] = shift of the live dims        (row n at params + n*params_stride;
 *                                                               params_stride = 0 broadcasts one row)
 *   live_idx[i] = column of x the i-th live parameter applies to; other columns are copied.
 *   live_idx == NULL means the contiguous range [live_start, live_start + n_live) (vectorised path).
 *   forward:  y = x*exp(ls) + sh       reverse:  y = (x - sh)*exp(-ls)
 *   ldj (nullable, [n_rows]):  ldj[n] = (ldj_accumulate ? ldj[n] : 0) + ldj_scale * sum_i ls[n,i]
 * Also serves st.Affine without coupling (live_idx = 0..dim-1). */
int sx_affine_coupling(const void *x, void *y, float *ldj, const float *params, int64_t params_stride,
                       const int32_t *live_idx, int32_t live_start, int32_t n_live, int64_t n_rows,
                       int32_t dim, int32_t dtype, int32_t reverse, int32_t ldj_accumulate,
                       float ldj_scale, void *stream);

/* Time-conditioned affine coupling (ContinuousAffineCoupling, stribor/flows/coupling.py:98-213):
 * as sx_affine_coupling, with every row's (log_scale, shift) multiplied by the time embedding of t[n] first:
 *   ls' = ls * phi(tscale[i] t),  sh' = sh * phi(tscale[n_live + i] t),  phi = the time net (net/time_net.py). */
#define SX_TIME_IDENTITY 0   /* TimeIdentity: t            */
#define SX_TIME_LINEAR   1   /* TimeLinear:   s t          */
#define SX_TIME_TANH     2   /* TimeTanh:     tanh(s t)    */
#define SX_TIME_LOG      3   /* TimeLog:      log(e^s t+1) */
int sx_time_affine_coupling(const void *x, void *y, float *ldj, const float *params, int64_t params_stride,
                            const float *t, const float *tscale, int32_t time_kind, const int32_t *live_idx,
                            int32_t live_start, int32_t n_live, int64_t n_rows, int32_t dim, int32_t dtype,
                            int32_t reverse, int32_t ldj_accumulate, float ldj_scale, void *stream);

/* Backward of sx_affine_coupling for training (layer-wise autograd path; the single-launch backward of whole
 * affine-coupling flows is the SX_STEP_COUPLING_AFFINE_BWD program): x, gy [n_rows, dim] fp32, gldj [n_rows];
 * gx: live columns receive dL/dx; gparams [n_rows, 2*n_live] = (dL/dlog_scale | dL/dshift). */
int sx_affine_coupling_bwd(const float *x, const float *gy, const float *gldj, const float *params,
                           int64_t params_stride, float *gx, float *gparams, const int32_t *live_idx,
                           int32_t live_start, int32_t n_live, int64_t n_rows, int32_t dim, int32_t reverse,
                           float ldj_scale, void *stream);

/* Rational-quadratic spline, element-wise part
 * (stribor/util/rational_quadratic_spline.py:11-251, util/search_sorted.py:3-5, flows/spline.py:82-86).
 *   params[n, i*(3K-1) + 0:K]    unnormalised widths  of live dim i
 *   params[n, i*(3K-1) + K:2K]   unnormalised heights
 *   params[n, i*(3K-1) + 2K:3K-1] unnormalised interior derivatives   (K = n_bins)
 *   domain [left,right] -> codomain [bottom,top]; outside the (input-side) interval: y = x, ljd = 0.
 *   ldiag (nullable, [n_rows, dim]): per-element log-diag-Jacobian (0 on non-live columns);
 *   the reverse direction returns the already-negated value like the reference (:234).
 *   ldj as in sx_affine_coupling (sum over live dims of the value written to ldiag). */
int sx_rqs_coupling(const void *x, void *y, float *ldj, float *ldiag, const float *params,
                    int64_t params_stride, const int32_t *live_idx, int32_t live_start, int32_t n_live, int32_t n_bins,
                    float left, float right, float bottom, float top, int64_t n_rows, int32_t dim,
                    int32_t dtype, int32_t reverse, int32_t ldj_accumulate, float ldj_scale,
                    uint32_t *err_flag, void *stream);

/* Backward of sx_rqs_coupling(reverse = 1) -- the direction log_prob evaluates -- for training:
 * reverse mode through rational_quadratic_spline.py:101-107,180-234.
 *   x [n_rows, dim] fp32: the values the inverse pass was given;  gout [n_rows, dim]: dL/d(output), live columns read;
 *   gldj [n_rows] (nullable): dL/d(row log-det), scaled by ldj_scale;  gldiag [n_rows, dim] (nullable; ABI v3): dL/d of the
 *   per-element log-derivative (the ldiag output: Spline.log_diag_jacobian, flows/spline.py:120-143, differentiated), added to
 *   the row adjoint of its element;  params as sx_rqs_coupling.
 *   gx [n_rows, dim]: live columns receive dL/dx;  gparams [n_rows, n_live*(3K-1)] (packed rows): dL/dparams. */
int sx_rqs_inverse_bwd(const float *x, const float *gout, const float *gldj, const float *gldiag, const float *params,
                       int64_t params_stride, float *gx, float *gparams, const int32_t *live_idx, int32_t live_start,
                       int32_t n_live, int32_t n_bins, float left, float right, float bottom, float top, int64_t n_rows,
                       int32_t dim, float ldj_scale, void *stream);

/* The same for sx_rqs_coupling(reverse = 0), the forward direction (forward / rsample of spline flows; reverse mode through
 * rational_quadratic_spline.py:101-107,180-207,236-248; the bin is searched on the widths, the input lives in [left, right]). */
int sx_rqs_forward_bwd(const float *x, const float *gout, const float *gldj, const float *gldiag, const float *params,
                       int64_t params_stride, float *gx, float *gparams, const int32_t *live_idx, int32_t live_start,
                       int32_t n_live, int32_t n_bins, float left, float right, float bottom, float top, int64_t n_rows,
                       int32_t dim, float ldj_scale, void *stream);

/* Backward of one rational-quadratic spline COUPLING with the parameter tensor never in HBM (training of
 * Coupling(Spline(spline_type='quadratic')), stribor/flows/coupling.py:69-95 + flows/spline.py:76-87): the spline's reverse
 * mode (as sx_rqs_inverse_bwd) fused with the last conditioner layer  params = h W2^T + b2  and that layer's backward.
 * A workgroup owns a slab of W2 -- the 3K-1 rows of two transformed columns -- over a range of rows (sx_rqs_slab.hip).
 *   x, gout [n_rows, dim], gldj [n_rows]: as sx_rqs_inverse_bwd;  h [n_rows, ld_h]: last hidden activation (`hidden` <= 256
 *   features; beyond 64 a workgroup holds one slab and runs one wave per SIMD, beyond 128 the layer takes two launches, each over
 *   half of the hidden tiles, and gh must not be NULL);  n_bins <= 16.
 *   xout: NULL for rational-quadratic splines.  Not NULL: MONOTONE CUBIC splines (cubic_spline.py:21-251; as
 *   sx_cubic_inverse_bwd: xout [n_rows, dim] is the inverse pass's output, domain [left, right] on both sides, 2K+2 parameters
 *   per element: slots 32 t + R of tile 2 carry the two boundary-derivative parameters at (R&3) + 4 (R>>3) = 0, 1).
 *   Slots: slab s (columns live[2s], live[2s+1]) has 96 slots; slot 32 t + R (t = 0 widths, 1 heights, 2 derivatives) is
 *   parameter (R&3) + 4 (R>>3) of column 2 s + ((R>>2)&1).  slot_rows[sx_rqs_slab_slots(n_live)]: slot -> row of W2 / b2 as
 *   handed to the packers, or -1 (padding).
 *   w_fwd: sx_pack_linear(W2, b2, row_idx = slot_rows, col_idx = hidden slots, m_tiles = slots/32, k_tiles = ceil(hidden/32),
 *          SX_GEMM_F16X3);  w_bwd: the same with transpose = 1 (m_tiles = ceil(hidden/32), k_tiles = slots/32, no bias).
 *   Outputs: gx [n_rows, dim] (transformed columns written), gh [n_rows, ld_gh] = dL/dh (NULL: the dh partials are left in
 *   `scratch` for sx_rqs_slab_l1_bwd), dW [rows of W2, ldw] and db
 *   (rows named by slot_rows written), all fp32.  scratch: sx_rqs_slab_scratch_floats(n_rows, n_live, hidden) floats,
 *   16-byte aligned, caller-owned.  err_flag (nullable) receives SX_FLAG_F16_RANGE when h or a parameter gradient leaves
 *   fp16's range (those rows' gx are NaN).
 *   tanh_hidden != 0: h = tanh(a) and gh receives dL/da = dL/dh (1 - h^2) (the conditioner's last activation folded in).
 *   scale (nullable, device, ONE float): the largest |gout| / |gldj| of this call, reduced by the caller on the device.  The
 *   kernels multiply both adjoints by S = 2^-ilogb(scale) on the way in and every output by 1/S on the way out (exact), so
 *   the parameter gradients -- operands of the fp16 x 3 GEMMs -- sit in fp16's normal range whatever the magnitude of the
 *   loss (dL/dlog_prob = 1/N underflows fp16 for large batches).  No host synchronisation. */
int32_t sx_rqs_slab_slots(int32_t n_live);
size_t sx_rqs_slab_scratch_floats(int64_t n_rows, int32_t n_live, int32_t hidden);
int sx_rqs_slab_bwd(const float *x, const float *gout, const float *gldj, const float *xout, const float *h, int64_t ld_h,
                    int32_t hidden,
                    const float *w_fwd, const float *w_bwd, const int32_t *slot_rows, float *gx, float *gh, int64_t ld_gh,
                    float *dW, int64_t ldw, float *db, const int32_t *live_idx, int32_t live_start, int32_t n_live,
                    int32_t n_bins, float left, float right, float bottom, float top, int64_t n_rows, int32_t dim,
                    float ldj_scale, int32_t tanh_hidden, const float *scale, float *scratch, uint32_t *err_flag,
                    void *stream);

/* Backward of the conditioner's FIRST layer behind sx_rqs_slab_bwd, for Linear - Tanh - Linear conditioners without a latent
 * input (net/mlp.py:26-58 with one hidden layer; coupling.py:61: the layer sees x * mask):  a = W1m x + b1, h = tanh(a),
 * W1m = W1 with the mask folded into its columns.  Call sx_rqs_slab_bwd with gh = NULL (the dh partials stay in its scratch),
 * then this on the same stream: one pass over the rows sums the partials, applies 1 - h^2, writes
 * gx[:, conditioning columns] = gout + W1m^T da (the transformed columns of gx were written by sx_rqs_slab_bwd) and contracts
 * dW1 += da x^T, db1 += sum da on the matrix pipe.
 *   slab_scratch: the scratch sx_rqs_slab_bwd was given (same n_rows, n_live, hidden);  h, x, gout, gx, scale: as there.
 *   w1t: sx_pack_linear(W1m [hidden, dim], NULL, row_idx = column slots, col_idx = hidden slots, m_tiles = ceil(dim/32),
 *        k_tiles = ceil(hidden/32), transpose = 1, SX_GEMM_F16X3).
 *   cond_mask: HOST pointer, 2 words: bit c of word t = column 32 t + c is a conditioning column.
 *   dW1 [hidden, ldw] and db1 [hidden] are ACCUMULATED into (zero them); col_map[ceil(dim/32)*32]: column -> column of dW1, or -1
 *   (transformed columns: their weights do not exist in W1m).  scratch: sx_rqs_slab_l1_scratch_floats(dim, hidden) floats.
 *   dim <= 64, hidden <= 128. */
size_t sx_rqs_slab_l1_scratch_floats(int32_t dim, int32_t hidden);
int sx_rqs_slab_l1_bwd(const float *slab_scratch, const float *h, int64_t ld_h, int32_t hidden, const float *x, const float *gout,
                       const float *w1t, const uint32_t *cond_mask, float *gx, float *dW1, int64_t ldw, float *db1,
                       const int32_t *col_map, int32_t n_live, int64_t n_rows, int32_t dim, const float *scale, float *scratch,
                       uint32_t *err_flag, void *stream);

/* One rational-quadratic spline COUPLING evaluated from the conditioner's LAST HIDDEN ACTIVATION, the parameter tensor never in HBM
 * (Coupling(Spline(spline_type='quadratic')).forward / inverse with their log-dets, stribor/flows/coupling.py:69-95 +
 * flows/spline.py:76-143 + util/rational_quadratic_spline.py:11-251, for conditioners whose last Linear is
 * params = h W2^T + b2 with up to 256 hidden units -- net/mlp.py:48-58 takes any width): the forward counterpart of sx_rqs_slab_bwd,
 * same slabs, same slots, same w_fwd pack (k_tiles = ceil(hidden/32) <= 8).
 *   x [n_rows, dim] fp32, n_bins <= 16;
 *   h, h_fragments = 0: [n_rows, ld_h] fp32, `hidden` <= 256 valid features (any conditioner: whatever produced its last hidden
 *   activation); h_fragments = 1: the fragments sx_rqs_slab_hidden wrote (ld_h ignored) -- every slab re-reads h, and in this form
 *   it is neither split into fp16 parts nor range-checked again;
 *   y [n_rows, dim]: the transformed columns (live_idx / live_start, n_live as sx_rqs_coupling) are written, and y = x in the
 *   n_pass columns pass_idx names (device array; the columns the coupling leaves alone; n_pass = 0: the caller fills them);
 *   reverse = 0: forward (bin searched on the widths, x in [left, right]), 1: inverse (heights, [bottom, top]);
 *   ldj (nullable) [n_rows] = (ldj_accumulate ? ldj : 0) + ldj_scale * sum over the transformed columns of log|d y / d x|
 *   (reverse: of the inverse map, as rational_quadratic_spline.py:232-234 returns it), summed in slab order;
 *   scratch: sx_rqs_slab_fwd_scratch_floats(n_rows, n_live) floats (needed when ldj is not NULL), caller-owned;
 *   err_flag (nullable) receives SX_FLAG_F16_RANGE when |h| leaves fp16's range (those rows' outputs are NaN);
 *   cubic != 0: MONOTONE CUBIC splines (util/cubic_spline.py:21-251, spline_type='cubic', the reference's default; as
 *   sx_cubic_coupling: domain [left, right] on both sides, 2K+2 parameters per element, slots as sx_rqs_slab_bwd's xout form;
 *   reverse = 2: the inverse with MINUS the forward log-det re-evaluated at the inverted point, flow.py:42-47).
 * sx_rqs_slab_hidden: the hidden layer of a single-hidden-layer conditioner (net/mlp.py:48-58; coupling.py:61-65: it sees
 * cat[x * mask, latent]),  h = act(W1 z + b1), as fp16 hi / lo MFMA fragments: h_frag [sx_rqs_slab_hidden_floats(n_rows, hidden)].
 *   w1: sx_pack_linear(W1, b1, row_idx = hidden slots, col_idx = input slots (slot q = column q of x, then of latent; -1 = masked),
 *   m_tiles = ceil(hidden/32), k_tiles = ceil((dim + latent_dim)/32), SX_GEMM_F16X3);  dim + latent_dim <= 128;  act: SX_ACT_*;
 *   cond_mask (HOST pointer, nullable = every slot): four 32-bit words, bit q = input slot q feeds the layer.  The other inputs are
 *   zeroed before the fp16 x 3 split -- coupling.py:61 multiplies them by mask = 0, whatever their magnitude --, and a sample whose
 *   conditioning input is beyond fp16's range is rescaled by a power of two inside the kernel: any finite fp32 row is valid
 *   (net/mlp.py:65).  err_flag receives SX_FLAG_F16_RANGE only for a hidden ACTIVATION beyond 65504 (unbounded activations). */
size_t sx_rqs_slab_hidden_floats(int64_t n_rows, int32_t hidden);
int sx_rqs_slab_hidden(const float *x, const float *latent, const float *w1, const uint32_t *cond_mask, float *h_frag,
                       int64_t n_rows, int32_t dim, int32_t latent_dim, int32_t hidden, int32_t act, uint32_t *err_flag, void *stream);
size_t sx_rqs_slab_fwd_scratch_floats(int64_t n_rows, int32_t n_live);
int sx_rqs_slab_fwd(const float *x, const float *h, int64_t ld_h, int32_t hidden, const float *w_fwd, float *y, float *ldj,
                    const int32_t *live_idx, int32_t live_start, int32_t n_live, const int32_t *pass_idx, int32_t n_pass,
                    int32_t n_bins, float left, float right,
                    float bottom, float top, int64_t n_rows, int32_t dim, int32_t reverse, float ldj_scale, int32_t ldj_accumulate,
                    int32_t h_fragments, int32_t cubic, float *scratch, uint32_t *err_flag, void *stream);

/* Monotone cubic spline, element-wise part -- spline_type='cubic', the reference's default
 * (stribor/util/cubic_spline.py:21-251, util/search_sorted.py:3-5, flows/spline.py:59-61,82-86).
 *   params[n, i*(2K+2) + 0:K]      unnormalised widths  of live dim i
 *   params[n, i*(2K+2) + K:2K]     unnormalised heights
 *   params[n, i*(2K+2) + 2K:2K+2]  unnormalised boundary derivatives (left, right)   (K = n_bins)
 *   domain = codomain = [lower, upper]; outside it: y = x, ljd = 0.  ldiag / ldj / reverse as in sx_rqs_coupling.
 *   reverse == 2: the inverse with the log-det the reference's Transform.inverse_and_log_det_jacobian returns
 *   (flow.py:42-47): MINUS the FORWARD log-derivative re-evaluated at the inverted point -- identical to the inverse's own
 *   value inside a bin, different where the inverted point rounds out of the domain (0) or into a neighbouring bin. */
int sx_cubic_coupling(const void *x, void *y, float *ldj, float *ldiag, const float *params,
                      int64_t params_stride, const int32_t *live_idx, int32_t live_start, int32_t n_live, int32_t n_bins,
                      float lower, float upper, int64_t n_rows, int32_t dim, int32_t dtype, int32_t reverse,
                      int32_t ldj_accumulate, float ldj_scale, void *stream);

/* Backward of sx_cubic_coupling(reverse = 1) for training: the cubic solve is differentiated implicitly, so the kernel is
 * handed the inverse pass's input `yin` AND its output `xout` (both [n_rows, dim] fp32); other arguments as
 * sx_rqs_inverse_bwd, gparams [n_rows, n_live*(2K+2)]. */
int sx_cubic_inverse_bwd(const float *yin, const float *xout, const float *gout, const float *gldj, const float *gldiag,
                         const float *params,
                         int64_t params_stride, float *gx, float *gparams, const int32_t *live_idx, int32_t live_start,
                         int32_t n_live, int32_t n_bins, float lower, float upper, int64_t n_rows, int32_t dim,
                         float ldj_scale, void *stream);

/* The same for sx_cubic_coupling(reverse = 0), the forward direction (out = f(t), ljd = log f'(t), bin searched on the
 * widths): x [n_rows, dim] the forward pass's input; no solve to differentiate. */
int sx_cubic_forward_bwd(const float *x, const float *gout, const float *gldj, const float *gldiag, const float *params,
                         int64_t params_stride,
                         float *gx, float *gparams, const int32_t *live_idx, int32_t live_start, int32_t n_live,
                         int32_t n_bins, float lower, float upper, int64_t n_rows, int32_t dim, float ldj_scale, void *stream);

/* Parameter-free element-wise flows: Sigmoid / Logit (stribor/flows/sigmoid.py:9-56), ELU / LeakyReLU
 * (flows/activations.py:11-101), Cumsum / Diff over the last axis (flows/cumsum.py:9-92).
 *   y (nullable for the element kinds): transformed values;  ldiag (nullable, [n_rows, dim]): per-element log-derivative;
 *   ldj (nullable, [n_rows]): its row sum, ldj[n] = (ldj_accumulate ? ldj[n] : 0) + sum_d ldiag[n, d].
 *   Forward kinds return the forward log-derivative at x; the *_INV kinds and LOGIT return what
 *   Transform.inverse_and_log_det_jacobian does (flow.py:42-47): minus the forward log-derivative at the value produced.
 *   param: LeakyReLU's negative_slope (SX_PW_LEAKY_RELU) or its reciprocal (SX_PW_LEAKY_RELU_INV); ignored otherwise. */
#define SX_PW_SIGMOID 1
#define SX_PW_LOGIT 2
#define SX_PW_ELU 3
#define SX_PW_ELU_INV 4
#define SX_PW_LEAKY_RELU 5
#define SX_PW_LEAKY_RELU_INV 6
#define SX_PW_CUMSUM 7
#define SX_PW_DIFF 8
int sx_pointwise(const void *x, void *y, float *ldj, float *ldiag, int64_t n_rows, int32_t dim, int32_t dtype,
                 int32_t kind, float param, int32_t ldj_accumulate, void *stream);

/* Backward of sx_pointwise (training): gx = gy * d(out)/dx + gldj[row] * d(log-derivative)/dx; x, gy, gx [n_rows, dim]
 * fp32, gldj [n_rows] nullable, gldiag [n_rows, dim] nullable (the adjoint of the per-element log-derivative, ABI v3).
 * Cumsum / Diff: the reversed scan of gy. */
int sx_pointwise_bwd(const float *x, const float *gy, const float *gldj, const float *gldiag, float *gx, int64_t n_rows, int32_t dim,
                     int32_t kind, float param, void *stream);

/* out[0] = max(out[0], max |a[0..na)|, max |b[0..nb)|) (b nullable; fp32; `out` holds a non-negative value, e.g. 0, on entry): the
 * magnitude of a backward pass's incoming adjoints (dL/dy [n_rows, dim] and dL/dlog-det [n_rows]), from which sx_rqs_slab_bwd
 * derives its power-of-two normalisation; one launch over every element. */
int sx_absmax2(const float *a, int64_t na, const float *b, int64_t nb, float *out, void *stream);

/* UnitNormal.log_prob + log-det accumulator (stribor/dist/normal.py:37,52-54; flow.py:128-129):
 *   out[n] = sum_d( -x[n,d]^2/2 ) - dim*log(sqrt(2*pi)) + (ldj ? ldj[n] : 0) */
int sx_unit_normal_logprob(const void *x, const float *ldj, float *out, int64_t n_rows, int32_t dim,
                           int32_t dtype, void *stream);

/* sum of an fp32 vector into ONE fp64 (block partials in fp64, one atomic per block);
 * `*out` must be zeroed by the caller (or hold a running total). */
int sx_sum_f64(const float *v, int64_t n, double *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * MFMA path: fragment packing + the fused flow kernel
 * ------------------------------------------------------------------------------------------ */

/* Number of floats sx_pack_linear writes for an (m_tiles*32) x (k_tiles*32) padded linear. */
size_t sx_packed_linear_floats(int32_t m_tiles, int32_t k_tiles);

/* Re-lays one nn.Linear (W: [out_dim, in_dim] row-major as torch stores it, b: [out_dim] or NULL)
 * into MFMA A-operand fragment order for the GEMM arithmetic `precision` (SX_GEMM_*), followed by the bias
 * in C-fragment order.  4 KiB per 32x32 tile pair in both modes.  With SX_GEMM_F16X3 a (scaled) weight beyond fp16's
 * range ORs SX_FLAG_F16_RANGE into `err_flag` (nullable) and packs as inf, so every product it enters is non-finite.
 *   row_idx[m_tiles*32]: output slot -> row of W, or -1 for a zero row (padding / pruned)
 *   col_idx[k_tiles*32]: input  slot -> column of W, or -1 for a zero column
 * With kmap(s,h) = (s&3) + 8*(s>>2) + 4*h (the C/D fragment row map):
 *   mode 0 (v_mfma_f32_32x32x2_f32), floats: A[m][kt][g][lane][e] = W[row_idx[32m + (lane&31)]][col_idx[32kt + kmap(4g+e, lane>>5)]]
 *   mode 1 (v_mfma_f32_32x32x16_f16), halfs: A[m][kt][s][p][lane][j] = part_p(W[row_idx[32m + (lane&31)]][col_idx[32kt + kmap(8s+j, lane>>5)]]),
 *          p = 0: hi = fp16(w), p = 1: lo = fp16(w - hi)
 *   then (floats) bias[m][h][r] = b[row_idx[32m + kmap(r,h)]].
 * Optional exact re-parametrisation (all NULL / 0 = plain copy):
 *   row_scale[m_tiles*32]  : A rows are multiplied by row_scale[slot]
 *   bias_scale[m_tiles*32], fold_ones: bias' = bias_scale[slot] * (b[row] + fold_ones * sum_live_cols W[row][col])
 * used to fold the constants of tanh(z) = 1 - 2/(exp2(2 log2(e) z) + 1) and exp(x) = exp2(log2(e) x) into the
 * weights so the kernel spends no VALU cycles on them (SX_ACT_TANH_FOLDED).
 * transpose != 0 packs W^T (row slots index W's columns, column slots W's rows; b must be NULL): the operands
 * of the backward pass (dh = W2^T dp, dz = W1^T dh_pre). */
int sx_pack_linear(const float *W, const float *b, int32_t out_dim, int32_t in_dim,
                   const int32_t *row_idx, const int32_t *col_idx, int32_t m_tiles, int32_t k_tiles,
                   const float *row_scale, const float *bias_scale, float fold_ones, int32_t transpose,
                   int32_t precision, uint32_t *err_flag, float *dst, void *stream);
/* The same, and the bound the spline phases' softmax relies on (replaces the running maximum of torch.softmax in
 * stribor/util/rational_quadratic_spline.py:101-105): *bound_out = max(*bound_out, max over the packed rows of the largest |output|
 * the row can produce from inputs in [0, 1]^k), i.e. |bias' + sum of the positive (negative) packed weights|.  bound_out: one device
 * float, zeroed once by the caller; it only grows (a re-pack after a parameter update keeps the larger value: conservative). */
int sx_pack_linear_bound(const float *W, const float *b, int32_t out_dim, int32_t in_dim,
                         const int32_t *row_idx, const int32_t *col_idx, int32_t m_tiles, int32_t k_tiles,
                         const float *row_scale, const float *bias_scale, float fold_ones, int32_t transpose,
                         int32_t precision, uint32_t *err_flag, float *dst, float *bound_out, void *stream);

/* A table of sx_pack_linear / sx_pack_linear_bound calls in ONE launch (a training step re-packs every Linear of a flow -- 13 per
 * spline coupling -- after each optimizer step; net/mlp.py:48-66 has no such step, it is the price of the fragment layout).  `jobs`:
 * DEVICE array of n_jobs records, every pointer a device pointer with the meaning of the same-named sx_pack_linear argument
 * (bound_out NULL = no bound); the records are read when the kernel runs, i.e. the table must stay alive and unchanged until then.
 * max_floats = the largest sx_packed_linear_floats(m_tiles, k_tiles) in the table.  Sizes and pointers inside the table are the
 * caller's contract (the kernel cannot report them): build it from the same values the single calls would take. */
typedef struct sx_pack_job {
    const float *W, *b;
    const int32_t *row_idx, *col_idx;
    const float *row_scale, *bias_scale;
    float *dst;
    float *bound_out;
    int32_t out_dim, in_dim, m_tiles, k_tiles, transpose;
    float fold_ones;
} sx_pack_job;
int sx_pack_linear_batch(const sx_pack_job *jobs, int32_t n_jobs, int32_t max_floats, int32_t precision, uint32_t *err_flag,
                         void *stream);

/* One step of a fused flow program.  The flow state lives in registers as 32-wide "tiles" of columns (tile t =
 * state slots 32t..32t+31); slots map to columns of x through in_col/out_col.  Field use per kind: */
#define SX_STEP_COUPLING_AFFINE      1  /* affine coupling (coupling.py:69-95 + affine.py:104-109): conditioner tiles [c0,c0+ct),
                                           transformed tiles [t0,t0+tt) (low / high halves or all tiles);
                                           blob = pack_linear(W1: h_tiles x ct) ++ pack_linear(W2: per transformed tile 32 log_scale
                                           rows then 32 shift rows, x h_tiles); act = SX_ACT_* or SX_ACT_TANH_FOLDED          */
#define SX_STEP_AFFINE_CONST         2  /* st.Affine without latent_net: blob = ls[tiles][2][16] ++ sh[tiles][2][16] (C-fragment) */
#define SX_STEP_LINEAR_TILE          3  /* blob = pack_linear(M, x_tiles m-tiles x tiles) ++ {log|det M| term, 1 float}: state = M . state
                                           + bias, ldj += blob's log-det term (already signed / scaled; it lives in the blob so
                                           that it is refreshed together with M), the whole layer in one step (act = x_tiles,
                                           t0 = 0; AffineLU affine.py:157-171, MatrixExponential affine.py:243-288)               */
#define SX_STEP_MLP_HIDDEN           5  /* blob = pack_linear(W, h_tiles x tiles): hidden = act(W . state + b)  (mlp.py:65)       */
#define SX_STEP_MLP_HIDDEN2          6  /* blob = pack_linear(W, h_tiles x h_tiles): hidden' = act(W . hidden + b)                 */
#define SX_STEP_MLP_OUT_TILE         7  /* blob = pack_linear(W, 1 x h_tiles): mlp_out[:, 32*t0 ..] = W[t0] . hidden + b (reverse = 1: += -- a later
                                         * hidden-unit chunk of a conditioner wider than the hidden tiles)                             */
#define SX_STEP_ROW_SCALE_EXP        9  /* blob = diag[tiles][2][16]: state *= exp(+-diag * t_row); t_row = row_t[n] or ldj_const;
                                           act != 0 applies log1p|t| (affine.py:239-240)                                          */
#define SX_STEP_RQS_HIDDEN          10  /* blob = pack_linear(W1, folded tanh): hidden of a spline coupling, kept for its phases.
                                         * pad_ bits 8..15 = the layer's ordinal L among the program's spline couplings: with mlp_out given the
                                         * step writes tanh h to rows [L n_rows, (L + 1) n_rows) of mlp_out, and with `side` given (programs
                                         * without a backward step) the state it received to rows [(L - 1) n_rows, L n_rows) of `side`
                                         * ([., dim] fp32, L >= 1): the saved tensors of a layer-by-layer backward from one forward launch.
                                         * Side outputs exist for pure spline programs with one hidden layer per conditioner (they run
                                         * their own kernel instances, so that the inference kernels carry none of that code).            */
#define SX_STEP_RQS_PHASE           11  /* one (8-column group, parameter block) slab of a rational-quadratic spline coupling:
                                           blob = pack_linear(W2 rows, 4 m-tiles) ++ {lo, hi} (phases 0 and 1: rows and bias
                                           times log2(e) -- the kernel's softmax runs in base 2); t0 = tile, c0 = group 0..3,
                                           ct = phase (0 search block, 1 select block, 2 derivatives + evaluate), tt = n_bins,
                                           pad_ = live mask of the tile's 32 slots.
                                           act = 1: the monotone CUBIC spline (util/cubic_spline.py:21-251, 2K+2 parameters per
                                           element): phase 0 = searched block (widths forward / heights inverse), 1 = the other
                                           block, 2 = the two boundary-derivative parameters ++ {lower, upper} + evaluate; a
                                           program holds phases of one spline type only                                          */
#define SX_STEP_COUPLING_AFFINE_BWD 12  /* backward of one affine coupling of a log_prob pass (training): state tiles
                                           [0,2) = x, [2,4) = dL/dx; blob = forward blob ++ pack(W2^T) ++ pack(W1^T); c0 = cond
                                           tile, t0 = transformed tile, tt = layer slot in `side` (see sx_flow_run)             */
#define SX_STEP_COUPLING_AFFINE_BWD_A 16 /* 128-column flows (x_tiles = 4, tiles = 8): the backward of one affine coupling in two steps, since
                                         * its four packed operands do not fit the LDS ring together.  A: blob = pack(W1') ++ pack(W2') (as
                                         * kind 12's first two): conditioner, un-transform, dL/d(log_scale, shift) (kept in registers for B);
                                         * c0 = first conditioner tile (0 or 2; the transformed tiles are the other half), tt = layer slot
                                         * of the side buffer.                                                                          */
#define SX_STEP_COUPLING_AFFINE_BWD_B 17 /* B: blob = pack(W2^T) ++ pack(W1^T): dL/dh, dL/dh_pre, adjoint of the conditioning tiles; same c0 / tt;
                                         * directly behind its A step (the kernel runs the pair in one iteration of its step loop)       */
#define SX_STEP_LINEAR_BWD          18  /* dense linear layer of a backward program (AffineLU / MatrixExponential, affine.py:156-171,243-288):
                                         * tiles [c0, c0 + 4) <- M . tiles + b, blob = pack_linear(M, 4 x 4) (c0 = 0: the x tiles, M = the
                                         * layer's forward matrix; c0 = 4: the adjoint tiles, M = W^T of the matrix log_prob applied).  The
                                         * factors of dL/dW = sum_n dL/du_n v_n^T are stored to the side buffer of layer slot tt at feature
                                         * offset 32 * t0: before the step when reverse = 1 (the adjoint), after it otherwise (v).  The two
                                         * steps of a layer come back to back, c0 = 0 first.                                             */
#define SX_STEP_POINTWISE           19  /* point-wise flow on the data tiles (Sigmoid / Logit sigmoid.py:9-56, ELU / LeakyReLU activations.py:11-101):
                                         * act = SX_PW_* kind (the direction is the kind), ldj_const = param (LeakyReLU slope or its reciprocal),
                                         * blob = live-slot mask [tiles][2][16] (C-fragment order: 1 = column, 0 = padding) ++ {log-slope};
                                         * the accumulator receives ldj_scale * (the kind's own log-derivative sum, as sx_pointwise).     */
#define SX_STEP_COUPLING_TIME       20  /* time-conditioned affine coupling (ContinuousAffineCoupling, coupling.py:184-213): conditioner over ALL tiles
                                         * (ct = tiles; the two slots behind the latent columns hold t and t0, filled from row_t and -- programs of
                                         * this kind only -- from `side`, read as a second [n_rows] time vector), transformed tiles [0, tt) = the data
                                         * tiles; blob = pack(W1, h_tiles x tiles) ++ pack(W2, 2 tt x h_tiles) ++ the time net's per-column constants
                                         * (see sx_flow_kernel.h); pad_ = time kind (0 identity, 1 linear, 2 tanh, 3 log, 4 fourier) | (which time
                                         * the embedding uses: 0 row_t, 1 side) << 8 | (fourier features) << 16.  Programs of these steps only.   */
#define SX_STEP_COUPLING_AFFINE_HC  21  /* one hidden-unit CHUNK of an affine coupling whose hidden layer is wider than the program's hidden tiles
                                        * (coupling.py:69-95 + mlp.py:65 with hidden > 32 h_tiles; Tanh conditioners): tiles as
                                        * SX_STEP_COUPLING_AFFINE; blob = pack(W1 rows of the chunk) ++ pack(W2 columns of the chunk, bias = b2
                                        * in the first chunk); pad_ bit 0 = first chunk, bit 1 = last chunk (applies the affine map).  The
                                        * chunks of a coupling come back to back. */
#define SX_STEP_WIDE_HIDDEN         22  /* programs on EIGHT state tiles (129 .. 256 columns; tiles = x_tiles = 8): the hidden layer of an affine coupling
                                        * whose mask splits the tiles, r = folded tanh(W1' . tiles [c0, c0 + 4) + b1') (c0 = 0 or 4, ct = 4), kept in
                                        * registers for the SX_STEP_WIDE_AFFINE_TILE steps behind it; blob = pack(W1', h_tiles x 4) */
#define SX_STEP_WIDE_AFFINE_TILE    23  /* ... and ONE transformed tile t0 of that coupling: (kk log_scale, shift) = W2'[rows of the tile] . r + b2', the
                                        * affine map (coupling.py:69-95, affine.py:104-109), the log-det; blob = pack(W2' rows, 2 x h_tiles) */
#define SX_STEP_MLP_INPUT           24  /* MLP programs (round 6): hidden = state -- the program's input tiles ARE the operand of its
                                         * SX_STEP_MLP_OUT_TILE steps, i.e. ONE nn.Linear y = W x + b (net/mlp.py:48-58 layer by layer, the
                                         * [N, D] x [D, D] products of the layer-wise training path) as a hand-written MFMA program
                                         * instead of a library GEMM; tiles <= h_tiles; first step of its program; blob: unused (>= 256 floats) */
#define SX_STEP_CPL_HIDDEN          13  /* deep conditioners (>= 2 hidden layers): hidden = act(W1 . state[c0..c0+ct) + b1), kept in
                                           registers for the next step; blob = pack_linear(W1, h_tiles x ct)                    */
#define SX_STEP_CPL_HIDDEN2         14  /* hidden = act(Wk . hidden + bk); blob = pack_linear(Wk, h_tiles x h_tiles)              */
                                        /* (in spline programs SX_STEP_RQS_HIDDEN with pad_ == 1 is the LAST hidden layer, fed by these) */
#define SX_STEP_COUPLING_AFFINE_DEEP 15 /* the coupling's last hidden layer (from the kept hidden state), output layer and affine
                                           map: blob = pack_linear(W_L, h_tiles x h_tiles) ++ pack_linear(W_out, 2*tt x h_tiles) */

#define SX_MAX_STEPS 128

typedef struct sx_step {
    int32_t  kind;        /* SX_STEP_*                                                        */
    int32_t  c0, ct;      /* conditioner input tiles [c0, c0+ct)    (other meanings: see the kinds)   */
    int32_t  t0, tt;      /* transformed / output tiles [t0, t0+tt) (other meanings: see the kinds)   */
    int32_t  reverse;     /* 1: inverse direction ((x-sh)*exp(-ls)), 0: forward               */
    int32_t  act;         /* SX_ACT_* of the hidden layer                                     */
    uint32_t blob_off;    /* offset of this step's blob in `blobs`, in floats (multiple of 256, >= 256) */
    uint32_t blob_floats; /* size of the blob in floats (multiple of 256: 1 KiB LDS-DMA pieces) */
    float    ldj_scale;   /* coefficient of this step's sum(log_scale) in the ldj accumulator  */
    float    ldj_const;   /* constant added to the ldj accumulator (AffineLU / MatrixExponential) */
    int32_t  pad_;        /* SX_STEP_RQS_PHASE: live-slot mask; else 0                         */
} sx_step;

typedef struct sx_program {
    int32_t n_steps;
    int32_t dim;          /* D: columns of x / y                                               */
    int32_t latent_dim;   /* columns of `latent` (0 = none); they occupy tiles after the data  */
    int32_t x_tiles;      /* tiles holding data columns                                        */
    int32_t tiles;        /* x_tiles + latent tiles: 1, 2 or 4 (8: backward programs of 128-column flows) */
    int32_t h_tiles;      /* hidden width / 32 rounded up to 1, 2 or 4                         */
    int32_t identity_cols;/* 1: state slot p <-> column p (vector loads), 0: use in_col/out_col */
    int32_t pad_;         /* 0, or the row stride of x in elements (>= dim; needs in_col): the program reads a column subset of wider rows */
    sx_step steps[SX_MAX_STEPS];
} sx_program;

/* Runs a fused program over n_rows samples: the whole flow stays in registers, weights stream
 * through LDS, every GEMM runs on the matrix cores in the arithmetic `precision` (SX_GEMM_*; `blobs` must have been
 * packed for the same one).  A program holds flow steps
 * (kinds 1, 2; + 3, 9 for dense linear layers; or 10, 11 for spline couplings), or conditioner steps (5-7),
 * or backward steps (12) -- the kernel variant is picked from the kinds present.
 * Replaces NormalizingFlow.{forward, inverse, forward_and_log_det_jacobian,
 * inverse_and_log_det_jacobian, log_prob} (stribor/flow.py:99-130) and, with a one-step program,
 * Coupling.{forward, inverse, log_det_jacobian} (stribor/flows/coupling.py:69-95).
 *   blobs    the steps' packed weights (sx_step.blob_off), behind a 1 KiB header: word 0 of the header holds SX_FLAG_*
 *            bits raised while packing (point sx_pack_linear's err_flag at it and zero it before re-packing); every
 *            launch ORs it into err_flag, so weights beyond the fp16 x 3 range are reported by each call that uses them
 *   x        [n_rows, dim]  input (dtype)
 *   latent   [n_rows, latent_dim] fp32 or NULL
 *   in_col   int32[x_tiles*32]: state slot -> column of x (-1 = zero pad); NULL if identity_cols
 *   out_col  int32[x_tiles*32]: state slot -> column of y (-1 = not stored); NULL if identity_cols
 *   y        [n_rows, dim]  final state (dtype), or NULL
 *   ldj_out  [n_rows] accumulated log-det terms, or NULL
 *   logp_out [n_rows] UnitNormal.log_prob(final state) + accumulated log-det terms, or NULL
 *   sum_out  one fp64: += sum_n logp_out[n] (or of ldj when logp_out is NULL); NULL to skip
 *   row_t    [n_rows] per-sample time of SX_STEP_ROW_SCALE_EXP steps (MatrixExponential with a tensor t,
 *            stribor/flows/affine.py:236-241), or NULL to use the step's constant
 *   side     training backward only (programs of SX_STEP_COUPLING_AFFINE_BWD steps):
 *            [n_layers, ceil(n_rows / 32), side_width, 32] fp32 -- 32-row groups, feature-major inside; the features
 *            of a row are [z(32) | tanh h(32*h_tiles) | dL/dh_pre(32*h_tiles) | dL/d(log_scale, shift)(64)] in slot
 *            order, from which sx_wgrad forms the weight gradients; row_t carries dL/dlog_prob;
 *            x is the flow's latent z, y receives dL/d(input)
 *   mlp_out  [n_rows, mlp_out_dim] (row stride mlp_out_stride) destination of SX_STEP_MLP_OUT_TILE
 *            steps, or NULL
 *   work     two uint32 {next-chunk ticket, finished workgroups}, zero before the first launch that uses them, owned by
 *            the caller and private to one stream (launches on a stream never overlap): the persistent workgroups take
 *            row chunks from it dynamically and the last one out zeroes the pair again, so launches need no memset and
 *            replay from HIP graphs.  NULL = static chunk stride.  After a failed launch zero the pair again.
 *   err_flag caller's flag word (device memory or host-mapped pinned memory), or NULL: SX_FLAG_F16_RANGE is OR-ed in
 *            (system-scope atomic) when a sample's operands left the fp16 x 3 range; such rows come back as NaN      */
int sx_flow_run(const sx_program *prog_host, const float *blobs, const void *x, const float *latent,
                const int32_t *in_col, const int32_t *out_col, void *y, float *ldj_out,
                float *logp_out, double *sum_out, float *mlp_out, int64_t mlp_out_stride,
                int32_t mlp_out_dim, const float *row_t, float *side, int64_t n_rows, int32_t dtype,
                int32_t precision, uint32_t *work, uint32_t *err_flag, void *stream);

/* sx_flow_run with the EXACT REDO PASS (round 6): any finite fp32 through the default arithmetic at fp32-grade error
 * (net/mlp.py:65, flows/affine.py:104-109,156-163 take any finite value; stribor has no operand range).
 *   blobs        packed for SX_GEMM_F16X3, blobs_exact the same program's weights packed for SX_GEMM_F32 (same offsets);
 *   redo         caller-owned DEVICE scratch of sx_flow_redo_words(n_rows) 32-bit words, zeroed once; the call leaves it zeroed
 *                (one list per stream: launches that share a list must be ordered).
 * precision must be SX_GEMM_F16X3.  The fp16 x 3 kernel does not flag a sample whose operand (flow state, conditioner input, an
 * unbounded hidden activation) exceeds 2048 (SX_REDO_ABOVE, sx_flow_kernel.h: far inside fp16's range -- a weight below 0.125 is held
 * to an absolute 3e-8 only, an error that grows with the entries it multiplies: 1.4e-4 relative measured on a row of 6.4e4, 4.5e-6 at
 * 2048) -- and does not rescale it either --: it appends the sample's 32-row group and a per-sample mask to `redo`,
 * and a second launch of the same program on the exact-fp32 kernel evaluates, stores (y, ldj_out, logp_out, mlp_out) and sums
 * (sum_out) exactly those samples.  On ordinary data the second launch reads one word per workgroup and ends.
 * Not for programs with side outputs (`side`).  MLP programs that ACCUMULATE into mlp_out (later hidden chunks) leave the named
 * samples' rows to the exact pass, which adds the chunk's exact contribution.
 * blobs_exact = redo = NULL: sx_flow_run (such samples come back as NaN + SX_FLAG_F16_RANGE). */
size_t sx_flow_redo_words(int64_t n_rows);
int sx_flow_run2(const sx_program *prog_host, const float *blobs, const float *blobs_exact, uint32_t *redo, const void *x,
                 const float *latent, const int32_t *in_col, const int32_t *out_col, void *y, float *ldj_out,
                 float *logp_out, double *sum_out, float *mlp_out, int64_t mlp_out_stride, int32_t mlp_out_dim,
                 const float *row_t, float *side, int64_t n_rows, int32_t dtype, int32_t precision,
                 uint32_t *work, uint32_t *err_flag, void *stream);

/* Training backward of log_prob for flows of affine couplings, layer-major: one launch per layer (or pair) with the weight gradients
 * contracted inside the kernel (the single-launch program of SX_STEP_COUPLING_AFFINE_BWD steps run through sx_flow_run
 * writes 896 B of per-row factors per layer for sx_wgrad_layer to read back; here only the state crosses HBM between
 * launches).  `prog_host`: 1 .. sx_flow_bwd_max_steps() SX_STEP_COUPLING_AFFINE_BWD steps (conditioner / transformed columns = the two
 * 32-column halves, hidden <= 64), fp16 x 3 blobs.
 *   z / frag_in   first launch: the flow's latent z [n_rows, dim] fp32 (the adjoint starts as -g z); later launches: the
 *                 state the previous launch left in frag_out -- [ceil(n_rows / 32)][4 tiles][4][64 lanes] float4
 *                 (x tiles 0, 1 then dL/dx tiles 0, 1; 16 KB per 32 rows)
 *   g             [n_rows] dL/dlog_prob
 *   frag_out / gy state for the next launch, or (last launch) dL/d(input) [n_rows, dim]
 *   acc_out       per step part_floats * n_part floats (sx_flow_bwd_partials): n_part tiles of [64 x 32 h_tiles | 64]
 *                 (dW2 rows: 32 log_scale then 32 shift slots; db2) followed by n_part tiles of [32 h_tiles x 32 |
 *                 32 h_tiles] (dW1, db1): the two inputs of sx_wgrad_reduce for that layer                         */
int sx_flow_bwd_max_steps(void);      /* layers per launch this build supports (register budget: 1) */
int sx_flow_bwd_partials(const sx_program *prog_host, int64_t n_rows, int32_t *n_part, int64_t *part_floats);
int sx_flow_bwd_run(const sx_program *prog_host, const float *blobs, const float *z, const float *g,
                    const float *frag_in, float *frag_out, float *gy, float *acc_out, int64_t n_rows,
                    uint32_t *work, uint32_t *err_flag, void *stream);
/* dW[rm(i)][cm(j)] += sum_p part[p][i * N32 + j] (i < m_valid, j < n_valid), db[rm(i)] += sum_p part[p][M32 * N32 + i];
 * part: n_part tiles of M32 * N32 + M32 floats; one writer per element (deterministic). */
int sx_wgrad_reduce(const float *part, int32_t n_part, int32_t M32, int32_t N32, float *dW, int64_t ldw, float *db,
                    int32_t m_valid, int32_t n_valid, const int32_t *row_map, const int32_t *col_map, void *stream);
/* A table of such reductions in ONE launch (the 2 L reductions that close a layer-major backward pass of L couplings).  `jobs`:
 * DEVICE array; a record's part / dW / db are FLOAT OFFSETS from part_base / out_base (db_off < 0: no bias), so a table built once
 * serves every step whatever buffers the step allocated; M32 <= 128; every record sums the same number of partials n_part;
 * max_elems = the largest M32 * N32 + M32 of the table.  The table must stay alive and unchanged until the kernel has run. */
typedef struct sx_reduce_job {
    int64_t part_off, dW_off, db_off, ldw;
    const int32_t *row_map, *col_map;
    int32_t M32, N32, m_valid, n_valid;
} sx_reduce_job;
int sx_wgrad_reduce_batch(const float *part_base, float *out_base, const sx_reduce_job *jobs, int32_t n_jobs, int32_t n_part,
                          int32_t max_elems, void *stream);

/* Weight-gradient contraction over the batch axis (training):
 *   dW[rm(i), cm(j)] += sum_n A[n, i] * B[n, j]   (dW row stride ldw),   db[rm(i)] += sum_n A[n, i]   (db may be NULL)
 * A has M features, B has Nc <= 128 features, fp32, in one of two layouts:
 *   SX_WGRAD_ROW_MAJOR   plain [n_rows, ld] matrices (row strides lda / ldb): what autograd holds for a Linear layer
 *                        (A = dL/d(output), B = the layer's input); M <= 2048 (128-feature slabs of A, B re-read)
 *   SX_WGRAD_ROW_GROUPS  slices of `side` (sx_flow_run): 32-row groups, feature-major inside -- element (row n,
 *                        feature i) of A is A[(n >> 5) * lda + i * 32 + (n & 31)] (lda / ldb = floats per group,
 *                        16-byte aligned; whole 32-feature tiles are read, rows of the last group past n_rows are
 *                        ignored); M <= 128
 * row_map [M] / col_map [Nc] (device int32, either may be NULL = identity) send the operands' feature order straight
 * to the parameter's own rows / columns; negative entries are dropped.  dW / db are accumulated into (zero them
 * first) by one writer per element: results do not depend on scheduling.  `scratch` (caller-owned, at least
 * sx_wgrad_scratch_floats(M, Nc, layout) floats, private to the stream until the call has run) holds the per-workgroup
 * partial tiles.  Replaces autograd's dense matmuls over the batch axis for nn.Linear
 * inside stribor/net/mlp.py:48-58 (a tall-skinny A^T B that library GEMMs run on a handful of workgroups). */
#define SX_WGRAD_ROW_MAJOR 0
#define SX_WGRAD_ROW_GROUPS 1
/* the row-group layout with the contraction on v_mfma_f32_32x32x16_f16 (operands split hi + lo in fp16, three products, fp32
 * accumulate: ~2^-22 per product).  For the factors a backward program stored: they are fp16 x 3 GEMM operands of that program
 * already (within fp16's range, or flagged by it); 5x fewer matrix-pipe cycles than the fp32 MFMA form. */
#define SX_WGRAD_ROW_GROUPS_F16X3 3
size_t sx_wgrad_scratch_floats(int32_t M, int32_t Nc, int32_t layout);
int sx_wgrad(const float *A, int64_t lda, int32_t M, const float *B, int64_t ldb, int32_t Nc, int64_t n_rows,
             int32_t layout, float *dW, int64_t ldw, float *db, const int32_t *row_map, const int32_t *col_map,
             float *scratch, void *stream);

/* Both weight gradients of ONE coupling layer of a backward program in one pass over its slot of `side`
 * (32-row groups of [z (32 c_tiles) | tanh h (32 h_tiles) | dL/dh_pre (32 h_tiles) | dL/d(log_scale, shift) (64 t_tiles)],
 * ld floats per group): dW2 [rows through row_map2, `hidden` columns] += dparams^T tanh_h, db2 += sum dparams,
 * dW1 [`hidden` rows, columns through col_map1] += dh_pre^T z, db1 += sum dh_pre -- what two sx_wgrad calls on the
 * slices compute, with the group streamed once.  Shapes: c_tiles = t_tiles = 1, h_tiles <= 2 (SX_E_UNSUPPORTED
 * otherwise: use sx_wgrad).  scratch: >= sx_wgrad_layer_scratch_floats(c_tiles, h_tiles, t_tiles) floats. */
size_t sx_wgrad_layer_scratch_floats(int32_t c_tiles, int32_t h_tiles, int32_t t_tiles);
int sx_wgrad_layer(const float *side, int64_t ld, int64_t n_rows, int32_t c_tiles, int32_t h_tiles, int32_t t_tiles,
                   int32_t hidden, float *dW2, int64_t ldw2, float *db2, const int32_t *row_map2, float *dW1,
                   int64_t ldw1, float *db1, const int32_t *col_map1, float *scratch, void *stream);

/* out[j] += sum_n A[n, j] for a row-major fp32 [n_rows, M] matrix (row stride lda): the bias gradient of a Linear
 * layer too wide for sx_wgrad (autograd's grad_output.sum(0) behind stribor/net/mlp.py:48-58).  Deterministic, no
 * atomics; zero `out` first.  scratch: >= 256 * M floats. */
int sx_colsum(const float *A, int64_t lda, int64_t n_rows, int32_t M, float *out, float *scratch, void *stream);

/* X[b] = T[b]^-1 for a batch of row-major fp64 D x D triangular matrices (D <= 128; lower != 0: lower triangular,
 * else upper; unit != 0: the diagonal is taken as 1).  Entries of T outside the triangle are not read; X is written
 * in full (zeros outside the triangle).  Parameter preprocessing of AffineLU / MatrixExponential in training: replaces
 * the triangular solves of stribor/flows/affine.py:159-163, 254-266 against the identity. */
int sx_tri_inverse_f64(const double *T, double *X, int32_t batch, int32_t D, int32_t lower, int32_t unit, void *stream);

/* LDS bytes and grid the launcher will use for a program (introspection for tests/bench). */
int sx_flow_launch_info(const sx_program *prog_host, int64_t n_rows, int32_t *grid, int32_t *block,
                        int32_t *lds_bytes);

#ifdef __cplusplus
}
#endif
#endif /* STRIBOR_HIP_H */
