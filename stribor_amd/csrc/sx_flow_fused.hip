// Host dispatcher of the fused flow kernel (device code: sx_flow_kernel.h, one object per tile pair).
#include "sx_flow_types.h"
#include <stdlib.h>

// experiment knobs, read once per process
static const bool g_no_pure_mode = sx_debug_knob("SX_NO_PURE_MODE", 0) != 0;
static const bool g_static_chunks = sx_debug_knob("SX_STATIC_CHUNKS", 0) != 0;
static const int g_blocks_per_cu = sx_debug_knob("SX_BLOCKS_PER_CU", 0);


// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int validate_and_convert(const sx_program *p, dprog *d, int *buf_floats, int *mlp_mode, int *side_width) {
    *side_width = 0;
    SX_REQUIRE(p != nullptr, "sx_flow_run: null program");
    SX_REQUIRE(p->n_steps >= 0 && p->n_steps <= SX_MAX_STEPS, "sx_flow_run: n_steps %d out of range", p->n_steps);
    SX_REQUIRE(p->tiles == 1 || p->tiles == 2 || p->tiles == 4 || (p->tiles == 8 && (p->x_tiles == 4 || p->x_tiles == 8)),
               "sx_flow_run: tiles must be 1, 2 or 4 (8 = 4 data + 4 adjoint tiles of a backward program, or 8 data tiles of a hidden-chunk program; got %d)", p->tiles);
    SX_REQUIRE(p->h_tiles == 1 || p->h_tiles == 2 || p->h_tiles == 4, "sx_flow_run: h_tiles must be 1, 2 or 4");
    SX_REQUIRE(p->tiles != 8 || p->x_tiles == 8 || p->h_tiles <= 2, "sx_flow_run: backward programs of 128-column flows are built for hidden <= 64");
    SX_REQUIRE(p->x_tiles >= 1 && p->x_tiles <= p->tiles, "sx_flow_run: bad x_tiles");
    SX_REQUIRE(p->dim >= 1 && p->dim <= 32 * p->x_tiles, "sx_flow_run: dim %d does not fit %d tiles", p->dim, p->x_tiles);
    SX_REQUIRE(p->latent_dim >= 0 && p->latent_dim <= 32 * (p->tiles - p->x_tiles), "sx_flow_run: latent_dim does not fit");
    SX_REQUIRE(!p->identity_cols || p->dim % 4 == 0, "sx_flow_run: identity_cols needs dim %% 4 == 0");
    d->n_steps = p->n_steps; d->dim = p->dim; d->latent_dim = p->latent_dim; d->x_tiles = p->x_tiles;
    // pad_: row stride of x in elements when the program reads a column subset of wider rows (MLP programs of couplings wider than
    // the state tiles: in_col then holds columns of the wide row); 0 = dim
    SX_REQUIRE(p->pad_ == 0 || (p->pad_ >= p->dim && !p->identity_cols), "sx_flow_run: x row stride %d needs in_col and >= dim", p->pad_);
    d->identity_cols = p->identity_cols; d->pad = p->pad_;
    int mx = 256;
    bool lin = false, rqs = false, aff = false, bwd = false, deep = false, cub = false, quadr = false;
    int n_bwd128 = 0;
    bool pw = false, timed = false, wide_rq = false, hc = false;
    int n_hc8 = 0;
    *mlp_mode = 0;
    for (int i = 0; i < p->n_steps; ++i) {
        const sx_step &s = p->steps[i];
        SX_REQUIRE(s.blob_off % 256 == 0 && s.blob_floats % 256 == 0 && s.blob_off >= 256,
                   "sx_flow_run: step %d blob not 1 KiB aligned behind the 1 KiB header", i);
        // (first: the sums below must not overflow -- found by the host sanitizer job with c0 = INT_MAX)
        // the device step keeps these in 8 bits
        SX_REQUIRE(s.c0 >= 0 && s.c0 < 256 && s.ct >= 0 && s.ct < 256 && s.t0 >= 0 && s.t0 < 256 && s.tt >= 0 && s.tt < 256 && s.act >= 0 && s.act < 256,
                   "sx_flow_run: step %d: tile fields out of range (c0 %d ct %d t0 %d tt %d act %d; each < 256)", i, s.c0, s.ct, s.t0, s.tt, s.act);
        SX_REQUIRE(s.kind == SX_STEP_RQS_PHASE || (s.c0 >= 0 && s.ct >= 0 && s.c0 + s.ct <= p->tiles && s.t0 >= 0 && s.tt >= 0),
                   "sx_flow_run: step %d bad tiles", i);
        size_t need = 0;
        switch (s.kind) {
            case SX_STEP_COUPLING_AFFINE: {
                const int T = p->tiles;
                const bool low = T >= 2 && s.c0 == 0 && s.ct == T / 2 && s.t0 == T / 2 && s.tt == T / 2;
                const bool high = T >= 2 && s.c0 == T / 2 && s.ct == T / 2 && s.t0 == 0 && s.tt == T / 2;
                const bool dense = s.c0 == 0 && s.ct == T && s.t0 == 0 && s.tt == T;
                SX_REQUIRE(low || high || dense, "sx_flow_run: step %d: coupling tiles must be low/high halves or dense", i);
                aff = true;
            }
                need = sx_packed_linear_floats(p->h_tiles, s.ct) + sx_packed_linear_floats(2 * s.tt, p->h_tiles);
                break;
            case SX_STEP_COUPLING_AFFINE_HC: {
                const int T = p->tiles;
                const bool low = T >= 2 && s.c0 == 0 && s.ct == T / 2 && s.t0 == T / 2 && s.tt == T / 2;
                const bool high = T >= 2 && s.c0 == T / 2 && s.ct == T / 2 && s.t0 == 0 && s.tt == T / 2;
                SX_REQUIRE(T <= 4 && (low || high), "sx_flow_run: step %d: hidden-chunk couplings condition one half of up to four tiles on the other", i);
                SX_REQUIRE(s.act == SX_ACT_TANH_FOLDED, "sx_flow_run: step %d: hidden-chunk steps carry the folded tanh", i);
                const int first = s.pad_ & 1, last = (s.pad_ >> 1) & 1;
                // the chunks of one coupling come back to back, same tiles and direction: first ... last
                SX_REQUIRE(first || (i > 0 && p->steps[i - 1].kind == SX_STEP_COUPLING_AFFINE_HC && !((p->steps[i - 1].pad_ >> 1) & 1) &&
                                     p->steps[i - 1].c0 == s.c0 && p->steps[i - 1].ct == s.ct && p->steps[i - 1].t0 == s.t0 &&
                                     (p->steps[i - 1].reverse != 0) == (s.reverse != 0)),
                           "sx_flow_run: step %d: a hidden-chunk step that is not the first follows the previous chunk of its coupling", i);
                SX_REQUIRE(last || (i + 1 < p->n_steps && p->steps[i + 1].kind == SX_STEP_COUPLING_AFFINE_HC && !(p->steps[i + 1].pad_ & 1)),
                           "sx_flow_run: step %d: a hidden-chunk step that is not the last is followed by the next chunk", i);
                aff = true; hc = true;
                need = sx_packed_linear_floats(p->h_tiles, s.ct) + sx_packed_linear_floats(2 * s.tt, p->h_tiles);
                break;
            }
            case SX_STEP_WIDE_HIDDEN:
                SX_REQUIRE(p->tiles == 8 && p->x_tiles == 8, "sx_flow_run: step %d: kind 22 belongs to programs on eight data tiles", i);
                SX_REQUIRE((s.c0 == 0 || s.c0 == 4) && s.ct == 4 && s.act == SX_ACT_TANH_FOLDED, "sx_flow_run: step %d: the hidden layer reads one half of the tiles (folded tanh)", i);
                SX_REQUIRE(i + 1 < p->n_steps && p->steps[i + 1].kind == SX_STEP_WIDE_AFFINE_TILE, "sx_flow_run: step %d: a WIDE_HIDDEN step is followed by its coupling's tile steps", i);
                need = sx_packed_linear_floats(p->h_tiles, 4); aff = true; hc = true; ++n_hc8; break;
            case SX_STEP_WIDE_AFFINE_TILE: {
                SX_REQUIRE(p->tiles == 8 && p->x_tiles == 8, "sx_flow_run: step %d: kind 23 belongs to programs on eight data tiles", i);
                SX_REQUIRE(s.t0 >= 0 && s.t0 < 8 && s.tt == 1, "sx_flow_run: step %d: one transformed tile per step", i);
                // the tile steps of a coupling follow its hidden step, all on the other half of the tiles
                int j = i - 1;
                while (j >= 0 && p->steps[j].kind == SX_STEP_WIDE_AFFINE_TILE) --j;
                SX_REQUIRE(j >= 0 && p->steps[j].kind == SX_STEP_WIDE_HIDDEN && (p->steps[j].c0 == 0) == (s.t0 >= 4),
                           "sx_flow_run: step %d: a tile step follows the hidden step of its coupling and transforms a tile of the other half", i);
                need = sx_packed_linear_floats(2, p->h_tiles); aff = true; hc = true; ++n_hc8; break;
            }
            case SX_STEP_AFFINE_CONST: need = 2 * 32 * p->tiles; ++n_hc8; break;
            case SX_STEP_MLP_HIDDEN:
                SX_REQUIRE(s.c0 == 0 && s.ct == p->tiles, "sx_flow_run: MLP_HIDDEN must read all tiles");
                need = sx_packed_linear_floats(p->h_tiles, p->tiles); *mlp_mode = 1; break;
            case SX_STEP_MLP_INPUT:
                SX_REQUIRE(i == 0 && p->tiles <= p->h_tiles && p->tiles <= 4, "sx_flow_run: MLP_INPUT is the first step of its program and needs tiles <= h_tiles");
                need = 0; *mlp_mode = 1; break;
            case SX_STEP_MLP_HIDDEN2: need = sx_packed_linear_floats(p->h_tiles, p->h_tiles); *mlp_mode = 1; break;
            case SX_STEP_MLP_OUT_TILE: need = sx_packed_linear_floats(1, p->h_tiles); *mlp_mode = 1; break;
            case SX_STEP_LINEAR_TILE:
                SX_REQUIRE(s.t0 == 0 && s.act == p->x_tiles,
                           "sx_flow_run: step %d: a linear step covers all %d output slabs (act = x_tiles, t0 = 0)", i, p->x_tiles);
                need = sx_packed_linear_floats(p->x_tiles, p->tiles) + 1; lin = true; break;      // + the layer's log-det
            case SX_STEP_ROW_SCALE_EXP: need = 32 * p->tiles; lin = true; break;
            case SX_STEP_CPL_HIDDEN: need = sx_packed_linear_floats(p->h_tiles, s.ct); deep = true; break;
            case SX_STEP_CPL_HIDDEN2: need = sx_packed_linear_floats(p->h_tiles, p->h_tiles); deep = true; break;
            case SX_STEP_COUPLING_AFFINE_DEEP:
                // (dense form with a latent input: the range covers the latent tile too, whose output rows are zero weights)
                SX_REQUIRE(s.tt >= 1 && s.t0 + s.tt <= p->tiles, "sx_flow_run: step %d: bad transformed tiles", i);
                need = sx_packed_linear_floats(p->h_tiles, p->h_tiles) + sx_packed_linear_floats(2 * s.tt, p->h_tiles);
                deep = true; aff = true; break;
            case SX_STEP_COUPLING_AFFINE_BWD: {
                SX_REQUIRE((p->tiles == 2 || p->tiles == 4) && p->x_tiles * 2 == p->tiles,
                           "sx_flow_run: step %d: backward programs carry x and dL/dx: tiles = 2 * x_tiles (2 or 4; 128-column flows use kinds 16 - 18)", i);
                const int XT = p->x_tiles;
                const bool low = XT == 2 && s.c0 == 0 && s.ct == 1 && s.t0 == 1;
                const bool high = XT == 2 && s.c0 == 1 && s.ct == 1 && s.t0 == 0;
                const bool dense = s.c0 == 0 && s.ct == XT && s.t0 == 0;
                SX_REQUIRE(low || high || dense, "sx_flow_run: step %d: backward coupling tiles must be halves or dense", i);
                const int ct = s.ct, tt = dense ? XT : 1;
                need = sx_packed_linear_floats(p->h_tiles, ct) + sx_packed_linear_floats(2 * tt, p->h_tiles) +
                       sx_packed_linear_floats(p->h_tiles, 2 * tt) + sx_packed_linear_floats(ct, p->h_tiles);
                const int sw = 32 * ct + 64 * p->h_tiles + 64 * tt;
                if (sw > *side_width) *side_width = sw;
                bwd = true; break;
            }
            case SX_STEP_COUPLING_AFFINE_BWD_A:
            case SX_STEP_COUPLING_AFFINE_BWD_B: {
                SX_REQUIRE(p->tiles == 8 && p->x_tiles == 4, "sx_flow_run: step %d: kinds 16 / 17 belong to backward programs on 4 + 4 tiles", i);
                SX_REQUIRE((s.c0 == 0 || s.c0 == 2) && s.ct == 2, "sx_flow_run: step %d: conditioner tiles must be one half of the data tiles", i);
                SX_REQUIRE(s.kind == SX_STEP_COUPLING_AFFINE_BWD_A || (i > 0 && p->steps[i - 1].kind == SX_STEP_COUPLING_AFFINE_BWD_A &&
                                                                      p->steps[i - 1].c0 == s.c0 && p->steps[i - 1].tt == s.tt),
                           "sx_flow_run: step %d: a BWD_B step follows its BWD_A step", i);
                // (the kernel runs A and B inside one iteration of its step loop)
                SX_REQUIRE(s.kind != SX_STEP_COUPLING_AFFINE_BWD_A || (i + 1 < p->n_steps && p->steps[i + 1].kind == SX_STEP_COUPLING_AFFINE_BWD_B),
                           "sx_flow_run: step %d: a BWD_A step is followed by its BWD_B step", i);
                need = s.kind == SX_STEP_COUPLING_AFFINE_BWD_A
                           ? sx_packed_linear_floats(p->h_tiles, 2) + sx_packed_linear_floats(4, p->h_tiles)
                           : sx_packed_linear_floats(p->h_tiles, 4) + sx_packed_linear_floats(2, p->h_tiles);
                const int sw = 32 * 2 + 64 * p->h_tiles + 64 * 2;
                if (sw > *side_width) *side_width = sw;
                bwd = true; ++n_bwd128; break;
            }
            case SX_STEP_LINEAR_BWD: {
                SX_REQUIRE(p->tiles == 8 && p->x_tiles == 4, "sx_flow_run: step %d: kind 18 belongs to backward programs on 4 + 4 tiles", i);
                SX_REQUIRE((s.c0 == 0 || s.c0 == 4) && s.t0 >= 0 && s.t0 + 4 <= 16, "sx_flow_run: step %d: bad tiles / side offset", i);
                // the two halves of a dense layer come back to back, x tiles first (one iteration of the kernel's step loop)
                SX_REQUIRE(s.c0 == 0 ? (i + 1 < p->n_steps && p->steps[i + 1].kind == SX_STEP_LINEAR_BWD && p->steps[i + 1].c0 == 4)
                                     : (i > 0 && p->steps[i - 1].kind == SX_STEP_LINEAR_BWD && p->steps[i - 1].c0 == 0),
                           "sx_flow_run: step %d: SX_STEP_LINEAR_BWD steps come in pairs (x tiles, then adjoint tiles)", i);
                need = sx_packed_linear_floats(4, 4);
                const int sw = 32 * (s.t0 + 4);
                if (sw > *side_width) *side_width = sw;
                bwd = true; ++n_bwd128; break;
            }
            case SX_STEP_COUPLING_TIME: {
                const int kind = s.pad_ & 0xff, K = (s.pad_ >> 16) & 0xff;
                SX_REQUIRE(s.c0 == 0 && s.ct == p->tiles && s.t0 == 0 && s.tt >= 1 &&
                           (s.tt == p->tiles || s.tt * 2 == p->tiles || s.tt * 4 == p->tiles),
                           "sx_flow_run: step %d: a time coupling conditions on all tiles and transforms the data tiles (tt = tiles, tiles/2 or tiles/4)", i);
                SX_REQUIRE(kind >= 0 && kind <= 4 && (kind != 4 || (K >= 1 && K <= 64)), "sx_flow_run: step %d: time kind %d / %d fourier features", i, kind, K);
                need = sx_packed_linear_floats(p->h_tiles, p->tiles) + sx_packed_linear_floats(2 * s.tt, p->h_tiles) +
                       (kind == 0 ? 0 : (kind == 4 ? (size_t)s.tt * K * 128 : (size_t)s.tt * 64));
                timed = true; break;
            }
            case SX_STEP_POINTWISE:
                SX_REQUIRE(s.act >= SX_PW_SIGMOID && s.act <= SX_PW_LEAKY_RELU_INV, "sx_flow_run: step %d: point-wise kind %d (1..6)", i, s.act);
                need = 32 * p->tiles + 1; pw = true; break;
            case SX_STEP_RQS_HIDDEN: {
                const int T = p->tiles;
                const bool low = T >= 2 && s.c0 == 0 && s.ct == T / 2;
                const bool high = T >= 2 && s.c0 == T / 2 && s.ct == T / 2;
                const bool dense = s.c0 == 0 && s.ct == T;
                const bool deep = (s.pad_ & 0xff) == 1;          // (bits 8..15 of pad_: the layer's ordinal, see SX_STEP_RQS_HIDDEN)
                SX_REQUIRE(deep || low || high || dense, "sx_flow_run: step %d: RQS conditioner tiles must be low/high halves or dense", i);
                need = sx_packed_linear_floats(p->h_tiles, deep ? p->h_tiles : s.ct); rqs = true; break;
            }
            case SX_STEP_RQS_PHASE:
                // bins: up to 16 (one output tile per element, four elements per step); rational-quadratic splines also 17 .. 32
                // (two tiles per element, two elements per step: act bit 1 = which pair of the lane's four elements)
                SX_REQUIRE(s.t0 < p->x_tiles && s.c0 >= 0 && s.c0 < 4 && s.ct >= 0 && s.ct < 3 && s.tt >= 1 && s.tt <= 32,
                           "sx_flow_run: step %d: bad RQS phase (tile %d group %d phase %d bins %d)", i, s.t0, s.c0, s.ct, s.tt);
                SX_REQUIRE(s.act == 0 || s.act == 1 || (s.act == 2 && s.tt > 16),
                           "sx_flow_run: step %d: spline phase kind %d (0 rational-quadratic, 1 cubic, 2 rational-quadratic second element pair of a 17..32-bin group)", i, s.act);
                SX_REQUIRE(s.tt <= 16 || s.act != 1, "sx_flow_run: step %d: cubic-spline phases carry up to 16 bins (got %d)", i, s.tt);
                if (s.tt > 16) wide_rq = true;
                if (s.act == 1) cub = true; else quadr = true;
                // the three parameter blocks of a group come back to back (search, select, evaluate): the kernels whose programs hold
                // spline couplings of one type run a triple inside ONE iteration of their step loop (the group's state is then local to it)
                SX_REQUIRE(s.ct == 0 ? (i + 2 < p->n_steps && p->steps[i + 1].kind == SX_STEP_RQS_PHASE && p->steps[i + 1].ct == 1 &&
                                        p->steps[i + 2].kind == SX_STEP_RQS_PHASE && p->steps[i + 2].ct == 2)
                                     : (i >= s.ct && p->steps[i - s.ct].kind == SX_STEP_RQS_PHASE && p->steps[i - s.ct].ct == 0 &&
                                        p->steps[i - s.ct].t0 == s.t0 && p->steps[i - s.ct].c0 == s.c0 && p->steps[i - s.ct].tt == s.tt &&
                                        p->steps[i - s.ct].act == s.act && (p->steps[i - s.ct].reverse != 0) == (s.reverse != 0)),
                           "sx_flow_run: step %d: spline phases come in triples (blocks 0, 1, 2 of one tile / group / bin count / direction)", i);
                need = sx_packed_linear_floats(4, p->h_tiles) + 4; rqs = true; break;
            default: sx_set_error("sx_flow_run: step %d has unsupported kind %d", i, s.kind); return SX_E_UNSUPPORTED;
        }
        SX_REQUIRE(s.blob_floats >= need, "sx_flow_run: step %d blob too small (%u < %zu floats)", i, s.blob_floats, need);
        // (one step's weights must fit one LDS buffer; checked here so that no size below is formed from an unbounded field)
        SX_REQUIRE((size_t)s.blob_floats * 8 <= 160 * 1024, "sx_flow_run: step %d needs %zu B of LDS per buffer (> 80 KiB)", i, (size_t)s.blob_floats * 4);
        if ((int)s.blob_floats > mx) mx = (int)s.blob_floats;
        dstep &o = d->steps[i];
        o.kind = (uint8_t)s.kind; o.c0 = (uint8_t)s.c0; o.ct = (uint8_t)s.ct; o.t0 = (uint8_t)s.t0; o.tt = (uint8_t)s.tt;
        o.reverse = (uint8_t)(s.reverse != 0); o.act = (uint8_t)s.act;
        o.pad = (uint8_t)(s.kind == SX_STEP_RQS_HIDDEN && (s.pad_ & 0xff) == 1);      // deep conditioner: source = kept hidden state
        if (s.kind == SX_STEP_COUPLING_AFFINE_HC) o.pad = (uint8_t)(s.pad_ & 3);        // first / last chunk
        o.blob_off = s.blob_off; o.blob_floats = s.blob_floats; o.ldj_scale = s.ldj_scale; o.ldj_const = s.ldj_const;
        o.mask = (uint32_t)s.pad_;
    }
    *buf_floats = mx;
    SX_REQUIRE(!(lin && *mlp_mode), "sx_flow_run: linear steps cannot be mixed with MLP-output steps");
    if (lin) *mlp_mode = 2;
    SX_REQUIRE(!(rqs && (lin || *mlp_mode == 1)), "sx_flow_run: spline steps cannot be mixed with linear / MLP-output steps");
    // mixed programs (kernel MODE 14): spline couplings beside affine couplings / point-wise steps, or both spline types
    const bool mixed = rqs && (aff || pw || (cub && quadr));
    SX_REQUIRE(!(wide_rq && mixed), "sx_flow_run: spline couplings of 17..32 bins run in programs of rational-quadratic couplings only");
    SX_REQUIRE(!(pw && (lin || *mlp_mode == 1 || bwd || (deep && !mixed))), "sx_flow_run: point-wise steps mix with couplings and element-wise affines only");
    if (rqs) *mlp_mode = mixed ? ((cub && !quadr) ? 16 : (quadr && !cub) ? 17 : 14) : (cub ? 12 : 3);      // 16 / 17: mixed programs with one spline type
    SX_REQUIRE(!(bwd && (rqs || lin || aff || *mlp_mode == 1)), "sx_flow_run: backward steps cannot be mixed with other step kinds");
    if (bwd) *mlp_mode = 4;
    SX_REQUIRE(!(timed && (rqs || lin || aff || bwd || deep || pw || *mlp_mode == 1)), "sx_flow_run: time couplings form programs of their own");
    if (timed) *mlp_mode = 15;
    SX_REQUIRE(p->tiles != 8 || (p->x_tiles == 4 && bwd && n_bwd128 == p->n_steps) || (p->x_tiles == 8 && hc && n_hc8 == p->n_steps),
               "sx_flow_run: 8 state tiles carry backward steps (kinds 16 - 18; 4 data + 4 adjoint tiles) or the couplings of kinds 22 / 23 and element-wise affines (8 data tiles) only");
    SX_REQUIRE(!(deep && (lin || bwd || *mlp_mode == 1)), "sx_flow_run: deep-conditioner steps only mix with couplings");
    if (deep && !mixed) *mlp_mode = rqs ? (cub ? 13 : 10) : 9;     // 10 / 13: the spline kernels with the deep-conditioner steps
    // MODE 20: affine couplings whose hidden layer runs as chunk steps (beside ordinary affine couplings / element-wise affines)
    SX_REQUIRE(!(hc && (rqs || lin || deep || bwd || timed || pw || *mlp_mode == 1)), "sx_flow_run: hidden-chunk couplings share a program with affine couplings and element-wise affines only");
    if (hc) *mlp_mode = 20;
    // MODE 5 / 6: nothing but tanh-folded affine couplings on half the tiles conditioned on the other half, all in
    // one direction (5 reverse, 6 forward) -- the plain RealNVP log_prob / sample program.  Its kernel carries two
    // straight-line arms only, which keeps the state in place (no phi copies) at 130 VGPRs.
    // MODE 7 / 8: the same couplings interleaved with dense linear layers (AffineLU / MatrixExponential -- cfg 4).
    if ((*mlp_mode == 0 || *mlp_mode == 2) && p->n_steps > 0 && p->tiles >= 2) {
        bool pure = true, any = false;
        int dir = -1;
        const int T = p->tiles;
        for (int i = 0; i < p->n_steps && pure; ++i) {
            const sx_step &s = p->steps[i];
            if (s.kind == SX_STEP_LINEAR_TILE || s.kind == SX_STEP_ROW_SCALE_EXP) { pure = *mlp_mode == 2; continue; }
            const bool low = s.c0 == 0 && s.ct == T / 2 && s.t0 == T / 2 && s.tt == T / 2;
            const bool high = s.c0 == T / 2 && s.ct == T / 2 && s.t0 == 0 && s.tt == T / 2;
            if (dir < 0) dir = s.reverse != 0;
            pure = s.kind == SX_STEP_COUPLING_AFFINE && s.act == SX_ACT_TANH_FOLDED && (low || high) &&
                   (s.reverse != 0) == (dir != 0);
            any = true;
        }
        if (pure && any && !g_no_pure_mode) *mlp_mode = (*mlp_mode == 2 ? 7 : 5) + (dir ? 0 : 1);
    }
    SX_REQUIRE((size_t)mx * 8 <= 160 * 1024, "sx_flow_run: a step needs %d B of LDS per buffer (> 80 KiB)", mx * 4);
    return SX_OK;
}

static int pick_grid(int64_t n_rows, int lds_bytes, int tiles, int mode) {
    int per_cu = (160 * 1024) / (lds_bytes > 0 ? lds_bytes : 1);
    int max_per_cu = SX_BLOCKS_FOR(tiles, mode);
    if (g_blocks_per_cu > 0) max_per_cu = g_blocks_per_cu;      // experiment knob
    if (per_cu > max_per_cu) per_cu = max_per_cu;
    if (per_cu < 1) per_cu = 1;
    const int rows_per_block = 32 * SX_BLOCK_WAVES(tiles, mode) * SX_NS_FOR(tiles);
    int64_t chunks = (n_rows + rows_per_block - 1) / rows_per_block;
    int64_t g = 256 * per_cu;
    if (g > chunks) g = chunks;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int sx_flow_launch_info(const sx_program *prog_host, int64_t n_rows, int32_t *grid, int32_t *block,
                                   int32_t *lds_bytes) {
    dprog d; int bf; int mm; int sw;
    int rc = validate_and_convert(prog_host, &d, &bf, &mm, &sw);
    if (rc) return rc;
    if (lds_bytes) *lds_bytes = bf * 8 + 16 + ((mm == 5 || mm == 6) ? 64 * SX_BLOCK_WAVES(prog_host->tiles, mm) * 8 : 0);
    if (block) *block = 64 * SX_BLOCK_WAVES(prog_host->tiles, mm);
    if (grid) *grid = pick_grid(n_rows, bf * 8 + 16 + ((mm == 5 || mm == 6) ? 64 * SX_BLOCK_WAVES(prog_host->tiles, mm) * 8 : 0), prog_host->tiles, mm);
    return SX_OK;
}


extern "C" size_t sx_flow_redo_words(int64_t n_rows) {
    if (n_rows < 0 || n_rows >= ((int64_t)1 << 36)) return 0;
    return (size_t)(2 + 2 * ((n_rows + 31) / 32));
}

extern "C" int sx_flow_run(const sx_program *prog_host, const float *blobs, const void *x, const float *latent,
                           const int32_t *in_col, const int32_t *out_col, void *y, float *ldj_out, float *logp_out,
                           double *sum_out, float *mlp_out, int64_t mlp_out_stride, int32_t mlp_out_dim,
                           const float *row_t, float *side, int64_t n_rows, int32_t dtype, int32_t precision,
                           uint32_t *work, uint32_t *err_flag, void *stream) {
    return sx_flow_run2(prog_host, blobs, nullptr, nullptr, x, latent, in_col, out_col, y, ldj_out, logp_out, sum_out, mlp_out, mlp_out_stride,
                        mlp_out_dim, row_t, side, n_rows, dtype, precision, work, err_flag, stream);
}

extern "C" int sx_flow_run2(const sx_program *prog_host, const float *blobs, const float *blobs_exact, uint32_t *redo, const void *x,
                            const float *latent, const int32_t *in_col, const int32_t *out_col, void *y, float *ldj_out,
                            float *logp_out, double *sum_out, float *mlp_out, int64_t mlp_out_stride, int32_t mlp_out_dim,
                            const float *row_t, float *side, int64_t n_rows, int32_t dtype, int32_t precision,
                            uint32_t *work, uint32_t *err_flag, void *stream) {
    dprog d; int bf; int mlp_mode; int sw;
    int rc = validate_and_convert(prog_host, &d, &bf, &mlp_mode, &sw);
    if (rc) return rc;
    // pure single-hidden-layer spline programs asked for their side outputs (tanh h, layer states: the training forward) run the
    // instances that carry that code
    if ((mlp_out != nullptr || side != nullptr) && (mlp_mode == 3 || mlp_mode == 12)) mlp_mode = mlp_mode == 3 ? 18 : 19;
    SX_REQUIRE(mlp_out == nullptr || mlp_mode == 1 || mlp_mode == 18 || mlp_mode == 19,
               "sx_flow_run: mlp_out is the output of MLP programs and the tanh-h side output of pure spline programs with one hidden layer (kernel mode %d has neither)", mlp_mode);
    SX_REQUIRE(x != nullptr && n_rows >= 0, "sx_flow_run: bad x / n_rows");
    SX_REQUIRE(blobs != nullptr || prog_host->n_steps == 0, "sx_flow_run: null blobs");
    SX_REQUIRE(dtype == SX_F32 || dtype == SX_BF16, "sx_flow_run: bad dtype");
    SX_REQUIRE(precision == SX_GEMM_F32 || precision == SX_GEMM_F16X3, "sx_flow_run: unknown precision %d", precision);
    SX_REQUIRE((blobs_exact == nullptr) == (redo == nullptr), "sx_flow_run2: blobs_exact and redo come together");
    SX_REQUIRE(redo == nullptr || precision == SX_GEMM_F16X3, "sx_flow_run2: the redo pass belongs to the fp16 x 3 arithmetic");
    SX_REQUIRE(redo == nullptr || (((uintptr_t)redo & 3) == 0 && side == nullptr), "sx_flow_run2: redo must be 4-byte aligned; programs with side outputs have no redo pass");
    SX_REQUIRE(prog_host->identity_cols || (in_col != nullptr && (y == nullptr || out_col != nullptr)),
               "sx_flow_run: in_col/out_col required when identity_cols == 0");
    SX_REQUIRE(prog_host->latent_dim == 0 || latent != nullptr, "sx_flow_run: latent_dim > 0 but latent is NULL");
    SX_REQUIRE(mlp_mode != 1 || (mlp_out != nullptr && mlp_out_dim > 0), "sx_flow_run: MLP steps need mlp_out");
    SX_REQUIRE(mlp_mode != 4 || (side != nullptr && row_t != nullptr && dtype == SX_F32 && prog_host->latent_dim == 0),
               "sx_flow_run: backward programs need side, row_t (dL/dlog_prob) and fp32 state");
    SX_REQUIRE(!prog_host->identity_cols || ((uintptr_t)x & 15) == 0, "sx_flow_run: x must be 16-byte aligned");
    if (n_rows == 0) return SX_OK;
    sx_flow_args a;
    a.prog = d; a.blobs = blobs; a.x = x; a.latent = latent; a.in_col = in_col; a.out_col = out_col; a.y = y;
    a.ldj_out = ldj_out; a.logp_out = logp_out; a.sum_out = sum_out; a.mlp_out = mlp_out;
    a.mlp_out_stride = mlp_out_stride; a.mlp_out_dim = mlp_out_dim; a.n_rows = n_rows; a.buf_floats = bf;
    // (MODE 5 / 6: + one fp64 slot per lane for the running batch sum, behind the ring and the ticket slots: sx_flow_kernel.h, LDS_SUM)
    a.bf16 = dtype == SX_BF16; a.mlp_mode = mlp_mode; a.lds = bf * 8 + 16 + ((mlp_mode == 5 || mlp_mode == 6) ? 64 * SX_BLOCK_WAVES(prog_host->tiles, mlp_mode) * 8 : 0);
    a.grid = pick_grid(n_rows, a.lds, prog_host->tiles, mlp_mode);
    a.stream = sx_stream(stream);
    a.row_t = row_t;
    a.side = side;
    a.side_width = sw;
    // dynamic chunk hand-out pays once a workgroup has several chunks; it needs a barrier per chunk (>= 1 step)
    const int rpb = 32 * SX_BLOCK_WAVES(prog_host->tiles, mlp_mode) * SX_NS_FOR(prog_host->tiles);
    const int64_t n_chunks = (n_rows + rpb - 1) / rpb;
    a.work = (prog_host->n_steps > 0 && n_chunks > 2 * (int64_t)a.grid && !g_static_chunks) ? work : nullptr;
    a.flags = err_flag;
    a.frag_in = nullptr; a.frag_out = nullptr; a.acc_out = nullptr;
    a.redo = redo; a.redo_pass = 0;
    const int T = prog_host->tiles, H = prog_host->h_tiles;
    const int fam = SX_MODE_FAMILY(mlp_mode);       // one object per kernel-MODE family (sx_flow_types.h)
    // One launch -- or, with a redo list, two: the fp16 x 3 kernel names the samples whose operands left fp16's range (32-row group +
    // per-sample mask, appended to the list) instead of flagging them, and the SAME program on the exact-fp32 kernel, fed with the
    // exact blobs, then evaluates, stores and sums exactly those samples (sx_flow_kernel.h: flow_kargs::redo).  The second launch
    // finds an empty list on ordinary data: its workgroups read one word and leave (a few microseconds per call).
    for (int pass = 0; pass < (redo != nullptr ? 2 : 1); ++pass) {
        const int prec = pass == 0 ? precision : SX_GEMM_F32;
        if (pass == 1) {
            a.blobs = blobs_exact; a.redo_pass = 1; a.work = nullptr; a.flags = err_flag;
            if (a.grid > 256) a.grid = 256;        // (the list is short, or empty)
        }
        int rc2 = SX_E_UNSUPPORTED;
        bool found = false;
#define SX_GOF(TT, HH, F) (prec == SX_GEMM_F16X3 ? sx_flow_launch_f16x3_t##TT##h##HH##_f##F(a) : sx_flow_launch_f32x_t##TT##h##HH##_f##F(a))
#define SX_GO(TT, HH) if (!found && T == TT && H == HH) { found = true; rc2 = fam == 0 ? SX_GOF(TT, HH, 0) : fam == 1 ? SX_GOF(TT, HH, 1) : SX_GOF(TT, HH, 2); }
        SX_GO(1, 1); SX_GO(1, 2); SX_GO(1, 4); SX_GO(2, 1); SX_GO(2, 2); SX_GO(2, 4); SX_GO(4, 1); SX_GO(4, 2); SX_GO(4, 4);
        SX_GO(8, 1); SX_GO(8, 2); SX_GO(8, 4);
#undef SX_GO
#undef SX_GOF
        if (!found) { sx_set_error("sx_flow_run: unsupported tile configuration"); return SX_E_UNSUPPORTED; }
        if (rc2 != SX_OK) return rc2;
    }
    return SX_OK;
}


// ------------------------------------------------------------------------------------------------
// Training backward, a PAIR of affine couplings per launch, weight gradients contracted in the kernel (MODE 11).
// ------------------------------------------------------------------------------------------------
static int bwd_check(const sx_program *p, dprog *d, int *bf) {
    int mm, sw;
    int rc = validate_and_convert(p, d, bf, &mm, &sw);
    if (rc) return rc;
    SX_REQUIRE(mm == 4 && p->n_steps >= 1 && p->n_steps <= SX_BWD_SLOTS && p->tiles == 4 && p->x_tiles == 2 && p->h_tiles <= 2,
               "sx_flow_bwd_run: a program of 1..%d SX_STEP_COUPLING_AFFINE_BWD steps on 2 + 2 state tiles, hidden <= 64", SX_BWD_SLOTS);
    for (int i = 0; i < p->n_steps; ++i)
        SX_REQUIRE(p->steps[i].ct == 1 && (p->steps[i].c0 == 0 || p->steps[i].c0 == 1) && p->steps[i].t0 == 1 - p->steps[i].c0,
                   "sx_flow_bwd_run: step %d: conditioner and transformed columns must be the two tile halves", i);
    return SX_OK;
}

extern "C" int sx_flow_bwd_max_steps(void) { return SX_BWD_SLOTS; }

extern "C" int sx_flow_bwd_partials(const sx_program *prog_host, int64_t n_rows, int32_t *n_part, int64_t *part_floats) {
    dprog d; int bf;
    int rc = bwd_check(prog_host, &d, &bf);
    if (rc) return rc;
    const int N2 = 32 * prog_host->h_tiles;
    if (n_part) *n_part = pick_grid(n_rows, prog_host->n_steps == 1 ? bf * 4 + 4 * 16384 + 16 : bf * 8 + 16, prog_host->tiles, 11);
    if (part_floats) *part_floats = (int64_t)(64 * N2 + 64) + (int64_t)(N2 * 32 + N2);
    return SX_OK;
}

extern "C" int sx_flow_bwd_run(const sx_program *prog_host, const float *blobs, const float *z, const float *g,
                               const float *frag_in, float *frag_out, float *gy, float *acc_out, int64_t n_rows,
                               uint32_t *work, uint32_t *err_flag, void *stream) {
    dprog d; int bf;
    int rc = bwd_check(prog_host, &d, &bf);
    if (rc) return rc;
    SX_REQUIRE(blobs && g && acc_out && n_rows >= 0, "sx_flow_bwd_run: null pointer");
    SX_REQUIRE((z != nullptr) != (frag_in != nullptr), "sx_flow_bwd_run: give z (first launch) or frag_in (later launches)");
    SX_REQUIRE((gy != nullptr) != (frag_out != nullptr), "sx_flow_bwd_run: give gy (last launch) or frag_out (earlier launches)");
    SX_REQUIRE(prog_host->identity_cols || z == nullptr, "sx_flow_bwd_run: z is read in the flow's own column order (identity_cols)");
    SX_REQUIRE((((uintptr_t)frag_in | (uintptr_t)frag_out | (uintptr_t)z | (uintptr_t)gy) & 15) == 0, "sx_flow_bwd_run: 16-byte alignment");
    if (n_rows == 0) return SX_OK;
    sx_flow_args a;
    a.prog = d; a.blobs = blobs; a.x = z ? (const void *)z : (const void *)frag_in; a.latent = nullptr; a.in_col = nullptr; a.out_col = nullptr;
    a.y = gy; a.ldj_out = nullptr; a.logp_out = nullptr; a.sum_out = nullptr; a.mlp_out = nullptr; a.mlp_out_stride = 0;
    a.mlp_out_dim = 0; a.n_rows = n_rows; a.buf_floats = bf; a.bf16 = 0; a.mlp_mode = 11;
    // single-step programs: resident weights + a 16 KB state landing zone per wave instead of the second weight buffer
    a.lds = prog_host->n_steps == 1 ? bf * 4 + 4 * 16384 + 16 : bf * 8 + 16;
    SX_REQUIRE(a.lds <= 160 * 1024, "sx_flow_bwd_run: the step needs %d B of LDS", a.lds);
    a.grid = pick_grid(n_rows, a.lds, prog_host->tiles, 11);
    a.stream = sx_stream(stream);
    a.row_t = g; a.side = nullptr; a.side_width = 0;
    a.redo = nullptr; a.redo_pass = 0;          // (the backward has no redo pass: an operand beyond fp16's range is flagged)
    const int rpb = 32 * SX_BLOCK_WAVES(prog_host->tiles, 11);
    const int64_t n_chunks = (n_rows + rpb - 1) / rpb;
    // single-step programs hand their chunks out statically: one workgroup per CU, every chunk the same work -- and without
    // the ticket a wave needs no barrier per chunk (measured: the same kernel time either way before that change)
    a.work = (prog_host->n_steps != 1 && n_chunks > 2 * (int64_t)a.grid && !g_static_chunks) ? work : nullptr;
    a.flags = err_flag;
    a.frag_in = frag_in; a.frag_out = frag_out; a.acc_out = acc_out;
    if (prog_host->h_tiles == 1) return sx_flow_launch_f16x3_t4h1_f2(a);
    return sx_flow_launch_f16x3_t4h2_f2(a);
}
