// Reverse mode of the rational-quadratic spline in the INVERSE direction (the direction log_prob evaluates), per element;
// shared by rqs_inverse_bwd_kernel (parameters staged in LDS, any bin count) and the fused slab backward of
// sx_rqs_slab.hip (parameters in MFMA accumulator registers, K <= 16).
// Reverse mode through exactly the operations of the forward kernel (rational_quadratic_spline.py:101-107, 180-234):
// quadratic root -> (a, b, c) -> (knots of bin b, its two derivatives) -> cumsum -> softmax / softplus.  The bin index is
// piecewise constant: no gradient.  Elements in the linear tails pass dL/d(out) through and contribute nothing to the
// parameters.  Ao = dL/d(out), Al = dL/d(log-derivative) of this element.
#pragma once
#ifndef RQS_MIN_BIN
#define RQS_MIN_BIN 1e-3f
#define RQS_MIN_DERIV 1e-3f
#define RQS_EPS 1e-6f
#endif
__device__ __forceinline__ float rqsb_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }
__device__ __forceinline__ float rqsb_softplus(float v) { return v > 20.f ? v : log1pf(expf(v)); }  // F.softplus
// exp(v) on v_exp_f32 with the product v*log2(e) carried to double-float accuracy (~1e-7 relative)
__device__ __forceinline__ float rqsb_exp(float v) {
    const float t = v * 1.44269504088896341f;
    const float r = fmaf(v, 1.44269504088896341f, -t) + v * 1.92596299e-8f;
    return __builtin_amdgcn_exp2f(t) * (1.f + r * 0.69314718055994531f);
}

// The scalar part: from the bin's geometry (knots cw_b / ch_b are only used through w_b, h_b) and the two raw derivative
// parameters to the adjoints of the four knots, the two derivative parameters and the input.
struct rqs_adj {
    float Acwb, Acwn, Achb, Achn;      // dL/d(knots of the bin): widths-side left / right, heights-side left / right
    float g_ub, g_un;                  // dL/d(raw derivative parameter at the bin's left / right knot)
    float Axin;                        // dL/d(input)
};
// FAST: divisions as v_rcp_f32 products, softplus / sigmoid on v_exp_f32 / v_log_f32 (~1e-7 relative, as the fused forward
// kernel evaluates them); otherwise IEEE divisions and libm, as the element-wise kernels do.
template <bool FAST>
__device__ __forceinline__ float rqsb_div(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }
template <bool FAST>
__device__ __forceinline__ float rqsb_softplus_p(float v) {
    if constexpr (FAST) return v > 20.f ? v : __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(v * 1.44269504088896341f)) * 0.69314718055994531f;
    else return rqsb_softplus(v);
}
template <bool FAST>
__device__ __forceinline__ float rqsb_sigmoid_p(float v) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
    else return rqsb_sigmoid(v);
}
template <bool FAST = false>
__device__ __forceinline__ rqs_adj rqs_inverse_bwd_core(float xin, float w_b, float h_b, float ch_b, float u_b, float u_n,
                                                        float Ao, float Al) {
    const float inv_wb = rqsb_div<FAST>(1.f, w_b);
    const float s_b = FAST ? h_b * inv_wb : h_b / w_b;
    const float d_b = RQS_MIN_DERIV + rqsb_softplus_p<FAST>(u_b), d_n = RQS_MIN_DERIV + rqsb_softplus_p<FAST>(u_n);
    const float dy = xin - ch_b;
    const float q = d_b + d_n - 2.f * s_b;
    const float a = dy * q + h_b * (s_b - d_b);
    const float bb = h_b * d_b - dy * q;
    const float c = -s_b * dy;
    const float disc = bb * bb - 4.f * a * c;
    const float sq = FAST ? __builtin_amdgcn_sqrtf(disc) : sqrtf(disc);
    const float D = -bb - sq;
    const float inv_D = rqsb_div<FAST>(1.f, D);
    const float r = FAST ? (2.f * c) * inv_D : (2.f * c) / D;
    const float tomt = r * (1.f - r), omr = 1.f - r;
    const float den = s_b + q * tomt;
    const float T = d_n * (r * r) + 2.f * s_b * tomt + d_b * (omr * omr);
    const float dnum = (s_b * s_b) * T;

    // ---- reverse ----
    const float Adnum = rqsb_div<FAST>(-Al, dnum), Aden = rqsb_div<FAST>(2.f * Al, den);
    float As = Adnum * (2.f * s_b * T + (s_b * s_b) * 2.f * tomt) + Aden;
    float Adn = Adnum * (s_b * s_b) * (r * r), Adb = Adnum * (s_b * s_b) * (omr * omr);
    float Atomt = Adnum * (s_b * s_b) * 2.f * s_b + Aden * q;
    float Ar = Adnum * (s_b * s_b) * (2.f * d_n * r) - Adnum * (s_b * s_b) * (2.f * d_b * omr);
    float Aq = Aden * tomt;
    Ar += Atomt * (1.f - 2.f * r) + Ao * w_b;
    float Awb = Ao * r, Acwb = Ao;
    float Ac = FAST ? Ar * 2.f * inv_D : Ar * 2.f / D;
    const float AD = FAST ? -Ar * r * inv_D : -Ar * r / D;
    float Abb = -AD;
    const float Adisc = (sq > 0.f) ? rqsb_div<FAST>(-AD, 2.f * sq) : 0.f;
    Abb += Adisc * 2.f * bb;
    const float Aa = -4.f * c * Adisc;
    Ac += -4.f * a * Adisc;
    As += -dy * Ac;
    float Ady = -s_b * Ac;
    float Ahb = d_b * Abb;
    Adb += h_b * Abb;
    Ady += -q * Abb;
    Aq += -dy * Abb;
    Ady += q * Aa;
    Aq += dy * Aa;
    Ahb += (s_b - d_b) * Aa;
    As += h_b * Aa;
    Adb += -h_b * Aa;
    Adb += Aq;
    Adn += Aq;
    As += -2.f * Aq;
    float Axin = Ady, Achb = -Ady;
    Ahb += FAST ? As * inv_wb : As / w_b;
    Awb += FAST ? -As * s_b * inv_wb : -As * s_b / w_b;
    float Acwn = Awb, Achn = Ahb;
    Acwb -= Awb;
    Achb -= Ahb;
    rqs_adj A;
    A.Acwb = Acwb; A.Acwn = Acwn; A.Achb = Achb; A.Achn = Achn;
    A.g_ub = Adb * rqsb_sigmoid_p<FAST>(u_b);          // d softplus = sigmoid
    A.g_un = Adn * rqsb_sigmoid_p<FAST>(u_n);
    A.Axin = Axin;
    return A;
}

// The same for the FORWARD direction (rational_quadratic_spline.py:236-248):  theta = (x - cw_b) / w_b,
//   out = ch_b + h_b (s theta^2 + d_b theta(1-theta)) / (s + (d_b + d_n - 2 s) theta(1-theta)),
//   ljd = log(s^2 (d_n theta^2 + 2 s theta(1-theta) + d_b (1-theta)^2)) - 2 log(den).
__device__ __forceinline__ rqs_adj rqs_forward_bwd_core(float xin, float w_b, float h_b, float cw_b, float u_b, float u_n, float Ao,
                                                        float Al) {
    const float s_b = h_b / w_b;
    const float d_b = RQS_MIN_DERIV + rqsb_softplus(u_b), d_n = RQS_MIN_DERIV + rqsb_softplus(u_n);
    const float th = (xin - cw_b) / w_b, omt = 1.f - th, tomt = th * omt;
    const float q = d_b + d_n - 2.f * s_b;
    const float inner = s_b * (th * th) + d_b * tomt;
    const float num = h_b * inner;
    const float den = s_b + q * tomt;
    const float T = d_n * (th * th) + 2.f * s_b * tomt + d_b * (omt * omt);
    const float dnum = (s_b * s_b) * T;
    // ---- reverse ----
    const float Anum = Ao / den;
    float Aden = -Ao * num / (den * den) - 2.f * Al / den;
    const float Adnum = Al / dnum;
    float As = Adnum * (2.f * s_b * T + (s_b * s_b) * 2.f * tomt) + Aden + Anum * h_b * (th * th);
    float Adn = Adnum * (s_b * s_b) * (th * th);
    float Adb = Adnum * (s_b * s_b) * (omt * omt) + Anum * h_b * tomt;
    float Atomt = Adnum * (s_b * s_b) * 2.f * s_b + Aden * q + Anum * h_b * d_b;
    float Ath = Adnum * (s_b * s_b) * (2.f * d_n * th - 2.f * d_b * omt) + Anum * h_b * s_b * 2.f * th;
    float Ahb = Anum * inner;
    const float Aq = Aden * tomt;
    Adb += Aq;
    Adn += Aq;
    As += -2.f * Aq;
    Ath += Atomt * (1.f - 2.f * th);
    const float Axin = Ath / w_b;
    float Acwb = -Ath / w_b;
    float Awb = -Ath * th / w_b;
    Ahb += As / w_b;
    Awb += -As * s_b / w_b;
    rqs_adj A;
    A.Acwn = Awb;
    A.Acwb = Acwb - Awb;
    A.Achn = Ahb;
    A.Achb = Ao - Ahb;                 // out = ch_b + ...
    A.g_ub = Adb * rqsb_sigmoid(u_b);
    A.g_un = Adn * rqsb_sigmoid(u_n);
    A.Axin = Axin;
    return A;
}

// Parameters in memory (LDS): the lane's 3K-1 un-normalised parameters sit at `uw` (widths, heights, derivatives back to
// back); they are overwritten with their gradients and dL/d(input) is returned.
// INVERSE: the direction log_prob evaluates (bin searched on the heights, input in [bottom, top]); otherwise the forward
// direction (bin searched on the widths, input in [left, right]).
template <bool INVERSE>
__device__ __forceinline__ float rqs_bwd_element(float *uw, int K, float xv, float Ao, float Al, float left, float right,
                                                 float bottom, float top, bool valid) {
    const float bconst = logf(expf(1.f - RQS_MIN_DERIV) - 1.f);
    const float norm = 1.f - RQS_MIN_BIN * (float)K;
    const float span_w = right - left, span_h = top - bottom;
    const float lo = INVERSE ? bottom : left, hi = INVERSE ? top : right;
    const bool inside = (xv >= lo) && (xv <= hi);
    const float xin = inside ? xv : lo;
    float *uh = uw + K, *ud = uh + K;

    // ---- forward: softmax pieces, knots, bin (heights), as rqs_kernel's generic path -----------------
    float mw = uw[0], mh = uh[0];
    for (int k = 1; k < K; ++k) { mw = fmaxf(mw, uw[k]); mh = fmaxf(mh, uh[k]); }
    float sw = 0.f, sh = 0.f;
    for (int k = 0; k < K; ++k) {          // exp once per parameter, kept in place (v_exp_f32, compensated argument)
        const float ew = rqsb_exp(uw[k] - mw), eh = rqsb_exp(uh[k] - mh);
        uw[k] = ew;
        uh[k] = eh;
        sw += ew;
        sh += eh;
    }
    const float inv_sw = 1.f / sw, inv_sh = 1.f / sh;
    int b = 0;
    float cw_b = left, ch_b = bottom, cw_n = right, ch_n = top;
    bool have_next = false;
    float csw = 0.f, csh = 0.f;
    for (int j = 1; j <= K; ++j) {
        const float wk = RQS_MIN_BIN + norm * (uw[j - 1] * inv_sw);
        const float hk = RQS_MIN_BIN + norm * (uh[j - 1] * inv_sh);
        csw += wk;
        csh += hk;
        const float kw = (j < K) ? span_w * csw + left : right;
        const float kh = (j < K) ? span_h * csh + bottom : top;
        const float ks = INVERSE ? kh : kw;                              // the searched side's knot (:71-73, search_sorted.py:4-5)
        const bool ge = xin >= ((j < K) ? ks : ks + RQS_EPS);
        if (ge && j < K) { b = j; cw_b = kw; ch_b = kh; }
        else if (!ge && !have_next) { cw_n = kw; ch_n = kh; have_next = true; }
    }
    const float w_b = cw_n - cw_b, h_b = ch_n - ch_b;
    const float u_b = (b == 0) ? bconst : ud[b - 1], u_n = (b + 1 == K) ? bconst : ud[b];
    const rqs_adj A = INVERSE ? rqs_inverse_bwd_core(xin, w_b, h_b, ch_b, u_b, u_n, Ao, Al)
                              : rqs_forward_bwd_core(xin, w_b, h_b, cw_b, u_b, u_n, Ao, Al);
    // knots -> cumsums -> widths / heights:  dL/dw_i = span * ([i < b] A(kw_b) + [i < b+1 < K] A(kw_{b+1}))
    const float Gw_lo = (b >= 1 ? span_w * A.Acwb : 0.f) + (b + 1 < K ? span_w * A.Acwn : 0.f);   // bins i < b
    const float Gw_b = (b + 1 < K ? span_w * A.Acwn : 0.f);                                       // bin i == b
    const float Gh_lo = (b >= 1 ? span_h * A.Achb : 0.f) + (b + 1 < K ? span_h * A.Achn : 0.f);
    const float Gh_b = (b + 1 < K ? span_h * A.Achn : 0.f);
    // softmax backward: du_i = p_i (G_i - sum_j G_j p_j), G_i = norm * dL/dw_i
    float dotw = 0.f, doth = 0.f;
    for (int k = 0; k < K; ++k) {
        const float pw = uw[k] * inv_sw, ph = uh[k] * inv_sh;
        dotw += pw * (k < b ? Gw_lo : (k == b ? Gw_b : 0.f));
        doth += ph * (k < b ? Gh_lo : (k == b ? Gh_b : 0.f));
    }
    const float gate = inside ? 1.f : 0.f;
    if (valid) {                       // (idle lanes of a ragged last group point at slice 0: they must not write)
        for (int k = 0; k < K; ++k) {
            const float pw = uw[k] * inv_sw, ph = uh[k] * inv_sh;
            uw[k] = gate * norm * pw * ((k < b ? Gw_lo : (k == b ? Gw_b : 0.f)) - dotw);
            uh[k] = gate * norm * ph * ((k < b ? Gh_lo : (k == b ? Gh_b : 0.f)) - doth);
        }
        for (int k = 0; k < K - 1; ++k) ud[k] = gate * ((k == b - 1 ? A.g_ub : 0.f) + (k == b ? A.g_un : 0.f));
    }
    return inside ? A.Axin : Ao;                                         // tails: out = x
}
__device__ __forceinline__ float rqs_inverse_bwd_element(float *uw, int K, float xv, float Ao, float Al, float left, float right,
                                                         float bottom, float top, bool valid) {
    return rqs_bwd_element<true>(uw, K, xv, Ao, Al, left, right, bottom, top, valid);
}

// Parameters in registers (one MFMA accumulator tile per block): Wp / Hp / Dp hold the element's K widths, K heights and
// K-1 derivative parameters in entries 0..K-1 (K <= 16; the other entries are ignored) and return their gradients (zeros
// in the unused entries and for an element that is not `valid`).  Every index is static -- the bin is applied through
// selects -- so the arrays never leave the register file; KC = 16 is the straight-line form, KC = 0 keeps K at run time.
// The same mathematics as rqs_inverse_bwd_element, arranged for the VALU budget of a kernel that runs beside MFMAs
// (~45 % of the instructions): hardware exp / rcp / log, fused multiply-adds, only the cumulative sums are selected in
// the sweep (the four knots are formed afterwards), and the softmax-backward dot products come from the selected cumulative
// sums:  sum_{k<b} p_k = (cs_b - b MIN) / norm,  p_b = (cs_{b+1} - cs_b - MIN) / norm.
// `hook(slot)`, slot = 0..47 (a constant once the loops are unrolled), is called once per iteration of the three 16-step
// loops (exp, knot sweep, output): the software-pipelined slab kernel issues the previous chunk's MFMAs from it, ~10 VALU apart.
struct rqsb_no_hook {
    __device__ __forceinline__ void operator()(int) const {}
};

typedef float rqsb_f16v __attribute__((ext_vector_type(16)));
template <int KC, class Hook = rqsb_no_hook>
__device__ __forceinline__ float rqs_inverse_bwd_regs(rqsb_f16v &Wp, rqsb_f16v &Hp, rqsb_f16v &Dp, int K, float xv, float Ao,
                                                      float Al, float left, float right, float bottom, float top, bool valid,
                                                      Hook &&hook = Hook{}) {
    constexpr float LOG2E = 1.44269504088896341f;
    const int Kn = KC ? KC : K;
    const float bconst = 0.5397424172369522f;                       // log(exp(1 - 1e-3) - 1), :81 boundary derivative constant
    const float norm = 1.f - RQS_MIN_BIN * (float)Kn;
    const float span_w = right - left, span_h = top - bottom;
    const bool inside = (xv >= bottom) && (xv <= top);
    const float xin = inside ? xv : bottom;
    float mw = Wp[0], mh = Hp[0];
#pragma unroll
    for (int k = 1; k < 16; ++k)
        if (KC ? (k < KC) : true) {
            const bool used = KC ? true : (k < K);
            mw = used ? fmaxf(mw, Wp[k]) : mw;
            mh = used ? fmaxf(mh, Hp[k]) : mh;
        }
    float sw = 0.f, sh = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (KC ? (k < KC) : true) {
            const bool used = KC ? true : (k < K);
            const float ew = used ? __builtin_amdgcn_exp2f((Wp[k] - mw) * LOG2E) : 0.f;
            const float eh = used ? __builtin_amdgcn_exp2f((Hp[k] - mh) * LOG2E) : 0.f;
            Wp[k] = ew;
            Hp[k] = eh;
            sw += ew;
            sh += eh;
        }
        hook(k);
    }
    const float inv_sw = __builtin_amdgcn_rcpf(sw), inv_sh = __builtin_amdgcn_rcpf(sh);
    const float nw = norm * inv_sw, nh = norm * inv_sh;              // bin size_k = MIN + e_k * n   (:101-105)
    // one sweep: cumulative sums; the heights' knots are compared with the input (search_sorted.py:4-5), and the cumulative
    // sums at the bin (`take`: last knot <= input) and after it (`nxt`: first knot > input) are kept
    int b = 0;
    float csw = 0.f, csh = 0.f, csw_b = 0.f, csw_n = 0.f, kh_b = bottom, kh_n = top;
    bool have_next = false;
#pragma unroll
    for (int j = 1; j <= 16; ++j) {
        if (KC ? (j <= KC) : true) {
            const bool used = KC ? true : (j <= K);
            const bool last = (j == Kn);
            csw += fmaf(Wp[j - 1], nw, RQS_MIN_BIN);
            csh += fmaf(Hp[j - 1], nh, RQS_MIN_BIN);
            const float kh = last ? top : fmaf(span_h, csh, bottom);     // ends pinned (:186-192)
            const bool ge = xin >= (last ? kh + RQS_EPS : kh);
            const bool take = used && ge && !last;
            const bool nxt = used && !ge && !have_next;
            b = take ? j : b;
            csw_b = take ? csw : csw_b;
            kh_b = take ? kh : kh_b;
            csw_n = nxt ? csw : csw_n;
            kh_n = nxt ? kh : kh_n;
            have_next = have_next || nxt;
        }
        hook(16 + j - 1);
    }
    const bool first = (b == 0), lastbin = (b + 1 == Kn);
    const float cw_b = first ? left : fmaf(span_w, csw_b, left);
    const float cw_n = lastbin ? right : fmaf(span_w, csw_n, left);
    const float ch_b = kh_b, ch_n = kh_n;
    const float w_b = cw_n - cw_b, h_b = ch_n - ch_b;
    float u_b = bconst, u_n = bconst;
#pragma unroll
    for (int k = 0; k < 15; ++k)
        if (KC ? (k < KC - 1) : true) {
            const bool used = KC ? true : (k < Kn - 1);
            u_n = (used && b == k) ? Dp[k] : u_n;
            u_b = (used && b == k + 1) ? Dp[k] : u_b;
        }
    const rqs_adj A = rqs_inverse_bwd_core<true>(xin, w_b, h_b, ch_b, u_b, u_n, Ao, Al);
    // knots -> cumsums -> bin sizes:  dL/dsize_i = span * ([i < b] A(knot_b) + [i < b+1 < K] A(knot_{b+1}))
    const float Gw_b = lastbin ? 0.f : span_w * A.Acwn, Gw_lo = (first ? 0.f : span_w * A.Acwb) + Gw_b;
    const float Gh_b = lastbin ? 0.f : span_h * A.Achn, Gh_lo = (first ? 0.f : span_h * A.Achb) + Gh_b;
    // softmax backward: du_i = norm p_i (G_i - sum_j G_j p_j)
    const float inv_norm = __builtin_amdgcn_rcpf(norm), bf = (float)b;
    const float inv_spw = __builtin_amdgcn_rcpf(span_w), inv_sph = __builtin_amdgcn_rcpf(span_h);
    const float cswb = (cw_b - left) * inv_spw, cshb = (ch_b - bottom) * inv_sph;
    const float Sw_lo = (cswb - bf * RQS_MIN_BIN) * inv_norm, pw_b = (w_b * inv_spw - RQS_MIN_BIN) * inv_norm;
    const float Sh_lo = (cshb - bf * RQS_MIN_BIN) * inv_norm, ph_b = (h_b * inv_sph - RQS_MIN_BIN) * inv_norm;
    const float dotw = Gw_lo * Sw_lo + Gw_b * pw_b, doth = Gh_lo * Sh_lo + Gh_b * ph_b;
    const bool on = valid && inside;
    const float fw = on ? nw : 0.f, fh = on ? nh : 0.f;               // norm / sum(e), gated
    const float w_lo = fw * (Gw_lo - dotw), w_at = fw * (Gw_b - dotw), w_hi = fw * -dotw;
    const float h_lo = fh * (Gh_lo - doth), h_at = fh * (Gh_b - doth), h_hi = fh * -doth;
    const float g_ub = on ? A.g_ub : 0.f, g_un = on ? A.g_un : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const bool used = KC ? (k < KC) : (k < K);
        const bool lt = k < b, eq = k == b;
        Wp[k] = used ? Wp[k] * (lt ? w_lo : (eq ? w_at : w_hi)) : 0.f;
        Hp[k] = used ? Hp[k] * (lt ? h_lo : (eq ? h_at : h_hi)) : 0.f;
        Dp[k] = (k < Kn - 1) ? (eq ? g_un : (k + 1 == b ? g_ub : 0.f)) : 0.f;
        hook(32 + k);
    }
    return inside ? A.Axin : Ao;                                         // tails: out = x
}

// ------------------------------------------------------------------------------------------------------------------------
// The monotone CUBIC spline's reverse mode (inverse direction) on registers, for the slab backward: Wp / Hp hold the element's
// K widths / heights, Dp[0..1] its two boundary-derivative parameters (cubic_spline.py:103-137); the same mathematics as
// cubic_bwd_kernel<true> (sx_rqs.hip: the solve differentiated implicitly at the forward's output `xo`, then reverse mode
// through the Steffen knot derivatives, cumsums and softmax), static indices, hardware exp / rcp.
#define CUBICB_MIN_BIN 1e-2f
template <int KC>
__device__ __forceinline__ float cubic_inverse_bwd_regs(rqsb_f16v &Wp, rqsb_f16v &Hp, rqsb_f16v &Dp, int K, float yv, float xo,
                                                        float Ao, float Al, float lower, float upper, bool valid) {
    constexpr float LOG2E = 1.44269504088896341f;
    const int Kn = KC ? KC : K;
    const float norm = 1.f - CUBICB_MIN_BIN * (float)Kn;
    const float span = upper - lower, inv_span = __builtin_amdgcn_rcpf(span);
    const bool inside = (yv >= lower) && (yv <= upper);
    const float yn = ((inside ? yv : lower) - lower) * inv_span;
    float mw = Wp[0], mh = Hp[0];
#pragma unroll
    for (int k = 1; k < 16; ++k)
        if (KC ? (k < KC) : true) {
            const bool used = KC ? true : (k < K);
            mw = used ? fmaxf(mw, Wp[k]) : mw;
            mh = used ? fmaxf(mh, Hp[k]) : mh;
        }
    float sw = 0.f, sh = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (KC ? (k < KC) : true) {
            const bool used = KC ? true : (k < K);
            const float ew = used ? __builtin_amdgcn_exp2f((Wp[k] - mw) * LOG2E) : 0.f;
            const float eh = used ? __builtin_amdgcn_exp2f((Hp[k] - mh) * LOG2E) : 0.f;
            Wp[k] = ew;
            Hp[k] = eh;
            sw += ew;
            sh += eh;
        }
    const float nw = norm * __builtin_amdgcn_rcpf(sw), nh = norm * __builtin_amdgcn_rcpf(sh);
    // one sweep: the bin (last lower edge <= y_n, by heights), its lower edge on the widths side, the sizes of bins b-1, b, b+1
    int b = 0;
    float cw = 0.f, ch = 0.f, cw_b = 0.f, w_b = 1.f, h_b = 1.f, w_m = 1.f, h_m = 1.f, w_p = 1.f, h_p = 1.f, w_last = 1.f, h_last = 1.f;
    bool need_next = false;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (KC ? (k < KC) : true) {
            const bool used = KC ? true : (k < K);
            const float wk = fmaf(Wp[k], nw, CUBICB_MIN_BIN), hk = fmaf(Hp[k], nh, CUBICB_MIN_BIN);
            const bool ge = used && yn >= ch;
            const bool nx = used && !ge && need_next;
            b = ge ? k : b;
            cw_b = ge ? cw : cw_b;
            w_b = ge ? wk : w_b; h_b = ge ? hk : h_b;
            w_m = ge ? w_last : w_m; h_m = ge ? h_last : h_m;
            w_p = nx ? wk : w_p; h_p = nx ? hk : h_p;
            need_next = used ? ge : need_next;                   // edges only grow: once false it stays false
            w_last = used ? wk : w_last; h_last = used ? hk : h_last;
            cw += used ? wk : 0.f;
            ch += used ? hk : 0.f;
        }
    const bool has_m = b > 0, has_p = b < Kn - 1;
    const float s_b = h_b * __builtin_amdgcn_rcpf(w_b), s_m = h_m * __builtin_amdgcn_rcpf(w_m), s_p = h_p * __builtin_amdgcn_rcpf(w_p);
    const float sgl = rqsb_sigmoid_p<true>(Dp[0]), sgr = rqsb_sigmoid_p<true>(Dp[1]);
    float dL, dR;
    bool L_m1 = false, L_first = false, R_m1 = false, R_first = false;
    if (!has_m) dL = sgl * 3.f * s_b;                                                      // :126
    else {
        const float m1 = fminf(s_m, s_b), m2 = 0.5f * (w_b * s_m + w_m * s_b) * __builtin_amdgcn_rcpf(w_m + w_b);    // :118-123
        L_m1 = m1 < m2; L_first = s_m < s_b;
        dL = fminf(m1, m2) * 2.f;                                                          // :124, :129 (slopes are positive)
    }
    if (!has_p) dR = sgr * 3.f * s_b;                                                      // :127
    else {
        const float m1 = fminf(s_b, s_p), m2 = 0.5f * (w_p * s_b + w_b * s_p) * __builtin_amdgcn_rcpf(w_b + w_p);
        R_m1 = m1 < m2; R_first = s_b < s_p;
        dR = fminf(m1, m2) * 2.f;
    }
    const float iw = __builtin_amdgcn_rcpf(w_b), iw2 = iw * iw;
    const float a = (dL + dR - 2.f * s_b) * iw2;                                           // :134
    const float bb = (3.f * s_b - 2.f * dL - dR) * iw;                                     // :135
    const float c = dL;                                                                    // :136
    const float t = (xo - lower) * inv_span - cw_b;
    const float fp = 3.f * a * (t * t) + 2.f * bb * t + c, fpp = 6.f * a * t + 2.f * bb;
    // ---- reverse: the solve f(t) = y_n differentiated implicitly --------------------------------------------------------
    const float Aon = Ao * span;                                   // x = out_n * span + lower
    const float ifp = __builtin_amdgcn_rcpf(fp);
    const float At = Aon - Al * fpp * ifp;                         // out_n = t + cw_b,  ljd = -log f'(t) + const
    // the root is an implicit function of the bin's polynomial: dt/dtheta = -(df/dtheta) / f'(t).  Where |a| < 1e-3 the reference
    // solves the QUADRATIC bb t^2 + c t + (d - y) = 0 instead (cubic_spline.py:216-222): its root does not depend on a at all and
    // the other derivatives carry 1 / q'(t), q' = 2 bb t + c -- not a rounding matter: da/dtheta carries 1 / w^2, so the a-path is
    // as large as the others, and a near-identity spline (every freshly initialised one) has a ~ 0 in every bin.
    const bool quad = fabsf(a) < 1e-3f;
    const float iq = quad ? __builtin_amdgcn_rcpf(2.f * bb * t + c) : ifp;
    const float Ay = At * iq * inv_span;                           // y_n = (y - lower) / span
    const float Aa = (quad ? 0.f : -At * (t * t * t) * iq) - Al * 3.f * (t * t) * ifp;
    const float Ab = -At * (t * t) * iq - Al * 2.f * t * ifp;
    const float Ac = -At * t * iq - Al * ifp;
    const float Achb = -At * iq;                                   // d = ch_b
    const float Acwb = Aon;
    const float AdL = Aa * iw2 - 2.f * Ab * iw + Ac;
    const float AdR = Aa * iw2 - Ab * iw;
    float As = -2.f * Aa * iw2 + 3.f * Ab * iw;
    float Aw = -2.f * a * Aa * iw - bb * Ab * iw;
    float Ah = 0.f, Awm = 0.f, Ahm = 0.f, Awp = 0.f, Ahp = 0.f, Asm = 0.f, Asp = 0.f, Audl = 0.f, Audr = 0.f;
    if (!has_m) { Audl = AdL * 3.f * s_b * sgl * (1.f - sgl); As += AdL * 3.f * sgl; }
    else if (L_m1) { if (L_first) Asm += 2.f * AdL; else As += 2.f * AdL; }
    else {
        const float iWd = __builtin_amdgcn_rcpf(w_m + w_b), N = w_b * s_m + w_m * s_b;      // dL = N / Wd
        Aw += AdL * (s_m * iWd - N * iWd * iWd);
        Awm += AdL * (s_b * iWd - N * iWd * iWd);
        Asm += AdL * w_b * iWd;
        As += AdL * w_m * iWd;
    }
    if (!has_p) { Audr = AdR * 3.f * s_b * sgr * (1.f - sgr); As += AdR * 3.f * sgr; }
    else if (R_m1) { if (R_first) As += 2.f * AdR; else Asp += 2.f * AdR; }
    else {
        const float iWd = __builtin_amdgcn_rcpf(w_b + w_p), N = w_p * s_b + w_b * s_p;      // dR = N / Wd
        Awp += AdR * (s_b * iWd - N * iWd * iWd);
        Aw += AdR * (s_p * iWd - N * iWd * iWd);
        As += AdR * w_p * iWd;
        Asp += AdR * w_b * iWd;
    }
    Ah += As * iw;  Aw += -As * s_b * iw;
    { const float im = __builtin_amdgcn_rcpf(w_m); Ahm += Asm * im; Awm += -Asm * s_m * im; }
    { const float ip = __builtin_amdgcn_rcpf(w_p); Ahp += Asp * ip; Awp += -Asp * s_p * ip; }
    if (!has_m) { Awm = 0.f; Ahm = 0.f; }
    if (!has_p) { Awp = 0.f; Ahp = 0.f; }
    // bin sizes: direct terms at b-1, b, b+1 + the cumsums' (cw_b, ch_b) on every bin below b; softmax backward
    float dotw = 0.f, doth = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (KC ? (k < KC) : true) {
            const float gw = (k < b ? Acwb : 0.f) + (k == b ? Aw : 0.f) + (k == b - 1 ? Awm : 0.f) + (k == b + 1 ? Awp : 0.f);
            const float gh = (k < b ? Achb : 0.f) + (k == b ? Ah : 0.f) + (k == b - 1 ? Ahm : 0.f) + (k == b + 1 ? Ahp : 0.f);
            dotw += Wp[k] * gw;                                    // unused entries hold 0
            doth += Hp[k] * gh;
        }
    const bool on = valid && inside;
    const float fw = on ? nw : 0.f, fh = on ? nh : 0.f;            // norm / sum(e), gated
    dotw *= nw * __builtin_amdgcn_rcpf(norm);                      // sum_k p_k G_k, p_k = e_k / sum(e)
    doth *= nh * __builtin_amdgcn_rcpf(norm);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const bool used = KC ? (k < KC) : (k < K);
        const float gw = (k < b ? Acwb : 0.f) + (k == b ? Aw : 0.f) + (k == b - 1 ? Awm : 0.f) + (k == b + 1 ? Awp : 0.f);
        const float gh = (k < b ? Achb : 0.f) + (k == b ? Ah : 0.f) + (k == b - 1 ? Ahm : 0.f) + (k == b + 1 ? Ahp : 0.f);
        Wp[k] = used ? fw * Wp[k] * (gw - dotw) : 0.f;
        Hp[k] = used ? fh * Hp[k] * (gh - doth) : 0.f;
        Dp[k] = k == 0 ? (on ? Audl : 0.f) : (k == 1 ? (on ? Audr : 0.f) : 0.f);
    }
    return inside ? Ay : Ao;                                          // tails: out = y
}
