// Monotone cubic spline (stribor/util/cubic_spline.py:21-251): the per-element arithmetic shared by the element-wise kernel
// (sx_rqs.hip: cubic_kernel) and the fused flow kernel's cubic phases (sx_flow_kernel.h).
#pragma once
#define CUBIC_MIN_BIN 1e-2f          // cubic_spline.py:13-14
#define CUBIC_EPS 1e-5f              // :15
#define CUBIC_QUAD_THRESHOLD 1e-3f   // :16

// ---- cubic_kernel's arithmetic: the hardware's 1-ulp exp2 / log2 / rcp / sqrt instead of libm calls and IEEE division
// sequences (the kernel is VALU-bound: 1,489 -> ~900 VALU instructions per element in the inverse direction, 791 -> ~500
// forward, tools/pmc_spline_kernels.sh).  Their errors (~1e-7 relative) are the size of the reference's own fp32 rounding;
// the inverse ends in Newton steps on the bin's cubic, so its result does not depend on how exactly the closed form ran.
__device__ __forceinline__ float cubic_fexp(float v) { return __builtin_amdgcn_exp2f(v * 1.44269504088896341f); }
__device__ __forceinline__ float cubic_frcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float cubic_flog(float v) { return __builtin_amdgcn_logf(v) * 0.69314718055994531f; }
__device__ __forceinline__ float cubic_fsigmoid(float v) { return cubic_frcp(1.f + cubic_fexp(-v)); }
__device__ __forceinline__ float cubic_fcbrt(float v) {        // :18-20  sign(x) * exp(log|x| / 3)
    const float m = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(fabsf(v)) * (1.0f / 3.0f));
    return (v == 0.f) ? 0.f : copysignf(m, v);
}
// Knot derivatives of bin b (:117-132) and its cubic a t^3 + bb t^2 + c t + d (:134-137) from the sizes of bins b-1, b, b+1.
// Bin sizes are >= 1e-2, so every slope is positive: sign(s_m) + sign(s_b) = 2 and the |.| of :118-119 are no-ops.
struct cubic_coef { float a, bb, c; };
__device__ __forceinline__ cubic_coef cubic_bin_coef(int b, int K, float w_b, float h_b, float w_m, float h_m, float w_p, float h_p,
                                                     float dpar0, float dpar1) {
    const float rwb = cubic_frcp(w_b);
    const float s_b = h_b * rwb;                                                           // :117
    const float s_m = h_m * cubic_frcp(w_m), s_p = h_p * cubic_frcp(w_p);
    const float mL = 0.5f * (w_b * s_m + w_m * s_b) * cubic_frcp(w_m + w_b);              // :120-123
    const float mR = 0.5f * (w_p * s_b + w_b * s_p) * cubic_frcp(w_b + w_p);
    const float dL = (b == 0) ? cubic_fsigmoid(dpar0) * 3.f * s_b : 2.f * fminf(fminf(s_m, s_b), mL);       // :126, :118-129
    const float dR = (b == K - 1) ? cubic_fsigmoid(dpar1) * 3.f * s_b : 2.f * fminf(fminf(s_b, s_p), mR);   // :127
    cubic_coef q;
    q.a = (dL + dR - 2.f * s_b) * (rwb * rwb);                                             // :134
    q.bb = (3.f * s_b - 2.f * dL - dR) * rwb;                                              // :135
    q.c = dL;                                                                              // :136
    return q;
}
// The inverse of the bin's cubic f(t) = a t^3 + bb t^2 + c t + d at y = xin (:154-222): the reference's closed forms (one-root
// Cardano form, three-root trigonometric form with the in-bin root picked in its order, quadratic formula where |a| < 1e-3),
// then Newton steps inside the bin.  Returns t = root - cw_b in [0, rcw - cw_b].
__device__ __forceinline__ float cubic_invert(float a, float bb, float c, float d, float xin, float cw_b, float rcw) {
            float o;
            const float ra = cubic_frcp(a);
            const float b_ = (bb * ra) * (1.0f / 3.0f);                                    // :154-156
            const float c_ = (c * ra) * (1.0f / 3.0f);
            const float d_ = (d - xin) * ra;
            const float delta_1 = -(b_ * b_) + c_;                                         // :158-160
            const float delta_2 = -c_ * b_ + d_;
            const float delta_3 = b_ * d_ - c_ * c_;
            const float disc = 4.f * delta_1 * delta_3 - delta_2 * delta_2;                // :162
            const float dep1 = -2.f * b_ * delta_1 + delta_2;                              // :164
            const bool three = disc > 0.f;                                                 // :167
            const float sqd = __builtin_amdgcn_sqrtf(fabsf(disc));
            // one root (:174-179)
            const float sq = three ? 0.f : sqd;
            const float one_root = (cubic_fcbrt((-dep1 + sq) * 0.5f) + cubic_fcbrt((-dep1 - sq) * 0.5f)) - b_ + cw_b;
            // three roots (:183-212): the first (order 1, 2, 3) that lies in the bin, root 1 if none does.
            // theta = atan2(sqrt(disc), -dep1) / 3 in [0, pi/3]: odd polynomial for atan on [0, 1] (1.3e-7), Taylor sums for
            // cos / sin on [0, pi/3] (< 4e-9) -- no range reduction is needed anywhere.
            const float ty = three ? sqd : 0.f, tx = -dep1;
            const float ax = fabsf(tx), hi_ = fmaxf(ax, ty), lo_ = fminf(ax, ty);
            const float qa = (hi_ > 0.f) ? lo_ * cubic_frcp(hi_) : 0.f, q2 = qa * qa;
            float at = -0.00405455706641078f;
            at = fmaf(at, q2, 0.021862920373678207f); at = fmaf(at, q2, -0.05591226741671562f); at = fmaf(at, q2, 0.09642192721366882f);
            at = fmaf(at, q2, -0.1390862762928009f); at = fmaf(at, q2, 0.19946564733982086f); at = fmaf(at, q2, -0.33329859375953674f);
            at = fmaf(at, q2, 0.9999993443489075f);
            at = at * qa;
            at = (ty > ax) ? 1.5707963267948966f - at : at;
            at = (tx < 0.f) ? 3.141592653589793f - at : at;
            const float theta = at * (1.0f / 3.0f), th2 = theta * theta;
            float cr1 = -2.755731922398589e-07f;                                           // cos: 1 - t^2/2! + ... - t^10/10!
            cr1 = fmaf(cr1, th2, 2.48015873015873e-05f); cr1 = fmaf(cr1, th2, -1.388888888888889e-03f); cr1 = fmaf(cr1, th2, 4.166666666666666e-02f);
            cr1 = fmaf(cr1, th2, -0.5f); cr1 = fmaf(cr1, th2, 1.f);
            float cr2 = -2.505210838544172e-08f;                                           // sin: t (1 - t^2/3! + ... - t^10/11!)
            cr2 = fmaf(cr2, th2, 2.755731922398589e-06f); cr2 = fmaf(cr2, th2, -1.984126984126984e-04f); cr2 = fmaf(cr2, th2, 8.333333333333333e-03f);
            cr2 = fmaf(cr2, th2, -1.666666666666667e-01f); cr2 = fmaf(cr2, th2, 1.f);
            cr2 = cr2 * theta;
            const float scale = 2.f * __builtin_amdgcn_sqrtf(three ? -delta_1 : 0.f), shift = -b_ + cw_b;
            const float r1 = cr1 * scale + shift;
            const float r2 = (-0.5f * cr1 - 0.5f * 1.7320508075688772f * cr2) * scale + shift;
            const float r3 = (-0.5f * cr1 + 0.5f * 1.7320508075688772f * cr2) * scale + shift;
            const float lo3 = cw_b - CUBIC_EPS, hi3 = rcw + CUBIC_EPS;
            const bool k1 = (lo3 < r1) && (r1 < hi3), k2 = (lo3 < r2) && (r2 < hi3), k3 = (lo3 < r3) && (r3 < hi3);
            const float pick = k1 ? r1 : (k2 ? r2 : (k3 ? r3 : r1));
            o = three ? pick : one_root;
            // a -> 0 (:216-222)
            if (fabsf(a) < CUBIC_QUAD_THRESHOLD) {
                // (-c + sqrt(c^2 - 4 bb qc)) / (2 bb) in its cancellation-free form -2 qc / (c + sqrt(.)), c = dL > 0: the same
                // root, but the reference's form loses everything as bb -> 0 (a near-identity spline has bb ~ 1e-5: 5 % of t,
                // 1e-2 of x in fp32 -- the reference's own fp32 path does that; its fp64 values are what this returns)
                const float qc = d - xin;
                o = (-2.f * qc) * cubic_frcp(c + __builtin_amdgcn_sqrtf(c * c - 4.f * bb * qc)) + cw_b;
            }
            // Newton steps on f(t) = a t^3 + bb t^2 + c t + d - y inside the bin (f is monotone there): the closed forms above lose
            // up to 5e-3 of the bin in fp32 where the cubic degenerates (the reference's fp32 path does too); two steps from
            // their result bring the residual to rounding level.  A non-finite start falls back to the bin's lower knot.
            float so = o - cw_b;                                                           // :224
            const float t_hi = rcw - cw_b, f0 = d - xin;
            so = fminf(fmaxf(so, 0.f), t_hi);                                              // (fmaxf(NaN, 0) = 0)
            const bool quad = fabsf(a) < CUBIC_QUAD_THRESHOLD;     // the reference's quadratic root stays as it is (it is not the cubic's)
            const float so_q = o - cw_b;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const float fv = fmaf(fmaf(fmaf(a, so, bb), so, c), so, f0);
                const float fd = fmaf(fmaf(3.f * a, so, 2.f * bb), so, c);
                const float st = fv * cubic_frcp(fd);
                so = (fd > 0.f) ? fminf(fmaxf(so - st, 0.f), t_hi) : so;
            }
            so = quad ? so_q : so;
            return so;
}
