// Spline-coupling phases of the fused flow kernel (rational-quadratic and monotone cubic; kernel MODEs 3 / 10 / 12 - 14 / 16 - 19).
// Part of sx_flow_kernel.h: included from there, inside namespace SX_PREC_NS, after the GEMM tile helpers -- not a header of its own.
// (Split out in round 4: the one 3,700-line header had become hard to navigate.  The objects still instantiate every MODE of a
//  (tiles, hidden-tiles) pair, so an edit here rebuilds them all; a per-family split of the OBJECTS is the open half)
// ------------------------------------------------------------------------------------------------
// Rational-quadratic spline coupling, fused (util/rational_quadratic_spline.py:11-251 + search_sorted.py).
// A layer is 1 RQS_HIDDEN step + 12 RQS_PHASE steps per transformed tile: the tile's 32 columns are handled in
// 4 groups of 8; per group three parameter blocks (the SEARCH block whose knots bracket the input, the SELECT
// block evaluated at the found bin, the derivatives) arrive as 4 MFMA output tiles each, laid out so that every
// lane receives, in the 16 registers of output tile q, the 16 parameters of its element q (row kmap(k, h) of tile q = parameter k
// of the lane-half-h element q; round 4: before, tile u held parameters 4u..4u+3 of all four elements, so no element could start
// before all four tiles were done): the 12 GiB [N, D*(3K-1)] parameter tensor of the reference (spline.py:82-86) only ever
// exists 64 registers at a time, and an element's arithmetic can run beside the MFMAs of the next element's tile.
// ------------------------------------------------------------------------------------------------
#define RQS_MIN 1e-3f
// One iteration of a K-generic sweep ends here: the running results pass through an empty volatile asm and a scheduling fence, so that
// an iteration's compares are consumed by its own selects (VCC) instead of being hoisted sixteen or thirty-two deep -- interleaved, the
// sweeps keep ~60 lane masks alive and the allocator spills them to VGPR lanes (v_writelane / v_readlane: vector instructions; the
// spline object reported 501 SGPR spills, round 5)
// The bin index of an element is the same value in its select and its evaluate block, and both compare it with every bin number: the
// optimizer shares those compares -- up to 2 x 31 lane masks kept alive across the step advance between the blocks, i.e. spilled to
// lanes and read back two v_readlane per select.  Each block takes the index through an empty asm instead and compares again.
#ifndef SX_NO_SWEEP_FENCE
#define SX_OPAQUE(v) asm volatile("" : "+v"(v))
#else
#define SX_OPAQUE(v) ((void)0)
#endif
#ifndef SX_NO_SWEEP_FENCE
#define SX_SWEEP_FENCE(a, b, c) do { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SX_SWEEP_FENCE(a, b, c) ((void)0)
#endif
struct rqs_elems {          // the 4 elements of the current group a lane owns
    float x[4];             // input values
    float a_b[4], a_w[4];   // searched sequence: knot at the bin, bin size
    float c_b[4], c_w[4];   // selected sequence: knot at the bin, bin size
    int b[4];               // bin index; + RQS_OUT when the input is outside the (input-side) interval (a bool here is a lane mask
};                          //  in SGPRs, carried across the blocks through v_writelane / v_readlane spills)
#define RQS_OUT 64

template <int HT>
__device__ __forceinline__ void rqs_gemm(const wptr w, const btile<1> (&bh)[HT], tile<1> (&acc)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        acc[u] = load_cfrag<1>(w.cb, 4 * HT * 1024 + u * 32);
#pragma unroll
        for (int m = 0; m < HT; ++m) gemm_tile<1>(w.wb, (u * HT + m) * 1024, bh[m], acc[u]);
    }
}
// parameter k (0..15) of element q: register k of output tile q
#define RQS_P(acc, q, k) (acc)[q].v[0][k]

// The three per-element routines are written branch-free (selects only) and take the bin count as a template
// argument: KC = 16 is the straight-line hot path of cfg 3, KC = 0 keeps K as a run-time value (k < K predicates).
template <int KC>
__device__ __forceinline__ bool rqs_has(int k, int K) { return KC ? (k < KC) : (k < K); }

// softmax numerators in place + the factor that turns them into bin sizes: size_k = MIN + e_k * inv.  The two softmax blocks'
// rows are packed times log2(e) (fused.py add_coupling_rqs), so exp(u - max) is a bare v_exp_f32.
// (W: output tiles per element -- 1: up to 16 bins; 2: up to 32 bins, element Q of the step's two = tiles 2Q, 2Q+1)
#define RQS_PW(acc, q, k) (acc)[W * (q) + ((k) >> 4)].v[0][(k) & 15]
template <int Q, int KC, int W = 1>
__device__ __forceinline__ float rqs_softmax(tile<1> (&acc)[4], int K) {
    float mx = RQS_PW(acc, Q, 0);
#pragma unroll
    for (int k = 1; k < 16 * W; ++k)
        if (KC ? (k < KC) : true) mx = fmaxf(mx, rqs_has<KC>(k, K) ? RQS_PW(acc, Q, k) : mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16 * W; ++k)
        if (KC ? (k < KC) : true) {
            const float e = rqs_has<KC>(k, K) ? __builtin_amdgcn_exp2f(RQS_PW(acc, Q, k) - mx) : 0.f;     // logits arrive in base 2
            RQS_PW(acc, Q, k) = e;
            sum += e;
        }
    const float Kf = KC ? (float)KC : (float)K;
    return (1.f - RQS_MIN * Kf) * fast_rcp(sum);         // :101-105
}

// phase 0: knots of the searched block (:180-192) and the bin search (search_sorted.py:4-5) in one sweep.
// The knots increase and `x >= knot_j` is true for a prefix of j, so
//   knot at the bin  = last knot with x >= knot  (running select),
//   next knot        = min over the knots with x < knot (min with `hi` where x >= knot).
// K = 16: the sixteen-step sweep as a two-level search -- which group of four bins (three compares against the knots at
// bins 4, 8, 12, formed from the groups' sums), then the bin inside the group (its four sizes picked by the group index): ~70
// VALU instructions per element instead of ~130 for the same knot values up to the order of the additions.
#ifndef SX_RQS_FLAT
template <int Q>
__device__ __forceinline__ void rqs_search16(tile<1> (&acc)[4], rqs_elems &e, float lo, float hi) {
    const float xv = e.x[Q];
    const bool in = (xv >= lo) && (xv <= hi);                       // :71 closed interval
    const float xin = in ? xv : lo;
    const float inv = rqs_softmax<Q, 16>(acc, 16);
    const float span = hi - lo;
    // cumulative sizes at the group boundaries: cs_4, cs_8, cs_12 (:180-192)
    const float S0 = 4.f * RQS_MIN + inv * ((RQS_P(acc, Q, 0) + RQS_P(acc, Q, 1)) + (RQS_P(acc, Q, 2) + RQS_P(acc, Q, 3)));
    const float S1 = 4.f * RQS_MIN + inv * ((RQS_P(acc, Q, 4) + RQS_P(acc, Q, 5)) + (RQS_P(acc, Q, 6) + RQS_P(acc, Q, 7)));
    const float S2 = 4.f * RQS_MIN + inv * ((RQS_P(acc, Q, 8) + RQS_P(acc, Q, 9)) + (RQS_P(acc, Q, 10) + RQS_P(acc, Q, 11)));
    const float C1 = S0, C2 = S0 + S1, C3 = (S0 + S1) + S2;
    const bool m1 = xin >= span * C1 + lo, m2 = xin >= span * C2 + lo, m3 = xin >= span * C3 + lo;       // a prefix: knots grow
    const float base = m3 ? C3 : (m2 ? C2 : (m1 ? C1 : 0.f));
    const int gb = m3 ? 12 : (m2 ? 8 : (m1 ? 4 : 0));
    auto pick = [&](int i) {
        return m3 ? RQS_P(acc, Q, 12 + i) : (m2 ? RQS_P(acc, Q, 8 + i) : (m1 ? RQS_P(acc, Q, 4 + i) : RQS_P(acc, Q, i)));
    };
    const float cs1 = base + (RQS_MIN + pick(0) * inv), cs2 = cs1 + (RQS_MIN + pick(1) * inv);
    const float cs3 = cs2 + (RQS_MIN + pick(2) * inv), cs4 = cs3 + (RQS_MIN + pick(3) * inv);
    const float k0 = m1 ? span * base + lo : lo;                    // ends pinned
    const float k1 = span * cs1 + lo, k2 = span * cs2 + lo, k3 = span * cs3 + lo;
    const float k4 = m3 ? hi : span * cs4 + lo;
    const bool g1 = xin >= k1, g2 = xin >= k2, g3 = xin >= k3;
    e.b[Q] = gb + (g3 ? 3 : (g2 ? 2 : (g1 ? 1 : 0))) + (in ? 0 : RQS_OUT);
    const float k_b = g3 ? k3 : (g2 ? k2 : (g1 ? k1 : k0));
    const float k_n = g3 ? k4 : (g2 ? k3 : (g1 ? k2 : k1));
    e.a_b[Q] = k_b;
    e.a_w[Q] = k_n - k_b;
}
// the other block at the found bin: the two knots around bin b from the group sums + the group's own sizes
template <int Q>
__device__ __forceinline__ void rqs_select16(tile<1> (&acc)[4], rqs_elems &e, float lo, float hi) {
    const float inv = rqs_softmax<Q, 16>(acc, 16);
    const float span = hi - lo;
    const int b = e.b[Q] & (RQS_OUT - 1);
    const float S0 = 4.f * RQS_MIN + inv * ((RQS_P(acc, Q, 0) + RQS_P(acc, Q, 1)) + (RQS_P(acc, Q, 2) + RQS_P(acc, Q, 3)));
    const float S1 = 4.f * RQS_MIN + inv * ((RQS_P(acc, Q, 4) + RQS_P(acc, Q, 5)) + (RQS_P(acc, Q, 6) + RQS_P(acc, Q, 7)));
    const float S2 = 4.f * RQS_MIN + inv * ((RQS_P(acc, Q, 8) + RQS_P(acc, Q, 9)) + (RQS_P(acc, Q, 10) + RQS_P(acc, Q, 11)));
    const bool m1 = b >= 4, m2 = b >= 8, m3 = b >= 12;
    const float base = m3 ? (S0 + S1) + S2 : (m2 ? S0 + S1 : (m1 ? S0 : 0.f));
    auto pick = [&](int i) {
        return m3 ? RQS_P(acc, Q, 12 + i) : (m2 ? RQS_P(acc, Q, 8 + i) : (m1 ? RQS_P(acc, Q, 4 + i) : RQS_P(acc, Q, i)));
    };
    const float cs1 = base + (RQS_MIN + pick(0) * inv), cs2 = cs1 + (RQS_MIN + pick(1) * inv);
    const float cs3 = cs2 + (RQS_MIN + pick(2) * inv), cs4 = cs3 + (RQS_MIN + pick(3) * inv);
    const int bl = b & 3;
    const float k0 = m1 ? span * base + lo : lo;
    const float k1 = span * cs1 + lo, k2 = span * cs2 + lo, k3 = span * cs3 + lo;
    const float k4 = m3 ? hi : span * cs4 + lo;
    const bool g1 = bl >= 1, g2 = bl >= 2, g3 = bl >= 3;
    const float k_b = g3 ? k3 : (g2 ? k2 : (g1 ? k1 : k0));
    const float k_n = g3 ? k4 : (g2 ? k3 : (g1 ? k2 : k1));
    e.c_b[Q] = k_b;
    e.c_w[Q] = k_n - k_b;
}
#endif
// ------------------------------------------------------------------------------------------------
// K = 16, BOUNDED logits (round 4): the lean form of the three phases above.  What changed and why (profiles/sq_cfg3.json: 256 of
// the 545 vector instructions per element were selects, compares and moves; 256 VGPRs + 68 B of scratch):
//   * no running maximum.  The softmax is shift-invariant and the shift only guards exp against overflow; the packer leaves a bound
//     on |logit| of the step's rows behind the spline bounds (sx_pack_linear_bound: |b' + the positive (negative) packed weights| --
//     the hidden activations are the folded tanh, in [0, 1]), and below RQS16_BOUND the phases run e_k = exp2(u_k) directly: 8
//     v_max3 + 16 v_sub per block and the canonicalising v_max pairs go away.  Above it (2^96 per bin: not seen with trained
//     conditioners) the KC = 0 sweep with the maximum runs instead;
//   * the sums of a group of four bins are formed left to right, so the three partial sums ARE the group's inner prefixes: the knots
//     inside the found group come from one pick of four prefixes each (12 selects) and one fma each, with the bounds folded into
//     per-phase constants (knot = c_span_inv * prefix + (k0 + i * span * MIN));
//   * every select takes named VALUES computed on both sides (a conditional EXPRESSION with arithmetic in an arm is a branch to
//     the front end -- the old code ran s_and_saveexec / s_cbranch_execz diamonds inside the element code --, and `c ? a : b` on
//     two lvalues selects the ADDRESS, which pins the operands to scratch);
//   * an element's 16 parameters are ONE output tile (RQS_P), so element q's arithmetic runs beside the MFMAs of tile q + 1
//     (rqs16_block): the matrix pipe and the VALU overlap inside the wave instead of only across the two waves of a SIMD.
// Arithmetic: the same knots up to the order of the additions (the reference: cumsum of MIN + (1 - K MIN) softmax, times the span,
// plus the lower bound, ends pinned -- rational_quadratic_spline.py:180-192).
// ------------------------------------------------------------------------------------------------
#define RQS16_BOUND 96.0f
// Round 6: the same code for ANY K <= 16 (`GEN`; K = 16 keeps its own instance).  K = 8 used to run 1.6 x SLOWER than K = 16 -- every
// bin count but sixteen took the K-generic sweeps.  The packer parks the unused slots of a tile where they cost nothing
// (fused.py add_coupling_rqs: logits of bins >= K at -1e30, so exp2 is 0 and every prefix sum is unchanged; derivative rows >= K - 1
// at the boundary constant, so D[K - 1] needs no select), and what is left for the kernel is the reference's end handling:
//   * knots of index >= K are never compared (rational_quadratic_spline.py:185-192 pins knot K to `hi` and search_sorted.py adds
//     1e-6 to it, so x = hi falls in bin K - 1): a group boundary 4 j >= K gets the constant +inf (free: it is the fma's addend);
//     inside the LAST reachable group the compares with local knots i >= K - 4 glast are masked (lane-mask and-not: scalar);
//   * the right knot of bin K - 1 is `hi` exactly: one compare of the bin index with K - 1 and one select per softmax block.
struct rqs16_c {            // per-phase constants (uniform; one set per phase step, shared by the lane's four elements)
    float lo, hi;
    float cs;               // (1 - K MIN) (hi - lo)
    float sm1, sm2, sm3, sm4;    // i MIN (hi - lo)
    float l4, l8, l12;      // lo + 4 j MIN (hi - lo): the group boundaries' constant part (GEN: +inf where 4 j >= K)
    int Km1, glast;         // GEN: K - 1; the last reachable group (K - 1) / 4
    bool u1, u2, u3;        // GEN: local knot i of the last group has index >= K
};
__device__ __forceinline__ rqs16_c rqs16_consts(float lo, float hi, int K = 16) {
    rqs16_c c;
    const float span = hi - lo, sm = RQS_MIN * span, inf = __builtin_inff();
    c.lo = lo; c.hi = hi; c.cs = (1.f - (float)K * RQS_MIN) * span;
    c.sm1 = sm; c.sm2 = 2.f * sm; c.sm3 = 3.f * sm; c.sm4 = 4.f * sm;
    c.l4 = 4 < K ? lo + 4.f * sm : inf; c.l8 = 8 < K ? lo + 8.f * sm : inf; c.l12 = 12 < K ? lo + 12.f * sm : inf;
    c.Km1 = K - 1; c.glast = (K - 1) >> 2;
    const int iK = K - 4 * c.glast;             // 1 .. 4: the local index of knot K in the last group
    c.u1 = 1 >= iK; c.u2 = 2 >= iK; c.u3 = 3 >= iK;
    return c;
}
// (a function taking VALUES: see above)
__device__ __forceinline__ float rqs16_pick(bool m1, bool m2, bool m3, float v0, float v1, float v2, float v3) {
    return m3 ? v3 : (m2 ? v2 : (m1 ? v1 : v0));
}
// ---- the MFMAs of the NEXT element's tile, issued from inside an element's arithmetic -------------------------------------------
// An element routine below calls hk.pt<0>() .. hk.pt<11>() at twelve points about eight vector instructions apart; a tile pipe
// issues its 6 HT MFMAs spread over those points (each followed by a scheduling fence, so the compiler keeps the interleave), the
// A fragments of k16-step T + 1 requested at the first MFMA of step T (two fragment buffers).  The prescriptive alternative,
// sched_group_barrier groups over one fenced region, was tried first: the solver honoured the first four MFMA / VALU groups and
// left the other eight MFMAs back to back in front of sixty vector instructions.
// (`tie...`: values the element's arithmetic has produced since the previous point.  Each passes through an empty volatile asm in
//  front of the point's MFMAs and the accumulator through one behind each MFMA: volatile asms keep their order, so the element's
//  vector work stays BETWEEN the MFMAs.  Round 6: with scheduling fences alone the optimizer sank the whole search and select
//  arithmetic of a group -- pure code whose results are first read by the evaluate block -- into that third block, kept all eight raw
//  logit tiles alive (128 registers) and left the first two blocks' 96 MFMAs bare: the ISA of <1,2,2,3>, LBB0_213 / 217 / 221.)
struct rqs_nohook {
    template <int P, class... T> __device__ __forceinline__ void pt(T &...) {}
};
template <class T> __device__ __forceinline__ void rqs_tie_one(T &v) { asm volatile("" : "+v"(v)); }
struct rqs_pinhook {        // a block's LAST element: no MFMAs to issue, but its arithmetic stays in its block all the same
    template <int P, class... T> __device__ __forceinline__ void pt(T &...tie) { (rqs_tie_one(tie), ...); }
};
template <int HT, int U>
struct rqs_tile_pipe {
    const char *wb, *cb;
    const btile<1> (&bh)[HT];
    tile<1> &acc;
#ifdef SX_F16X3
    u32x4 fh[2], fl[2];
#endif
    __device__ __forceinline__ rqs_tile_pipe(const wptr w, const btile<1> (&bh_)[HT], tile<1> &acc_) : wb(w.wb), cb(w.cb), bh(bh_), acc(acc_) {}
#ifdef SX_F16X3
    template <int T> __device__ __forceinline__ void load() {       // fragments of k16-step T (k-tile T / 2, half T % 2)
        constexpr int a_off = (U * HT + T / 2) * 1024, s2 = T % 2;
        fh[T & 1] = *reinterpret_cast<const u32x4 *>(wb + (a_off + (2 * s2) * 256) * 4);
        fl[T & 1] = *reinterpret_cast<const u32x4 *>(wb + (a_off + (2 * s2 + 1) * 256) * 4);
    }
    template <int J> __device__ __forceinline__ void mfma() {
        constexpr int T = J / 3, k = J % 3, m = T / 2, s2 = T % 2;
        if constexpr (k == 0 && T + 1 < 2 * HT) load<T + 1>();
        const h8 ah = __builtin_bit_cast(h8, fh[T & 1]), al = __builtin_bit_cast(h8, fl[T & 1]);
        if constexpr (k == 0) acc.v[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[m].hi[0][s2], acc.v[0], 0, 0, 0);      // smallest terms first
        else if constexpr (k == 1) acc.v[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[m].lo[0][s2], acc.v[0], 0, 0, 0);
        else acc.v[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[m].hi[0][s2], acc.v[0], 0, 0, 0);
        asm volatile("" : "+v"(acc.v[0]));
    }
    template <int J, int JE> __device__ __forceinline__ void run() {
        if constexpr (J < JE) { mfma<J>(); run<J + 1, JE>(); }
    }
    __device__ __forceinline__ void start() {
        acc = load_cfrag<1>(cb, 4 * HT * 1024 + U * 32);
        load<0>();
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int P, class... T> __device__ __forceinline__ void pt(T &...tie) {
        constexpr int N = 6 * HT;
        (rqs_tie_one(tie), ...);
        __builtin_amdgcn_sched_barrier(0);
        run<P * N / 12, (P + 1) * N / 12>();
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    // exact fp32: v_mfma_f32_32x32x2_f32 executes on the VALU itself -- nothing to overlap: the tile's GEMM up front
    __device__ __forceinline__ void start() {
        acc = load_cfrag<1>(cb, 4 * HT * 1024 + U * 32);
#pragma unroll
        for (int m = 0; m < HT; ++m) gemm_tile<1>(wb, (U * HT + m) * 1024, bh[m], acc);
    }
    template <int P, class... T> __device__ __forceinline__ void pt(T &...) {}
#endif
};
// exp2 of a tile's 16 logits, the groups' inner prefixes (e0, e0+e1, e0+e1+e2, the group's sum), the group boundaries' knots
struct rqs16_s {
    float a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3, d0, d1, d2, d3;
    float T1, T2, T3;       // knots at bins 4, 8, 12
    float sinv;             // span (1 - 16 MIN) / sum
};
template <class H>
__device__ __forceinline__ rqs16_s rqs16_sums(const f32x16 &u, const rqs16_c &c, H &hk) {
    rqs16_s s;
    hk.template pt<0>();
    s.a0 = __builtin_amdgcn_exp2f(u[0]); s.a1 = s.a0 + __builtin_amdgcn_exp2f(u[1]); s.a2 = s.a1 + __builtin_amdgcn_exp2f(u[2]); s.a3 = s.a2 + __builtin_amdgcn_exp2f(u[3]);
    hk.template pt<1>(s.a3);
    s.b0 = __builtin_amdgcn_exp2f(u[4]); s.b1 = s.b0 + __builtin_amdgcn_exp2f(u[5]); s.b2 = s.b1 + __builtin_amdgcn_exp2f(u[6]); s.b3 = s.b2 + __builtin_amdgcn_exp2f(u[7]);
    hk.template pt<2>(s.b3);
    s.c0 = __builtin_amdgcn_exp2f(u[8]); s.c1 = s.c0 + __builtin_amdgcn_exp2f(u[9]); s.c2 = s.c1 + __builtin_amdgcn_exp2f(u[10]); s.c3 = s.c2 + __builtin_amdgcn_exp2f(u[11]);
    hk.template pt<3>(s.c3);
    s.d0 = __builtin_amdgcn_exp2f(u[12]); s.d1 = s.d0 + __builtin_amdgcn_exp2f(u[13]); s.d2 = s.d1 + __builtin_amdgcn_exp2f(u[14]); s.d3 = s.d2 + __builtin_amdgcn_exp2f(u[15]);
    hk.template pt<4>(s.d3);
    const float G2 = s.a3 + s.b3, G3 = G2 + s.c3;
    s.sinv = c.cs * fast_rcp(G3 + s.d3);
    s.T1 = __builtin_fmaf(s.sinv, s.a3, c.l4);
    s.T2 = __builtin_fmaf(s.sinv, G2, c.l8);
    s.T3 = __builtin_fmaf(s.sinv, G3, c.l12);
    return s;
}
// the knots around the group chosen by the nested masks m1 >= m2 >= m3 (group >= 1, 2, 3); points 6 .. 8
template <bool GEN, class H>
__device__ __forceinline__ void rqs16_knots(const rqs16_s &s, const rqs16_c &c, bool m1, bool m2, bool m3, float &k0, float &k1, float &k2,
                                            float &k3, float &k4, H &hk) {
    k0 = rqs16_pick(m1, m2, m3, c.lo, s.T1, s.T2, s.T3);
    float p1 = rqs16_pick(m1, m2, m3, s.a0, s.b0, s.c0, s.d0);
    hk.template pt<6>(k0, p1);
    const float p2 = rqs16_pick(m1, m2, m3, s.a1, s.b1, s.c1, s.d1);
    float p3 = rqs16_pick(m1, m2, m3, s.a2, s.b2, s.c2, s.d2);
    k1 = __builtin_fmaf(s.sinv, p1, k0 + c.sm1);
    k2 = __builtin_fmaf(s.sinv, p2, k0 + c.sm2);
    hk.template pt<7>(k1, k2, p3);
    const float p4 = rqs16_pick(m1, m2, m3, s.a3, s.b3, s.c3, s.d3);
    k3 = __builtin_fmaf(s.sinv, p3, k0 + c.sm3);
    const float k4c = __builtin_fmaf(s.sinv, p4, k0 + c.sm4);
    const float hi_ = c.hi;
    k4 = GEN ? k4c : (m3 ? hi_ : k4c);                                  // ends pinned (:189-192; GEN: by the bin index, below)
    hk.template pt<8>(k3, k4);
}
template <int Q, bool GEN, class H>
__device__ __forceinline__ void rqs16_search(const f32x16 &u, rqs_elems &e, const rqs16_c &c, H &hk) {
    rqs16_s s = rqs16_sums(u, c, hk);
    hk.template pt<5>(s.T1, s.T2, s.T3);
    const float xv = e.x[Q];
    const bool in = (xv >= c.lo) && (xv <= c.hi);                       // :71 closed interval
    const float lo_ = c.lo;
    const float xin = in ? xv : lo_;
    const bool m1 = xin >= s.T1, m2 = xin >= s.T2, m3 = xin >= s.T3;    // a prefix: the knots grow
    float k0, k1, k2, k3, k4;
    rqs16_knots<GEN>(s, c, m1, m2, m3, k0, k1, k2, k3, k4, hk);
    bool g1 = xin >= k1, g2 = xin >= k2, g3 = xin >= k3;
    if constexpr (GEN) {
        // groups beyond glast are unreachable (their boundaries are +inf), so "in the last group" is the mask of boundary glast
        const bool last = c.glast == 0 ? true : (c.glast == 1 ? m1 : (c.glast == 2 ? m2 : m3));
        g1 = g1 && !(last && c.u1); g2 = g2 && !(last && c.u2); g3 = g3 && !(last && c.u3);
    }
    int bg = 4 * ((int)m1 + (int)m2 + (int)m3) + (in ? 0 : RQS_OUT);
    hk.template pt<9>(bg);
    e.b[Q] = bg + ((int)g1 + (int)g2 + (int)g3);
    float k_b = g3 ? k3 : (g2 ? k2 : (g1 ? k1 : k0));
    hk.template pt<10>(k_b, e.b[Q]);
    float k_n = g3 ? k4 : (g2 ? k3 : (g1 ? k2 : k1));
    if constexpr (GEN) { const float hi_ = c.hi; k_n = e.b[Q] == c.Km1 ? hi_ : k_n; }     // (an outside input's index carries RQS_OUT: its knots are unused)
    e.a_b[Q] = k_b;
    e.a_w[Q] = k_n - k_b;
    hk.template pt<11>(e.a_b[Q], e.a_w[Q]);
}
template <int Q, bool GEN, class H>
__device__ __forceinline__ void rqs16_select(const f32x16 &u, rqs_elems &e, const rqs16_c &c, H &hk) {
    rqs16_s s = rqs16_sums(u, c, hk);
    hk.template pt<5>(s.T1, s.T2, s.T3);
    const int b = e.b[Q] & (RQS_OUT - 1), bl = b & 3;
    const bool m1 = b >= 4, m2 = b >= 8, m3 = b >= 12;
    float k0, k1, k2, k3, k4;
    rqs16_knots<GEN>(s, c, m1, m2, m3, k0, k1, k2, k3, k4, hk);
    const bool g1 = bl >= 1, g2 = bl >= 2, g3 = bl >= 3;
    hk.template pt<9>();
    float k_b = g3 ? k3 : (g2 ? k2 : (g1 ? k1 : k0));
    hk.template pt<10>(k_b);
    float k_n = g3 ? k4 : (g2 ? k3 : (g1 ? k2 : k1));
    if constexpr (GEN) { const float hi_ = c.hi; k_n = b == c.Km1 ? hi_ : k_n; }
    e.c_b[Q] = k_b;
    e.c_w[Q] = k_n - k_b;
    hk.template pt<11>(e.c_b[Q], e.c_w[Q]);
}
template <int Q, int KC, int W = 1>
__device__ __forceinline__ void rqs_search(tile<1> (&acc)[4], rqs_elems &e, int K, float lo, float hi) {
#ifndef SX_RQS_FLAT
    if constexpr (KC == 16 && W == 1) { rqs_search16<Q>(acc, e, lo, hi); return; }
#endif
    const float xv = e.x[Q];
    const bool in = (xv >= lo) && (xv <= hi);                       // :71 closed interval
    const float xin = in ? xv : lo;
    const float inv = rqs_softmax<Q, KC, W>(acc, K);
    const int Kn = KC ? KC : K;
    int b = 0;
    float k_b = lo, k_n = hi, cs = 0.f;
#pragma unroll
    for (int j = 1; j <= 16 * W; ++j) {
        if (KC ? (j <= KC) : true) {
            cs += RQS_MIN + RQS_PW(acc, Q, j - 1) * inv;
            const bool last = (j == Kn);
            const bool used = KC ? true : (j <= K);
            const float knot = last ? hi : (hi - lo) * cs + lo;          // ends pinned
            const bool ge = used && (xin >= (last ? knot + 1e-6f : knot));
            const bool take = ge && !last;                               // j == K only clamps (see rqs_kernel)
            b = take ? j : b;
            k_b = take ? knot : k_b;
            k_n = fminf(k_n, (ge || !used) ? hi : knot);
            SX_SWEEP_FENCE(b, k_b, k_n);
        }
    }
    e.b[Q] = b + (in ? 0 : RQS_OUT);
    e.a_b[Q] = k_b;
    e.a_w[Q] = k_n - k_b;
}
// phase 1: knots of the other block at the found bin
template <int Q, int KC, int W = 1>
__device__ __forceinline__ void rqs_select(tile<1> (&acc)[4], rqs_elems &e, int K, float lo, float hi) {
#ifndef SX_RQS_FLAT
    if constexpr (KC == 16 && W == 1) { rqs_select16<Q>(acc, e, lo, hi); return; }
#endif
    const float inv = rqs_softmax<Q, KC, W>(acc, K);
    const int Kn = KC ? KC : K;
    int b = e.b[Q] & (RQS_OUT - 1);
    SX_OPAQUE(b);       // (see SX_OPAQUE)
    // only two knots are needed: select their cumulative sums in the sweep and rescale those two afterwards
    float cs = 0.f, cs_b = 0.f, cs_n = 0.f;
    bool has_b = false, has_n = false;
#pragma unroll
    for (int j = 1; j <= 16 * W; ++j) {
        if (KC ? (j < KC) : true) {                                      // knot K is `hi` (the default k_n)
            cs += RQS_MIN + RQS_PW(acc, Q, j - 1) * inv;
            const bool used = KC ? true : (j < Kn);
            const bool is_b = used && j == b, is_n = used && j == b + 1;
            cs_b = is_b ? cs : cs_b;
            cs_n = is_n ? cs : cs_n;
            has_b = has_b || is_b;
            has_n = has_n || is_n;
            SX_SWEEP_FENCE(cs_b, cs_n, cs);
        }
    }
    const float k_b = has_b ? (hi - lo) * cs_b + lo : lo;
    const float k_n = has_n ? (hi - lo) * cs_n + lo : hi;
    e.c_b[Q] = k_b;
    e.c_w[Q] = k_n - k_b;
}
// F.softplus(v) = log1p(exp(v)) (threshold 20) on v_exp_f32 / v_log_f32: |abs err| <~ 1e-7, far below the 1e-4
// conditioning noise of the spline's log-derivative
__device__ __forceinline__ float fast_log(float v) { return __builtin_amdgcn_logf(v) * 0.69314718055994531f; }
__device__ __forceinline__ float rqs_softplus(float v) { return v > 20.f ? v : fast_log(1.f + fast_exp(v)); }
// phase 2: the two knot derivatives at the bin (:107,:206-207), then the rational-quadratic (:236-248) or its
// inverse (:212-234; the returned log-derivative is already negated like the reference's).
// the evaluation behind the derivative pick: softplus of the two picked parameters (:107), the rational-quadratic (:236-248) or its
// inverse (:212-234; the returned log-derivative is already negated like the reference's), the linear tails (:86-87)
template <int Q, bool REV, class H>
__device__ __forceinline__ void rqs_eval_core(float r_b, float r_n, bool in, const rqs_elems &e, float &out, float &ljd, H &hk) {
    float d_b = RQS_MIN + rqs_softplus(r_b);
    hk.template pt<4>(d_b);
    float d_n = RQS_MIN + rqs_softplus(r_n);
    hk.template pt<5>(d_n);
    // REV: the searched block is the heights (codomain side), the selected one the widths
    const float cw_b = REV ? e.c_b[Q] : e.a_b[Q], w_b = REV ? e.c_w[Q] : e.a_w[Q];
    const float ch_b = REV ? e.a_b[Q] : e.c_b[Q], h_b = REV ? e.a_w[Q] : e.c_w[Q];
    const float s_b = h_b * fast_rcp(w_b);
    const float xin = in ? e.x[Q] : (REV ? ch_b : cw_b);
    if constexpr (REV) {
        const float dy = xin - ch_b;
        float q = d_b + d_n - 2.f * s_b;
        hk.template pt<6>(q);
        const float a = dy * q + h_b * (s_b - d_b);
        const float bb = h_b * d_b - dy * q;
        const float c = -s_b * dy;
        float disc = bb * bb - 4.f * a * c;
        hk.template pt<7>(disc);
        // (disc >= 0 in exact arithmetic -- the spline is monotone --; rounding can leave it a few ulps below zero where the root
        //  sits on a knot and the reference's own fp32 evaluation stays at or above it: clamp instead of returning NaN, :223)
        // the root is a position inside the bin: rounding can also leave it an ulp outside [0, 1], where a steep bin next
        // to a flat one (slope ~1e3, knot derivative ~1e-3) turns the derivative's numerator negative and its log into NaN
        const float root = __builtin_amdgcn_fmed3f((2.f * c) * fast_rcp(-bb - __builtin_amdgcn_sqrtf(__builtin_fmaxf(disc, 0.f))), 0.f, 1.f);
        out = root * w_b + cw_b;
        hk.template pt<8>(out);
        const float tomt = root * (1.f - root), omr = 1.f - root;
        float den = s_b + q * tomt;
        hk.template pt<9>(den);
        float dnum = (s_b * s_b) * (d_n * (root * root) + 2.f * s_b * tomt + d_b * (omr * omr));
        hk.template pt<10>(dnum);
        ljd = -fast_log(dnum) + 2.f * fast_log(den);
    } else {
        float theta = (xin - cw_b) * fast_rcp(w_b);
        hk.template pt<6>(theta);
        const float tomt = theta * (1.f - theta), omt = 1.f - theta;
        float num = h_b * (s_b * (theta * theta) + d_b * tomt);
        hk.template pt<7>(num);
        float den = s_b + (d_b + d_n - 2.f * s_b) * tomt;
        out = ch_b + num * fast_rcp(den);
        hk.template pt<8>(out, den);
        float dnum = (s_b * s_b) * (d_n * (theta * theta) + 2.f * s_b * tomt + d_b * (omt * omt));
        hk.template pt<9>(dnum);
        hk.template pt<10>();
        ljd = fast_log(dnum) - 2.f * fast_log(den);
    }
    out = in ? out : e.x[Q];                                    // :86-87 linear tails
    ljd = in ? ljd : 0.f;
    hk.template pt<11>(out, ljd);
}
template <int Q, bool REV, int KC, class H>
__device__ __forceinline__ void rqs_eval(const f32x16 &u, const rqs_elems &e, int K, float &out, float &ljd, H &hk) {
    hk.template pt<0>();
    int b = e.b[Q] & (RQS_OUT - 1);
    if constexpr (KC != 16) SX_OPAQUE(b);
    const bool in = e.b[Q] < RQS_OUT;
    const int Kn = KC ? KC : K;
    const float cst = 0.5397424172369522f;                      // log(exp(1 - 1e-3) - 1), :81 boundary derivative constant
    float r_b = cst, r_n = cst;
    // derivative k sits between bins k and k+1: it is the RIGHT knot's of bin k and the LEFT knot's of bin k+1, so one
    // compare per bin index serves both selections
#ifndef SX_RQS_FLAT
    if constexpr (KC == 16) {
        // two-level pick (see rqs_search16): D[-1] = D[15] = cst, r_b = D[b - 1], r_n = D[b]; the five candidates D[4g-1 .. 4g+3]
        // of the bin's group first, then the pair inside it.  (Scalars and macros on purpose: the same code with a float[5] and
        // a lambda became a stack object -- scratch in the phase loop, cfg 3 6.4 -> 27.6 ms.)
        const bool m1 = b >= 4, m2 = b >= 8, m3 = b >= 12;
        // (named copies first: a select between two array elements is canonicalised into a select of ADDRESSES, which keeps
        //  the accumulator tiles in memory -- 384 B of scratch)
        const float d0 = u[0], d1 = u[1], d2 = u[2], d3 = u[3], d4 = u[4];
        const float d5 = u[5], d6 = u[6], d7 = u[7], d8 = u[8], d9 = u[9];
        const float d10 = u[10], d11 = u[11], d12 = u[12], d13 = u[13], d14 = u[14];
        float v0 = m3 ? d11 : (m2 ? d7 : (m1 ? d3 : cst)), v1 = m3 ? d12 : (m2 ? d8 : (m1 ? d4 : d0));
        hk.template pt<1>(v0, v1);
        float v2 = m3 ? d13 : (m2 ? d9 : (m1 ? d5 : d1)), v3 = m3 ? d14 : (m2 ? d10 : (m1 ? d6 : d2));
        float v4 = m3 ? cst : (m2 ? d11 : (m1 ? d7 : d3));
        hk.template pt<2>(v2, v3, v4);
        const int bl = b & 3;
        const bool g1 = bl >= 1, g2 = bl >= 2, g3 = bl >= 3;
        r_b = g3 ? v3 : (g2 ? v2 : (g1 ? v1 : v0));
        r_n = g3 ? v4 : (g2 ? v3 : (g1 ? v2 : v1));
        hk.template pt<3>(r_b, r_n);
    } else
#endif
    {
    bool is_prev = (b == 0);             // "bin index == k" carried to the next iteration as "bin index - 1 == k - 1"
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        if (KC ? (k < KC - 1) : true) {
            const bool used = KC ? true : (k < Kn - 1);
            const float v = u[k];
            const bool is_next = (b == k + 1);
            r_n = (used && is_prev) ? v : r_n;          // k == b
            r_b = (used && is_next) ? v : r_b;          // k == b - 1
            is_prev = is_next;
        }
    }
    }
    rqs_eval_core<Q, REV>(r_b, r_n, in, e, out, ljd, hk);
}
// 17..32 bins: the fifteen-step pick above over the element's two tiles
template <int Q, bool REV>
__device__ __forceinline__ void rqs_eval_wide(const f32x16 &u0, const f32x16 &u1, const rqs_elems &e, int K, float &out, float &ljd) {
    int b = e.b[Q] & (RQS_OUT - 1);
    SX_OPAQUE(b);
    const bool in = e.b[Q] < RQS_OUT;
    const float cst = 0.5397424172369522f;
    float r_b = cst, r_n = cst;
    bool is_prev = (b == 0);
#pragma unroll
    for (int k = 0; k < 31; ++k) {
        const bool used = k < K - 1;
        const float v = k < 16 ? u0[k & 15] : u1[k & 15];
        const bool is_next = (b == k + 1);
        r_n = (used && is_prev) ? v : r_n;          // k == b
        r_b = (used && is_next) ? v : r_b;          // k == b - 1
        is_prev = is_next;
        SX_SWEEP_FENCE(r_n, r_b, r_b);
    }
    rqs_nohook nh;
    rqs_eval_core<Q, REV>(r_b, r_n, in, e, out, ljd, nh);
}
template <int Q, bool REV, int KC>
__device__ __forceinline__ void rqs_eval(const f32x16 &u, const rqs_elems &e, int K, float &out, float &ljd) {
    rqs_nohook nh;
    rqs_eval<Q, REV, KC>(u, e, K, out, ljd, nh);
}

template <int TX, int HT, int KC>
__device__ __forceinline__ void rqs_phase_k(tile<1> (&acc)[4], tile<1> (&xs)[TX], rqs_elems &e, const dstep &st,
                                            float lo, float hi, float &ldj, int h) {
    const int g = st.c0, K = st.tt;
    if (st.ct == 0) {
        // fetch the group's 4 inputs out of the state tile (wave-uniform selects keep register indices static)
#pragma unroll
        for (int t = 0; t < TX; ++t)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                if (t == st.t0 && gg == g) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) e.x[q] = xs[t].v[0][4 * gg + q];
                }
#if !defined(SX_RQS_FLAT) && !defined(SX_RQS_OLD16)
        if constexpr (KC == 16) {
            const rqs16_c c = rqs16_consts(lo, hi);
            rqs_nohook nh16;
            rqs16_search<0, false>(acc[0].v[0], e, c, nh16); __builtin_amdgcn_sched_barrier(0);
            rqs16_search<1, false>(acc[1].v[0], e, c, nh16); __builtin_amdgcn_sched_barrier(0);
            rqs16_search<2, false>(acc[2].v[0], e, c, nh16); __builtin_amdgcn_sched_barrier(0);
            rqs16_search<3, false>(acc[3].v[0], e, c, nh16);
        } else
#endif
        {
        rqs_search<0, KC>(acc, e, K, lo, hi);
        rqs_search<1, KC>(acc, e, K, lo, hi);
        rqs_search<2, KC>(acc, e, K, lo, hi);
        rqs_search<3, KC>(acc, e, K, lo, hi);
        }
    } else if (st.ct == 1) {
#if !defined(SX_RQS_FLAT) && !defined(SX_RQS_OLD16)
        if constexpr (KC == 16) {
            const rqs16_c c = rqs16_consts(lo, hi);
            rqs_nohook nh16;
            rqs16_select<0, false>(acc[0].v[0], e, c, nh16); __builtin_amdgcn_sched_barrier(0);
            rqs16_select<1, false>(acc[1].v[0], e, c, nh16); __builtin_amdgcn_sched_barrier(0);
            rqs16_select<2, false>(acc[2].v[0], e, c, nh16); __builtin_amdgcn_sched_barrier(0);
            rqs16_select<3, false>(acc[3].v[0], e, c, nh16);
        } else
#endif
        {
        rqs_select<0, KC>(acc, e, K, lo, hi);
        rqs_select<1, KC>(acc, e, K, lo, hi);
        rqs_select<2, KC>(acc, e, K, lo, hi);
        rqs_select<3, KC>(acc, e, K, lo, hi);
        }
    } else {
        float out[4], lj[4];
        if (st.reverse) {
            rqs_eval<0, true, KC>(acc[0].v[0], e, K, out[0], lj[0]);
            rqs_eval<1, true, KC>(acc[1].v[0], e, K, out[1], lj[1]);
            rqs_eval<2, true, KC>(acc[2].v[0], e, K, out[2], lj[2]);
            rqs_eval<3, true, KC>(acc[3].v[0], e, K, out[3], lj[3]);
        } else {
            rqs_eval<0, false, KC>(acc[0].v[0], e, K, out[0], lj[0]);
            rqs_eval<1, false, KC>(acc[1].v[0], e, K, out[1], lj[1]);
            rqs_eval<2, false, KC>(acc[2].v[0], e, K, out[2], lj[2]);
            rqs_eval<3, false, KC>(acc[3].v[0], e, K, out[3], lj[3]);
        }
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool live = (st.mask >> (q + 8 * g + 4 * h)) & 1u;     // slot kmap(4g+q, h) of the tile
            out[q] = live ? out[q] : e.x[q];
            s += live ? lj[q] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < TX; ++t)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                if (t == st.t0 && gg == g) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) xs[t].v[0][4 * gg + q] = out[q];
                }
        ldj += st.ldj_scale * s;
    }
}

template <int TX, int HT>
__device__ __forceinline__ void rqs_phase(tile<1> (&xs)[TX], const btile<1> (&bh)[HT], rqs_elems &e, const wptr w,
                                          const dstep &st, float &ldj, int lane, bool &group_lean) {
    SX_DEP_MARK_SPLINE;
    const int h = lane >> 5;
    tile<1> acc[4];
    rqs_gemm<HT>(w, bh, acc);
    const float lo = *reinterpret_cast<const float *>(w.cb - h * 64 + (4 * HT * 1024 + 128) * 4);
    const float hi = *reinterpret_cast<const float *>(w.cb - h * 64 + (4 * HT * 1024 + 129) * 4);
    // the straight-line K = 16 code needs the step's logits bounded (no running maximum in its softmax: see rqs16_sums); the
    // packer leaves the bound of the rows it packed behind the spline bounds (the derivative block's slot stays 0)
    // (one bound per group, in its FIRST block's blob: the decision made there holds for the group's select block as well)
    const float bound = *reinterpret_cast<const float *>(w.cb - h * 64 + (4 * HT * 1024 + 130) * 4);
    if (st.ct == 0) group_lean = __builtin_amdgcn_readfirstlane((int)(bound < RQS16_BOUND)) != 0;
    const bool lean = st.ct == 2 || group_lean;
    if (st.tt == 16 && lean) rqs_phase_k<TX, HT, 16>(acc, xs, e, st, lo, hi, ldj, h);
    else rqs_phase_k<TX, HT, 0>(acc, xs, e, st, lo, hi, ldj, h);
}

// The three phases of one group inside ONE iteration of the step loop (kernels whose programs hold rational-quadratic couplings
// only: MODE 3 / 10 / 18).  `advance()` ends the current step and begins the next one in place (weight-ring flip, wait + barrier,
// refill, descriptor fetch) and returns that step's descriptor and LDS pointers.  As three iterations, the group's state (28
// registers) was loop-carried beside the flow state and the B fragments, every arm of the step switch had to agree with every
// other on where all of it lives (some 60 register moves at the head of each phase), and the group's four inputs were fetched /
// stored by 32 wave-uniform selects over the whole state; here the group state is local, the phase kind is not a branch, and
// the four inputs move through a scalar switch on (tile, group).
#define RQS_GROUP_CASES(OP)                                                                                              \
    switch (tg) {                                                                                                        \
        default: OP(0, 0) break;                                                                                         \
        case 1: OP(0, 1) break;                                                                                          \
        case 2: OP(0, 2) break;                                                                                          \
        case 3: OP(0, 3) break;                                                                                          \
        case 4: if constexpr (TX > 1) { OP(1, 0) } break;                                                                \
        case 5: if constexpr (TX > 1) { OP(1, 1) } break;                                                                \
        case 6: if constexpr (TX > 1) { OP(1, 2) } break;                                                                \
        case 7: if constexpr (TX > 1) { OP(1, 3) } break;                                                                \
        case 8: if constexpr (TX > 2) { OP(2, 0) } break;                                                                \
        case 9: if constexpr (TX > 2) { OP(2, 1) } break;                                                                \
        case 10: if constexpr (TX > 2) { OP(2, 2) } break;                                                               \
        case 11: if constexpr (TX > 2) { OP(2, 3) } break;                                                               \
        case 12: if constexpr (TX > 3) { OP(3, 0) } break;                                                               \
        case 13: if constexpr (TX > 3) { OP(3, 1) } break;                                                               \
        case 14: if constexpr (TX > 3) { OP(3, 2) } break;                                                               \
        case 15: if constexpr (TX > 3) { OP(3, 3) } break;                                                               \
    }
// (the empty asm keeps each arm a real scalar branch: folded into selects the switch is the 32-select form again)
#define RQS_FETCH(T, G) { asm volatile("" ::: "memory"); _Pragma("unroll") for (int q = 0; q < 4; ++q) e.x[q] = xs[T].v[0][4 * G + q]; }
#define RQS_STORE(T, G) { asm volatile("" ::: "memory"); _Pragma("unroll") for (int q = 0; q < 4; ++q) xs[T].v[0][4 * G + q] = out[q]; }
// lo, hi and the logit bound of a block's blob
template <int HT>
__device__ __forceinline__ void rqs_block_scalars(const wptr w, int h, float &lo, float &hi, bool &lean) {
    lo = *reinterpret_cast<const float *>(w.cb - h * 64 + (4 * HT * 1024 + 128) * 4);
    hi = *reinterpret_cast<const float *>(w.cb - h * 64 + (4 * HT * 1024 + 129) * 4);
    const float bound = *reinterpret_cast<const float *>(w.cb - h * 64 + (4 * HT * 1024 + 130) * 4);
    lean = __builtin_amdgcn_readfirstlane((int)(bound < RQS16_BOUND)) != 0;
}
// One block of the K = 16 hot path, software-pipelined over the four elements: the MFMAs of element q + 1's tile are issued from
// inside the arithmetic of element q (rqs_tile_pipe; two accumulator tiles alternate).  PH: 0 search, 1 select, 2 evaluate.
// (A pipeline over the whole group -- the next block's first tile beside this block's last element, with the step advance in
//  between, and the bias / first fragments of tile q + 2 requested from element q -- was built and measured: the in-kernel stamps
//  moved (first tiles 25 % -> 8 % of the wave cycles, last elements 8 % -> 4 %) but the kernel did not (5.31 -> 5.36 ms per 2^20
//  rows): what one wave leaves idle the SIMD's other wave was already using.  It cost 41 registers and is not kept.)
template <int PH, bool REV, int Q, bool GEN = false, class H>
__device__ __forceinline__ void rqs16_unit(const f32x16 &u, rqs_elems &e, const rqs16_c &c, float (&out)[4], float (&lj)[4], H &hk) {
    if constexpr (PH == 0) rqs16_search<Q, GEN>(u, e, c, hk);
    else if constexpr (PH == 1) rqs16_select<Q, GEN>(u, e, c, hk);
    else rqs_eval<Q, REV, 16>(u, e, 16, out[Q], lj[Q], hk);       // (K < 16: rows >= K - 1 of the derivative tile hold the boundary constant)
}
// ------------------------------------------------------------------------------------------------
// 17 .. 32 bins, bounded logits (round 6; VERDICT r5 #1: K 16 -> 24 cost 2.4 x, the sweeps over 32 slots carried ~60 lane masks and the
// allocator spilled them to VGPR lanes -- v_writelane / v_readlane, vector instructions).  The K <= 16 form one level up: an element's
// parameters are TWO output tiles (32 slots), eight groups of four bins; the group comes from seven compares against the group
// boundaries' knots (a prefix: the knots grow), the five values around the group from seven-deep select chains on those masks, the
// bin inside the group as before.  The packer parks slots >= K (logits -1e30, derivative rows >= K - 1 at the boundary constant);
// boundaries 4 j >= K get +inf, local knots of index >= K in the last reachable group are masked, the last bin's right knot is the
// bound itself (rqs16_c).  ~145 vector instructions per softmax block and element, no lane mask outlives its select chain.
// ------------------------------------------------------------------------------------------------
struct rqs32_c {
    float lo, hi, cs, sm1, sm2, sm3, sm4;
    float l[7];             // lo + 4 j MIN (hi - lo), j = 1 .. 7 (+inf where 4 j >= K)
    int Km1, glast;
    bool u1, u2, u3;
};
__device__ __forceinline__ rqs32_c rqs32_consts(float lo, float hi, int K) {
    rqs32_c c;
    const float span = hi - lo, sm = RQS_MIN * span, inf = __builtin_inff();
    c.lo = lo; c.hi = hi; c.cs = (1.f - (float)K * RQS_MIN) * span;
    c.sm1 = sm; c.sm2 = 2.f * sm; c.sm3 = 3.f * sm; c.sm4 = 4.f * sm;
#pragma unroll
    for (int j = 1; j <= 7; ++j) c.l[j - 1] = 4 * j < K ? lo + 4.f * (float)j * sm : inf;
    c.Km1 = K - 1; c.glast = (K - 1) >> 2;
    const int iK = K - 4 * c.glast;
    c.u1 = 1 >= iK; c.u2 = 2 >= iK; c.u3 = 3 >= iK;
    return c;
}
struct rqs32_s {
    float p[8][4];          // the groups' inner prefixes: e0, e0 + e1, e0 + e1 + e2, the group's sum
    float T[7];             // knots at bins 4, 8, .. 28
    float sinv;
};
// v0 .. v7 by the nested masks m[0] >= m[1] >= .. >= m[6] (group >= 1 .. 7): a chain of seven selects (VALUES: see rqs16_pick)
__device__ __forceinline__ float rqs32_pick(const bool (&m)[7], float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
    float v = m[0] ? v1 : v0;
    v = m[1] ? v2 : v; v = m[2] ? v3 : v; v = m[3] ? v4 : v; v = m[4] ? v5 : v; v = m[5] ? v6 : v; v = m[6] ? v7 : v;
    return v;
}
template <class H>
__device__ __forceinline__ void rqs32_sums(const f32x16 &u0, const f32x16 &u1, const rqs32_c &c, rqs32_s &s, H &hk) {
    hk.template pt<0>();
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const f32x16 &u = g < 4 ? u0 : u1;
        const int o = 4 * (g & 3);
        s.p[g][0] = __builtin_amdgcn_exp2f(u[o]);
        s.p[g][1] = s.p[g][0] + __builtin_amdgcn_exp2f(u[o + 1]);
        s.p[g][2] = s.p[g][1] + __builtin_amdgcn_exp2f(u[o + 2]);
        s.p[g][3] = s.p[g][2] + __builtin_amdgcn_exp2f(u[o + 3]);
        if (g == 1) hk.template pt<1>(s.p[0][3], s.p[1][3]);
        if (g == 3) hk.template pt<2>(s.p[2][3], s.p[3][3]);
        if (g == 5) hk.template pt<3>(s.p[4][3], s.p[5][3]);
        if (g == 7) hk.template pt<4>(s.p[6][3], s.p[7][3]);
    }
    float G[8];
    G[0] = s.p[0][3];
#pragma unroll
    for (int g = 1; g < 8; ++g) G[g] = G[g - 1] + s.p[g][3];
    s.sinv = c.cs * fast_rcp(G[7]);
#pragma unroll
    for (int j = 0; j < 7; ++j) s.T[j] = __builtin_fmaf(s.sinv, G[j], c.l[j]);
    hk.template pt<5>(s.T[0], s.T[3], s.T[6]);
}
template <class H>
__device__ __forceinline__ void rqs32_knots(const rqs32_s &s, const rqs32_c &c, const bool (&m)[7], float &k0, float &k1, float &k2, float &k3,
                                            float &k4, H &hk) {
    k0 = rqs32_pick(m, c.lo, s.T[0], s.T[1], s.T[2], s.T[3], s.T[4], s.T[5], s.T[6]);
    float p1 = rqs32_pick(m, s.p[0][0], s.p[1][0], s.p[2][0], s.p[3][0], s.p[4][0], s.p[5][0], s.p[6][0], s.p[7][0]);
    hk.template pt<6>(k0, p1);
    float p2 = rqs32_pick(m, s.p[0][1], s.p[1][1], s.p[2][1], s.p[3][1], s.p[4][1], s.p[5][1], s.p[6][1], s.p[7][1]);
    k1 = __builtin_fmaf(s.sinv, p1, k0 + c.sm1);
    hk.template pt<7>(k1, p2);
    float p3 = rqs32_pick(m, s.p[0][2], s.p[1][2], s.p[2][2], s.p[3][2], s.p[4][2], s.p[5][2], s.p[6][2], s.p[7][2]);
    k2 = __builtin_fmaf(s.sinv, p2, k0 + c.sm2);
    hk.template pt<8>(k2, p3);
    const float p4 = rqs32_pick(m, s.p[0][3], s.p[1][3], s.p[2][3], s.p[3][3], s.p[4][3], s.p[5][3], s.p[6][3], s.p[7][3]);
    k3 = __builtin_fmaf(s.sinv, p3, k0 + c.sm3);
    k4 = __builtin_fmaf(s.sinv, p4, k0 + c.sm4);
    hk.template pt<9>(k3, k4);
}
template <int Q, class H>
__device__ __forceinline__ void rqs32_search(const f32x16 &u0, const f32x16 &u1, rqs_elems &e, const rqs32_c &c, H &hk) {
    rqs32_s s;
    rqs32_sums(u0, u1, c, s, hk);
    const float xv = e.x[Q];
    const bool in = (xv >= c.lo) && (xv <= c.hi);                       // :71 closed interval
    const float lo_ = c.lo;
    const float xin = in ? xv : lo_;
    bool m[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) m[j] = xin >= s.T[j];                   // a prefix: the knots grow (boundaries 4 j >= K are +inf)
    float k0, k1, k2, k3, k4;
    rqs32_knots(s, c, m, k0, k1, k2, k3, k4, hk);
    bool g1 = xin >= k1, g2 = xin >= k2, g3 = xin >= k3;
    // groups beyond glast are unreachable, so "in the last group" is the mask of boundary glast
    const bool last = c.glast <= 4 ? m[3] : (c.glast == 5 ? m[4] : (c.glast == 6 ? m[5] : m[6]));      // (17 .. 32 bins: glast = 4 .. 7)
    g1 = g1 && !(last && c.u1); g2 = g2 && !(last && c.u2); g3 = g3 && !(last && c.u3);
    int bg = in ? 0 : RQS_OUT;
#pragma unroll
    for (int j = 0; j < 7; ++j) bg += m[j] ? 4 : 0;
    hk.template pt<10>(bg);
    e.b[Q] = bg + ((int)g1 + (int)g2 + (int)g3);
    float k_b = g3 ? k3 : (g2 ? k2 : (g1 ? k1 : k0));
    float k_n = g3 ? k4 : (g2 ? k3 : (g1 ? k2 : k1));
    const float hi_ = c.hi;
    k_n = e.b[Q] == c.Km1 ? hi_ : k_n;                                  // ends pinned (:189-192)
    e.a_b[Q] = k_b;
    e.a_w[Q] = k_n - k_b;
    hk.template pt<11>(e.b[Q], e.a_b[Q], e.a_w[Q]);
}
template <int Q, class H>
__device__ __forceinline__ void rqs32_select(const f32x16 &u0, const f32x16 &u1, rqs_elems &e, const rqs32_c &c, H &hk) {
    rqs32_s s;
    rqs32_sums(u0, u1, c, s, hk);
    const int b = e.b[Q] & (RQS_OUT - 1), bl = b & 3;
    bool m[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) m[j] = b >= 4 * (j + 1);
    float k0, k1, k2, k3, k4;
    rqs32_knots(s, c, m, k0, k1, k2, k3, k4, hk);
    const bool g1 = bl >= 1, g2 = bl >= 2, g3 = bl >= 3;
    hk.template pt<10>();
    float k_b = g3 ? k3 : (g2 ? k2 : (g1 ? k1 : k0));
    float k_n = g3 ? k4 : (g2 ? k3 : (g1 ? k2 : k1));
    const float hi_ = c.hi;
    k_n = b == c.Km1 ? hi_ : k_n;
    e.c_b[Q] = k_b;
    e.c_w[Q] = k_n - k_b;
    hk.template pt<11>(e.c_b[Q], e.c_w[Q]);
}
// the two knot derivatives of the bin: D[-1] = D[K - 1] = the boundary constant (the packer parks it in the rows >= K - 1), r_b = D[b - 1],
// r_n = D[b]; the five candidates D[4g - 1 .. 4g + 3] of the bin's group by the eight-way chains, then the pair inside it
template <int Q, bool REV, class H>
__device__ __forceinline__ void rqs32_eval(const f32x16 &u0, const f32x16 &u1, const rqs_elems &e, float &out, float &ljd, H &hk) {
    hk.template pt<0>();
    const int b = e.b[Q] & (RQS_OUT - 1), bl = b & 3;
    const bool in = e.b[Q] < RQS_OUT;
    const float cst = 0.5397424172369522f;                      // log(exp(1 - 1e-3) - 1), :81 boundary derivative constant
    bool m[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) m[j] = b >= 4 * (j + 1);
#define RQS32_D(k) ((k) < 0 ? cst : ((k) < 16 ? u0[(k) & 15] : ((k) < 32 ? u1[(k) & 15] : cst)))
    // (named copies: a select between two vector elements is canonicalised into a select of ADDRESSES -- rqs_eval)
    float d[33];
#pragma unroll
    for (int k = -1; k < 32; ++k) d[k + 1] = RQS32_D(k);
#undef RQS32_D
    float v0 = rqs32_pick(m, d[0], d[4], d[8], d[12], d[16], d[20], d[24], d[28]);
    float v1 = rqs32_pick(m, d[1], d[5], d[9], d[13], d[17], d[21], d[25], d[29]);
    hk.template pt<1>(v0, v1);
    float v2 = rqs32_pick(m, d[2], d[6], d[10], d[14], d[18], d[22], d[26], d[30]);
    float v3 = rqs32_pick(m, d[3], d[7], d[11], d[15], d[19], d[23], d[27], d[31]);
    hk.template pt<2>(v2, v3);
    float v4 = rqs32_pick(m, d[4], d[8], d[12], d[16], d[20], d[24], d[28], d[32]);
    const bool g1 = bl >= 1, g2 = bl >= 2, g3 = bl >= 3;
    float r_b = g3 ? v3 : (g2 ? v2 : (g1 ? v1 : v0));
    float r_n = g3 ? v4 : (g2 ? v3 : (g1 ? v2 : v1));
    hk.template pt<3>(r_b, r_n);
    rqs_eval_core<Q, REV>(r_b, r_n, in, e, out, ljd, hk);
}
// two tiles' MFMAs from one element's twelve points
template <int HT, int UA, int UB>
struct rqs_pair_pipe {
    rqs_tile_pipe<HT, UA> a;
    rqs_tile_pipe<HT, UB> b;
    __device__ __forceinline__ rqs_pair_pipe(const wptr w, const btile<1> (&bh)[HT], tile<1> &ta, tile<1> &tb) : a(w, bh, ta), b(w, bh, tb) {}
    __device__ __forceinline__ void start() { a.start(); b.start(); }
    template <int P, class... T> __device__ __forceinline__ void pt(T &...tie) {
        a.template pt<P>(tie...);
        b.template pt<P>();
    }
};
template <int PH, bool REV, int Q, class H>
__device__ __forceinline__ void rqs32_unit(const f32x16 &u0, const f32x16 &u1, rqs_elems &e, const rqs32_c &c, float (&out)[4], float (&lj)[4], H &hk) {
    if constexpr (PH == 0) rqs32_search<Q>(u0, u1, e, c, hk);
    else if constexpr (PH == 1) rqs32_select<Q>(u0, u1, e, c, hk);
    else rqs32_eval<Q, REV>(u0, u1, e, out[Q], lj[Q], hk);
}
// a step's two elements (tiles 0, 1 and 2, 3): element 0 beside the MFMAs of tiles 2, 3 (tiles 0, 1 are in A0 / A1 on entry), element 1
// is left to the caller (beside the next step's first two tiles, or alone at the end of the group)
template <int HT, int PH, bool REV>
__device__ __forceinline__ void rqs32_first(const wptr w, const btile<1> (&bh)[HT], rqs_elems &e, const rqs32_c &c, float (&out)[4], float (&lj)[4],
                                            tile<1> &A0, tile<1> &A1, tile<1> &B0, tile<1> &B1) {
    rqs_pair_pipe<HT, 2, 3> p(w, bh, B0, B1);
    p.start();
    rqs32_unit<PH, REV, 0>(A0.v[0], A1.v[0], e, c, out, lj, p);
}
template <int HT, bool GEN_UNUSED, class ADV>
__device__ __forceinline__ void rqs32_group(const wptr w0, const btile<1> (&bh)[HT], rqs_elems &e, int K, float lo, float hi, int h,
                                            ADV &&advance, float (&out)[4], float (&lj)[4], uint32_t &live_mask, float &ldj_scale) {
    wptr w;
    dstep st;
    bool lean;
    tile<1> A0, A1, B0, B1;
    A0 = load_cfrag<1>(w0.cb, 4 * HT * 1024);
    A1 = load_cfrag<1>(w0.cb, 4 * HT * 1024 + 32);
#pragma unroll
    for (int m = 0; m < HT; ++m) { gemm_tile<1>(w0.wb, m * 1024, bh[m], A0); gemm_tile<1>(w0.wb, (HT + m) * 1024, bh[m], A1); }
    const rqs32_c c0 = rqs32_consts(lo, hi, K);
    rqs32_first<HT, 0, false>(w0, bh, e, c0, out, lj, A0, A1, B0, B1);
    advance(st, w);
    rqs_block_scalars<HT>(w, h, lo, hi, lean);
    const rqs32_c c1 = rqs32_consts(lo, hi, K);
    { rqs_pair_pipe<HT, 0, 1> p(w, bh, A0, A1); p.start(); rqs32_unit<0, false, 1>(B0.v[0], B1.v[0], e, c0, out, lj, p); }
    rqs32_first<HT, 1, false>(w, bh, e, c1, out, lj, A0, A1, B0, B1);
    advance(st, w);
    { rqs_pair_pipe<HT, 0, 1> p(w, bh, A0, A1); p.start(); rqs32_unit<1, false, 1>(B0.v[0], B1.v[0], e, c1, out, lj, p); }
    if (st.reverse) {
        rqs32_first<HT, 2, true>(w, bh, e, c1, out, lj, A0, A1, B0, B1);
        rqs_pinhook nh; rqs32_unit<2, true, 1>(B0.v[0], B1.v[0], e, c1, out, lj, nh);
    } else {
        rqs32_first<HT, 2, false>(w, bh, e, c1, out, lj, A0, A1, B0, B1);
        rqs_pinhook nh; rqs32_unit<2, false, 1>(B0.v[0], B1.v[0], e, c1, out, lj, nh);
    }
    live_mask = st.mask; ldj_scale = st.ldj_scale;
}
template <int HT, int PH, bool REV>
__device__ __forceinline__ void rqs16_block(const wptr w, const btile<1> (&bh)[HT], rqs_elems &e, const rqs16_c &c, float (&out)[4],
                                            float (&lj)[4], prof_t &pf) {
    tile<1> A, B;
    {   // tile 0: nothing to run beside it yet
        A = load_cfrag<1>(w.cb, 4 * HT * 1024);
#pragma unroll
        for (int m = 0; m < HT; ++m) gemm_tile<1>(w.wb, m * 1024, bh[m], A);
    }
    SX_STAMP(pf, 3);     // a block's first tile (not overlapped)
    { rqs_tile_pipe<HT, 1> p(w, bh, B); p.start(); rqs16_unit<PH, REV, 0>(A.v[0], e, c, out, lj, p); }
    { rqs_tile_pipe<HT, 2> p(w, bh, A); p.start(); rqs16_unit<PH, REV, 1>(B.v[0], e, c, out, lj, p); }
    { rqs_tile_pipe<HT, 3> p(w, bh, B); p.start(); rqs16_unit<PH, REV, 2>(A.v[0], e, c, out, lj, p); }
    SX_STAMP(pf, 4);     // three (tile GEMM, element) pairs
    { rqs_pinhook nh; rqs16_unit<PH, REV, 3>(B.v[0], e, c, out, lj, nh); }
    SX_STAMP(pf, 5);     // the last element (no MFMAs beside it)
}
// elements 0 .. 2 of a block beside its tiles 1 .. 3 (tile 0 is in A on entry, tile 3 in B on exit)
template <int HT, int PH, bool REV, bool GEN>
__device__ __forceinline__ void rqs16_mid(const wptr w, const btile<1> (&bh)[HT], rqs_elems &e, const rqs16_c &c, float (&out)[4],
                                          float (&lj)[4], tile<1> &A, tile<1> &B) {
    { rqs_tile_pipe<HT, 1> p(w, bh, B); p.start(); rqs16_unit<PH, REV, 0, GEN>(A.v[0], e, c, out, lj, p); }
    { rqs_tile_pipe<HT, 2> p(w, bh, A); p.start(); rqs16_unit<PH, REV, 1, GEN>(B.v[0], e, c, out, lj, p); }
    { rqs_tile_pipe<HT, 3> p(w, bh, B); p.start(); rqs16_unit<PH, REV, 2, GEN>(A.v[0], e, c, out, lj, p); }
}
// the lean form of a whole group (three blocks, two step advances): see rqs_triple
template <int HT, bool GEN, class ADV>
__device__ __forceinline__ void rqs16_group(const wptr w0, const btile<1> (&bh)[HT], rqs_elems &e, int K, float lo, float hi, int h,
                                            ADV &&advance, float (&out)[4], float (&lj)[4], uint32_t &live_mask, float &ldj_scale) {
    wptr w;
    dstep st;
    bool lean;
    // round 6: the pipeline runs over the whole group -- a block's LAST element is evaluated beside the NEXT block's first tile
    // (behind the step advance: that tile's weights are the next step's), so of a group's twelve (tile, element) pairs only the
    // first tile and the last element stand alone
    tile<1> A, B;
    A = load_cfrag<1>(w0.cb, 4 * HT * 1024);
#pragma unroll
    for (int m = 0; m < HT; ++m) gemm_tile<1>(w0.wb, m * 1024, bh[m], A);
    const rqs16_c c0 = rqs16_consts(lo, hi, GEN ? K : 16);
    rqs16_mid<HT, 0, false, GEN>(w0, bh, e, c0, out, lj, A, B);
    advance(st, w);
    rqs_block_scalars<HT>(w, h, lo, hi, lean);
    const rqs16_c c1 = rqs16_consts(lo, hi, GEN ? K : 16);
    { rqs_tile_pipe<HT, 0> p(w, bh, A); p.start(); rqs16_unit<0, false, 3, GEN>(B.v[0], e, c0, out, lj, p); }
    rqs16_mid<HT, 1, false, GEN>(w, bh, e, c1, out, lj, A, B);
    advance(st, w);
    { rqs_tile_pipe<HT, 0> p(w, bh, A); p.start(); rqs16_unit<1, false, 3, GEN>(B.v[0], e, c1, out, lj, p); }
    if (st.reverse) {
        rqs16_mid<HT, 2, true, GEN>(w, bh, e, c1, out, lj, A, B);
        rqs_pinhook nh; rqs16_unit<2, true, 3, GEN>(B.v[0], e, c1, out, lj, nh);
    } else {
        rqs16_mid<HT, 2, false, GEN>(w, bh, e, c1, out, lj, A, B);
        rqs_pinhook nh; rqs16_unit<2, false, 3, GEN>(B.v[0], e, c1, out, lj, nh);
    }
    live_mask = st.mask; ldj_scale = st.ldj_scale;
}
template <int TX, int HT, class ADV>
__device__ __forceinline__ void rqs_triple(tile<1> (&xs)[TX], const btile<1> (&bh)[HT], const wptr w0, const dstep &st0, float &ldj,
                                           int lane, ADV &&advance, prof_t &pf) {
    SX_DEP_MARK_SPLINE;
    const int h = lane >> 5;
    const int K = st0.tt, tg = 4 * st0.t0 + st0.c0;
    rqs_elems e;
    float lo, hi, out[4], lj[4];
    bool lean;
    uint32_t live_mask;     // of the group's last step (the evaluation): scalars, not the descriptor -- a struct merged from the two
    float ldj_scale;        // paths below is a stack object
    RQS_GROUP_CASES(RQS_FETCH)
    rqs_block_scalars<HT>(w0, h, lo, hi, lean);     // (the bound in the first block's blob covers both softmax blocks of the group)
    if (K <= 16 && lean) {
#ifndef SX_RQS_NO_CHAIN
        if (K == 16) rqs16_group<HT, false>(w0, bh, e, K, lo, hi, h, advance, out, lj, live_mask, ldj_scale);
        else rqs16_group<HT, true>(w0, bh, e, K, lo, hi, h, advance, out, lj, live_mask, ldj_scale);
#else
        wptr w;
        dstep st;
        rqs16_block<HT, 0, false>(w0, bh, e, rqs16_consts(lo, hi), out, lj, pf);
        advance(st, w);
        rqs_block_scalars<HT>(w, h, lo, hi, lean);
        const rqs16_c c1 = rqs16_consts(lo, hi);
        rqs16_block<HT, 1, false>(w, bh, e, c1, out, lj, pf);
        advance(st, w);
        if (st.reverse) rqs16_block<HT, 2, true>(w, bh, e, c1, out, lj, pf);
        else rqs16_block<HT, 2, false>(w, bh, e, c1, out, lj, pf);
        live_mask = st.mask; ldj_scale = st.ldj_scale;
#endif
    } else if (K > 16) {
        // 17 .. 32 bins (round 4: the one-launch tier used to stop at 16 and such layers ran conditioner program + element-wise kernel
        // through HBM, 8x slower): an element's parameters are TWO output tiles, so a step carries two of the lane's four elements
        // and a group is two triples (step.act bit 1 = which pair); the sweeps with the running maximum, over 32 slots
        const int half = st0.act >> 1;
        rqs_elems e2;
        e2.x[0] = half ? e.x[2] : e.x[0]; e2.x[1] = half ? e.x[3] : e.x[1];
        float o0, o1, l0, l1;
        if (lean) {
            // bounded logits (the packed bound of the group's two softmax blocks): the two-level form, pipelined over the triple
            float o2[4], l2[4];
            rqs32_group<HT, true>(w0, bh, e2, K, lo, hi, h, advance, o2, l2, live_mask, ldj_scale);
            o0 = o2[0]; o1 = o2[1]; l0 = l2[0]; l1 = l2[1];
        } else {
        wptr w;
        dstep st;
        tile<1> acc[4];
        rqs_gemm<HT>(w0, bh, acc);
        rqs_search<0, 0, 2>(acc, e2, K, lo, hi); asm volatile("" : "+v"(e2.b[0]), "+v"(e2.a_b[0]), "+v"(e2.a_w[0])); __builtin_amdgcn_sched_barrier(0);
        rqs_search<1, 0, 2>(acc, e2, K, lo, hi); asm volatile("" : "+v"(e2.b[1]), "+v"(e2.a_b[1]), "+v"(e2.a_w[1])); __builtin_amdgcn_sched_barrier(0);
        advance(st, w);
        rqs_block_scalars<HT>(w, h, lo, hi, lean);
        rqs_gemm<HT>(w, bh, acc);
        rqs_select<0, 0, 2>(acc, e2, K, lo, hi); asm volatile("" : "+v"(e2.c_b[0]), "+v"(e2.c_w[0])); __builtin_amdgcn_sched_barrier(0);
        rqs_select<1, 0, 2>(acc, e2, K, lo, hi); asm volatile("" : "+v"(e2.c_b[1]), "+v"(e2.c_w[1])); __builtin_amdgcn_sched_barrier(0);
        advance(st, w);
        rqs_gemm<HT>(w, bh, acc);
        if (st.reverse) {
            rqs_eval_wide<0, true>(acc[0].v[0], acc[1].v[0], e2, K, o0, l0); __builtin_amdgcn_sched_barrier(0);
            rqs_eval_wide<1, true>(acc[2].v[0], acc[3].v[0], e2, K, o1, l1);
        } else {
            rqs_eval_wide<0, false>(acc[0].v[0], acc[1].v[0], e2, K, o0, l0); __builtin_amdgcn_sched_barrier(0);
            rqs_eval_wide<1, false>(acc[2].v[0], acc[3].v[0], e2, K, o1, l1);
        }
        live_mask = st.mask; ldj_scale = st.ldj_scale;
        }
        out[0] = half ? e.x[0] : o0; out[1] = half ? e.x[1] : o1; out[2] = half ? o0 : e.x[2]; out[3] = half ? o1 : e.x[3];
        lj[0] = half ? 0.f : l0; lj[1] = half ? 0.f : l1; lj[2] = half ? l0 : 0.f; lj[3] = half ? l1 : 0.f;
    } else {
        // up to 16 bins, softmax with its running maximum: block by block, one element at a time (interleaved by the scheduler the
        // four sweeps keep ~60 lane masks alive: hundreds of SGPR spills and nine VGPRs of spill lanes in the whole kernel)
        // (the empty asm pins each element's results where they are computed: without it the optimizer sinks all four sweeps of a
        //  block below the step advance that follows, and the register allocation of the whole kernel pays for that)
#define RQS_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); __builtin_amdgcn_sched_barrier(0)
#define RQS_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b)); __builtin_amdgcn_sched_barrier(0)
        wptr w;
        dstep st;
        tile<1> acc[4];
        rqs_gemm<HT>(w0, bh, acc);
        rqs_search<0, 0>(acc, e, K, lo, hi); RQS_PIN3(e.b[0], e.a_b[0], e.a_w[0]);
        rqs_search<1, 0>(acc, e, K, lo, hi); RQS_PIN3(e.b[1], e.a_b[1], e.a_w[1]);
        rqs_search<2, 0>(acc, e, K, lo, hi); RQS_PIN3(e.b[2], e.a_b[2], e.a_w[2]);
        rqs_search<3, 0>(acc, e, K, lo, hi); RQS_PIN3(e.b[3], e.a_b[3], e.a_w[3]);
        advance(st, w);
        rqs_block_scalars<HT>(w, h, lo, hi, lean);
        rqs_gemm<HT>(w, bh, acc);
        rqs_select<0, 0>(acc, e, K, lo, hi); RQS_PIN2(e.c_b[0], e.c_w[0]);
        rqs_select<1, 0>(acc, e, K, lo, hi); RQS_PIN2(e.c_b[1], e.c_w[1]);
        rqs_select<2, 0>(acc, e, K, lo, hi); RQS_PIN2(e.c_b[2], e.c_w[2]);
        rqs_select<3, 0>(acc, e, K, lo, hi); RQS_PIN2(e.c_b[3], e.c_w[3]);
        advance(st, w);
        rqs_gemm<HT>(w, bh, acc);
        if (st.reverse) {
            rqs_eval<0, true, 0>(acc[0].v[0], e, K, out[0], lj[0]); RQS_PIN2(out[0], lj[0]);
            rqs_eval<1, true, 0>(acc[1].v[0], e, K, out[1], lj[1]); RQS_PIN2(out[1], lj[1]);
            rqs_eval<2, true, 0>(acc[2].v[0], e, K, out[2], lj[2]); RQS_PIN2(out[2], lj[2]);
            rqs_eval<3, true, 0>(acc[3].v[0], e, K, out[3], lj[3]);
        } else {
            rqs_eval<0, false, 0>(acc[0].v[0], e, K, out[0], lj[0]); RQS_PIN2(out[0], lj[0]);
            rqs_eval<1, false, 0>(acc[1].v[0], e, K, out[1], lj[1]); RQS_PIN2(out[1], lj[1]);
            rqs_eval<2, false, 0>(acc[2].v[0], e, K, out[2], lj[2]); RQS_PIN2(out[2], lj[2]);
            rqs_eval<3, false, 0>(acc[3].v[0], e, K, out[3], lj[3]);
        }
#undef RQS_PIN3
#undef RQS_PIN2
        live_mask = st.mask; ldj_scale = st.ldj_scale;
    }
    float sl = 0.f;
    const int g = st0.c0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool live = (live_mask >> (q + 8 * g + 4 * h)) & 1u;     // slot kmap(4g+q, h) of the tile
        out[q] = live ? out[q] : e.x[q];
        sl += live ? lj[q] : 0.f;
    }
    RQS_GROUP_CASES(RQS_STORE)
    ldj += ldj_scale * sl;
}

// ------------------------------------------------------------------------------------------------
// Monotone cubic spline coupling, fused (util/cubic_spline.py:21-251; the reference's default spline_type) -- kernel MODE 12
// (13: with deep conditioners).  Same step layout as the rational-quadratic spline: per group of 8 columns three parameter
// blocks of 4 MFMA output tiles -- the SEARCHED block (widths forward, heights inverse: bin + the sizes of bins b-1, b, b+1 and
// the knot at b), the OTHER block (the same four numbers at the found bin), the two boundary-derivative parameters -- so the
// [N, D(2K+2)] parameter tensor the unfused path writes and reads (4.6 GB per layer at 2^20 x 64, K = 16) never exists.
// The arithmetic is sx_cubic_core.h's (shared with cubic_kernel).  The inverse returns what the reference's
// Transform.inverse_and_log_det_jacobian does (flow.py:42-47): minus the forward log-derivative at the inverted point -- in
// the solved bin (the spline is C1: at a knot both bins give the same derivative to rounding), 0 where that point leaves the
// domain.
// ------------------------------------------------------------------------------------------------
struct cubic_elems {        // the 4 elements of the current group a lane owns
    float x[4];
    float s_k[4], s_m[4], s_b[4], s_p[4];   // searched sequence: knot at the bin, sizes of bins b-1, b, b+1
    float o_k[4], o_m[4], o_b[4], o_p[4];   // other sequence, same
    int b[4];               // bin index (+ RQS_OUT: the input is outside the domain; see rqs_elems)
};
template <int Q, int KC>
__device__ __forceinline__ float cub_softmax(tile<1> (&acc)[4], int K) {
    float mx = RQS_P(acc, Q, 0);
#pragma unroll
    for (int k = 1; k < 16; ++k)
        if (KC ? (k < KC) : true) mx = fmaxf(mx, rqs_has<KC>(k, K) ? RQS_P(acc, Q, k) : mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (KC ? (k < KC) : true) {
            const float e = rqs_has<KC>(k, K) ? __builtin_amdgcn_exp2f(RQS_P(acc, Q, k) - mx) : 0.f;     // logits arrive in base 2
            RQS_P(acc, Q, k) = e;
            sum += e;
        }
    const float Kf = KC ? (float)KC : (float)K;
    return (1.f - CUBIC_MIN_BIN * Kf) * cubic_frcp(sum);         // :103-104, :110-111
}
__device__ __forceinline__ float cub_norm(float xv, bool in, float lo, float hi) { return ((in ? xv : lo) - lo) * cubic_frcp(hi - lo); }   // :98-101
#ifndef SX_CUB_FLAT
// K = 16: two-level forms of the two sweeps below (see rqs_search16): the group of four bins from the groups' sums, then the six
// sizes around it (bins 4g-1 .. 4g+4) picked by the group index, the bin and its neighbours inside them
#define CUB_CL(k) ((k) < 0 ? 0 : ((k) > 15 ? 15 : (k)))        /* (the out-of-range neighbours are never used) */
#define CUB_Z(i) (CUBIC_MIN_BIN + inv * (m3 ? RQS_P(acc, Q, CUB_CL(11 + (i))) : (m2 ? RQS_P(acc, Q, CUB_CL(7 + (i))) : \
                                        (m1 ? RQS_P(acc, Q, CUB_CL(3 + (i))) : RQS_P(acc, Q, CUB_CL((i) - 1))))))
#define CUB_GROUP_SUMS()                                                                                                        \
    const float S0 = 4.f * CUBIC_MIN_BIN + inv * ((RQS_P(acc, Q, 0) + RQS_P(acc, Q, 1)) + (RQS_P(acc, Q, 2) + RQS_P(acc, Q, 3)));    \
    const float S1 = 4.f * CUBIC_MIN_BIN + inv * ((RQS_P(acc, Q, 4) + RQS_P(acc, Q, 5)) + (RQS_P(acc, Q, 6) + RQS_P(acc, Q, 7)));    \
    const float S2 = 4.f * CUBIC_MIN_BIN + inv * ((RQS_P(acc, Q, 8) + RQS_P(acc, Q, 9)) + (RQS_P(acc, Q, 10) + RQS_P(acc, Q, 11)))
template <int Q>
__device__ __forceinline__ void cub_search16(tile<1> (&acc)[4], cubic_elems &e, float lo, float hi) {
    const float xv = e.x[Q];
    const bool in = (xv >= lo) && (xv <= hi);                       // :40 closed interval
    const float xn = cub_norm(xv, in, lo, hi);
    const float inv = cub_softmax<Q, 16>(acc, 16);
    CUB_GROUP_SUMS();
    const bool m1 = xn >= S0, m2 = xn >= S0 + S1, m3 = xn >= (S0 + S1) + S2;
    const float k0 = m3 ? (S0 + S1) + S2 : (m2 ? S0 + S1 : (m1 ? S0 : 0.f));
    const float z0 = CUB_Z(0), z1 = CUB_Z(1), z2 = CUB_Z(2), z3 = CUB_Z(3), z4 = CUB_Z(4), z5 = CUB_Z(5);
    const float k1 = k0 + z1, k2 = k1 + z2, k3 = k2 + z3;
    const bool g1 = xn >= k1, g2 = xn >= k2, g3 = xn >= k3;
    e.b[Q] = (m3 ? 12 : (m2 ? 8 : (m1 ? 4 : 0))) + (g3 ? 3 : (g2 ? 2 : (g1 ? 1 : 0))) + (in ? 0 : RQS_OUT);
    e.s_k[Q] = g3 ? k3 : (g2 ? k2 : (g1 ? k1 : k0));
    e.s_m[Q] = g3 ? z3 : (g2 ? z2 : (g1 ? z1 : z0));
    e.s_b[Q] = g3 ? z4 : (g2 ? z3 : (g1 ? z2 : z1));
    e.s_p[Q] = g3 ? z5 : (g2 ? z4 : (g1 ? z3 : z2));
}
template <int Q>
__device__ __forceinline__ void cub_select16(tile<1> (&acc)[4], cubic_elems &e) {
    const float inv = cub_softmax<Q, 16>(acc, 16);
    const int b = e.b[Q] & (RQS_OUT - 1), bl = b & 3;
    CUB_GROUP_SUMS();
    const bool m1 = b >= 4, m2 = b >= 8, m3 = b >= 12;
    const float k0 = m3 ? (S0 + S1) + S2 : (m2 ? S0 + S1 : (m1 ? S0 : 0.f));
    const float z0 = CUB_Z(0), z1 = CUB_Z(1), z2 = CUB_Z(2), z3 = CUB_Z(3), z4 = CUB_Z(4), z5 = CUB_Z(5);
    const float k1 = k0 + z1, k2 = k1 + z2, k3 = k2 + z3;
    const bool g1 = bl >= 1, g2 = bl >= 2, g3 = bl >= 3;
    e.o_k[Q] = g3 ? k3 : (g2 ? k2 : (g1 ? k1 : k0));
    e.o_m[Q] = g3 ? z3 : (g2 ? z2 : (g1 ? z1 : z0));
    e.o_b[Q] = g3 ? z4 : (g2 ? z3 : (g1 ? z2 : z1));
    e.o_p[Q] = g3 ? z5 : (g2 ? z4 : (g1 ? z3 : z2));
}
#endif
// phase 0: sizes + running knots of the searched block and the bin search (search_sorted.py:4-5) in one sweep
template <int Q, int KC>
__device__ __forceinline__ void cub_search(tile<1> (&acc)[4], cubic_elems &e, int K, float lo, float hi) {
#ifndef SX_CUB_FLAT
    if constexpr (KC == 16) { cub_search16<Q>(acc, e, lo, hi); return; }
#endif
    const float xv = e.x[Q];
    const bool in = (xv >= lo) && (xv <= hi);                       // :40 closed interval
    const float xn = cub_norm(xv, in, lo, hi);
    const float inv = cub_softmax<Q, KC>(acc, K);
    int b = 0;
    float k_b = 0.f, s_b = 0.f, s_m = 1.f, s_p = 1.f, cum = 0.f, last = 1.f;
    bool need = false;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (KC ? (k < KC) : true) {
            const bool used = KC ? true : (k < K);
            const float sz = CUBIC_MIN_BIN + RQS_P(acc, Q, k) * inv;
            const bool ge = used && (xn >= cum);                    // lower knot of bin k (knot 0 = 0): knots only grow
            const bool nx = used && !ge && need;
            b = ge ? k : b; k_b = ge ? cum : k_b; s_b = ge ? sz : s_b; s_m = ge ? last : s_m;
            s_p = nx ? sz : s_p;
            need = ge;
            last = sz;
            cum += sz;
        }
    }
    e.b[Q] = b + (in ? 0 : RQS_OUT); e.s_k[Q] = k_b; e.s_m[Q] = s_m; e.s_b[Q] = s_b; e.s_p[Q] = s_p;
}
// phase 1: the other block at the found bin
template <int Q, int KC>
__device__ __forceinline__ void cub_select(tile<1> (&acc)[4], cubic_elems &e, int K) {
#ifndef SX_CUB_FLAT
    if constexpr (KC == 16) { cub_select16<Q>(acc, e); return; }
#endif
    const float inv = cub_softmax<Q, KC>(acc, K);
    const int b = e.b[Q] & (RQS_OUT - 1);
    float k_b = 0.f, o_b = 0.f, o_m = 1.f, o_p = 1.f, cum = 0.f;
    bool is_prev = false;                 // b == k - 1
    bool is_cur = (b == 0);               // b == k
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (KC ? (k < KC) : true) {
            const bool used = KC ? true : (k < K);
            const float sz = CUBIC_MIN_BIN + RQS_P(acc, Q, k) * inv;
            const bool is_next = (b == k + 1);
            k_b = (used && is_cur) ? cum : k_b;
            o_b = (used && is_cur) ? sz : o_b;
            o_m = (used && is_next) ? sz : o_m;     // bin k is b - 1
            o_p = (used && is_prev) ? sz : o_p;     // bin k is b + 1
            is_prev = is_cur;
            is_cur = is_next;
            cum += sz;
        }
    }
    e.o_k[Q] = k_b; e.o_m[Q] = o_m; e.o_b[Q] = o_b; e.o_p[Q] = o_p;
}
// phase 2: knot derivatives of the bin (:117-132), its cubic (:134-137), the polynomial or its inverse
template <int Q, bool REV, class H>
__device__ __forceinline__ void cub_eval(const f32x16 &u, const cubic_elems &e, int K, float lo, float hi, float &out, float &ljd, H &hk) {
    hk.template pt<0>();
    const int b = e.b[Q] & (RQS_OUT - 1);
    const bool in = e.b[Q] < RQS_OUT;
    const float w_m = REV ? e.o_m[Q] : e.s_m[Q], w_b = REV ? e.o_b[Q] : e.s_b[Q], w_p = REV ? e.o_p[Q] : e.s_p[Q];
    const float h_m = REV ? e.s_m[Q] : e.o_m[Q], h_b = REV ? e.s_b[Q] : e.o_b[Q], h_p = REV ? e.s_p[Q] : e.o_p[Q];
    const float cw_b = REV ? e.o_k[Q] : e.s_k[Q], ch_b = REV ? e.s_k[Q] : e.o_k[Q];
    cubic_coef cf = cubic_bin_coef(b, K, w_b, h_b, w_m, h_m, w_p, h_p, u[0], u[1]);
    hk.template pt<1>(cf.a, cf.bb, cf.c); hk.template pt<2>();
    const float a = cf.a, bb = cf.bb, c = cf.c, d = ch_b;
    const float xn = cub_norm(e.x[Q], in, lo, hi), span = hi - lo;
    if constexpr (REV) {
        float rcw = (b == K - 1) ? 1.f : cw_b + w_b;                                       // :107 (last knot pinned)
        hk.template pt<3>(rcw); hk.template pt<4>(); hk.template pt<5>();
        float so = cubic_invert(a, bb, c, d, xn, cw_b, rcw);
        hk.template pt<6>(so); hk.template pt<7>(); hk.template pt<8>(); hk.template pt<9>();
        out = fminf(fmaxf((so + cw_b) * span + lo, lo), hi);                               // :235 (clamped: see cubic_kernel)
        const bool in2 = (out >= lo) && (out <= hi);
        float t2 = cub_norm(out, in2, lo, hi) - cw_b;
        hk.template pt<10>(t2);
        ljd = in2 ? -cubic_flog(3.f * a * (t2 * t2) + 2.f * bb * t2 + c) : 0.f;            // flow.py:42-47
    } else {
        hk.template pt<3>(); hk.template pt<4>(); hk.template pt<5>();
        float t = xn - cw_b;                                                               // :229
        out = (a * (t * t * t) + bb * (t * t) + c * t + d) * span + lo;                    // :230-233, :238
        hk.template pt<6>(out, t); hk.template pt<7>(); hk.template pt<8>(); hk.template pt<9>(); hk.template pt<10>();
        ljd = cubic_flog(3.f * a * (t * t) + 2.f * bb * t + c);                            // :235-237
    }
    out = in ? out : e.x[Q];                                                               // :46-48 linear tails
    ljd = in ? ljd : 0.f;
    hk.template pt<11>(out, ljd);
}
template <int Q, bool REV>
__device__ __forceinline__ void cub_eval(tile<1> (&acc)[4], const cubic_elems &e, int K, float lo, float hi, float &out, float &ljd) {
    rqs_nohook nh;
    cub_eval<Q, REV>(acc[Q].v[0], e, K, lo, hi, out, ljd, nh);
}

// ---- K = 16, bounded logits: the lean forms of the two sweeps (as rqs16_*: no running maximum below the packed logit bound, value
// selects, one element = one output tile, the next tile's MFMAs issued from the element's twelve points).  Normalised coordinates:
// knot_j = j MIN + inv P_j on [0, 1] (:103-111); a bin's neighbours b - 1 / b + 1 are needed beside it (:117-132), so the SIZES of
// the six bins around the found group are picked (18 selects) instead of RQ's four prefixes.
struct cub16_s {
    float e[16];            // exp2 of the logits (named through the accessors below only with constant indices)
    float T1, T2, T3;       // knots at bins 4, 8, 12
    float inv;              // (1 - 16 MIN) / sum
};
// (any K <= 16, round 6: as rqs16_c -- the packer parks the logits of bins >= K at -1e30; group boundaries 4 j >= K get +inf, the
//  compares with local knots of index >= K in the last reachable group are masked; cubic_bin_coef / cub_eval take K itself)
struct cub16_c {
    float cK;               // 1 - K MIN
    float t4, t8, t12;      // 4 j MIN (GEN: +inf where 4 j >= K)
    int K, glast;
    bool u1, u2, u3;
};
__device__ __forceinline__ cub16_c cub16_consts(int K = 16) {
    cub16_c c;
    const float inf = __builtin_inff();
    c.cK = 1.f - (float)K * CUBIC_MIN_BIN;
    c.t4 = 4 < K ? 4.f * CUBIC_MIN_BIN : inf; c.t8 = 8 < K ? 8.f * CUBIC_MIN_BIN : inf; c.t12 = 12 < K ? 12.f * CUBIC_MIN_BIN : inf;
    c.K = K; c.glast = (K - 1) >> 2;
    const int iK = K - 4 * c.glast;
    c.u1 = 1 >= iK; c.u2 = 2 >= iK; c.u3 = 3 >= iK;
    return c;
}
template <class H>
__device__ __forceinline__ void cub16_sums(const f32x16 &u, const cub16_c &cc, float (&ev)[16], float &T1, float &T2, float &T3, float &inv, H &hk) {
    hk.template pt<0>();
    ev[0] = __builtin_amdgcn_exp2f(u[0]); ev[1] = __builtin_amdgcn_exp2f(u[1]); ev[2] = __builtin_amdgcn_exp2f(u[2]); ev[3] = __builtin_amdgcn_exp2f(u[3]);
    float g0 = (ev[0] + ev[1]) + (ev[2] + ev[3]);
    hk.template pt<1>(g0);
    ev[4] = __builtin_amdgcn_exp2f(u[4]); ev[5] = __builtin_amdgcn_exp2f(u[5]); ev[6] = __builtin_amdgcn_exp2f(u[6]); ev[7] = __builtin_amdgcn_exp2f(u[7]);
    float g1 = (ev[4] + ev[5]) + (ev[6] + ev[7]);
    hk.template pt<2>(g1);
    ev[8] = __builtin_amdgcn_exp2f(u[8]); ev[9] = __builtin_amdgcn_exp2f(u[9]); ev[10] = __builtin_amdgcn_exp2f(u[10]); ev[11] = __builtin_amdgcn_exp2f(u[11]);
    float g2 = (ev[8] + ev[9]) + (ev[10] + ev[11]);
    hk.template pt<3>(g2);
    ev[12] = __builtin_amdgcn_exp2f(u[12]); ev[13] = __builtin_amdgcn_exp2f(u[13]); ev[14] = __builtin_amdgcn_exp2f(u[14]); ev[15] = __builtin_amdgcn_exp2f(u[15]);
    float g3 = (ev[12] + ev[13]) + (ev[14] + ev[15]);
    hk.template pt<4>(g3);
    const float G2 = g0 + g1, G3 = G2 + g2;
    inv = cc.cK * cubic_frcp(G3 + g3);
    T1 = __builtin_fmaf(inv, g0, cc.t4);
    T2 = __builtin_fmaf(inv, G2, cc.t8);
    T3 = __builtin_fmaf(inv, G3, cc.t12);
}
// the group's start knot k0, its three inner knots and the sizes z0..z5 of bins 4g-1 .. 4g+4 (the out-of-range neighbours of the
// first and the last group are never used: any finite value)
template <class H>
__device__ __forceinline__ void cub16_group(const float (&ev)[16], float T1, float T2, float T3, float inv, bool m1, bool m2, bool m3,
                                            float &k0, float &k1, float &k2, float &k3, float (&z)[6], H &hk) {
    k0 = rqs16_pick(m1, m2, m3, 0.f, T1, T2, T3);
    float q0 = rqs16_pick(m1, m2, m3, ev[0], ev[3], ev[7], ev[11]);
    float q1 = rqs16_pick(m1, m2, m3, ev[0], ev[4], ev[8], ev[12]);
    hk.template pt<6>(k0, q0, q1);
    float q2 = rqs16_pick(m1, m2, m3, ev[1], ev[5], ev[9], ev[13]);
    float q3 = rqs16_pick(m1, m2, m3, ev[2], ev[6], ev[10], ev[14]);
    z[0] = __builtin_fmaf(inv, q0, CUBIC_MIN_BIN); z[1] = __builtin_fmaf(inv, q1, CUBIC_MIN_BIN);
    hk.template pt<7>(z[0], z[1], q2, q3);
    const float q4 = rqs16_pick(m1, m2, m3, ev[3], ev[7], ev[11], ev[15]);
    const float q5 = rqs16_pick(m1, m2, m3, ev[4], ev[8], ev[12], ev[15]);
    z[2] = __builtin_fmaf(inv, q2, CUBIC_MIN_BIN); z[3] = __builtin_fmaf(inv, q3, CUBIC_MIN_BIN);
    hk.template pt<8>(z[2], z[3]);
    z[4] = __builtin_fmaf(inv, q4, CUBIC_MIN_BIN); z[5] = __builtin_fmaf(inv, q5, CUBIC_MIN_BIN);
    k1 = k0 + z[1]; k2 = k1 + z[2]; k3 = k2 + z[3];
}
__device__ __forceinline__ float cub16_pick4(bool g1, bool g2, bool g3, float v0, float v1, float v2, float v3) {
    return g3 ? v3 : (g2 ? v2 : (g1 ? v1 : v0));
}
template <int Q, bool GEN, class H>
__device__ __forceinline__ void cub16_search(const f32x16 &u, cubic_elems &e, const cub16_c &cc, float lo, float hi, H &hk) {
    float ev[16], T1, T2, T3, inv;
    cub16_sums(u, cc, ev, T1, T2, T3, inv, hk);
    hk.template pt<5>(T1, T2, T3);
    const float xv = e.x[Q];
    const bool in = (xv >= lo) && (xv <= hi);                       // :40 closed interval
    const float xn = cub_norm(xv, in, lo, hi);
    const bool m1 = xn >= T1, m2 = xn >= T2, m3 = xn >= T3;
    float k0, k1, k2, k3, z[6];
    cub16_group(ev, T1, T2, T3, inv, m1, m2, m3, k0, k1, k2, k3, z, hk);
    bool g1 = xn >= k1, g2 = xn >= k2, g3 = xn >= k3;
    if constexpr (GEN) {
        const bool last = cc.glast == 0 ? true : (cc.glast == 1 ? m1 : (cc.glast == 2 ? m2 : m3));
        g1 = g1 && !(last && cc.u1); g2 = g2 && !(last && cc.u2); g3 = g3 && !(last && cc.u3);
    }
    int bg = 4 * ((int)m1 + (int)m2 + (int)m3) + (in ? 0 : RQS_OUT);
    hk.template pt<9>(bg, z[4], z[5]);
    e.b[Q] = bg + ((int)g1 + (int)g2 + (int)g3);
    e.s_k[Q] = cub16_pick4(g1, g2, g3, k0, k1, k2, k3);
    e.s_m[Q] = cub16_pick4(g1, g2, g3, z[0], z[1], z[2], z[3]);
    hk.template pt<10>(e.b[Q], e.s_k[Q], e.s_m[Q]);
    e.s_b[Q] = cub16_pick4(g1, g2, g3, z[1], z[2], z[3], z[4]);
    e.s_p[Q] = cub16_pick4(g1, g2, g3, z[2], z[3], z[4], z[5]);
    hk.template pt<11>(e.s_b[Q], e.s_p[Q]);
}
template <int Q, class H>
__device__ __forceinline__ void cub16_select(const f32x16 &u, cubic_elems &e, const cub16_c &cc, H &hk) {
    float ev[16], T1, T2, T3, inv;
    cub16_sums(u, cc, ev, T1, T2, T3, inv, hk);
    hk.template pt<5>(T1, T2, T3);
    const int b = e.b[Q] & (RQS_OUT - 1), bl = b & 3;
    const bool m1 = b >= 4, m2 = b >= 8, m3 = b >= 12;
    float k0, k1, k2, k3, z[6];
    cub16_group(ev, T1, T2, T3, inv, m1, m2, m3, k0, k1, k2, k3, z, hk);
    const bool g1 = bl >= 1, g2 = bl >= 2, g3 = bl >= 3;
    hk.template pt<9>(z[4], z[5]);
    e.o_k[Q] = cub16_pick4(g1, g2, g3, k0, k1, k2, k3);
    e.o_m[Q] = cub16_pick4(g1, g2, g3, z[0], z[1], z[2], z[3]);
    hk.template pt<10>(e.o_k[Q], e.o_m[Q]);
    e.o_b[Q] = cub16_pick4(g1, g2, g3, z[1], z[2], z[3], z[4]);
    e.o_p[Q] = cub16_pick4(g1, g2, g3, z[2], z[3], z[4], z[5]);
    hk.template pt<11>(e.o_b[Q], e.o_p[Q]);
}
template <int PH, bool REV, int Q, bool GEN, class H>
__device__ __forceinline__ void cub16_unit(const f32x16 &u, cubic_elems &e, const cub16_c &cc, float lo, float hi, float (&out)[4], float (&lj)[4], H &hk) {
    if constexpr (PH == 0) cub16_search<Q, GEN>(u, e, cc, lo, hi, hk);
    else if constexpr (PH == 1) cub16_select<Q>(u, e, cc, hk);
    else cub_eval<Q, REV>(u, e, GEN ? cc.K : 16, lo, hi, out[Q], lj[Q], hk);
}
template <int HT, int PH, bool REV>
__device__ __forceinline__ void cub16_block(const wptr w, const btile<1> (&bh)[HT], cubic_elems &e, float lo, float hi, float (&out)[4],
                                            float (&lj)[4]) {
    const cub16_c cc = cub16_consts();
    tile<1> A, B;
    {
        A = load_cfrag<1>(w.cb, 4 * HT * 1024);
#pragma unroll
        for (int m = 0; m < HT; ++m) gemm_tile<1>(w.wb, m * 1024, bh[m], A);
    }
    { rqs_tile_pipe<HT, 1> p(w, bh, B); p.start(); cub16_unit<PH, REV, 0, false>(A.v[0], e, cc, lo, hi, out, lj, p); }
    { rqs_tile_pipe<HT, 2> p(w, bh, A); p.start(); cub16_unit<PH, REV, 1, false>(B.v[0], e, cc, lo, hi, out, lj, p); }
    { rqs_tile_pipe<HT, 3> p(w, bh, B); p.start(); cub16_unit<PH, REV, 2, false>(A.v[0], e, cc, lo, hi, out, lj, p); }
    { rqs_pinhook nh; cub16_unit<PH, REV, 3, false>(B.v[0], e, cc, lo, hi, out, lj, nh); }
}
template <int HT, int PH, bool REV, bool GEN>
__device__ __forceinline__ void cub16_mid(const wptr w, const btile<1> (&bh)[HT], cubic_elems &e, const cub16_c &cc, float lo, float hi,
                                          float (&out)[4], float (&lj)[4], tile<1> &A, tile<1> &B) {
    { rqs_tile_pipe<HT, 1> p(w, bh, B); p.start(); cub16_unit<PH, REV, 0, GEN>(A.v[0], e, cc, lo, hi, out, lj, p); }
    { rqs_tile_pipe<HT, 2> p(w, bh, A); p.start(); cub16_unit<PH, REV, 1, GEN>(B.v[0], e, cc, lo, hi, out, lj, p); }
    { rqs_tile_pipe<HT, 3> p(w, bh, B); p.start(); cub16_unit<PH, REV, 2, GEN>(A.v[0], e, cc, lo, hi, out, lj, p); }
}
template <int HT, bool GEN, class ADV>
__device__ __forceinline__ void cub16_group(const wptr w0, const btile<1> (&bh)[HT], cubic_elems &e, int K, float lo, float hi, int h,
                                            ADV &&advance, float (&out)[4], float (&lj)[4], uint32_t &live_mask, float &ldj_scale) {
    // (the group-long pipeline of rqs16_group)
    wptr w;
    dstep st;
    bool lean;
    const cub16_c cc = cub16_consts(GEN ? K : 16);
    tile<1> A, B;
    A = load_cfrag<1>(w0.cb, 4 * HT * 1024);
#pragma unroll
    for (int m = 0; m < HT; ++m) gemm_tile<1>(w0.wb, m * 1024, bh[m], A);
    const float lo0 = lo, hi0 = hi;
    cub16_mid<HT, 0, false, GEN>(w0, bh, e, cc, lo0, hi0, out, lj, A, B);
    advance(st, w);
    { rqs_tile_pipe<HT, 0> p(w, bh, A); p.start(); cub16_unit<0, false, 3, GEN>(B.v[0], e, cc, lo0, hi0, out, lj, p); }
    cub16_mid<HT, 1, false, GEN>(w, bh, e, cc, lo0, hi0, out, lj, A, B);
    advance(st, w);
    rqs_block_scalars<HT>(w, h, lo, hi, lean);
    { rqs_tile_pipe<HT, 0> p(w, bh, A); p.start(); cub16_unit<1, false, 3, GEN>(B.v[0], e, cc, lo0, hi0, out, lj, p); }
    if (st.reverse) {
        cub16_mid<HT, 2, true, GEN>(w, bh, e, cc, lo, hi, out, lj, A, B);
        rqs_pinhook nh; cub16_unit<2, true, 3, GEN>(B.v[0], e, cc, lo, hi, out, lj, nh);
    } else {
        cub16_mid<HT, 2, false, GEN>(w, bh, e, cc, lo, hi, out, lj, A, B);
        rqs_pinhook nh; cub16_unit<2, false, 3, GEN>(B.v[0], e, cc, lo, hi, out, lj, nh);
    }
    live_mask = st.mask; ldj_scale = st.ldj_scale;
}
// The three blocks of a group in one iteration of the step loop (see rqs_triple)
#define CUB_FETCH(T, G) { asm volatile("" ::: "memory"); _Pragma("unroll") for (int q = 0; q < 4; ++q) e.x[q] = xs[T].v[0][4 * G + q]; }
template <int TX, int HT, class ADV>
__device__ __forceinline__ void cubic_triple(tile<1> (&xs)[TX], const btile<1> (&bh)[HT], const wptr w0, const dstep &st0, float &ldj,
                                             int lane, ADV &&advance) {
    SX_DEP_MARK_SPLINE;
    const int h = lane >> 5;
    const int K = st0.tt, tg = 4 * st0.t0 + st0.c0;
    cubic_elems e;
    float lo, hi, out[4], lj[4];
    bool lean;
    uint32_t live_mask;
    float ldj_scale;
    RQS_GROUP_CASES(CUB_FETCH)
    rqs_block_scalars<HT>(w0, h, lo, hi, lean);
    if (K <= 16 && lean) {
#ifndef SX_RQS_NO_CHAIN
        if (K == 16) cub16_group<HT, false>(w0, bh, e, K, lo, hi, h, advance, out, lj, live_mask, ldj_scale);
        else cub16_group<HT, true>(w0, bh, e, K, lo, hi, h, advance, out, lj, live_mask, ldj_scale);
#else
        wptr w;
        dstep st;
        cub16_block<HT, 0, false>(w0, bh, e, lo, hi, out, lj);
        advance(st, w);
        cub16_block<HT, 1, false>(w, bh, e, lo, hi, out, lj);
        advance(st, w);
        rqs_block_scalars<HT>(w, h, lo, hi, lean);
        if (st.reverse) cub16_block<HT, 2, true>(w, bh, e, lo, hi, out, lj);
        else cub16_block<HT, 2, false>(w, bh, e, lo, hi, out, lj);
        live_mask = st.mask; ldj_scale = st.ldj_scale;
#endif
    } else {
#define CUB_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); __builtin_amdgcn_sched_barrier(0)
        wptr w;
        dstep st;
        tile<1> acc[4];
        rqs_gemm<HT>(w0, bh, acc);
        cub_search<0, 0>(acc, e, K, lo, hi); CUB_PIN4(e.b[0], e.s_k[0], e.s_m[0], e.s_b[0]);
        cub_search<1, 0>(acc, e, K, lo, hi); CUB_PIN4(e.b[1], e.s_k[1], e.s_m[1], e.s_b[1]);
        cub_search<2, 0>(acc, e, K, lo, hi); CUB_PIN4(e.b[2], e.s_k[2], e.s_m[2], e.s_b[2]);
        cub_search<3, 0>(acc, e, K, lo, hi); CUB_PIN4(e.b[3], e.s_k[3], e.s_m[3], e.s_b[3]);
        advance(st, w);
        rqs_gemm<HT>(w, bh, acc);
        cub_select<0, 0>(acc, e, K); CUB_PIN4(e.o_k[0], e.o_m[0], e.o_b[0], e.o_p[0]);
        cub_select<1, 0>(acc, e, K); CUB_PIN4(e.o_k[1], e.o_m[1], e.o_b[1], e.o_p[1]);
        cub_select<2, 0>(acc, e, K); CUB_PIN4(e.o_k[2], e.o_m[2], e.o_b[2], e.o_p[2]);
        cub_select<3, 0>(acc, e, K); CUB_PIN4(e.o_k[3], e.o_m[3], e.o_b[3], e.o_p[3]);
        advance(st, w);
        rqs_block_scalars<HT>(w, h, lo, hi, lean);
        rqs_gemm<HT>(w, bh, acc);
        if (st.reverse) {
            cub_eval<0, true>(acc, e, K, lo, hi, out[0], lj[0]); cub_eval<1, true>(acc, e, K, lo, hi, out[1], lj[1]);
            cub_eval<2, true>(acc, e, K, lo, hi, out[2], lj[2]); cub_eval<3, true>(acc, e, K, lo, hi, out[3], lj[3]);
        } else {
            cub_eval<0, false>(acc, e, K, lo, hi, out[0], lj[0]); cub_eval<1, false>(acc, e, K, lo, hi, out[1], lj[1]);
            cub_eval<2, false>(acc, e, K, lo, hi, out[2], lj[2]); cub_eval<3, false>(acc, e, K, lo, hi, out[3], lj[3]);
        }
#undef CUB_PIN4
        live_mask = st.mask; ldj_scale = st.ldj_scale;
    }
    float sl = 0.f;
    const int g = st0.c0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool live = (live_mask >> (q + 8 * g + 4 * h)) & 1u;
        out[q] = live ? out[q] : e.x[q];
        sl += live ? lj[q] : 0.f;
    }
    RQS_GROUP_CASES(RQS_STORE)
    ldj += ldj_scale * sl;
}

template <int TX, int HT, int KC>
__device__ __forceinline__ void cubic_phase_k(tile<1> (&acc)[4], tile<1> (&xs)[TX], cubic_elems &e, const dstep &st,
                                              float lo, float hi, float &ldj, int h) {
    const int g = st.c0, K = st.tt;
    if (st.ct == 0) {
#pragma unroll
        for (int t = 0; t < TX; ++t)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                if (t == st.t0 && gg == g) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) e.x[q] = xs[t].v[0][4 * gg + q];
                }
        // (one element at a time: interleaving the four two-level searches keeps 4 x 12 picked values alive at once -- scratch)
        cub_search<0, KC>(acc, e, K, lo, hi); __builtin_amdgcn_sched_barrier(0);
        cub_search<1, KC>(acc, e, K, lo, hi); __builtin_amdgcn_sched_barrier(0);
        cub_search<2, KC>(acc, e, K, lo, hi); __builtin_amdgcn_sched_barrier(0);
        cub_search<3, KC>(acc, e, K, lo, hi);
    } else if (st.ct == 1) {
        cub_select<0, KC>(acc, e, K); __builtin_amdgcn_sched_barrier(0);
        cub_select<1, KC>(acc, e, K); __builtin_amdgcn_sched_barrier(0);
        cub_select<2, KC>(acc, e, K); __builtin_amdgcn_sched_barrier(0);
        cub_select<3, KC>(acc, e, K);
    } else {
        float out[4], lj[4];
        if (st.reverse) {
            cub_eval<0, true>(acc, e, K, lo, hi, out[0], lj[0]);
            cub_eval<1, true>(acc, e, K, lo, hi, out[1], lj[1]);
            cub_eval<2, true>(acc, e, K, lo, hi, out[2], lj[2]);
            cub_eval<3, true>(acc, e, K, lo, hi, out[3], lj[3]);
        } else {
            cub_eval<0, false>(acc, e, K, lo, hi, out[0], lj[0]);
            cub_eval<1, false>(acc, e, K, lo, hi, out[1], lj[1]);
            cub_eval<2, false>(acc, e, K, lo, hi, out[2], lj[2]);
            cub_eval<3, false>(acc, e, K, lo, hi, out[3], lj[3]);
        }
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool live = (st.mask >> (q + 8 * g + 4 * h)) & 1u;     // slot kmap(4g+q, h) of the tile
            out[q] = live ? out[q] : e.x[q];
            s += live ? lj[q] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < TX; ++t)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                if (t == st.t0 && gg == g) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) xs[t].v[0][4 * gg + q] = out[q];
                }
        ldj += st.ldj_scale * s;
    }
}
template <int TX, int HT>
__device__ __forceinline__ void cubic_phase(tile<1> (&xs)[TX], const btile<1> (&bh)[HT], cubic_elems &e, const wptr w,
                                            const dstep &st, float &ldj, int lane) {
    SX_DEP_MARK_SPLINE;
    const int h = lane >> 5;
    tile<1> acc[4];
    rqs_gemm<HT>(w, bh, acc);
    const float lo = *reinterpret_cast<const float *>(w.cb - h * 64 + (4 * HT * 1024 + 128) * 4);
    const float hi = *reinterpret_cast<const float *>(w.cb - h * 64 + (4 * HT * 1024 + 129) * 4);
    if (st.tt == 16) cubic_phase_k<TX, HT, 16>(acc, xs, e, st, lo, hi, ldj, h);
    else cubic_phase_k<TX, HT, 0>(acc, xs, e, st, lo, hi, ldj, h);
}

