// Training backward steps of the fused flow kernel (kernel MODEs 4 and 11: coupling / dense-layer backward, in-kernel weight
// gradients).  Part of sx_flow_kernel.h: included from there, inside namespace SX_PREC_NS -- not a header of its own.
// ------------------------------------------------------------------------------------------------
// Training: backward of one affine coupling of a log_prob pass.
// The forward (log_prob) direction computed x_out = (x_in - sh) * exp(-ls) on the transformed tile with
// (ls, sh) = net(x_cond) and added -sum(ls) to the log-prob.  Flows are invertible, so nothing was saved: given
// x_out and a = dL/dx_out this step recomputes the conditioner, un-transforms the state (x_in = x_out*exp(ls) + sh)
// and propagates the adjoint; the per-row factors of the weight gradients (z, tanh h, dL/dh_pre, dL/dparams) go to
// HBM, where the caller contracts them over the batch with plain GEMMs.
// State tiles: [0,2) = x, [2,4) = dL/dx.  C = conditioning x tile, T = 1 - C the transformed one.
// ------------------------------------------------------------------------------------------------
// side layout: 32-row groups, feature-major inside ([feature][32 rows]): every store of a C-fragment register is two
// 128 B runs (one per lane half) -- row-major rows would take 64 separate 16 B pieces per store, and the L2
// write-request rate, not the bytes, then bounds the kernel (measured 2.10 -> 1.83 ms on cfg 2).  sx_wgrad reads
// whole 4 KB feature tiles and turns them through LDS.
__device__ __forceinline__ void store_ctile(float *row_base, int off, const f32x16 &v, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) row_base[(off + (r & 3) + 8 * (r >> 2) + 4 * h) * 32] = v[r];
}
// XT data tiles (1 or 2); state tiles [0,XT) = x, [XT,2XT) = dL/dx.  Conditioner tiles [C0,C0+CT), transformed tiles
// [T0,T0+TT) of the data tiles: pruned halves (XT = 2) or dense (any mask; zero weights outside the mask).
template <int XT, int HT, int C0, int CT, int T0, int TT>
__device__ __forceinline__ void coupling_affine_bwd(tile<1> (&xs)[2 * XT], const wptr w, float g, float *side_row,
                                                    int lane, rng_t &rg) {
    SX_DEP_MARK_BWD;
    constexpr int F1 = 0;                                            // forward pack(W1', HT x CT)
    constexpr int F2 = HT * CT * 1024 + HT * 32;                     // forward pack(W2', 2TT x HT)
    constexpr int F2B = F2 + 2 * TT * HT * 1024;                     //   its bias
    constexpr int B2 = F2B + 2 * TT * 32;                            // pack(W2^T, HT x 2TT)
    constexpr int B1 = B2 + HT * 2 * TT * 1024 + HT * 32;            // pack(W1^T, CT x HT)
    const int h = lane >> 5;
    // The per-row factors of the weight gradients -- [z (CT) | tanh h (HT) | dL/dh_pre (HT) | dL/dls_t, dL/dsh_t (2TT)]
    // -- are stored as soon as each is final, so that the stores drain behind the GEMM phases that follow instead
    // of queueing up at the end of the step (one wave per SIMD: nothing else hides them).
    if (side_row != nullptr) {
#pragma unroll
        for (int c = 0; c < CT; ++c) store_ctile(side_row, 32 * c, xs[C0 + c].v[0], h);
    }
    // 1. recompute the conditioner (folded tanh: r = (1 - tanh)/2)
    tile<1> hid[HT];
    hidden_layer<1, 2 * XT, HT, C0, CT, true>(xs, hid, w, F1, SX_ACT_TANH_FOLDED, rg);
#pragma unroll
    for (int r = 0; r < 16; ++r) hid[HT - 1].v[0][r] = fast_sig2(hid[HT - 1].v[0][r]);
    btile<1> bh[HT];
#pragma unroll
    for (int m = 0; m < HT; ++m) bh[m] = make_btile<1>(hid[m]);
    // 2. per transformed tile: (kk*log_scale, shift), un-transform, adjoints
    tile<1> dls[TT], dsh[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        tile<1> ls = load_cfrag<1>(w.cb, F2B + (2 * t) * 32), sh = load_cfrag<1>(w.cb, F2B + (2 * t + 1) * 32);
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            gemm_tile<1>(w.wb, F2 + ((2 * t) * HT + m) * 1024, bh[m], ls);          // kk*log_scale, kk = -log2 e
            gemm_tile<1>(w.wb, F2 + ((2 * t + 1) * HT + m) * 1024, bh[m], sh);      // shift
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(ls.v[0][r]);                     // exp(-log_scale)
            const float xo = xs[T0 + t].v[0][r], al = xs[XT + T0 + t].v[0][r];
            xs[T0 + t].v[0][r] = xo * __builtin_amdgcn_exp2f(-ls.v[0][r]) + sh.v[0][r];      // x_in
            const float ai = al * e;                                                // dL/dx_in
            xs[XT + T0 + t].v[0][r] = ai;
            dsh[t].v[0][r] = -ai;                                                   // dL/dshift
            dls[t].v[0][r] = -al * xo - g;                                          // dL/dlog_scale (incl. -sum(ls))
        }
        if (side_row != nullptr) {
            store_ctile(side_row, 32 * CT + 64 * HT + 64 * t, dls[t].v[0], h);
            store_ctile(side_row, 32 * CT + 64 * HT + 64 * t + 32, dsh[t].v[0], h);
        }
    }
    // 3. dh = W2^T [dls_0; dsh_0; dls_1; ...],  dh_pre = dh * (1 - tanh^2)
    tile<1> dh[HT];
#pragma unroll
    for (int m = 0; m < HT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[m].v[0][r] = 0.f;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const btile<1> b0 = make_btile<1>(dls[t], rg), b1 = make_btile<1>(dsh[t], rg);
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            gemm_tile<1>(w.wb, B2 + (m * 2 * TT + 2 * t) * 1024, b0, dh[m]);
            gemm_tile<1>(w.wb, B2 + (m * 2 * TT + 2 * t + 1) * 1024, b1, dh[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < HT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float th = 1.f - 2.f * hid[m].v[0][r];               // tanh
            hid[m].v[0][r] = th;
            dh[m].v[0][r] *= (1.f - th * th);
        }
    if (side_row != nullptr) {
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            store_ctile(side_row, 32 * CT + 32 * m, hid[m].v[0], h);
            store_ctile(side_row, 32 * CT + 32 * HT + 32 * m, dh[m].v[0], h);
        }
    }
    // 4. adjoint of the conditioning tiles: += W1^T dh_pre
    {
        btile<1> bd[HT];
#pragma unroll
        for (int m = 0; m < HT; ++m) bd[m] = make_btile<1>(dh[m], rg);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            tile<1> dz;
#pragma unroll
            for (int r = 0; r < 16; ++r) dz.v[0][r] = 0.f;
#pragma unroll
            for (int m = 0; m < HT; ++m) gemm_tile<1>(w.wb, B1 + (c * HT + m) * 1024, bd[m], dz);
#pragma unroll
            for (int r = 0; r < 16; ++r) xs[XT + C0 + c].v[0][r] += dz.v[0][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same backward step for 128-column flows (XT = 4 data tiles + 4 adjoint tiles, cfg 4), in TWO steps: all four packed
// operands of a D = 128 coupling (W1', W2', W2^T, W1^T: 100 KB) do not fit the double-buffered LDS ring, so step A carries the
// forward operands (recompute the conditioner, un-transform, parameter adjoints) and step B the transposed ones (dh, dh_pre,
// adjoint of the conditioning tiles); r = (1 - tanh h) / 2 and dL/d(log_scale, shift) wait for step B in registers (`keep`).
// Side features of a row: [z (32 CT) | tanh h (32 HT) | dL/dh_pre (32 HT) | per transformed tile dL/dls (32), dL/dsh (32)].
// ------------------------------------------------------------------------------------------------
template <int XT, int HT, int C0, int CT, int T0, int TT>
__device__ __forceinline__ void coupling_affine_bwd_a(tile<1> (&xs)[2 * XT], const wptr w, float g, float *side_row, int lane,
                                                      rng_t &rg, tile<1> (&keep)[HT + 2 * TT]) {
    SX_DEP_MARK_BWD;
    constexpr int F1 = 0;                                            // pack(W1', HT x CT)
    constexpr int F2 = HT * CT * 1024 + HT * 32;                     // pack(W2', 2TT x HT)
    constexpr int F2B = F2 + 2 * TT * HT * 1024;                     //   its bias
    const int h = lane >> 5;
    if (side_row != nullptr) {
#pragma unroll
        for (int c = 0; c < CT; ++c) store_ctile(side_row, 32 * c, xs[C0 + c].v[0], h);
    }
    tile<1> hid[HT];
#if defined(SX_F16X3) && !defined(SX_BWD_MMAJOR)
    // (one wave per SIMD: the A fragments of gemm tile q + 1 are requested before the MFMAs of tile q -- gemm_tile_pf -- through the
    //  hidden layer and the output layer; the sigmoid of hidden tile m - 1 rides between the MFMAs of tile m's first k-chunk)
    constexpr int NHT = HT * CT, NOT_ = 2 * TT * HT;
    auto off_of = [](int q) {            // q-th gemm tile: hidden m-major, then per transformed tile t and hidden tile m: log_scale, shift
        if (q < NHT) return F1 + q * 1024;
        const int o = q - NHT;
        if (o >= NOT_) return -1;
        const int t = o / (2 * HT), m = (o / 2) % HT, which = o & 1;
        return F2 + ((2 * t + which) * HT + m) * 1024;
    };
    auto none = [](int) {};
    afr cura = afr_load(w.wb, off_of(0));
    int q = 0;
    {
        btile<1> bsrc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) bsrc[c] = make_btile<1>(xs[C0 + c], rg);
        __builtin_amdgcn_sched_barrier(0);
        const int bias = F1 + HT * CT * 1024;
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            tile<1> acc = load_cfrag<1>(w.cb, bias + m * 32);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (m > 0 && c == 0)
                    gemm_tile_pf<1, false>(w.wb, cura, off_of(q + 1), bsrc[c], acc, [&](int i) {
                        if (!(i & 1)) { hid[m > 0 ? m - 1 : 0].v[0][i] = pinned(fast_sig2(hid[m > 0 ? m - 1 : 0].v[0][i]));
                                        hid[m > 0 ? m - 1 : 0].v[0][i + 1] = pinned(fast_sig2(hid[m > 0 ? m - 1 : 0].v[0][i + 1])); }
                    });
                else
                    gemm_tile_pf<1, false>(w.wb, cura, off_of(q + 1), bsrc[c], acc, none);
                ++q;
            }
            hid[m] = acc;
        }
    }
#else
    hidden_layer<1, 2 * XT, HT, C0, CT, true>(xs, hid, w, F1, SX_ACT_TANH_FOLDED, rg);
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) hid[HT - 1].v[0][r] = fast_sig2(hid[HT - 1].v[0][r]);
    btile<1> bh[HT];
#pragma unroll
    for (int m = 0; m < HT; ++m) { bh[m] = make_btile<1>(hid[m]); keep[m] = hid[m]; }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        tile<1> ls = load_cfrag<1>(w.cb, F2B + (2 * t) * 32), sh = load_cfrag<1>(w.cb, F2B + (2 * t + 1) * 32);
#pragma unroll
        for (int m = 0; m < HT; ++m) {
#if defined(SX_F16X3) && !defined(SX_BWD_MMAJOR)
            gemm_tile_pf<1, false>(w.wb, cura, off_of(q + 1), bh[m], ls, none); ++q;       // kk*log_scale, kk = -log2 e
            gemm_tile_pf<1, false>(w.wb, cura, off_of(q + 1), bh[m], sh, none); ++q;       // shift
#else
            gemm_tile<1>(w.wb, F2 + ((2 * t) * HT + m) * 1024, bh[m], ls);          // kk*log_scale, kk = -log2 e
            gemm_tile<1>(w.wb, F2 + ((2 * t + 1) * HT + m) * 1024, bh[m], sh);      // shift
#endif
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(ls.v[0][r]);                     // exp(-log_scale)
            const float xo = xs[T0 + t].v[0][r], al = xs[XT + T0 + t].v[0][r];
            xs[T0 + t].v[0][r] = xo * __builtin_amdgcn_exp2f(-ls.v[0][r]) + sh.v[0][r];      // x_in
            const float ai = al * e;                                                // dL/dx_in
            xs[XT + T0 + t].v[0][r] = ai;
            keep[HT + 2 * t + 1].v[0][r] = -ai;                                     // dL/dshift
            keep[HT + 2 * t].v[0][r] = -al * xo - g;                                // dL/dlog_scale (incl. -sum(ls))
        }
        if (side_row != nullptr) {
            store_ctile(side_row, 32 * CT + 64 * HT + 64 * t, keep[HT + 2 * t].v[0], h);
            store_ctile(side_row, 32 * CT + 64 * HT + 64 * t + 32, keep[HT + 2 * t + 1].v[0], h);
        }
    }
}
template <int XT, int HT, int C0, int CT, int T0, int TT>
__device__ __forceinline__ void coupling_affine_bwd_b(tile<1> (&xs)[2 * XT], const wptr w, float *side_row, int lane, rng_t &rg,
                                                      tile<1> (&keep)[HT + 2 * TT]) {
    SX_DEP_MARK_BWD;
    constexpr int B2 = 0;                                            // pack(W2^T, HT x 2TT)
    constexpr int B1 = HT * 2 * TT * 1024 + HT * 32;                 // pack(W1^T, CT x HT)
    const int h = lane >> 5;
    tile<1> dh[HT];
#pragma unroll
    for (int m = 0; m < HT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[m].v[0][r] = 0.f;
#if defined(SX_F16X3) && !defined(SX_BWD_MMAJOR)
    constexpr int NDH = 2 * TT * HT, NDZ = CT * HT;
    auto off_of = [](int q) {            // q-th gemm tile: dh per (t, m): the log_scale and the shift adjoint's; then dz per (c, m)
        if (q < NDH) { const int t = q / (2 * HT), m = (q / 2) % HT, which = q & 1; return B2 + (m * 2 * TT + 2 * t + which) * 1024; }
        const int o = q - NDH;
        return o < NDZ ? B1 + o * 1024 : -1;
    };
    auto none = [](int) {};
    afr cura = afr_load(w.wb, off_of(0));
    int q = 0;
#endif
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const btile<1> b0 = make_btile<1>(keep[HT + 2 * t], rg), b1 = make_btile<1>(keep[HT + 2 * t + 1], rg);
#pragma unroll
        for (int m = 0; m < HT; ++m) {
#if defined(SX_F16X3) && !defined(SX_BWD_MMAJOR)
            gemm_tile_pf<1, false>(w.wb, cura, off_of(q + 1), b0, dh[m], none); ++q;
            gemm_tile_pf<1, false>(w.wb, cura, off_of(q + 1), b1, dh[m], none); ++q;
#else
            gemm_tile<1>(w.wb, B2 + (m * 2 * TT + 2 * t) * 1024, b0, dh[m]);
            gemm_tile<1>(w.wb, B2 + (m * 2 * TT + 2 * t + 1) * 1024, b1, dh[m]);
#endif
        }
    }
#pragma unroll
    for (int m = 0; m < HT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float th = 1.f - 2.f * keep[m].v[0][r];              // tanh
            keep[m].v[0][r] = th;
            dh[m].v[0][r] *= (1.f - th * th);
        }
    if (side_row != nullptr) {
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            store_ctile(side_row, 32 * CT + 32 * m, keep[m].v[0], h);
            store_ctile(side_row, 32 * CT + 32 * HT + 32 * m, dh[m].v[0], h);
        }
    }
    btile<1> bd[HT];
#pragma unroll
    for (int m = 0; m < HT; ++m) bd[m] = make_btile<1>(dh[m], rg);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        tile<1> dz;
#pragma unroll
        for (int r = 0; r < 16; ++r) dz.v[0][r] = 0.f;
#pragma unroll
        for (int m = 0; m < HT; ++m) {
#if defined(SX_F16X3) && !defined(SX_BWD_MMAJOR)
            gemm_tile_pf<1, false>(w.wb, cura, off_of(q + 1), bd[m], dz, none); ++q;
#else
            gemm_tile<1>(w.wb, B1 + (c * HT + m) * 1024, bd[m], dz);
#endif
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) xs[XT + C0 + c].v[0][r] += dz.v[0][r];
    }
}
// Dense linear layer of the backward pass (AffineLU / MatrixExponential, affine.py:156-171,243-288), one step per half of the
// state: tiles [T0, T0 + XT) <- M . tiles + b (blob = pack_linear(M, XT x XT) + bias).  On the x tiles M is the layer's FORWARD
// matrix (the log_prob pass applied its inverse: u = W v + c; the step recovers v), on the adjoint tiles it is W^T (dL/dv = W^T
// dL/du).  The factors of dL/dW = sum_n dL/du_n v_n^T go to the side buffer: the adjoint tiles BEFORE their step (store_before),
// the x tiles AFTER theirs, at feature offset `soff`.
template <int XT, int T0>
__device__ __forceinline__ void linear_bwd_half(tile<1> (&xs)[2 * XT], const wptr w, float *side_row, int soff, bool store_before,
                                                int lane, rng_t &rg) {
    SX_DEP_MARK_BWD;
    const int h = lane >> 5;
    if (side_row != nullptr && store_before) {
#pragma unroll
        for (int c = 0; c < XT; ++c) store_ctile(side_row, soff + 32 * c, xs[T0 + c].v[0], h);
    }
#if defined(SX_F16X3) && !defined(SX_BWD_MMAJOR)
    {
        // k-major over XT live accumulators, as the forward layer (sx_flow_kernel.h, SX_STEP_LINEAR_TILE): the fp16 split of source
        // tile c + 1 rides between the MFMAs of k-tile c, the A fragments of the next gemm tile are requested before the MFMAs of the
        // current one.  This kernel runs ONE wave per SIMD: nothing else covers a tile's LDS round trip or its split (round 5)
        tile<1> acc[XT];
#pragma unroll
        for (int m = 0; m < XT; ++m) acc[m] = load_cfrag<1>(w.cb, XT * XT * 1024 + m * 32);
        afr cura = afr_load(w.wb, 0);
        float mx = 0.f;
        btile<1> bcur = make_btile_mx<1>(xs[T0], mx);
        __builtin_amdgcn_sched_barrier(0);
        u32x4 nhi[1][2], nlo[1][2];
        constexpr int PPM = 8 / XT;
#pragma unroll
        for (int c = 0; c < XT; ++c) {
#pragma unroll
            for (int m = 0; m < XT; ++m) {
                const int qn = c * XT + m + 1;
                const int next_off = qn < XT * XT ? ((qn % XT) * XT + qn / XT) * 1024 : -1;
                gemm_tile_pf<1, false>(w.wb, cura, next_off, bcur, acc[m], [&](int i) {
                    if (c + 1 < XT) {
                        if (PPM == 2) { if (i == 0 || i == 8) split_pair<1, true>(xs[T0 + (c + 1 < XT ? c + 1 : 0)], nhi, nlo, PPM * m + (i >> 3), mx); }
                        else { if (i == 0 || i == 3 || i == 8 || i == 11) split_pair<1, true>(xs[T0 + (c + 1 < XT ? c + 1 : 0)], nhi, nlo, PPM * m + (i >> 3) * 2 + ((i & 7) != 0), mx); }
                    }
                });
            }
            if (c + 1 < XT) bcur = btile_of<1>(nhi, nlo);
        }
        rng_note(rg, mx);
#pragma unroll
        for (int m = 0; m < XT; ++m) xs[T0 + m] = acc[m];
    }
#else
    btile<1> bx[XT];
#pragma unroll
    for (int c = 0; c < XT; ++c) bx[c] = make_btile<1>(xs[T0 + c], rg);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < XT; ++m) {
        tile<1> acc = load_cfrag<1>(w.cb, XT * XT * 1024 + m * 32);
#pragma unroll
        for (int c = 0; c < XT; ++c) gemm_tile<1>(w.wb, (m * XT + c) * 1024, bx[c], acc);
        xs[T0 + m] = acc;
    }
#endif
    if (side_row != nullptr && !store_before) {
#pragma unroll
        for (int c = 0; c < XT; ++c) store_ctile(side_row, soff + 32 * c, xs[T0 + c].v[0], h);
    }
}

// ------------------------------------------------------------------------------------------------
// Training backward with the weight gradients contracted IN the kernel (fp16 x 3 build only).
//
// The single-launch backward above writes 224 floats of per-row factors per layer to HBM (7.5 GB per 2^20-row cfg-2
// step) and sx_wgrad_layer reads them back: ~28x the step's algorithmic traffic.  Here a launch covers a PAIR of
// layers and keeps dW2 / dW1 of both in accumulator registers over all the chunks a wave processes (2 x 6 tiles =
// 192 registers: one wave per SIMD owns the 512-register file), so only the state (x | dL/dx) crosses HBM between
// launches, in fragment order (1 KB per load / store instruction).
//
// The contraction runs over the BATCH, which sits on the MFMA lanes; an MFMA sums over registers.  The factor tiles
// already exist as fp16 hi / lo B fragments (they feed the step's own GEMMs); used as the A operand against a 0/1
// selection matrix, X^T . I = X^T (cdna_hip_programming.md, 'An accumulator tile as the next MFMA's operand'), the
// matrix pipe itself turns a tile: 2 MFMAs per part, exact (fp16 x 1.0 in an fp32 accumulator), no LDS.  The turned
// parts convert back to fp16 exactly, and dW[i][j] += sum_n A[n,i] B[n,j] is the usual 3-product split GEMM with
// k = sample.  tanh h = 1 - 2r is never formed: sum_n dp_n (1 - 2 r_n)^T = (sum_n dp_n) 1^T - 2 sum_n dp_n r_n^T is
// applied once per workgroup at the end.
// ------------------------------------------------------------------------------------------------
#ifdef SX_F16X3
struct sel_t {                 // I_s as a B operand: lane (c, h), element j = [16 s + 8 (j >> 2) + 4 h + (j & 3) == c]
    h8 s[2];
};
__device__ __forceinline__ sel_t make_sel(int lane) {
    sel_t r;
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int j = 0; j < 8; ++j) r.s[st][j] = (16 * st + 8 * (j >> 2) + 4 * h + (j & 3) == c) ? (_Float16)1.0f : (_Float16)0.0f;
    return r;
}
struct tfrag {                 // one turned tile: lane = feature, k = sample; hi / lo parts, two k16 steps
    h8 hi[2], lo[2];
};
__device__ __forceinline__ void turn_part(const h8 (&p)[2], const sel_t &sel, h8 (&out)[2], float &colsum) {
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.f;
    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(p[0], sel.s[0], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(p[1], sel.s[1], t, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32x4 u;
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = pk_f16(t[8 * s + 2 * q], t[8 * s + 2 * q + 1]);     // exact: the values are fp16
        out[s] = __builtin_bit_cast(h8, u);
    }
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += t[r];
    colsum += a;
}
// turned fragments of a factor tile + (optionally) the per-feature sum over this wave's samples
__device__ __forceinline__ tfrag turn_tile(const btile<1> &b, const sel_t &sel, float &colsum) {
    tfrag t;
    turn_part(b.hi[0], sel, t.hi, colsum);
    turn_part(b.lo[0], sel, t.lo, colsum);
    return t;
}
// acc[i][j] += sum_n A[n, i] B[n, j]   (rows = A's features, columns = B's features)
__device__ __forceinline__ void contract(const tfrag &a, const tfrag &b, f32x16 &acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo[s], b.hi[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[s], b.lo[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[s], b.hi[s], acc, 0, 0, 0);
    }
}
template <int HT>
struct wacc {                  // one layer's weight-gradient accumulators (pruned halves: CT = TT = 1)
    f32x16 c2[2][HT];          // [dp tile (ls, sh)][hidden tile]: sum_n dp_n r_n^T
    f32x16 c1[HT];             // [hidden tile]:                    sum_n dh_pre_n z_n^T
    float b2[2], b1[HT];       // per-lane (= per feature) sums of dp and dh_pre over this wave's samples
};
template <int HT>
__device__ __forceinline__ void wacc_zero(wacc<HT> &a) {
    SX_DEP_MARK_BWD;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        a.b2[p] = 0.f;
#pragma unroll
        for (int m = 0; m < HT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) a.c2[p][m][r] = 0.f;
    }
#pragma unroll
    for (int m = 0; m < HT; ++m) {
        a.b1[m] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a.c1[m][r] = 0.f;
    }
}

// One affine coupling of the backward pass (as coupling_affine_bwd, XT = 2, pruned halves) with its weight gradients
// accumulated into `A`.  `live`: this lane's sample is a real row (a padded tail row contributes nothing).
template <int HT, int C0, int T0>
__device__ __forceinline__ void coupling_affine_bwd_acc(tile<1> (&xs)[4], const wptr w, float g, bool live, wacc<HT> &A,
                                                        const sel_t &sel, rng_t &rg) {
    SX_DEP_MARK_BWD;
    constexpr int XT = 2, CT = 1, TT = 1;
    constexpr int F1 = 0;
    constexpr int F2 = HT * CT * 1024 + HT * 32;
    constexpr int F2B = F2 + 2 * TT * HT * 1024;
    constexpr int B2 = F2B + 2 * TT * 32;
    constexpr int B1 = B2 + HT * 2 * TT * 1024 + HT * 32;
    // Register budget: state 64 + two layers' accumulators 192 of the 512; every fragment below is formed as late and
    // dropped as early as possible (z is split twice, r is rebuilt from its fp16 parts instead of being kept in fp32).
    // 1. conditioner (folded tanh: r = (1 - tanh) / 2)
    btile<1> bh[HT];
    {
        tile<1> hid[HT];
        hidden_layer<1, 4, HT, C0, CT, true>(xs, hid, w, F1, SX_ACT_TANH_FOLDED, rg);
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[HT - 1].v[0][r] = fast_sig2(hid[HT - 1].v[0][r]);
#pragma unroll
        for (int m = 0; m < HT; ++m) bh[m] = make_btile<1>(hid[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // 2. (kk log_scale, shift), un-transform, adjoints of the parameters
    btile<1> b0, b1;
    {
        tile<1> ls = load_cfrag<1>(w.cb, F2B), sh = load_cfrag<1>(w.cb, F2B + 32);
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            gemm_tile<1>(w.wb, F2 + m * 1024, bh[m], ls);
            gemm_tile<1>(w.wb, F2 + (HT + m) * 1024, bh[m], sh);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(ls.v[0][r]);                     // exp(-log_scale)
            const float xo = xs[T0].v[0][r], al = xs[XT + T0].v[0][r];
            xs[T0].v[0][r] = xo * __builtin_amdgcn_exp2f(-ls.v[0][r]) + sh.v[0][r];  // x_in
            const float ai = al * e;                                                // dL/dx_in
            xs[XT + T0].v[0][r] = ai;
            sh.v[0][r] = live ? -ai : 0.f;                                          // dL/dshift
            ls.v[0][r] = live ? -al * xo - g : 0.f;                                 // dL/dlog_scale (incl. -sum(ls))
        }
        b0 = make_btile<1>(ls, rg);
        b1 = make_btile<1>(sh, rg);
    }
    __builtin_amdgcn_sched_barrier(0);
    // 3. dW2 partial: sum_n dp_n r_n^T (turned tiles; the tanh fix-up happens once per workgroup), one hidden tile at a time
    if (!(SX_X & 512)) {        // (SX_X & 512: timing experiment without the weight-gradient contraction)
        float dummy = 0.f;
        const tfrag t0 = turn_tile(b0, sel, A.b2[0]), t1 = turn_tile(b1, sel, A.b2[1]);
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            const tfrag th = turn_tile(bh[m], sel, dummy);
            contract(t0, th, A.c2[0][m]);
            contract(t1, th, A.c2[1][m]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // 4. dh = W2^T [dls; dsh];  dh_pre = dh (1 - tanh^2) with tanh = 1 - 2 r, r = hi + lo of its fp16 parts
    btile<1> bd[HT];
#pragma unroll
    for (int m = 0; m < HT; ++m) {
        tile<1> dh;
#pragma unroll
        for (int r = 0; r < 16; ++r) dh.v[0][r] = 0.f;
        gemm_tile<1>(w.wb, B2 + (m * 2) * 1024, b0, dh);
        gemm_tile<1>(w.wb, B2 + (m * 2 + 1) * 1024, b1, dh);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float rr = (float)bh[m].hi[0][s][j] + (float)bh[m].lo[0][s][j];
                const float th = 1.f - 2.f * rr;
                dh.v[0][8 * s + j] *= (1.f - th * th);
            }
        bd[m] = make_btile<1>(dh, rg);
        __builtin_amdgcn_sched_barrier(0);
    }
    // 5. adjoint of the conditioning tile += W1^T dh_pre;  dW1 partial: sum_n dh_pre_n z_n^T
    {
        tile<1> dz;
#pragma unroll
        for (int r = 0; r < 16; ++r) dz.v[0][r] = 0.f;
#pragma unroll
        for (int m = 0; m < HT; ++m) gemm_tile<1>(w.wb, B1 + m * 1024, bd[m], dz);
#pragma unroll
        for (int r = 0; r < 16; ++r) xs[XT + C0].v[0][r] += dz.v[0][r];
        float dummy = 0.f;
        if (SX_X & 512) return;
        const tfrag tz = turn_tile(make_btile<1>(xs[C0]), sel, dummy);      // z again (the tile itself is unchanged)
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            const tfrag td = turn_tile(bd[m], sel, A.b1[m]);
            contract(td, tz, A.c1[m]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
#endif   // SX_F16X3
