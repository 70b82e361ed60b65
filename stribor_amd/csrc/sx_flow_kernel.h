// Fused flow kernel: a whole NormalizingFlow (or one Coupling, or one conditioner MLP) in ONE launch.
//
// Replaces the Python loop of stribor/flow.py:99-130 and, per coupling layer, the torch-op chain
//   mask -> x*mask -> Linear -> Tanh -> Linear -> chunk -> (x-shift)*exp(-log_scale) -> blend -> sum
// of stribor/flows/coupling.py:48-95 + flows/affine.py:59-123 + net/mlp.py:65.
//
// Mapping to CDNA4 (MI355X_MICROARCH.md, cdna_hip_programming.md §3):
//   * one wave = 32 samples.  Samples sit on the MFMA column (lane&31), features on the C rows, so the
//     flow state x[D] of a sample is 16*D/32 VGPRs per lane in v_mfma_f32_32x32x2_f32 C-fragment order
//     (lane half h = lane>>5 owns features kmap(r,h), r = 0..15, of every 32-wide tile);
//   * a C tile is directly the B operand of the next GEMM (k-step s <-> feature kmap(s,h); the weights
//     are pre-permuted by sx_pack_linear), so x -> hidden -> (log_scale, shift) -> x' never leaves
//     registers: no LDS transposes, no HBM round trips between layers;
//   * GEMMs run on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32: an fp32 fma chain, so parity
//     with the CPU reference holds at ~1e-7); tanh/exp are v_exp_f32 + v_rcp_f32 on the VALU, which
//     overlaps the matrix pipe of the co-resident wave;
//   * weights of one step (<= ~25 KB for D=64,H=64) stream L2 -> LDS by LDS-DMA (global_load_lds x16 B),
//     double-buffered: step s+1 lands while step s computes; A operands come from LDS as ds_read_b128
//     (4 k-steps per read, conflict-free: lane-linear 16 B);
//   * 256-thread workgroups (4 waves = 128 samples per pass), persistent grid-stride over sample chunks,
//     2 workgroups per CU so that one wave's VALU phase hides under its SIMD partner's MFMA phase;
//   * per-sample log-det / log-prob: in-lane sums + ONE cross-half shuffle; optional batch sum as fp64
//     block partials + one atomic per workgroup (flow.py:129 + the multi-GPU all-reduce operand).
#include "sx_common.h"
#include "sx_flow_types.h"

#define SX_ROWS_PER_BLOCK 128
#define SX_HALF_LOG_2PI 0.91893853320467274178f

__host__ __device__ static inline int sx_kmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

extern __shared__ __attribute__((aligned(16))) float smem[];

typedef __attribute__((address_space(3))) void lds_void;

// ---- weights: L2 -> LDS by LDS-DMA, 1 KiB per wave-instruction, lane-linear -----------------------------
__device__ __forceinline__ void stage_blob(const float *__restrict__ g, int lds_float_off, uint32_t n_floats) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n_bytes = n_floats * 4u;    // multiple of 1024 (host pads blobs to 256 floats)
    const char *gsrc = reinterpret_cast<const char *>(g);
    char *ldst = reinterpret_cast<char *>(smem + lds_float_off);
    for (uint32_t off = wave * 1024u; off < n_bytes; off += 4u * 1024u) {
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(gsrc + off + lane * 16),
                                         (lds_void *)(ldst + off), 16, 0, 0);
    }
}

// ---- one 32x32 output tile += A(32 x 32) . B(32 x 32 samples): 16 k-steps, A from LDS ------------------
__device__ __forceinline__ f32x16 gemm_tile(int a_off, const f32x16 &b, f32x16 acc, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 a = reinterpret_cast<const f32x4 *>(smem)[(a_off >> 2) + g * 64 + lane];   // ds_read_b128
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * g + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * g + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * g + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * g + 3], acc, 0, 0, 0);
    }
    return acc;
}

// bias / per-feature constants in C-fragment order: [h][16] floats at off
__device__ __forceinline__ f32x16 load_cfrag(int off, int h) {
    f32x16 v;
    const f32x4 *p = reinterpret_cast<const f32x4 *>(smem) + ((off >> 2) + h * 4);
    const f32x4 a = p[0], b = p[1], c = p[2], d = p[3];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
    v[12] = d.x; v[13] = d.y; v[14] = d.z; v[15] = d.w;
    return v;
}

__device__ __forceinline__ float act_one(float v, int act) {
    switch (act) {
        case SX_ACT_RELU: return fmaxf(v, 0.f);
        case SX_ACT_SIGMOID: return fast_rcp(1.f + fast_exp(-v));
        case SX_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case SX_ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));
        case SX_ACT_LEAKYRELU: return v > 0.f ? v : 0.01f * v;
        case SX_ACT_SILU: return v * fast_rcp(1.f + fast_exp(-v));
        case SX_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        default: return v;
    }
}
__device__ __forceinline__ void activate(f32x16 &v, int act) {
    if (act == SX_ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fast_tanh(v[r]);
    } else if (act != SX_ACT_IDENTITY) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = act_one(v[r], act);
    }
}

// hidden[m] = act(W . state[C0..C0+CT) + b),  blob = pack_linear(W, HT m-tiles, CT k-tiles)
template <int TX, int HT, int C0, int CT>
__device__ __forceinline__ void hidden_from_state(const f32x16 (&xs)[TX], f32x16 (&hid)[HT], int base, int act,
                                                  int lane) {
    const int h = lane >> 5;
    const int bias = base + HT * CT * 1024;
#pragma unroll
    for (int m = 0; m < HT; ++m) {
        f32x16 acc = load_cfrag(bias + m * 32, h);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc = gemm_tile(base + (m * CT + c) * 1024, xs[C0 + c], acc, lane);
        activate(acc, act);
        hid[m] = acc;
    }
}

// Affine coupling step (affine.py:104-109 through coupling.py:69-95), conditioner evaluated once (quirk Q2).
template <int TX, int HT, int C0, int CT, int T0, int TT>
__device__ __forceinline__ void coupling_affine(f32x16 (&xs)[TX], int base, const dstep &st, float &ldj, int lane) {
    const int h = lane >> 5;
    f32x16 hid[HT];
    hidden_from_state<TX, HT, C0, CT>(xs, hid, base, st.act, lane);
    const int a2 = base + HT * CT * 1024 + HT * 32;   // pack_linear(W2: 2*TT m-tiles, HT k-tiles)
    const int b2 = a2 + 2 * TT * HT * 1024;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        f32x16 ls = load_cfrag(b2 + (2 * t) * 32, h);
        f32x16 sh = load_cfrag(b2 + (2 * t + 1) * 32, h);
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            ls = gemm_tile(a2 + ((2 * t) * HT + m) * 1024, hid[m], ls, lane);
            sh = gemm_tile(a2 + ((2 * t + 1) * HT + m) * 1024, hid[m], sh, lane);
        }
        f32x16 &x = xs[T0 + t];
        if (st.reverse) {
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = (x[r] - sh[r]) * fast_exp(-ls[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = x[r] * fast_exp(ls[r]) + sh[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s += ls[r];
    }
    ldj += st.ldj_scale * s;
}

// Elementwise affine with per-feature constants (st.Affine without latent_net, affine.py:63-64,104-109)
template <int TX>
__device__ __forceinline__ void affine_const(f32x16 (&xs)[TX], int base, const dstep &st, int x_tiles, float &ldj,
                                             int lane) {
    const int h = lane >> 5;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TX; ++t) {
        if (t < x_tiles) {
            const f32x16 ls = load_cfrag(base + t * 32, h);
            const f32x16 sh = load_cfrag(base + (TX + t) * 32, h);
            if (st.reverse) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xs[t][r] = (xs[t][r] - sh[r]) * fast_exp(-ls[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) xs[t][r] = xs[t][r] * fast_exp(ls[r]) + sh[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s += ls[r];     // padding slots carry log_scale = 0
        }
    }
    ldj += st.ldj_scale * s;
}

__device__ __forceinline__ float ld_elem(const void *p, int64_t off, int bf16) {
    if (bf16) return bf16_to_f32(reinterpret_cast<const uint16_t *>(p)[off]);
    return reinterpret_cast<const float *>(p)[off];
}
__device__ __forceinline__ void st_elem(void *p, int64_t off, float v, int bf16) {
    if (bf16) reinterpret_cast<uint16_t *>(p)[off] = f32_to_bf16(v);
    else reinterpret_cast<float *>(p)[off] = v;
}

// MODE 0: flow programs (coupling / affine-const steps); MODE 1: + persistent hidden state (MLP programs);
// MODE 2: flow programs with dense linear layers (AffineLU / MatrixExponential): + a second state tile set
template <int TX, int HT, int MODE>
__global__ __launch_bounds__(256, 2) void flow_fused_kernel(
    const dprog prog, const float *__restrict__ blobs, const void *__restrict__ x,
    const float *__restrict__ latent, const int32_t *__restrict__ in_col, const int32_t *__restrict__ out_col,
    void *__restrict__ y, float *__restrict__ ldj_out, float *__restrict__ logp_out, double *__restrict__ sum_out,
    float *__restrict__ mlp_out, int64_t mlp_out_stride, int mlp_out_dim, int64_t n_rows, int buf_floats,
    int bf16, const float *__restrict__ row_t) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int dim = prog.dim, x_tiles = prog.x_tiles, n_steps = prog.n_steps;
    const int64_t n_chunks = (n_rows + SX_ROWS_PER_BLOCK - 1) / SX_ROWS_PER_BLOCK;
    double block_sum = 0.0;

    // prologue: first step's weights into buffer 0
    int cur = 0;
    if ((int64_t)blockIdx.x < n_chunks && n_steps > 0 && prog.steps[0].blob_floats)
        stage_blob(blobs + prog.steps[0].blob_off, 0, prog.steps[0].blob_floats);

    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int64_t row = chunk * SX_ROWS_PER_BLOCK + wave * 32 + j;
        const int64_t lrow = row < n_rows ? row : n_rows - 1;      // clamp loads, mask stores
        const bool has_next_chunk = chunk + gridDim.x < n_chunks;

        // ---- load the state tiles in C-fragment order ---------------------------------------------------
        f32x16 xs[TX];
#pragma unroll
        for (int t = 0; t < TX; ++t) {
            if (t < x_tiles) {
                if (prog.identity_cols) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = 32 * t + 8 * q + 4 * h;
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (c + 3 < dim) {
                            if (bf16) {
                                const u16x4 u = *reinterpret_cast<const u16x4 *>(
                                    reinterpret_cast<const uint16_t *>(x) + lrow * dim + c);
                                v = f32x4{bf16_to_f32(u.x), bf16_to_f32(u.y), bf16_to_f32(u.z), bf16_to_f32(u.w)};
                            } else {
                                v = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(x) + lrow * dim + c);
                            }
                        }
                        xs[t][4 * q + 0] = v.x; xs[t][4 * q + 1] = v.y; xs[t][4 * q + 2] = v.z; xs[t][4 * q + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = in_col[32 * t + sx_kmap(r, h)];
                        xs[t][r] = c >= 0 ? ld_elem(x, lrow * dim + c, bf16) : 0.f;
                    }
                }
            } else {   // latent tiles (fp32), conditioner-only inputs (coupling.py:64-65)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = 32 * (t - x_tiles) + sx_kmap(r, h);
                    xs[t][r] = (latent != nullptr && c < prog.latent_dim) ? latent[lrow * prog.latent_dim + c] : 0.f;
                }
            }
        }

        float ldj = 0.f;
        float ldj_c = 0.f;
        f32x16 hid[MODE == 1 ? HT : 1];
        f32x16 xnew[MODE == 2 ? TX : 1];

        for (int s = 0; s < n_steps; ++s) {
            // (1) this step's weights were issued one step ago (or in the prologue): wait for MY pieces, then
            //     the barrier makes every wave's pieces visible AND guarantees all waves left step s-1,
            //     i.e. nobody still reads buffer cur^1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // (2) refill buffer cur^1 with the next step's weights; the DMA flies under this step's MFMAs.
            const int nxt = (s + 1 < n_steps) ? s + 1 : 0;
            if ((s + 1 < n_steps || has_next_chunk) && prog.steps[nxt].blob_floats)
                stage_blob(blobs + prog.steps[nxt].blob_off, (cur ^ 1) * buf_floats, prog.steps[nxt].blob_floats);

            const dstep st = prog.steps[s];
            const int base = cur * buf_floats;
            switch (st.kind) {
                case SX_STEP_COUPLING_AFFINE:
                    if constexpr (TX >= 2) {
                        if (st.ct == TX / 2 && st.c0 == 0 && st.t0 == TX / 2) {          // cond = low tiles
                            coupling_affine<TX, HT, 0, TX / 2, TX / 2, TX / 2>(xs, base, st, ldj, lane);
                            break;
                        }
                        if (st.ct == TX / 2 && st.c0 == TX / 2 && st.t0 == 0) {          // cond = high tiles
                            coupling_affine<TX, HT, TX / 2, TX / 2, 0, TX / 2>(xs, base, st, ldj, lane);
                            break;
                        }
                    }
                    coupling_affine<TX, HT, 0, TX, 0, TX>(xs, base, st, ldj, lane);       // dense
                    break;
                case SX_STEP_AFFINE_CONST:
                    affine_const<TX>(xs, base, st, x_tiles, ldj, lane);
                    break;
                case SX_STEP_MLP_HIDDEN:
                    if constexpr (MODE == 1) hidden_from_state<TX, HT, 0, TX>(xs, hid, base, st.act, lane);
                    break;
                case SX_STEP_MLP_HIDDEN2:
                    if constexpr (MODE == 1) {
                        f32x16 nh[HT];
                        const int bias = base + HT * HT * 1024;
#pragma unroll
                        for (int m = 0; m < HT; ++m) {
                            f32x16 acc = load_cfrag(bias + m * 32, h);
#pragma unroll
                            for (int c = 0; c < HT; ++c) acc = gemm_tile(base + (m * HT + c) * 1024, hid[c], acc, lane);
                            activate(acc, st.act);
                            nh[m] = acc;
                        }
#pragma unroll
                        for (int m = 0; m < HT; ++m) hid[m] = nh[m];
                    }
                    break;
                case SX_STEP_MLP_OUT_TILE:
                    if constexpr (MODE == 1) {
                        f32x16 acc = load_cfrag(base + HT * 1024, h);
#pragma unroll
                        for (int c = 0; c < HT; ++c) acc = gemm_tile(base + c * 1024, hid[c], acc, lane);
                        if (row < n_rows) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int c = 32 * st.t0 + 8 * q + 4 * h;
                                float *o = mlp_out + row * mlp_out_stride + c;
                                if (c + 3 < mlp_out_dim && (mlp_out_stride & 3) == 0) {
                                    *reinterpret_cast<f32x4 *>(o) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (c + e < mlp_out_dim) o[e] = acc[4 * q + e];
                                }
                            }
                        }
                    }
                    break;
                case SX_STEP_LINEAR_TILE:
                    // one 32-row slab of y = M . x + b (AffineLU affine.py:157,159-163; MatrixExponential
                    // affine.py:243-270 with the triangular solves folded into M on the host, in fp64)
                    if constexpr (MODE == 2) {
                        f32x16 acc = load_cfrag(base + TX * 1024, h);
#pragma unroll
                        for (int c = 0; c < TX; ++c) acc = gemm_tile(base + c * 1024, xs[c], acc, lane);
#pragma unroll
                        for (int t = 0; t < TX; ++t)
                            if (t == st.t0) xnew[t] = acc;
                        if (st.tt) {   // last slab: commit
#pragma unroll
                            for (int t = 0; t < TX; ++t)
                                if (t < x_tiles) xs[t] = xnew[t];
                        }
                    }
                    break;
                case SX_STEP_ROW_SCALE_EXP:
                    // x *= exp(+-diag * t_row)  (affine.py:263), t_row optionally log1p|t| (affine.py:239-240)
                    if constexpr (MODE == 2) {
                        float tr = row_t != nullptr ? row_t[lrow] : st.ldj_const;
                        if (st.act) tr = log1pf(fabsf(tr));
                        const float sg = st.reverse ? -tr : tr;
                        float sd = 0.f;
#pragma unroll
                        for (int t = 0; t < TX; ++t) {
                            if (t < x_tiles) {
                                const f32x16 dg = load_cfrag(base + t * 32, h);
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    xs[t][r] *= fast_exp(dg[r] * sg);
                                    sd += dg[r];
                                }
                            }
                        }
                        ldj += st.ldj_scale * sd * tr;
                    }
                    break;
                default: break;
            }
            if (st.kind != SX_STEP_ROW_SCALE_EXP) ldj_c += st.ldj_const;
            cur ^= 1;
        }

        // ---- epilogue: outputs -----------------------------------------------------------------------------
        if (y != nullptr && row < n_rows) {
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                if (t < x_tiles) {
                    if (prog.identity_cols) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int c = 32 * t + 8 * q + 4 * h;
                            if (c + 3 < dim) {
                                if (bf16) {
                                    u16x4 u{f32_to_bf16(xs[t][4 * q]), f32_to_bf16(xs[t][4 * q + 1]),
                                            f32_to_bf16(xs[t][4 * q + 2]), f32_to_bf16(xs[t][4 * q + 3])};
                                    *reinterpret_cast<u16x4 *>(reinterpret_cast<uint16_t *>(y) + row * dim + c) = u;
                                } else {
                                    *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(y) + row * dim + c) =
                                        f32x4{xs[t][4 * q], xs[t][4 * q + 1], xs[t][4 * q + 2], xs[t][4 * q + 3]};
                                }
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int c = out_col[32 * t + sx_kmap(r, h)];
                            if (c >= 0) st_elem(y, row * dim + c, xs[t][r], bf16);
                        }
                    }
                }
            }
        }
        if (ldj_out != nullptr || logp_out != nullptr || sum_out != nullptr) {
            float sq = 0.f;
            if (logp_out != nullptr) {
#pragma unroll
                for (int t = 0; t < TX; ++t)
                    if (t < x_tiles) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sq += xs[t][r] * xs[t][r];
                    }
            }
            const float part_lp = ldj - 0.5f * sq;
            const float tot = part_lp + __shfl_xor(part_lp, 32, 64);   // the two lane halves of one sample
            const float ldj_tot = ldj + __shfl_xor(ldj, 32, 64) + ldj_c;
            const float lp = tot + ldj_c - (float)dim * SX_HALF_LOG_2PI;
            if (row < n_rows && h == 0) {
                if (ldj_out != nullptr) ldj_out[row] = ldj_tot;
                if (logp_out != nullptr) logp_out[row] = lp;
                block_sum += (double)(logp_out != nullptr ? lp : ldj_tot);
            }
        }
    }

    if (sum_out != nullptr) {
        double *part = reinterpret_cast<double *>(smem);   // no second __shared__ object beside the DMA ring
        block_sum = wave_sum_f64(block_sum);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (lane == 0) part[wave] = block_sum;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sum_out, (part[0] + part[1]) + (part[2] + part[3]));
    }
}


template <int TX, int HT>
static int sx_flow_launch_impl(const sx_flow_args &a) {
#define SX_FL(MD)                                                                                              \
    do {                                                                                                       \
        auto k = flow_fused_kernel<TX, HT, MD>;                                                                \
        if (a.lds > 48 * 1024) {                                                                               \
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, a.lds); \
            if (e != hipSuccess) { sx_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; } \
        }                                                                                                      \
        hipLaunchKernelGGL(k, dim3(a.grid), dim3(256), a.lds, a.stream, a.prog, a.blobs, a.x, a.latent, a.in_col, \
                           a.out_col, a.y, a.ldj_out, a.logp_out, a.sum_out, a.mlp_out, a.mlp_out_stride,      \
                           a.mlp_out_dim, a.n_rows, a.buf_floats, a.bf16, a.row_t);                                     \
    } while (0)
    if (a.mlp_mode == 1) SX_FL(1); else if (a.mlp_mode == 2) SX_FL(2); else SX_FL(0);
#undef SX_FL
    SX_LAUNCH_CHECK();
    return SX_OK;
}
