// Weight-gradient contractions over the batch axis (training):  dW[i][j] += sum_n A[n][i] * B[n][j],
// db[i] += sum_n A[n][i]  with n_rows ~ 1e6 and a few dozen to a few hundred features.
//
// A tall-skinny "A^T B" whose reduction axis is the batch: library GEMMs run it on a handful of workgroups
// (measured 1.5 ms per 64 x 64 x 2^20 product = 9 % of the HBM rate).  Here the batch is split over the whole chip:
//   * 32-row groups are dealt to the waves of 256 workgroups.  v_mfma_f32_32x32x2_f32 wants A^T[i = lane & 31][k = lane >> 5]
//     and B[k][j = lane & 31] with k = the row, i.e. lane = feature.  A wave reads a 32-row x 32-feature tile with four
//     coalesced 16 B-per-lane loads -- 4 KB contiguous in the backward flow kernel's row-group layout (element (row n,
//     feature f) at (n >> 5) * ld + f * 32 + (n & 31)), eight whole 128 B segments per load for row-major torch
//     tensors -- and turns it through a private LDS patch: lane (i, kk) ends up with rows 16 kk .. 16 kk + 15 of
//     feature i and feeds them to 16 MFMAs (the pairing of rows into k-steps is free as long as A and B agree).
//   * each wave keeps its (M/32) x (Nc/32) output tiles in registers over all its groups; the waves of a workgroup
//     then sum their tiles in LDS by turns, and the workgroup stores ONE partial tile to a per-stream scratch;
//   * a second small kernel sums the <= 256 partials per element and adds them to dW / db through optional row /
//     column maps (one writer per element: deterministic).
// The first version added the waves' tiles with float atomics: global (4096 waves x 4096 contended adds took twice as
// long as the data pass) and then LDS (ds_add_f32 serialises its lanes: 40 us per launch).
#include "sx_common.h"
#include <mutex>
#include <vector>

constexpr int PATCH = 32 * 36;       // one 32-feature x 32-row tile, feature rows padded to 36 floats

// ---- fp16 x 3 contraction (SX_WGRAD_ROW_GROUPS_F16X3): the operands of a backward program are fp16 x 3 GEMM operands already
// (scaled into fp16's range, range-tracked by the kernel that wrote them), so the batch contraction can run on the matrix pipe as
// well: a lane's 16 rows of a feature split hi + lo (hi = f16(v), lo = f16(v - hi), to nearest: sx_flow_kernel.h pk_f16), two 16-row steps of
// v_mfma_f32_32x32x16_f16 with three products each -- 6 MFMAs x 32 cycles per tile pair and 32 rows instead of 16 x 64.
typedef __attribute__((address_space(3))) void lds_void;
typedef _Float16 wg_h8 __attribute__((ext_vector_type(8)));
typedef uint32_t wg_u4 __attribute__((ext_vector_type(4)));
struct wg_split { wg_h8 hi[2], lo[2]; };
__device__ __forceinline__ uint32_t wg_pk(float a, float b) {      // v_cvt_pk_f16_f32, round to nearest even (a truncating split biases a sum over 2^20 rows)
    typedef float wg_f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 wg_h2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((wg_f2){a, b}, wg_h2));
}
__device__ __forceinline__ wg_split wg_make(const f32x4 (&v)[4]) {
    wg_split o;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        wg_u4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v0 = v[2 * s + (q >> 1)][2 * (q & 1)], v1 = v[2 * s + (q >> 1)][2 * (q & 1) + 1];
            const uint32_t ph = wg_pk(v0, v1);
            float l0, l1;
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(ph), "v"(v0));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(ph), "v"(v1));
            hi[q] = ph;
            lo[q] = wg_pk(l0, l1);
        }
        o.hi[s] = __builtin_bit_cast(wg_h8, hi);
        o.lo[s] = __builtin_bit_cast(wg_h8, lo);
    }
    return o;
}
__device__ __forceinline__ void wg_contract(const wg_split &a, const wg_split &b, f32x16 &acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo[s], b.hi[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[s], b.lo[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi[s], b.hi[s], acc, 0, 0, 0);
    }
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// coalesced global -> registers: instr q, lane L holds feature 8q + (L >> 3), rows 4 (L & 7) .. + 3 of the tile
__device__ __forceinline__ void load_tile(const float *tile_base, bool ok, int lane, f32x4 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        v[q] = ok ? *reinterpret_cast<const f32x4 *>(tile_base + q * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
}
// registers -> LDS patch -> registers: lane (i, kk) gets rows 16 kk .. 16 kk + 15 of feature i.  DS operations of
// one wave execute in order, so the patch is reused tile after tile without barriers.
__device__ __forceinline__ void turn_tile(float *patch, int lane, f32x4 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(patch + (8 * q + (lane >> 3)) * 36 + 4 * (lane & 7)) = v[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4 *>(patch + (lane & 31) * 36 + 16 * (lane >> 5) + 4 * q);
}

// row-major tiles (32 rows x 32 features = 32 segments of 128 B, row stride ld): instr q, lane L holds row
// 8q + (L >> 3), features 4 (L & 7) .. + 3 -- eight whole 128 B segments per load; rows past `rows_left` read as zero
__device__ __forceinline__ void load_tile_rows(const float *tile_base, int64_t ld, int rows_left, bool f_ok, int lane,
                                               f32x4 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 8 * q + (lane >> 3);
        v[q] = (f_ok && row < rows_left) ? *reinterpret_cast<const f32x4 *>(tile_base + row * ld + 4 * (lane & 7))
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
// the patch is written row-major (conflict-free 16 B writes) and read down its columns: lane (i, kk) gets feature i
// of rows 16 kk .. 16 kk + 15 (lanes of a half read 32 consecutive floats)
__device__ __forceinline__ void turn_tile_rows(float *patch, int lane, f32x4 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(patch + (8 * q + (lane >> 3)) * 36 + 4 * (lane & 7)) = v[q];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[q][c] = patch[(16 * (lane >> 5) + 4 * q + c) * 36 + (lane & 31)];
}

// LAYOUT 1: SX_WGRAD_ROW_GROUPS; 2: row-major [n_rows, ld] with 16 B loads through the LDS patch (16-byte aligned rows);
// 0: row-major with per-element loads (any alignment: lane (i, kk) loads element (row 2s + kk, feature i), 128 B
// contiguous per lane half).  blockIdx.y = 128-feature slab of A (row-major only: M up to 2048, B is re-read per slab
// from L2 / MALL).
template <int MT, int NT, int WB, int LAYOUT>
__global__ __launch_bounds__(64 * WB) void wgrad_kernel(const float *__restrict__ A, int64_t lda,
                                                        const float *__restrict__ B, int64_t ldb, int64_t n_rows,
                                                        float *__restrict__ part, int m_total, int n_valid) {
    constexpr int M32 = 32 * MT, N32 = 32 * NT;
    constexpr int RED = M32 * N32 + M32;
    constexpr bool GROUPS = LAYOUT == 1 || LAYOUT == 3, VEC_ROWS = LAYOUT == 2, F16 = LAYOUT == 3;
    // fp16 x 3 form: the next group's tiles come by LDS-DMA while the current group's MFMAs run (MT + NT tiles of 4 KB per
    // wave; no patch, no register staging -- 256 accumulator registers of a 128 x 128 output leave no room for one).
    // All vector-memory traffic of the loop is then LDS-DMA with COUNTED waits: beside an LDS-DMA in flight the compiler
    // waits vmcnt(0) for any ordinary load, i.e. it would drain the prefetch at the first use of a tile.
    constexpr bool DMA_B = F16;
    constexpr int STAGE = DMA_B ? WB * (MT + NT) * 1024 : WB * PATCH;
    // epilogue: the waves' accumulator tiles meet in LDS in C-FRAGMENT order (one ds_write_b128 per four registers), in as many
    // separate regions as fit (NREG): the first NREG waves only write, later turns add in place, the last pass sums the regions
    constexpr int FRAG = MT * NT * 1024;
    constexpr int NREG = WB * FRAG <= 34816 ? WB : (2 * FRAG <= 34816 ? 2 : 1);          // 136 KB of tiles at most
    constexpr int EPI = NREG * FRAG + WB * M32;                                            // + the waves' bias sums
    constexpr int LDS_FLOATS = ((GROUPS || VEC_ROWS) && STAGE > EPI) ? STAGE : EPI;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    A += 128 * blockIdx.y;
    const int m_valid = m_total - 128 * (int)blockIdx.y;                 // >= M32 for all but the last slab
    part += (int64_t)blockIdx.y * gridDim.x * RED;
    // (the wave index through readfirstlane: group indices and tile base pointers are then provably wave-uniform -- scalar
    //  registers and scalar-base loads instead of a 64-bit address pair per lane and tile)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, kk = lane >> 5;
    [[maybe_unused]] float *patch = lds + wave * PATCH;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    float bsum[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) bsum[m] = 0.f;

    // 32-row groups are dealt to the waves, grid-strided
    const int64_t n_waves = (int64_t)gridDim.x * WB;
    const int64_t w_id = (int64_t)blockIdx.x * WB + wave;
    const int64_t n_groups = (n_rows + 31) >> 5;
    // features past m_valid / n_valid and rows past n_rows are masked after the turn (lane i = feature)
    // fp16 x 3 form: the tiles of a wave's NEXT group are requested as soon as the current ones have been split (their
    // registers are free from then on), so the loads fly under the current group's MFMAs -- with 128 .. 256 accumulator
    // registers a SIMD holds one or two waves, and load -> wait -> compute in turn left HBM idle two thirds of the time
    constexpr int PFB = DMA_B ? 0 : NT;
    f32x4 a[MT][4], b[NT][4];
    // LDS-DMA of a B tile: lane L of instruction q fills 16-byte slot 64 q + L of the tile's 4 KB -- and FETCHES the chunk that
    // belongs there: (feature f, 4-row chunk rc) lives in slot 8 f + ((rc + f) & 7).  The rotation makes the turned read
    // (lane = feature, its 16 rows = 4 chunks) conflict-free: eight neighbouring features hit eight different 16-byte columns.
    [[maybe_unused]] float *aland = lds + wave * ((MT + NT) * 1024), *bland = aland + MT * 1024;     // wave-uniform (m0)
    [[maybe_unused]] auto issue_tile = [&](const float *gtile, float *ltile) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gtile), 0, 4096, 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = 8 * q + (lane >> 3), rc = ((lane & 7) - f) & 7;
            // (MUBUF form: a FLAT-encoded LDS-DMA in flight turns every compiler-placed wait into a full one, sx_flow_kernel.h stage_blob)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(ltile + q * 256), 16, (f * 32 + rc * 4) * 4, 0, 0, 0);
        }
    };
    // the turned read: lane (feature i, half kk) takes rows 16 kk .. 16 kk + 15 of its feature = chunks 4 kk .. 4 kk + 3
    [[maybe_unused]] auto read_tile = [&](const float *ltile, f32x4 (&v)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4 *>(ltile + i * 32 + (((4 * kk + q) + i) & 7) * 4);
    };
    if constexpr (F16) {
        const bool first = w_id < n_groups;
        if constexpr (DMA_B) {
            if (first) {            // B tiles first, then A, as in the loop (the counted waits below rely on the order)
#pragma unroll
                for (int n = 0; n < NT; ++n) issue_tile(B + w_id * ldb + n * 1024, bland + n * 1024);
#pragma unroll
                for (int m = 0; m < MT; ++m) issue_tile(A + w_id * lda + m * 1024, aland + m * 1024);
            }
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) load_tile(A + w_id * lda + m * 1024, first, lane, a[m]);
#pragma unroll
            for (int n = 0; n < PFB; ++n) load_tile(B + w_id * ldb + n * 1024, first, lane, b[n]);
        }
    }
    for (int64_t g = w_id; g < n_groups; g += n_waves) {
        const int64_t rem = n_rows - 32 * g;
        [[maybe_unused]] const int64_t gn = g + n_waves;
        [[maybe_unused]] const bool more = gn < n_groups;
        if constexpr (GROUPS) {
            if constexpr (!F16) {
#pragma unroll
                for (int m = 0; m < MT; ++m) load_tile(A + g * lda + m * 1024, true, lane, a[m]);
            }
            if constexpr (!DMA_B) {
#pragma unroll
                for (int n = F16 ? PFB : 0; n < NT; ++n) load_tile(B + g * ldb + n * 1024, true, lane, b[n]);
            } else {
                // this group's B tiles have landed: vector-memory operations retire in order, and the only younger ones are the
                // 4 MT DMAs of this group's A tiles (each is waited for where the m loop reads it)
                wait_vm<4 * MT>();
            }
            const int left = (rem < 32 ? (int)rem : 32) - 16 * kk;      // valid rows among this lane's 16
            // (fp16 x 3 form: an A tile is turned right before its split, in the m loop below -- the tile requested last is then
            //  also needed last, a whole iteration later)
            auto turn_a = [&](int m) {
                turn_tile(patch, lane, a[m]);
                const bool f_ok = 32 * m + i < m_valid;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[m][q][c] = (f_ok && 4 * q + c < left) ? a[m][q][c] : 0.f;
            };
            if constexpr (!F16) {
#pragma unroll
                for (int m = 0; m < MT; ++m) turn_a(m);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if constexpr (DMA_B) {
                    read_tile(bland + n * 1024, b[n]);
                } else {
                    turn_tile(patch, lane, b[n]);
                }
                const bool f_ok = 32 * n + i < n_valid;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) b[n][q][c] = (f_ok && 4 * q + c < left) ? b[n][q][c] : 0.f;
            }
        } else if constexpr (VEC_ROWS) {
            const int rows_left = rem < 32 ? (int)rem : 32;
#pragma unroll
            for (int m = 0; m < MT; ++m)
                load_tile_rows(A + 32 * g * lda + 32 * m, lda, rows_left, 32 * m + 4 * (lane & 7) < m_valid, lane, a[m]);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                load_tile_rows(B + 32 * g * ldb + 32 * n, ldb, rows_left, 32 * n + 4 * (lane & 7) < n_valid, lane, b[n]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                turn_tile_rows(patch, lane, a[m]);
                if (32 * m + i >= m_valid) {                            // a quad straddling the last feature
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                turn_tile_rows(patch, lane, b[n]);
                if (32 * n + i >= n_valid) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[n][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        } else {
            const float *ra = A + (32 * g + kk) * lda + i, *rb = B + (32 * g + kk) * ldb + i;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int s = 4 * q + c;                             // k-step s pairs rows 2s, 2s + 1
                    const bool ok = 2 * s + kk < rem;
#pragma unroll
                    for (int m = 0; m < MT; ++m) a[m][q][c] = (ok && 32 * m + i < m_valid) ? ra[2 * s * lda + 32 * m] : 0.f;
#pragma unroll
                    for (int n = 0; n < NT; ++n) b[n][q][c] = (ok && 32 * n + i < n_valid) ? rb[2 * s * ldb + 32 * n] : 0.f;
                }
        }
        if constexpr (F16) {
            wg_split bs[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                bs[n] = wg_make(b[n]);
                if (n < PFB) load_tile(B + gn * ldb + n * 1024, more, lane, b[n]);
            }
            if constexpr (DMA_B) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // every read of the landing tiles has returned
                if (more) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) issue_tile(B + gn * ldb + n * 1024, bland + n * 1024);
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if constexpr (GROUPS) {
                    const int left = (rem < 32 ? (int)rem : 32) - 16 * kk;
                    if constexpr (DMA_B) {
                        // tile m of this group: younger operations are the rest of this group's A tiles, the next group's B tiles
                        // and the next group's A tiles issued so far -- 4 (MT - 1 + NT) in the steady state, 4 (MT - 1 - m) in a wave's last group
                        if (more) wait_vm<4 * (MT - 1) + 4 * NT>();
                        else if (m == 0) wait_vm<4 * (MT - 1)>();
                        else if (m == 1) wait_vm<(MT > 2 ? 4 * (MT - 2) : 0)>();
                        else if (m == 2) wait_vm<(MT > 3 ? 4 * (MT - 3) : 0)>();
                        else wait_vm<0>();
                        read_tile(aland + m * 1024, a[m]);
                    } else {
                        turn_tile(patch, lane, a[m]);
                    }
                    const bool f_ok = 32 * m + i < m_valid;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) a[m][q][c] = (f_ok && 4 * q + c < left) ? a[m][q][c] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) bsum[m] += (a[m][q][0] + a[m][q][1]) + (a[m][q][2] + a[m][q][3]);
                const wg_split as = wg_make(a[m]);
                if constexpr (DMA_B) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (more) issue_tile(A + gn * lda + m * 1024, aland + m * 1024);
                } else {
                    load_tile(A + gn * lda + m * 1024, more, lane, a[m]);
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) wg_contract(as, bs[n], acc[m][n]);
            }
        } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    bsum[m] += a[m][q][c];
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q][c], b[n][q][c], acc[m][n], 0, 0, 0);
                }
        }
    }
    __syncthreads();                                    // every wave is done with its patch / landing tiles
    // C layout: lane (col j = lane & 31, half kk) holds rows kmap(r, kk) of a tile in register r; four registers = one 16-byte
    // LDS slot [tile][r / 4][lane].  (Round 2 added element by element, one wave after the other: read - add - write round trips
    // of 4 bytes per lane, 37 us of a 128 x 128 launch regardless of the row count.)
    {
        float *reg = lds + (wave % NREG) * FRAG;
        for (int turn = 0; turn < WB / NREG; ++turn) {
            if (wave / NREG == turn) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 *slot = reinterpret_cast<f32x4 *>(reg + (m * NT + n) * 1024 + g * 256 + lane * 4);
                            f32x4 v = {acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
                            if (turn > 0) { const f32x4 o = *slot; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                            *slot = v;
                        }
            }
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float t = bsum[m] + __shfl_xor(bsum[m], 32, 64);
            if (kk == 0) lds[NREG * FRAG + wave * M32 + 32 * m + i] = t;
        }
        __syncthreads();
    }
    float *dst = part + (int64_t)blockIdx.x * (M32 * N32 + M32);
    for (int e = threadIdx.x; e < M32 * N32; e += 64 * WB) {
        const int row = e / N32, col = e - row * N32;
        const int m = row >> 5, rr = row & 31, n = col >> 5, j = col & 31;
        const int hk = (rr >> 2) & 1, r = (rr & 3) + 4 * (rr >> 3);               // rr = (r & 3) + 8 (r >> 2) + 4 hk
        const int f = (m * NT + n) * 1024 + (r >> 2) * 256 + (j + 32 * hk) * 4 + (r & 3);
        float v = lds[f];
#pragma unroll
        for (int q = 1; q < NREG; ++q) v += lds[q * FRAG + f];
        dst[e] = v;
    }
    for (int e = threadIdx.x; e < M32; e += 64 * WB) {
        float v = lds[NREG * FRAG + e];
#pragma unroll
        for (int q = 1; q < WB; ++q) v += lds[NREG * FRAG + q * M32 + e];
        dst[M32 * N32 + e] = v;
    }
}

// dW[rm(row)][cm(col)] += sum_p part[p][row * N32 + col]; db[rm(row)] += sum_p part[p][M32 * N32 + row], with
// rm / cm the optional row / column maps (negative = dropped).  256 threads = 32 adjacent elements x 8 partial
// groups (128 B coalesced reads, G / 8 independent loads per thread); one writer per element, no atomics.
__device__ __forceinline__ void wgrad_reduce_body(const float *__restrict__ part, int n_part, int M32, int N32,
                                                  float *__restrict__ dW, int64_t ldw, float *__restrict__ db,
                                                  int m_valid, int n_valid, const int32_t *__restrict__ row_map,
                                                  const int32_t *__restrict__ col_map, int slab) {
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int E = M32 * N32 + M32;
    const int e = blockIdx.x * 32 + el;
    const int slab_row = 128 * slab;                                     // 128-row slab of dW
    part += (int64_t)slab * n_part * E;
    m_valid -= slab_row;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < E) {
        const float *src = part + e;
        int p = grp;
        for (; p + 24 < n_part; p += 32) {
            s0 += src[(int64_t)p * E];
            s1 += src[(int64_t)(p + 8) * E];
            s2 += src[(int64_t)(p + 16) * E];
            s3 += src[(int64_t)(p + 24) * E];
        }
        for (; p < n_part; p += 8) s0 += src[(int64_t)p * E];
    }
    red[grp][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && e < E) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g][el];
        if (e < M32 * N32) {
            int row = e / N32, col = e % N32;
            if (row < m_valid && col < n_valid) {
                row += slab_row;
                if (row_map != nullptr) row = row_map[row];
                if (col_map != nullptr) col = col_map[col];
                if (row >= 0 && col >= 0) dW[(int64_t)row * ldw + col] += t;
            }
        } else if (db != nullptr && e - M32 * N32 < m_valid) {
            int row = e - M32 * N32 + slab_row;
            if (row_map != nullptr) row = row_map[row];
            if (row >= 0) db[row] += t;
        }
    }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int n_part, int M32, int N32,
                                                           float *__restrict__ dW, int64_t ldw, float *__restrict__ db,
                                                           int m_valid, int n_valid, const int32_t *__restrict__ row_map,
                                                           const int32_t *__restrict__ col_map) {
    wgrad_reduce_body(part, n_part, M32, N32, dW, ldw, db, m_valid, n_valid, row_map, col_map, blockIdx.y);      // blockIdx.y = slab
}
// a table of reductions in one launch (blockIdx.y = job): the 2 L reductions of a layer-major backward pass
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const float *__restrict__ part_base, float *__restrict__ out_base,
                                                                 const sx_reduce_job *__restrict__ jobs, int n_part) {
    const sx_reduce_job j = jobs[blockIdx.y];
    if ((int)blockIdx.x * 32 >= j.M32 * j.N32 + j.M32) return;          // (uniform per workgroup: before the barrier of the body)
    wgrad_reduce_body(part_base + j.part_off, n_part, j.M32, j.N32, out_base + j.dW_off, j.ldw, j.db_off >= 0 ? out_base + j.db_off : nullptr,
                      j.m_valid, j.n_valid, j.row_map, j.col_map, 0);
}

extern "C" int sx_wgrad_reduce_batch(const float *part_base, float *out_base, const sx_reduce_job *jobs, int32_t n_jobs,
                                     int32_t n_part, int32_t max_elems, void *stream) {
    SX_REQUIRE(part_base && out_base && jobs && n_jobs >= 1 && n_jobs <= 65535 && n_part >= 1 && max_elems >= 1,
               "sx_wgrad_reduce_batch: bad arguments");
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((max_elems + 31) / 32, n_jobs), dim3(256), 0, sx_stream(stream), part_base,
                       out_base, jobs, (int)n_part);
    SX_LAUNCH_CHECK();
    return SX_OK;
}

// Second stage on its own: sums `n_part` partial tiles ([M32 x N32 | M32] floats each, produced by sx_flow_bwd_run)
// into a parameter's rows / columns.
extern "C" int sx_wgrad_reduce(const float *part, int32_t n_part, int32_t M32, int32_t N32, float *dW, int64_t ldw,
                               float *db, int32_t m_valid, int32_t n_valid, const int32_t *row_map,
                               const int32_t *col_map, void *stream) {
    SX_REQUIRE(part && dW && n_part >= 1 && M32 >= 1 && M32 <= 128 && N32 >= 1, "sx_wgrad_reduce: bad arguments");
    const int E = M32 * N32 + M32;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((E + 31) / 32, 1), dim3(256), 0, sx_stream(stream), part, (int)n_part,
                       (int)M32, (int)N32, dW, ldw, db, (int)m_valid, (int)n_valid, row_map, col_map);
    SX_LAUNCH_CHECK();
    return SX_OK;
}

// The per-workgroup partial tiles go through a caller-owned scratch (sx_wgrad_scratch_floats): the library keeps no
// per-stream state and never allocates.
static void wgrad_shape(int32_t M, int32_t Nc, int32_t layout, int *slabs, int *mt, int *nt) {
    *slabs = (M + 127) / 128;
    *mt = *slabs > 1 ? 4 : (M + 31) / 32;
    *nt = (Nc + 31) / 32;
}
extern "C" size_t sx_wgrad_scratch_floats(int32_t M, int32_t Nc, int32_t layout) {
    if (M < 1 || Nc < 1) return 0;
    int slabs, mt, nt;
    wgrad_shape(M, Nc, layout, &slabs, &mt, &nt);
    return (size_t)256 * (32 * mt * 32 * nt + 32 * mt) * slabs;
}
extern "C" size_t sx_wgrad_layer_scratch_floats(int32_t c_tiles, int32_t h_tiles, int32_t t_tiles) {
    const size_t E2 = (size_t)32 * 2 * t_tiles * 32 * h_tiles + 32 * 2 * t_tiles, E1 = (size_t)32 * h_tiles * 32 * c_tiles + 32 * h_tiles;
    return 256 * (E2 + E1);
}

extern "C" int sx_wgrad(const float *A, int64_t lda, int32_t M, const float *B, int64_t ldb, int32_t Nc,
                        int64_t n_rows, int32_t layout, float *dW, int64_t ldw, float *db, const int32_t *row_map,
                        const int32_t *col_map, float *scratch, void *stream) {
    SX_REQUIRE(A && B && dW && scratch, "sx_wgrad: null pointer");
    SX_REQUIRE(layout == SX_WGRAD_ROW_MAJOR || layout == SX_WGRAD_ROW_GROUPS || layout == SX_WGRAD_ROW_GROUPS_F16X3,
               "sx_wgrad: unknown layout %d", layout);
    const bool f16 = layout == SX_WGRAD_ROW_GROUPS_F16X3;
    const bool groups = layout == SX_WGRAD_ROW_GROUPS || f16;
    SX_REQUIRE(M >= 1 && Nc >= 1 && Nc <= 128 && n_rows >= 0 && M <= (groups ? 128 : 2048),
               "sx_wgrad: Nc must be in 1..128, M in 1..128 (row groups) or 1..2048 (row-major)");
    SX_REQUIRE(!groups || (((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && lda % 4 == 0 && ldb % 4 == 0),
               "sx_wgrad: A, B must be 16-byte aligned row groups (ld a multiple of 4 floats)");
    if (n_rows == 0) return SX_OK;
    // row-major operands whose rows are 16-byte aligned (torch tensors with a multiple of 4 features) load 16 B per lane
    const bool vec_rows = !groups && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && lda % 4 == 0 && ldb % 4 == 0;
    const int slabs = (M + 127) / 128;
    const int mt = slabs > 1 ? 4 : (M + 31) / 32, nt = (Nc + 31) / 32;
    hipStream_t st = sx_stream(stream);
    const int E = 32 * mt * 32 * nt + 32 * mt;
    // waves per workgroup by register budget (accumulators = 16 * MT * NT VGPRs); 256 workgroups fill the CUs
#define SX_WG_L(MT_, NT_, GR_)                                                                                      \
    {                                                                                                              \
        constexpr int WB = GR_ == 3 ? ((MT_ * NT_ <= 4 && MT_ + NT_ <= 4) ? 8 : 4)                                   \
                                    : (MT_ * NT_ <= 2 ? 16 : (MT_ * NT_ <= 8 ? 8 : 4));                            \
        constexpr int GMAX = 256;                                                                                  \
        int64_t g = (n_rows + 32 * WB - 1) / (32 * WB);                                                            \
        if (g > GMAX) g = GMAX;                                                                                    \
        float *part = scratch;                                                                                     \
        hipLaunchKernelGGL((wgrad_kernel<MT_, NT_, WB, GR_>), dim3((int)g, slabs), dim3(64 * WB), 0, st, A, lda,   \
                           B, ldb, n_rows, part, M, Nc);                                                           \
        SX_LAUNCH_CHECK();                                                                                         \
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((E + 31) / 32, slabs), dim3(256), 0, st, part, (int)g,        \
                           32 * MT_, 32 * NT_, dW, ldw, db, M, Nc, row_map, col_map);                              \
        SX_LAUNCH_CHECK();                                                                                         \
        return SX_OK;                                                                                              \
    }
#define SX_WG(MT_, NT_)                                                                                            \
    if (mt == MT_ && nt == NT_) {                                                                                  \
        if (f16) SX_WG_L(MT_, NT_, 3) else if (groups) SX_WG_L(MT_, NT_, 1) else if (vec_rows) SX_WG_L(MT_, NT_, 2) else SX_WG_L(MT_, NT_, 0) \
    }
    SX_WG(1, 1) SX_WG(1, 2) SX_WG(2, 1) SX_WG(2, 2) SX_WG(2, 4) SX_WG(4, 2) SX_WG(4, 1) SX_WG(1, 4) SX_WG(4, 4)
    SX_WG(3, 1) SX_WG(3, 2) SX_WG(3, 3) SX_WG(3, 4) SX_WG(1, 3) SX_WG(2, 3) SX_WG(4, 3)
#undef SX_WG
#undef SX_WG_L
    sx_set_error("sx_wgrad: unsupported tile shape %d x %d", mt, nt);
    return SX_E_UNSUPPORTED;
}

// Both weight gradients of one coupling layer of the backward flow program in ONE pass over its per-row factors:
// a 32-row group holds [z (CT tiles) | tanh h (HT) | dL/dh_pre (HT) | dL/d(log_scale, shift) (2 TT)] contiguously, so a
// wave streams the whole group (28 KB for cfg 2) instead of two strided subsets in two launches:
//   dW2 (2 TT x HT tiles) += dparams^T tanh_h,  db2 += sum dparams;   dW1 (HT x CT tiles) += dh_pre^T z,  db1 += sum dh_pre.
template <int CT, int HT, int TT, int WB>
__global__ __launch_bounds__(64 * WB) void wgrad_layer_kernel(const float *__restrict__ side, int64_t ld, int64_t n_rows,
                                                              float *__restrict__ part2, float *__restrict__ part1) {
    constexpr int PT = 2 * TT;
    constexpr int E2 = 32 * PT * 32 * HT + 32 * PT, E1 = 32 * HT * 32 * CT + 32 * HT;
    constexpr int RED = E2 + E1;
    __shared__ __attribute__((aligned(16))) float lds[(RED > WB * PATCH ? RED : WB * PATCH)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    float *patch = lds + wave * PATCH;
    f32x16 acc2[PT][HT], acc1[HT][CT];
    float bs2[PT], bs1[HT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        bs2[p] = 0.f;
#pragma unroll
        for (int n = 0; n < HT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[p][n][r] = 0.f;
    }
#pragma unroll
    for (int m = 0; m < HT; ++m) {
        bs1[m] = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[m][c][r] = 0.f;
    }
    const int64_t n_waves = (int64_t)gridDim.x * WB, w_id = (int64_t)blockIdx.x * WB + wave;
    const int64_t n_groups = (n_rows + 31) >> 5;
    for (int64_t g = w_id; g < n_groups; g += n_waves) {
        const float *base = side + g * ld;
        f32x4 z[CT][4], hh[HT][4], dh[HT][4], dp[PT][4];
#pragma unroll
        for (int c = 0; c < CT; ++c) load_tile(base + 1024 * c, true, lane, z[c]);
#pragma unroll
        for (int m = 0; m < HT; ++m) load_tile(base + 1024 * (CT + m), true, lane, hh[m]);
#pragma unroll
        for (int m = 0; m < HT; ++m) load_tile(base + 1024 * (CT + HT + m), true, lane, dh[m]);
#pragma unroll
        for (int p = 0; p < PT; ++p) load_tile(base + 1024 * (CT + 2 * HT + p), true, lane, dp[p]);
        const int64_t rem = n_rows - 32 * g;
        const int left = (rem < 32 ? (int)rem : 32) - 16 * kk;          // valid rows among this lane's 16
        auto turn = [&](f32x4 (&v)[4]) {
            turn_tile(patch, lane, v);
            if (left < 16) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[q][c] = (4 * q + c < left) ? v[q][c] : 0.f;
            }
        };
#pragma unroll
        for (int n = 0; n < HT; ++n) turn(hh[n]);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            turn(dp[p]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    bs2[p] += dp[p][q][c];
#pragma unroll
                    for (int n = 0; n < HT; ++n)
                        acc2[p][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[p][q][c], hh[n][q][c], acc2[p][n], 0, 0, 0);
                }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) turn(z[c]);
#pragma unroll
        for (int m = 0; m < HT; ++m) {
            turn(dh[m]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    bs1[m] += dh[m][q][c4];
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        acc1[m][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(dh[m][q][c4], z[c][q][c4], acc1[m][c], 0, 0, 0);
                }
        }
    }
    __syncthreads();                                    // every wave is done with its patch
    float *red = lds;
    constexpr int N2 = 32 * HT, N1 = 32 * CT;
    for (int w = 0; w < WB; ++w) {
        if (wave == w) {
#pragma unroll
            for (int p = 0; p < PT; ++p) {
#pragma unroll
                for (int n = 0; n < HT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int e = (32 * p + (r & 3) + 8 * (r >> 2) + 4 * kk) * N2 + 32 * n + i;
                        red[e] = (w == 0 ? 0.f : red[e]) + acc2[p][n][r];
                    }
                const float t = bs2[p] + __shfl_xor(bs2[p], 32, 64);
                const int e = 32 * PT * N2 + 32 * p + i;
                if (kk == 0) red[e] = (w == 0 ? 0.f : red[e]) + t;
            }
#pragma unroll
            for (int m = 0; m < HT; ++m) {
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int e = E2 + (32 * m + (r & 3) + 8 * (r >> 2) + 4 * kk) * N1 + 32 * c + i;
                        red[e] = (w == 0 ? 0.f : red[e]) + acc1[m][c][r];
                    }
                const float t = bs1[m] + __shfl_xor(bs1[m], 32, 64);
                const int e = E2 + 32 * HT * N1 + 32 * m + i;
                if (kk == 0) red[e] = (w == 0 ? 0.f : red[e]) + t;
            }
        }
        __syncthreads();
    }
    float *d2 = part2 + (int64_t)blockIdx.x * E2, *d1 = part1 + (int64_t)blockIdx.x * E1;
    for (int e = threadIdx.x; e < E2; e += 64 * WB) d2[e] = red[e];
    for (int e = threadIdx.x; e < E1; e += 64 * WB) d1[e] = red[E2 + e];
}

extern "C" int sx_wgrad_layer(const float *side, int64_t ld, int64_t n_rows, int32_t c_tiles, int32_t h_tiles,
                              int32_t t_tiles, int32_t hidden, float *dW2, int64_t ldw2, float *db2,
                              const int32_t *row_map2, float *dW1, int64_t ldw1, float *db1, const int32_t *col_map1,
                              float *scratch, void *stream) {
    SX_REQUIRE(side && dW2 && dW1 && scratch, "sx_wgrad_layer: null pointer");
    SX_REQUIRE(((uintptr_t)side & 15) == 0 && ld % 4 == 0 && ld >= 32 * (c_tiles + 2 * h_tiles + 2 * t_tiles) * 32,
               "sx_wgrad_layer: side must be 16-byte aligned row groups of at least the layer's features");
    SX_REQUIRE(hidden >= 1 && hidden <= 32 * h_tiles && n_rows >= 0, "sx_wgrad_layer: bad hidden width");
    if (n_rows == 0) return SX_OK;
    hipStream_t st = sx_stream(stream);
    constexpr int WB = 8, GMAX = 256;
    int64_t g = (n_rows + 32 * WB - 1) / (32 * WB);
    if (g > GMAX) g = GMAX;
#define SX_WL(CT_, HT_, TT_)                                                                                       \
    if (c_tiles == CT_ && h_tiles == HT_ && t_tiles == TT_) {                                                      \
        constexpr int E2 = 32 * 2 * TT_ * 32 * HT_ + 32 * 2 * TT_, E1 = 32 * HT_ * 32 * CT_ + 32 * HT_;             \
        float *part = scratch;                                                                                     \
        float *part1 = part + (size_t)GMAX * E2;                                                                   \
        hipLaunchKernelGGL((wgrad_layer_kernel<CT_, HT_, TT_, WB>), dim3((int)g), dim3(64 * WB), 0, st, side, ld,  \
                           n_rows, part, part1);                                                                   \
        SX_LAUNCH_CHECK();                                                                                         \
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((E2 + 31) / 32, 1), dim3(256), 0, st, part, (int)g,           \
                           64 * TT_, 32 * HT_, dW2, ldw2, db2, 64 * TT_, hidden, row_map2, (const int32_t *)nullptr); \
        SX_LAUNCH_CHECK();                                                                                         \
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((E1 + 31) / 32, 1), dim3(256), 0, st, part1, (int)g,          \
                           32 * HT_, 32 * CT_, dW1, ldw1, db1, hidden, 32 * CT_, (const int32_t *)nullptr, col_map1); \
        SX_LAUNCH_CHECK();                                                                                         \
        return SX_OK;                                                                                              \
    }
    SX_WL(1, 1, 1) SX_WL(1, 2, 1)
#undef SX_WL
    sx_set_error("sx_wgrad_layer: unsupported tile shape (c %d, h %d, t %d)", c_tiles, h_tiles, t_tiles);
    return SX_E_UNSUPPORTED;
}

// Column sums of a row-major [n_rows, M] matrix (the bias gradient of a wide Linear layer: out[j] += sum_n A[n][j]).
// Two stages like sx_wgrad: 256 row slices x column blocks are summed in registers -- a workgroup is Q column lanes
// (16 B each when the rows are 16-byte aligned: a wave reads 1 KB of a row) x 256 / Q row lanes, folded through LDS --
// then one writer per column.  The result does not depend on scheduling and the launch pair replays from a HIP graph
// (torch's multi-block column sum did not on this build).
template <bool VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ A, int64_t lda, int64_t n_rows, int M,
                                                     int Q, float *__restrict__ part) {
    constexpr int W = VEC ? 4 : 1;
    __shared__ float red[256 * W];
    const int R = 256 / Q, cl = threadIdx.x % Q, rl = threadIdx.x / Q;
    const int c = (blockIdx.y * Q + cl) * W;
    float s[4][W];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int w = 0; w < W; ++w) s[u][w] = 0.f;
    if (c < M) {
        const int64_t step = (int64_t)gridDim.x * R;
        int64_t r = (int64_t)blockIdx.x * R + rl;
        for (; r + 3 * step < n_rows; r += 4 * step) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float *p = A + (r + u * step) * lda + c;
                if constexpr (VEC) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(p);
                    s[u][0] += v.x; s[u][1] += v.y; s[u][2] += v.z; s[u][3] += v.w;
                } else {
                    s[u][0] += *p;
                }
            }
        }
        for (; r < n_rows; r += step) {
            const float *p = A + r * lda + c;
#pragma unroll
            for (int w = 0; w < W; ++w) s[0][w] += p[w];
        }
    }
#pragma unroll
    for (int w = 0; w < W; ++w) red[threadIdx.x * W + w] = (s[0][w] + s[1][w]) + (s[2][w] + s[3][w]);
    __syncthreads();
    if (rl == 0 && c < M) {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            float t = 0.f;
            for (int k = 0; k < R; ++k) t += red[(k * Q + cl) * W + w];
            part[(int64_t)blockIdx.x * M + c + w] = t;
        }
    }
}
// out[c] += sum_p part[p][c]: 256 threads = 32 columns x 8 partial groups
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float *__restrict__ part, int n_part, int M,
                                                            float *__restrict__ out) {
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < M) {
        int p = grp;
        for (; p + 24 < n_part; p += 32) {
            s0 += part[(int64_t)p * M + c]; s1 += part[(int64_t)(p + 8) * M + c];
            s2 += part[(int64_t)(p + 16) * M + c]; s3 += part[(int64_t)(p + 24) * M + c];
        }
        for (; p < n_part; p += 8) s0 += part[(int64_t)p * M + c];
    }
    red[grp][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && c < M) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g][el];
        out[c] += t;
    }
}

extern "C" int sx_colsum(const float *A, int64_t lda, int64_t n_rows, int32_t M, float *out, float *scratch, void *stream) {
    SX_REQUIRE(A && out && scratch, "sx_colsum: null pointer");
    SX_REQUIRE(M >= 1 && n_rows >= 0 && lda >= M, "sx_colsum: bad shape");
    if (n_rows == 0) return SX_OK;
    hipStream_t st = sx_stream(stream);
    const bool vec = ((uintptr_t)A & 15) == 0 && lda % 4 == 0 && M % 4 == 0;
    const int lanes = vec ? (M + 3) / 4 : M;                       // column lanes needed
    int Q = 256;
    while (Q > 1 && Q / 2 >= lanes) Q /= 2;                        // power of two: 256 / Q row lanes per workgroup
    const int gy = (lanes + Q - 1) / Q;
    int gx = 1024 / gy;
    gx = gx < 64 ? 64 : (gx > 256 ? 256 : gx);
    const int64_t slices = (n_rows + (256 / Q) - 1) / (256 / Q);
    if (gx > slices) gx = (int)slices;
    float *part = scratch;
    if (vec) hipLaunchKernelGGL(colsum_kernel<true>, dim3(gx, gy), dim3(256), 0, st, A, lda, n_rows, (int)M, Q, part);
    else hipLaunchKernelGGL(colsum_kernel<false>, dim3(gx, gy), dim3(256), 0, st, A, lda, n_rows, (int)M, Q, part);
    SX_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((M + 31) / 32), dim3(256), 0, st, part, gx, (int)M, out);
    SX_LAUNCH_CHECK();
    return SX_OK;
}
