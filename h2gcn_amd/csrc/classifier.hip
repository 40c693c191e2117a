// classifier.hip -- the classifier side of an H2GCN training step as hand-written gfx950 kernels on the fp32 matrix cores:
//
//     Z = (X .* M / keep) @ W + b          (forward)          X: [N, K] concat buffer, K = 7 * hidden (448), W: [K, C], C <= 64
//     dX = (G @ W^T) .* M / keep           (backward, data)   G: [N, C]
//     dW = (X .* M / keep)^T @ G           (backward, weights)
//
// Reference: keras `Dropout(rate)` followed by the output `Dense` -- `D0.5-MO` of the network-setup DSL (reference
// h2gcn/models/H2GCN.py:235-257 builds the two layers, :308-325 calls them in order; SURVEY.md 8(f) rank 2/3: "the final
// dense classifier", "training loop on device").  With stock kernels this is five passes over the 4.3 GB buffer of the
// products shape (dropout forward, skinny GEMM, two backward GEMMs, dropout backward: ~15 of a 68 ms step, the GEMMs at
// 1.1-1.9 TB/s because C = 47 outputs starve a general GEMM tile).  Here every pass streams X (or writes dX) ONCE:
//   * the dropout mask is a COUNTER-BASED function of (seed, step, row, column) -- one keyed round of a 32-bit avalanche hash
//     per group of four elements -- recomputed wherever it is needed instead of stored or applied in a pass of its own;
//   * the products run on v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain), C padded to 16-column tiles (47 -> 48),
//     W staged through LDS in 64-row chunks whose layout makes the B-fragment reads conflict-free;
//   * dW is accumulated per workgroup in registers over a contiguous row range and reduced in fixed order (deterministic).
// Where the time goes (products shape, measured by knocking parts out -- tools/classifier_kernels.py): the fp32
// matrix pipe alone needs 0.84 ms per pass at the clock the chip holds under this load (50.4 M MFMAs x 32 cycles over 1024
// SIMDs at ~1.9-2.1 GHz; SQ_VALU_MFMA_BUSY_CYCLES confirms the count), the X stream alone 0.86 ms (5 TB/s); the mask VALU work
// shares the issue port with the MFMAs (+0.1-0.3 ms); the two do not overlap perfectly at 3-4 waves per SIMD:
// 1.44 / 1.60 / 1.33 ms (forward / dX / dW) = 0.58 / 0.52 / 0.63 of the matrix-pipe floor, 3.0 / 2.7 / 3.2 TB/s of X.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "capi_internal.h"
#include "h2gcn_hip.h"

namespace {

using h2gcn::fail;
using f32x4 = float __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kThreads = 256;   // 4 waves
constexpr int kRowsPerWave = 32;
constexpr int kRowsPerGroup = 4 * kRowsPerWave;
#ifndef H2GCN_CLS_KC
#define H2GCN_CLS_KC 64
#endif
constexpr int kKC = H2GCN_CLS_KC;   // rows of W per LDS chunk (forward) / output columns per chunk (dX): 64 keeps 5 waves per SIMD resident
#ifndef H2GCN_CLS_DW_SUB
#define H2GCN_CLS_DW_SUB 1
#endif
#ifndef H2GCN_CLS_DW_WG
#define H2GCN_CLS_DW_WG 3
#endif
constexpr int kDwSub = H2GCN_CLS_DW_SUB;       // 4-row reduction sub-steps fetched together (measured: 1 beats 2 and 4, 1.44 vs 2.0 / 1.8 ms)
constexpr int kDwRowsPerStep = 4 * kDwSub;

__host__ __device__ constexpr int lds_stride(int nt) { return nt == 1 ? 16 : (nt <= 3 ? 48 : 80); }  // floats; stride % 32 == 16
constexpr int kDxStride = kKC + 16;                                                                      // 80 (144 for 128-column chunks): % 32 == 16

// ---- the mask generator (documented in include/h2gcn_hip.h; the test-side restatement reproduces it bit for bit) -------
// One hash per aligned GROUP of four columns of a row (gid = row * ceil(K/4) + col/4, 64-bit): a keyed avalanche round whose
// second key is injected between its two multiplies.  keep_prob a multiple of 1/256 (0.5, 0.75, 0.9375, ...): the four BYTES
// of the word are the four elements' fields -- one round, two v_mul_lo_u32 per four elements; any other keep_prob: a second
// round yields a second word and the fields are 16 bits wide.  (History: hashing every element separately made the VALU work
// exceed the MFMA work; three rounds per group and a multiply by 1/keep per element still cost 0.29 of 1.15 ms -- the scale
// is now applied to the finished sums instead.)
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    return h ^ (h >> 16);
}
struct MaskKey {
    uint32_t k0, k1, thr;   // thr: threshold of a field (8-bit fields: keep_prob * 256, 16-bit fields: keep_prob * 65536)
    int bytes;              // 1: 8-bit fields
    int on;
    int64_t groups_per_row;
};
__device__ __forceinline__ MaskKey make_key(uint64_t seed, const int64_t* step_dev, uint32_t thr16, int on, int K) {
    const uint64_t step = (on && step_dev) ? (uint64_t)*step_dev : 0;
    const uint32_t k0 = mix32((uint32_t)seed ^ mix32((uint32_t)step + 0x9E3779B9u));
    const uint32_t k1 = mix32((uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32) ^ k0 ^ 0x85EBCA6Bu);
    const int bytes = (thr16 & 0xFFu) == 0;
    return MaskKey{k0, k1, bytes ? thr16 >> 8 : thr16, bytes, on, (int64_t)((K + 3) / 4)};
}
// first hash word of the group holding (row, col)
__device__ __forceinline__ uint32_t group_word(const MaskKey& m, int64_t row, int col) {
    const uint64_t gid = (uint64_t)(row * m.groups_per_row + (col >> 2));
    const uint32_t hi = (uint32_t)(gid >> 32);
    uint32_t h = (uint32_t)gid ^ ((hi << 16) | (hi >> 16)) ^ m.k0;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= m.k1;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    return h ^ (h >> 16);
}

// ---- weight packing ------------------------------------------------------------------------------------------------------
// forward image: row r of chunk c <-> k = 128 c + 16 g + 4 kq + j with r % 128 = (g*4 + j)*4 + kq: the four k's one
// v_mfma_f32_16x16x4 step consumes (kq = lane >> 4) sit in adjacent LDS rows; [Kpad][stride], zero beyond K / C
__global__ void pack_w_fwd_kernel(const float* __restrict__ w, int K, int C, int Kpad, int stride, float* __restrict__ out) {
    const int total = Kpad * stride;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int r = t / stride, col = t - r * stride;
        const int c = r / kKC, rr = r % kKC;
        const int kq = rr & 3, j = (rr >> 2) & 3, g = rr >> 4;
        const int k = c * kKC + 16 * g + 4 * kq + j;
        out[t] = (k < K && col < C) ? w[(int64_t)k * C + col] : 0.f;
    }
}
// backward-data image: chunk v of 128 output columns, row cc = 4 s + kq (the c index), column kk: W[128 v + kk][cc];
// [n_chunks][Cpad][kDxStride], zero beyond K / C
__global__ void pack_w_dx_kernel(const float* __restrict__ w, int K, int C, int Cpad, int n_chunks, float* __restrict__ out) {
    const int total = n_chunks * Cpad * kDxStride;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int v = t / (Cpad * kDxStride), rem = t - v * (Cpad * kDxStride);
        const int cc = rem / kDxStride, kk = rem - cc * kDxStride;
        const int k = v * kKC + kk;
        out[t] = (kk < kKC && k < K && cc < C) ? w[(int64_t)k * C + cc] : 0.f;
    }
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// float4 of row `row` at columns k .. k+3 (k % 4 == 0; zero beyond the matrix).  Branch-light and with compile-time register
// indices only: a runtime-indexed tail loop would send the caller's fragment array to scratch.
__device__ __forceinline__ f4u load_row4(const float* __restrict__ X, int64_t ldx, int64_t n_rows, int K, int64_t row, int k) {
    f4u v = {0.f, 0.f, 0.f, 0.f};
    if (row < n_rows && k < K) {
        const float* p = X + row * ldx + k;
        if (k + 4 <= K) {
            v = *reinterpret_cast<const f4u*>(p);
        } else {   // the last, partial group of a row whose width is not a multiple of 4
            v[0] = p[0];
            if (k + 1 < K) v[1] = p[1];
            if (k + 2 < K) v[2] = p[2];
        }
    }
    return v;
}
// dropout of the group (row, k .. k+3): dropped elements zeroed.  The 1 / keep_prob scale of the survivors is applied by the
// callers to the finished sums (forward, dW) or at the store (dX) -- the same value up to one rounding, bit-identical for
// keep_prob = 0.5
// MASK: 0 = no dropout (evaluation), 1 = 8-bit fields, 2 = 16-bit fields -- a compile-time mode: a run-time branch inside the
// unrolled MFMA bodies costs the scheduler its interleaving (measured on the dW kernel: 1.45 -> 1.60 ms)
template <int MASK>
__device__ __forceinline__ f4u apply_mask(f4u v, const MaskKey& mk, int64_t row, int k) {
    if constexpr (MASK == 1) {
        const uint32_t w0 = group_word(mk, row, k);
        v[0] = (w0 & 0xFFu) < mk.thr ? v[0] : 0.f;
        v[1] = ((w0 >> 8) & 0xFFu) < mk.thr ? v[1] : 0.f;
        v[2] = ((w0 >> 16) & 0xFFu) < mk.thr ? v[2] : 0.f;
        v[3] = (w0 >> 24) < mk.thr ? v[3] : 0.f;
    } else if constexpr (MASK == 2) {
        const uint32_t w0 = group_word(mk, row, k);
        const uint32_t w1 = mix32(w0 ^ 0x85EBCA6Bu);
        v[0] = (w0 & 0xFFFFu) < mk.thr ? v[0] : 0.f;
        v[1] = (w0 >> 16) < mk.thr ? v[1] : 0.f;
        v[2] = (w1 & 0xFFFFu) < mk.thr ? v[2] : 0.f;
        v[3] = (w1 >> 16) < mk.thr ? v[3] : 0.f;
    }
    return v;
}


// ---- small operands (the reference's own datasets: Cora 2 708 rows, citeseer 3 327, syn-products 10 000) -----------------
// At a few thousand rows the kernels above are one latency chain each -- 7 K-chunks x (fragment fetch, LDS fill, barrier) in
// a handful of workgroups, plus a weight-packing launch in front: ~18 us per pass on Cora, where a whole training epoch is
// 0.3 ms (profiles/r04_cora_epoch_kernels.txt: 30 % of it).  Below ~12 k rows (and C <= 16) three plain VALU kernels
// serve the forward and dX with the same contract -- same mask generator, same placement of the 1 / keep scale, fp32 FMAs,
// deterministic -- and the shortest chain each: no packing launch, W^T staged once per workgroup in LDS (12.5 KB on Cora).
// (dW stays on the matrix-core kernel + its fixed-order reduction: a column-slab walk over all rows in plain code was tried and
// is slower, 26.7 vs 18.2 us on Cora.)
constexpr int kSmallCP = 16;   // classes the small kernels keep in registers

// W^T staged in LDS as Wt[c][Kp] (Kp = K rounded up to 4, zero-padded): a lane / thread that owns the column group k .. k+3
// reads its four weights of class c with ONE 16-byte LDS read, consecutive lanes consecutive addresses (no bank conflicts);
// straight from memory the same access is a 112-byte-stride gather over W's rows
__device__ __forceinline__ void stage_wt(const float* __restrict__ W, int K, int C, int Kp, float* __restrict__ wt) {
    for (int t = threadIdx.x; t < Kp * C; t += kThreads) {
        const int k = t / C, c = t - k * C;                 // coalesced read of W[k][c]
        wt[c * Kp + k] = k < K ? W[t] : 0.f;
    }
    __syncthreads();
}

// forward: one wave per row (kSmallRowsPerWave rows in turn); lane l owns the column groups l, l + 64, ...; 16 class partials
// per lane, folded across the wave
constexpr int kSmallRowsPerWave = 1;   // (2 / 4 rows per wave: 11-13 / 18-19 us per pass on Cora instead of 8-10: the launch is one wave's chain)
constexpr int kSmallMaxK = 512;      // forward: two column groups per lane in registers (wider operands: matrix-core kernels)
template <int MASK>
__global__ __launch_bounds__(kThreads) void small_fwd_kernel(const float* __restrict__ X, int64_t ldx, int64_t n_rows, int K,
                                                             const float* __restrict__ W, const float* __restrict__ bias, int C,
                                                             float inv_keep, uint32_t thr, int mask_on, uint64_t seed,
                                                             const int64_t* step_dev, float* __restrict__ Y, int64_t ldy) {
    extern __shared__ float wt[];   // [C][Kp]
    const int Kp = (K + 3) & ~3;
    const int lane = threadIdx.x & 63;
    const MaskKey mk = make_key(seed, step_dev, thr, mask_on, K);
    const int64_t row0 = ((int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * kSmallRowsPerWave;
    // this wave's fragments of X first (K <= 512: the groups 4*lane and 256 + 4*lane of each row): their latency runs under the
    // staging of W
    f4u a[kSmallRowsPerWave][2];
#pragma unroll
    for (int r = 0; r < kSmallRowsPerWave; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) a[r][h] = load_row4(X, ldx, n_rows, K, row0 + r, 256 * h + 4 * lane);
    stage_wt(W, K, C, Kp, wt);
#pragma unroll
    for (int r = 0; r < kSmallRowsPerWave; ++r) {
        const int64_t row = row0 + r;
        float acc[kSmallCP];
#pragma unroll
        for (int c = 0; c < kSmallCP; ++c) acc[c] = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 256 * h + 4 * lane;
            if (k < K) {
                const f4u x = apply_mask<MASK>(a[r][h], mk, row, k);
#pragma unroll
                for (int c = 0; c < kSmallCP; ++c) {
                    if (c < C) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(wt + c * Kp + k);
                        acc[c] = fmaf(x[0], w[0], acc[c]);
                        acc[c] = fmaf(x[1], w[1], acc[c]);
                        acc[c] = fmaf(x[2], w[2], acc[c]);
                        acc[c] = fmaf(x[3], w[3], acc[c]);
                    }
                }
            }
        }
        float mine = 0.f;
#pragma unroll
        for (int c = 0; c < kSmallCP; ++c) {
            if (c < C) {
                float v = acc[c];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);   // fixed butterfly: deterministic
                mine = lane == c ? v : mine;
            }
        }
        if (lane < C && row < n_rows) Y[row * ldy + lane] = mine * inv_keep + (bias ? bias[lane] : 0.f);
    }
}

// backward, data: one thread per (row, group of four columns)
template <int MASK>
__global__ __launch_bounds__(kThreads) void small_dx_kernel(const float* __restrict__ G, int64_t ldg, int64_t n_rows, int K, int C,
                                                            const float* __restrict__ W, float inv_keep, uint32_t thr, int mask_on,
                                                            uint64_t seed, const int64_t* step_dev, float* __restrict__ dX, int64_t lddx) {
    extern __shared__ float wt[];   // [C][Kp]
    const int Kp = (K + 3) & ~3;
    const int groups = Kp / 4;
    stage_wt(W, K, C, Kp, wt);
    const MaskKey mk = make_key(seed, step_dev, thr, mask_on, K);
    const int64_t total = n_rows * groups;
    for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += (int64_t)gridDim.x * kThreads) {
        const int64_t row = t / groups;
        const int k = (int)(t - row * groups) * 4;
        f32x4 out = {0.f, 0.f, 0.f, 0.f};
        const float* g = G + row * ldg;
#pragma unroll
        for (int c = 0; c < kSmallCP; ++c) {
            if (c < C) {
                const float gc = g[c];
                const f32x4 w = *reinterpret_cast<const f32x4*>(wt + c * Kp + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) out[j] = fmaf(gc, w[j], out[j]);
            }
        }
        f4u keep = {inv_keep, inv_keep, inv_keep, inv_keep};
        keep = apply_mask<MASK>(keep, mk, row, k);        // inv_keep where kept, 0 where dropped
        float* dst = dX + row * lddx + k;
        if (k + 4 <= K) {
            f4u o = {out[0] * keep[0], out[1] * keep[1], out[2] * keep[2], out[3] * keep[3]};
            *reinterpret_cast<f4u*>(dst) = o;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k + j < K) dst[j] = out[j] * keep[j];
        }
    }
}

int64_t g_small_rows = 12288;   // operands of at most this many rows (and C <= kSmallCP, K <= kSmallMaxK) take the small kernels; 0 = never
                                // (measured crossover at K = 448, C = 7: 8 192 rows 16 / 14 us small vs 19 / 19 matrix-core, 16 384 rows 27 / 22 vs
                                //  20 / 19, 32 768 rows 51 / 47 vs 23 / 22 -- profiles/r04_cora_epoch_kernels.txt)
size_t small_lds_bytes(int K, int C) { return (size_t)((K + 3) & ~3) * C * 4; }
bool small_operand(int64_t n_rows, int K, int C) { return n_rows <= g_small_rows && C <= kSmallCP && K <= kSmallMaxK; }

// ---- forward -------------------------------------------------------------------------------------------------------------
// One workgroup = 4 waves x 32 rows; a wave owns 2 row tiles x NT column tiles of 16x16 accumulators.  W chunks of 128 rows
// are double-buffered in LDS (one barrier per chunk).
template <int NT, int MASK>
__global__ __launch_bounds__(kThreads) void dropout_dense_fwd_kernel(const float* __restrict__ X, int64_t ldx, int64_t n_rows, int K,
                                                                     const float* __restrict__ Wp, int Kpad, const float* __restrict__ bias,
                                                                     int C, float inv_keep, uint32_t thr, int mask_on, uint64_t seed,
                                                                     const int64_t* step_dev, float* __restrict__ Y, int64_t ldy) {
    constexpr int S = lds_stride(NT);
    extern __shared__ float lds[];   // 2 x kKC x S
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const MaskKey mk = make_key(seed, step_dev, thr, mask_on, K);
    const int n_chunks = Kpad / kKC;
    const int64_t n_groups = (n_rows + kRowsPerGroup - 1) / kRowsPerGroup;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t row_base = grp * kRowsPerGroup + (int64_t)wave * kRowsPerWave;
        f32x4 acc[2][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < n_chunks; ++c) {
            float* buf = lds + (c & 1) * (kKC * S);
            // the wave's own X fragments first: their latency runs under the LDS fill and the barrier.  (Fetching them a whole
            // chunk ahead was measured too: the second fragment buffer costs a wave per SIMD, 1.44 -> 1.56 ms; 32-row chunks with
            // the look-ahead 1.53 ms.)
            f4u a[2][kKC / 16];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < kKC / 16; ++g) a[t][g] = load_row4(X, ldx, n_rows, K, row_base + 16 * t + i, c * kKC + 16 * g + 4 * kq);
            // this buffer was last read two chunks ago; every wave has passed the barrier of the previous chunk since
            {
                const f32x4* src = reinterpret_cast<const f32x4*>(Wp + (int64_t)c * kKC * S);
                f32x4* dst = reinterpret_cast<f32x4*>(buf);
                for (int t = threadIdx.x; t < kKC * S / 4; t += kThreads) dst[t] = src[t];
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < kKC / 16; ++g) {
                const f4u a0 = apply_mask<MASK>(a[0][g], mk, row_base + i, c * kKC + 16 * g + 4 * kq);
                const f4u a1 = apply_mask<MASK>(a[1][g], mk, row_base + 16 + i, c * kKC + 16 * g + 4 * kq);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* brow = buf + ((g * 4 + j) * 4 + kq) * S + i;
#pragma unroll
                    for (int u = 0; u < NT; ++u) {
                        const float b = brow[16 * u];
                        acc[0][u] = mfma16(a0[j], b, acc[0][u]);
                        acc[1][u] = mfma16(a1[j], b, acc[1][u]);
                    }
                }
            }
        }
        __syncthreads();   // the next group's first fill must not overtake this group's last reads of buffer 0
        // C/D fragment: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int col = 16 * u + i;
                if (col >= C) continue;
                const float bv = bias ? bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row_base + 16 * t + 4 * kq + r;
                    if (row < n_rows) Y[row * ldy + col] = acc[t][u][r] * inv_keep + bv;
                }
            }
    }
}

// ---- backward, data: dX = (G W^T) .* M / keep ---------------------------------------------------------------------------------
template <int NT, int MASK>
__global__ __launch_bounds__(kThreads) void dropout_dense_dx_kernel(const float* __restrict__ G, int64_t ldg, int64_t n_rows, int K, int C,
                                                                    const float* __restrict__ Wtp, int n_chunks, float inv_keep,
                                                                    uint32_t thr, int mask_on, uint64_t seed, const int64_t* step_dev,
                                                                    float* __restrict__ dX, int64_t lddx) {
    constexpr int CP = NT * 16, NS = NT * 4;
    extern __shared__ float lds[];   // 2 x CP x kDxStride
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const MaskKey mk = make_key(seed, step_dev, thr, mask_on, K);
    const int64_t n_groups = (n_rows + kRowsPerGroup - 1) / kRowsPerGroup;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t row_base = grp * kRowsPerGroup + (int64_t)wave * kRowsPerWave;
        float ga[2][NS];   // A fragments: G[row_base + 16 t + i][4 s + kq]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t row = row_base + 16 * t + i;
                const int cc = 4 * s + kq;
                ga[t][s] = (row < n_rows && cc < C) ? G[row * ldg + cc] : 0.f;
            }
        for (int v = 0; v < n_chunks; ++v) {
            float* buf = lds + (v & 1) * (CP * kDxStride);
            {
                const f32x4* src = reinterpret_cast<const f32x4*>(Wtp + (int64_t)v * CP * kDxStride);
                f32x4* dst = reinterpret_cast<f32x4*>(buf);
                for (int t = threadIdx.x; t < CP * kDxStride / 4; t += kThreads) dst[t] = src[t];
            }
            __syncthreads();
#pragma unroll 2
            for (int vv = 0; vv < kKC / 16; ++vv) {
                const int c0 = v * kKC + 16 * vv;
                if (c0 >= K) break;   // wave-uniform: tiles beyond the matrix
                // TRANSPOSED tile: D = W-fragment (16 output columns x 4 c) * G^T (4 c x 16 rows), so that a lane ends up with
                // FOUR CONSECUTIVE COLUMNS of one row (D[ii = 4 kq + r][n = i]): one 16-byte store and exactly one mask group
                // per lane and row tile -- no cross-lane traffic, a quarter of the store instructions
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const float wf = buf[(4 * s + kq) * kDxStride + 16 * vv + i];   // A[ii = i][kk = kq] = W[c0 + i][4 s + kq]
                    acc0 = mfma16(wf, ga[0][s], acc0);                              // B[kk = kq][n = i] = G[row][4 s + kq]
                    acc1 = mfma16(wf, ga[1][s], acc1);
                }
                const int col = c0 + 4 * kq;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int64_t row = row_base + 16 * t + i;
                    if (row >= n_rows || col >= K) continue;
                    f4u o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = t == 0 ? acc0[r] : acc1[r];
                    if constexpr (MASK != 0) {
                        o = apply_mask<MASK>(o, mk, row, col);
                        o *= inv_keep;
                    }
                    float* dst = dX + row * lddx + col;
                    if (col + 4 <= K) {
                        __builtin_nontemporal_store(o, reinterpret_cast<f4u*>(dst));
                    } else {
                        dst[0] = o[0];
                        if (col + 1 < K) dst[1] = o[1];
                        if (col + 2 < K) dst[2] = o[2];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- backward, weights: dW = (X .* M / keep)^T G, per-workgroup partials ------------------------------------------------------
// grid.y = blocks of 512 columns of X (8 segments of 64); a wave owns one PAIR of adjacent segments.  Narrow inputs (K <= 256:
// fewer than four pairs) split the workgroup's rows over the otherwise idle waves instead (row_split = 2 or 4 sub-ranges,
// each with its own partial result).  M index of a tile (segment, j): ii <-> k = 64 seg + 4 ii + j.  The reduction runs over
// the row range in steps of 4 rows.
// launch bounds: 3 waves per SIMD (2 with four column tiles) -- without them the epilogue's accumulator read-out (96 AGPRs -> VGPRs
// at once) sets the allocation and the kernel drops to 2 waves per SIMD (measured 1.45 -> 1.76 ms)
template <int NT, int MASK>
__global__ __launch_bounds__(kThreads, NT <= 3 ? 3 : 2) void dropout_dense_dw_kernel(const float* __restrict__ X, int64_t ldx, int64_t n_rows, int K,
                                                                    const float* __restrict__ G, int64_t ldg, int C, float inv_keep,
                                                                    uint32_t thr, int mask_on, uint64_t seed, const int64_t* step_dev,
                                                                    int64_t rows_per_wg, float* __restrict__ partial, int Kp, int row_split) {
    constexpr int CP = NT * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const MaskKey mk = make_key(seed, step_dev, thr, mask_on, K);
    const int pairs = 4 / row_split, rsub = wave / pairs;
    const int seg0 = blockIdx.y * 8 + 2 * (wave % pairs);          // this wave's segments: seg0, seg0 + 1
    const int64_t wg_begin = (int64_t)blockIdx.x * rows_per_wg, wg_end = min(wg_begin + rows_per_wg, n_rows);
    const int64_t sub = ((rows_per_wg / row_split) + 3) / 4 * 4;   // rows of one sub-range (multiple of 4 rows)
    const int64_t r_begin = min(wg_begin + rsub * sub, wg_end), r_end = rsub == row_split - 1 ? wg_end : min(r_begin + sub, wg_end);
    f32x4 acc[2][4][NT];
#pragma unroll
    for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[w2][j][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    // software pipeline: a step covers kDwSub sub-steps of 4 rows; the operands of step s+1 are requested before the
    // kDwSub * 8 * NT MFMAs of step s are issued (bytes in flight per wave: 2 steps x kDwSub x 2 KiB of X)
    auto fetch = [&](int64_t r0, f4u (&xa)[kDwSub][2], float (&gb)[kDwSub][NT]) {
#pragma unroll
        for (int q = 0; q < kDwSub; ++q) {
            const int64_t row = r0 + 4 * q + kq;
            const bool row_ok = row < r_end;
#pragma unroll
            for (int w2 = 0; w2 < 2; ++w2) xa[q][w2] = load_row4(X, ldx, row_ok ? n_rows : 0, K, row, 64 * (seg0 + w2) + 4 * i);
#pragma unroll
            for (int u = 0; u < NT; ++u) gb[q][u] = (row_ok && 16 * u + i < C) ? G[row * ldg + 16 * u + i] : 0.f;
        }
    };
    f4u xa_n[kDwSub][2];
    float gb_n[kDwSub][NT];
    if (r_begin < r_end) fetch(r_begin, xa_n, gb_n);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += kDwRowsPerStep) {
        f4u xa[kDwSub][2];
        float gb[kDwSub][NT];
#pragma unroll
        for (int q = 0; q < kDwSub; ++q) {
            xa[q][0] = xa_n[q][0];
            xa[q][1] = xa_n[q][1];
#pragma unroll
            for (int u = 0; u < NT; ++u) gb[q][u] = gb_n[q][u];
        }
        if (r0 + kDwRowsPerStep < r_end) fetch(r0 + kDwRowsPerStep, xa_n, gb_n);
#pragma unroll
        for (int q = 0; q < kDwSub; ++q)
#pragma unroll
            for (int w2 = 0; w2 < 2; ++w2) {
                const f4u xm = apply_mask<MASK>(xa[q][w2], mk, r0 + 4 * q + kq, 64 * (seg0 + w2) + 4 * i);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int u = 0; u < NT; ++u) acc[w2][j][u] = mfma16(xm[j], gb[q][u], acc[w2][j][u]);
            }
    }
    float* out = partial + ((int64_t)blockIdx.x * row_split + rsub) * Kp * CP;
#pragma unroll
    for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 64 * (seg0 + w2) + 4 * (4 * kq + r) + j;
                    if (k < Kp) out[(int64_t)k * CP + 16 * u + i] = acc[w2][j][u][r] * inv_keep;
                }
}

// dW[k][c] = sum over the per-workgroup partials in a fixed order (deterministic): a workgroup owns 64 outputs, its four waves
// each sum a quarter of the partials in ascending order, the quarters are combined as (q0 + q1) + (q2 + q3)
__global__ __launch_bounds__(256) void reduce_dw_kernel(const float* __restrict__ partial, int n_parts, int Kp, int CP, int K, int C,
                                                        float* __restrict__ dW) {
    __shared__ float quarter[4][64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + o;
    float s = 0.f;
    if (t < K * C) {
        const int k = t / C, c = t - k * C;
        const int per = (n_parts + 3) / 4;
        const int p_begin = min(q * per, n_parts), p_end = min(p_begin + per, n_parts);
        const float* src = partial + (int64_t)k * CP + c;
        const int64_t stride = (int64_t)Kp * CP;
#pragma unroll 8
        for (int p = p_begin; p < p_end; ++p) s += src[p * stride];
    }
    quarter[q][o] = s;
    __syncthreads();
    if (q == 0 && t < K * C) dW[t] = (quarter[0][o] + quarter[1][o]) + (quarter[2][o] + quarter[3][o]);
}

struct Shape {
    int nt, kpad, cp, n_chunks, kp, gy, row_split;
    int64_t gx, rows_per_wg;
    size_t off_wfwd, off_wdx, off_partial, total;
};

int cu_count() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
}

// grid of a persistent kernel (workgroups loop over row groups): exactly the workgroups that are resident at once -- a grid
// larger than that runs its excess as a second, nearly empty round (measured: 6 per CU requested, 5 resident: +25 % time)
template <typename Kern>
unsigned persistent_grid(Kern kern, size_t lds_bytes, int64_t n_groups) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, kThreads, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 4;
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_groups, (int64_t)cu_count() * per_cu));
}

Shape shape_of(int64_t n_rows, int K, int C) {
    Shape s;
    s.nt = (C + 15) / 16;
    s.cp = s.nt * 16;
    s.kpad = (K + kKC - 1) / kKC * kKC;
    s.n_chunks = s.kpad / kKC;
    s.gy = (K + 511) / 512;
    s.kp = s.gy * 512;
    const int n_pairs = ((K + 63) / 64 + 1) / 2;                       // segment pairs that hold columns of X
    s.row_split = s.gy > 1 ? 1 : (n_pairs <= 1 ? 4 : (n_pairs == 2 ? 2 : 1));
    // dW: enough workgroups to fill the chip, each over a contiguous row range that is a multiple of 4 rows
    const int64_t want = std::max<int64_t>(1, (int64_t)cu_count() * H2GCN_CLS_DW_WG / s.gy);   // 3 workgroups per CU: the register budget of the 96 accumulators
    int64_t per = (n_rows + want - 1) / want;
    per = std::max<int64_t>(64, (per + 15) / 16 * 16);
    s.rows_per_wg = per;
    s.gx = std::max<int64_t>(1, (n_rows + per - 1) / per);
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    s.off_wfwd = 0;
    s.off_wdx = al((size_t)s.kpad * lds_stride(s.nt) * 4);
    s.off_partial = s.off_wdx + al((size_t)s.n_chunks * s.cp * kDxStride * 4);
    s.total = s.off_partial + al((size_t)s.gx * s.row_split * s.kp * s.cp * 4);
    return s;
}

uint32_t keep_threshold(float keep_prob) {   // 16-bit fields: kept iff field < keep_prob * 65536
    const double t = (double)keep_prob * 65536.0;
    return t >= 65536.0 ? 65536u : (uint32_t)t;
}

int check_common(const void* X, int64_t ld, int64_t n_rows, int K, const void* W, int C, float keep_prob) {
    if (n_rows < 0 || K < 1 || C < 1 || C > 64) return fail(H2GCN_ERR_INVALID_ARGUMENT, "dropout_dense: n_rows %lld, K %d, C %d (C <= 64)", (long long)n_rows, K, C);
    if (ld < K) return fail(H2GCN_ERR_INVALID_ARGUMENT, "dropout_dense: row stride %lld < K = %d", (long long)ld, K);
    if ((!X && n_rows > 0) || !W) return fail(H2GCN_ERR_INVALID_ARGUMENT, "dropout_dense: NULL operand");
    if (!(keep_prob > 0.f) || keep_prob > 1.f) return fail(H2GCN_ERR_INVALID_ARGUMENT, "dropout_dense: keep_prob = %g outside (0, 1]", (double)keep_prob);
    return H2GCN_OK;
}

int mask_mode(float keep_prob) { return keep_prob < 1.f ? ((keep_threshold(keep_prob) & 0xFFu) == 0 ? 1 : 2) : 0; }

// f(integral_constant<NT>, integral_constant<MASK>)
template <typename F>
int with_nt_mask(int nt, int mask, F&& f) {
    auto with_mask = [&](auto nt_c) -> int {
        switch (mask) {
            case 0: return f(nt_c, std::integral_constant<int, 0>());
            case 1: return f(nt_c, std::integral_constant<int, 1>());
            default: return f(nt_c, std::integral_constant<int, 2>());
        }
    };
    switch (nt) {
        case 1: return with_mask(std::integral_constant<int, 1>());
        case 2: return with_mask(std::integral_constant<int, 2>());
        case 3: return with_mask(std::integral_constant<int, 3>());
        default: return with_mask(std::integral_constant<int, 4>());
    }
}

}  // namespace

extern "C" {

int64_t h2gcn_dropout_dense_small_rows(int64_t rows) {
    const int64_t old = g_small_rows;
    if (rows >= 0) g_small_rows = rows;
    return old;
}

size_t h2gcn_dropout_dense_workspace_bytes(int64_t n_rows, int32_t k, int32_t c) {
    if (n_rows < 0 || k < 1 || c < 1 || c > 64) return 0;
    return shape_of(n_rows, k, c).total;
}

int h2gcn_dropout_dense_f32(const float* X, int64_t ldx, int64_t n_rows, int32_t K, const float* W, int32_t C, const float* bias,
                            float keep_prob, uint64_t seed, const int64_t* step_dev, float* Y, int64_t ldy, void* workspace,
                            size_t workspace_bytes, void* stream_v) {
    int st = check_common(X, ldx, n_rows, K, W, C, keep_prob);
    if (st != H2GCN_OK) return st;
    if (n_rows == 0) return H2GCN_OK;
    if (!Y || ldy < C) return fail(H2GCN_ERR_INVALID_ARGUMENT, "dropout_dense: bad output (ldy %lld)", (long long)ldy);
    const Shape s = shape_of(n_rows, K, C);
    if (!workspace || workspace_bytes < s.total || ((uintptr_t)workspace & 15u))
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "dropout_dense: workspace of %zu bytes (16-byte aligned) needed, got %zu", s.total, workspace_bytes);
    hipStream_t stream = (hipStream_t)stream_v;
    if (small_operand(n_rows, K, C)) {   // latency-bound operand: one plain kernel, no packing
        const int64_t rows_per_block = (int64_t)(kThreads / 64) * kSmallRowsPerWave;
        const unsigned blocks = (unsigned)((n_rows + rows_per_block - 1) / rows_per_block);
        const size_t lds = small_lds_bytes(K, C);
        switch (mask_mode(keep_prob)) {
            case 0: hipLaunchKernelGGL(small_fwd_kernel<0>, dim3(blocks), dim3(kThreads), lds, stream, X, ldx, n_rows, (int)K, W, bias, (int)C, 1.f / keep_prob, keep_threshold(keep_prob), 0, seed, step_dev, Y, ldy); break;
            case 1: hipLaunchKernelGGL(small_fwd_kernel<1>, dim3(blocks), dim3(kThreads), lds, stream, X, ldx, n_rows, (int)K, W, bias, (int)C, 1.f / keep_prob, keep_threshold(keep_prob), 1, seed, step_dev, Y, ldy); break;
            default: hipLaunchKernelGGL(small_fwd_kernel<2>, dim3(blocks), dim3(kThreads), lds, stream, X, ldx, n_rows, (int)K, W, bias, (int)C, 1.f / keep_prob, keep_threshold(keep_prob), 1, seed, step_dev, Y, ldy); break;
        }
        H2GCN_HIP_TRY(hipGetLastError());
        return H2GCN_OK;
    }
    float* wp = (float*)((char*)workspace + s.off_wfwd);
    const int S = lds_stride(s.nt);
    hipLaunchKernelGGL(pack_w_fwd_kernel, dim3(64), dim3(256), 0, stream, W, (int)K, (int)C, s.kpad, S, wp);
    H2GCN_HIP_TRY(hipGetLastError());
    const int mask_on = keep_prob < 1.f ? 1 : 0;
    const size_t lds_bytes = (size_t)2 * kKC * S * 4;
    const int64_t n_groups = (n_rows + kRowsPerGroup - 1) / kRowsPerGroup;
    return with_nt_mask(s.nt, mask_mode(keep_prob), [&](auto nt_c, auto mask_c) -> int {
        constexpr int NT = decltype(nt_c)::value, MASK = decltype(mask_c)::value;
        H2GCN_HIP_TRY(hipFuncSetAttribute((const void*)dropout_dense_fwd_kernel<NT, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        const unsigned grid = persistent_grid(dropout_dense_fwd_kernel<NT, MASK>, lds_bytes, n_groups);
        hipLaunchKernelGGL((dropout_dense_fwd_kernel<NT, MASK>), dim3(grid), dim3(kThreads), lds_bytes, stream, X, ldx, n_rows, (int)K, (const float*)wp,
                           s.kpad, bias, (int)C, 1.f / keep_prob, keep_threshold(keep_prob), mask_on, seed, step_dev, Y, ldy);
        H2GCN_HIP_TRY(hipGetLastError());
        return H2GCN_OK;
    });
}

int h2gcn_dropout_dense_backward_f32(const float* X, int64_t ldx, int64_t n_rows, int32_t K, const float* W, int32_t C, const float* G,
                                     int64_t ldg, float keep_prob, uint64_t seed, const int64_t* step_dev, float* dX, int64_t lddx,
                                     float* dW, void* workspace, size_t workspace_bytes, void* stream_v) {
    int st = check_common(X, ldx, n_rows, K, W, C, keep_prob);
    if (st != H2GCN_OK) return st;
    if (!G && n_rows > 0) return fail(H2GCN_ERR_INVALID_ARGUMENT, "dropout_dense_backward: G is NULL");
    if (ldg < C || (dX && lddx < K)) return fail(H2GCN_ERR_INVALID_ARGUMENT, "dropout_dense_backward: bad strides");
    const Shape s = shape_of(n_rows, K, C);
    if (!workspace || workspace_bytes < s.total || ((uintptr_t)workspace & 15u))
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "dropout_dense_backward: workspace of %zu bytes (16-byte aligned) needed, got %zu", s.total, workspace_bytes);
    hipStream_t stream = (hipStream_t)stream_v;
    const int mask_on = keep_prob < 1.f ? 1 : 0;
    const float inv_keep = 1.f / keep_prob;
    const uint32_t thr = keep_threshold(keep_prob);
    if (n_rows == 0) {
        if (dW) H2GCN_HIP_TRY(hipMemsetAsync(dW, 0, (size_t)K * C * 4, stream));
        return H2GCN_OK;
    }
    if (dX && small_operand(n_rows, K, C)) {   // latency-bound operand (see small_fwd_kernel); dW below, on the matrix cores
        const int64_t total = n_rows * ((K + 3) / 4);
        const unsigned blocks = (unsigned)std::min<int64_t>((total + kThreads - 1) / kThreads, 2 * (int64_t)cu_count());   // W^T is staged once per workgroup
        const size_t lds = small_lds_bytes(K, C);
        switch (mask_mode(keep_prob)) {
            case 0: hipLaunchKernelGGL(small_dx_kernel<0>, dim3(blocks), dim3(kThreads), lds, stream, G, ldg, n_rows, (int)K, (int)C, W, inv_keep, thr, 0, seed, step_dev, dX, lddx); break;
            case 1: hipLaunchKernelGGL(small_dx_kernel<1>, dim3(blocks), dim3(kThreads), lds, stream, G, ldg, n_rows, (int)K, (int)C, W, inv_keep, thr, 1, seed, step_dev, dX, lddx); break;
            default: hipLaunchKernelGGL(small_dx_kernel<2>, dim3(blocks), dim3(kThreads), lds, stream, G, ldg, n_rows, (int)K, (int)C, W, inv_keep, thr, 1, seed, step_dev, dX, lddx); break;
        }
        H2GCN_HIP_TRY(hipGetLastError());
        dX = nullptr;   // done
    }
    if (dX) {
        float* wtp = (float*)((char*)workspace + s.off_wdx);
        hipLaunchKernelGGL(pack_w_dx_kernel, dim3(64), dim3(256), 0, stream, W, (int)K, (int)C, s.cp, s.n_chunks, wtp);
        H2GCN_HIP_TRY(hipGetLastError());
        const size_t lds_bytes = (size_t)2 * s.cp * kDxStride * 4;
        const int64_t n_groups = (n_rows + kRowsPerGroup - 1) / kRowsPerGroup;
        st = with_nt_mask(s.nt, mask_mode(keep_prob), [&](auto nt_c, auto mask_c) -> int {
            constexpr int NT = decltype(nt_c)::value, MASK = decltype(mask_c)::value;
            H2GCN_HIP_TRY(hipFuncSetAttribute((const void*)dropout_dense_dx_kernel<NT, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            const unsigned grid = persistent_grid(dropout_dense_dx_kernel<NT, MASK>, lds_bytes, n_groups);
            hipLaunchKernelGGL((dropout_dense_dx_kernel<NT, MASK>), dim3(grid), dim3(kThreads), lds_bytes, stream, G, ldg, n_rows, (int)K, (int)C,
                               (const float*)wtp, s.n_chunks, inv_keep, thr, mask_on, seed, step_dev, dX, lddx);
            H2GCN_HIP_TRY(hipGetLastError());
            return H2GCN_OK;
        });
        if (st != H2GCN_OK) return st;
    }
    if (dW) {
        float* part = (float*)((char*)workspace + s.off_partial);
        st = with_nt_mask(s.nt, mask_mode(keep_prob), [&](auto nt_c, auto mask_c) -> int {
            constexpr int NT = decltype(nt_c)::value, MASK = decltype(mask_c)::value;
            hipLaunchKernelGGL((dropout_dense_dw_kernel<NT, MASK>), dim3((unsigned)s.gx, (unsigned)s.gy), dim3(kThreads), 0, stream, X, ldx, n_rows, (int)K,
                               G, ldg, (int)C, inv_keep, thr, mask_on, seed, step_dev, s.rows_per_wg, part, s.kp, s.row_split);
            H2GCN_HIP_TRY(hipGetLastError());
            return H2GCN_OK;
        });
        if (st != H2GCN_OK) return st;
        hipLaunchKernelGGL(reduce_dw_kernel, dim3((unsigned)((K * C + 63) / 64)), dim3(256), 0, stream, (const float*)part, (int)s.gx * s.row_split,
                           s.kp, s.cp, (int)K, (int)C, dW);
        H2GCN_HIP_TRY(hipGetLastError());
    }
    return H2GCN_OK;
}

}  // extern "C"
