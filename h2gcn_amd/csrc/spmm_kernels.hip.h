// spmm_kernels.hip.h -- gfx950 device code of the fused multi-hop CSR x dense aggregation.
//
// What it computes (reference h2gcn/models/_layers.py:78-81, GCNLayer.call):
//     Y[i, s, :] = sum_j A_s[i, j] * X[j, :]          for every selected hop s, fused in ONE launch,
// and, in SUM mode (the adjoint, reference gradient of SparseTensorDenseMatMul wrt its dense operand,
// reached from h2gcn/models/H2GCN.py:66-74):
//     dX[i, :]   = sum_s sum_j At_s[i, j] * dY[j, s, :].
//
// Hardware mapping (MI355X / CDNA4, wave64; the path is HBM-gather bound -- no MFMA on purpose):
//   * one wavefront walks a few consecutive rows ("CSR-adaptive, wave per row"); the feature row it gathers
//     per nonzero is contiguous (d*4 bytes), read with one 16-byte load per lane: LPR = d/4 lanes cover a
//     row, so a wave gathers G = 64/LPR neighbour rows per load instruction (d=128: 2 x 512 B = 1 KiB);
//   * column ids / values of a row are fetched 64 at a time with one coalesced load per wave and handed to
//     the lane groups with ds_bpermute (G>1) or v_readlane + scalar base addressing (G==1);
//   * loads are issued in batches of UNROLL before the first FMA so that >= 8 KiB per wave is in flight;
//   * partial sums of the G lane groups are folded with gfx950 v_permlane32_swap / v_permlane16_swap
//     (no LDS round trip), then LPR lanes write the output row with 16-byte non-temporal stores;
//   * (row,hop) segments with >= long_row_threshold nonzeros are taken out of the regular path and split
//     over the 4 waves of a workgroup: LDS-staged partial sums, summed in fixed wave order (deterministic);
//   * CSR-adaptive by SEGMENT CLASS: the plan bins every segment as short (<= 16 nonzeros) / medium / long.  A launch whose
//     segments are short throughout keeps the tile walk with G consecutive short rows per round (in-tile short-row mode);
//     a MIXED launch (mean >= 16 but >= 5 % of the segments short: a sparse hop next to a dense one, short rows scattered
//     among long ones) is list-driven -- each class has its own walk (one lane group / one wave / one workgroup per
//     segment, see short_list_blocks), wherever the short segments sit;
//   * the block -> row-tile map gives every XCD (private L2) a contiguous range of rows.
//
// Floating point: fp32 multiply-add per nonzero, ONE CANONICAL SUMMATION TREE per output element:
//     neighbour j of a (row, hop) segment (j = its position in the row's ascending column order, the reference's
//     order) is added, in ascending j, into partial  P[j mod 4];  the row's value is  (P0 + P1) + (P2 + P3).
// Every kernel in this file produces exactly that tree whatever its lane geometry: with 64-column slices the four
// lane groups of a wave each own one partial and the fold is v_permlane16_swap / v_permlane32_swap; with 128-column
// slices a lane group keeps two partials in registers (even / odd steps), with 256-column slices and in the generic
// column-tiled kernel a lane keeps all four; the short-row mode keeps four per lane group.  (A segment with >=
// long_row_threshold nonzeros is split over the 4 waves of a workgroup in 64-neighbour chunks: each wave builds the
// same tree over its chunks, the 4 wave totals are added in wave order.)  The bits of Y therefore depend only on the
// row's own nonzeros and on long_row_threshold -- never on the slice / feature-chunk width, the scratch copy, the
// segment-walk variant, grid geometry, rows_per_wave or the row partition: a P-GPU run reproduces the 1-GPU result
// bit-for-bit with ANY chunking (SURVEY.md 8(e) "Determinism").  There is no exception: widths below 64 columns run on
// the 64-column geometry with a masked slice (the 32 / 16-column kernels of rounds 1-2, whose 8 / 16 lane groups needed
// a wider tree, were slower on every width and have been removed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "h2gcn_hip.h"

namespace h2gcn {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr int kNumXcd = 8;
constexpr int kMaxRowsPerWave = 7;  // (rows_per_wave + 1) * n_hops row pointers must fit one wave-wide load
#ifndef H2GCN_MIN_WAVES
#define H2GCN_MIN_WAVES 6
#endif
constexpr int kMinWavesPerSimd = H2GCN_MIN_WAVES;
#ifndef H2GCN_SHORT_MIN_WAVES
#define H2GCN_SHORT_MIN_WAVES 4
#endif  // __launch_bounds__ of the in-tile short-row kernels (waves per SIMD the register budget must allow)
#ifndef H2GCN_SHORT_FB4_MIN_WAVES
#define H2GCN_SHORT_FB4_MIN_WAVES 7
#endif  // ... of the forward in-tile short-row kernels with shallow fallback batches and of the list-driven kernels with
        // shallow load batches (FB = 4; memory-resident operands whose nonzeros sit mostly in short segments: occupancy
        // buys bandwidth -- even at the price of 12 B of scratch in the one instantiation that needs 78 registers, the
        // forward 128-column kernel with 64-bit offsets: lowdeg 6.97 ms at 7 waves vs 7.21 at 6 without scratch)
#ifndef H2GCN_MAIN_MAXB
#define H2GCN_MAIN_MAXB 8
#endif  // deepest load batch of the bandwidth kernels (8 x 16 B per lane in flight)
#ifndef H2GCN_WIDE_MAXB
#define H2GCN_WIDE_MAXB H2GCN_MAIN_MAXB
#endif  // deepest load batch of the plain walk of the 128 / 256-column slices
#ifndef H2GCN_WIDE_MIN_WAVES
#define H2GCN_WIDE_MIN_WAVES H2GCN_MIN_WAVES
#endif  // ... of the 128 / 256-column slices, whose lanes keep 2 / 4 partials of the canonical tree
constexpr int kShortMax = 16;       // longest segment of the binned short class (one index fetch per lane group of 16 lanes)
constexpr int kShortSumHops = 4;    // SUM mode serves listed rows only for selections of at most this many hops (LDS staging)
#ifndef H2GCN_SHORT_PREFETCH
#define H2GCN_SHORT_PREFETCH 2
#endif  // index fetches the list walk keeps in flight ahead of the round that is gathering
#ifndef H2GCN_OFF64_HEAVY_MIN_WAVES
#define H2GCN_OFF64_HEAVY_MIN_WAVES 5
#endif  // ... of the 64-bit-offset kernels with a general store or a 128 / 256-column slice (two address VGPRs per load in
        // flight): 82-88 VGPRs at 5 waves per SIMD; pinned to 6 they sit on the 80-register cap with 12-28 bytes of scratch.
        // Same-box A/B (profiles/r04_ab_off64_launch_bounds.txt): 5 waves without scratch is level with or 0.1-0.5 % ahead of 6
        // waves with it (slice 128, forced 64-bit offsets: 19.74-19.80 vs 19.84-19.85 ms; training step at hidden 100: 115.0 vs
        // 115.4-119.0 ms) -- and the 64-bit kernels cost nothing over the 32-bit ones on the same operand (17.63 vs 17.63 ms)
constexpr int kMaxTileCols = 256;   // columns one pass of a wave covers at most (64 lanes x float4)
constexpr int kTreeParts = 4;       // partials of the canonical summation tree (see the header comment)

// NP = partials a lane keeps for the canonical tree: the G = 64/LPR lane groups of a wave own G of the 4 partials per
// step, so a lane cycles through 4/G of them (G = 4: one, G = 2: two, G = 1: all four)
template <int LPR>
struct Tree {
    static constexpr int G = kWave / LPR;
    static_assert(G == 1 || G == 2 || G == 4, "lane geometries: 64 / 128 / 256 columns per slice");
    static constexpr int NP = kTreeParts / G;
};

struct HopCsr {
    const int64_t* rowptr;
    const int32_t* colidx;
    const float* vals;
};

struct LaunchParams {
    HopCsr hop[H2GCN_MAX_HOPS];          // the selected hops, packed
    int64_t src_hop_off[H2GCN_MAX_HOPS]; // element offset added to the gather source for hop s
    int64_t dst_hop_off[H2GCN_MAX_HOPS]; // element offset added to the output for hop s (ignored in SUM mode)
    int n_sel;
    int d;           // feature columns of the output
    int d_src;       // readable columns of a gather-source row (== d in place; d rounded up to 4 in the zero-padded scratch)
    int64_t n_rows;  // rows of the output
    const float* src;
    int64_t ld_src;
    int64_t src_slice_stride;  // EXACT kernels: element offset of column slice q in the gather source = q * this
                               // (row-major source: slice_cols; slice-major scratch copy: n_src_rows * slice_cols)
    float* dst;
    int64_t ld_dst;
    const int64_t* long_list; // forward: (row << 4 | s) per long segment; SUM: row per long row
    int n_long;
    int long_threshold;
    int rows_per_wave;
    int64_t n_tiles;       // row tiles of rows_per_wave * kWavesPerBlock rows
    int64_t tiles_per_xcd; // ceil(n_tiles / 8)
    int n_slices;          // column slices of slice_cols features, processed slice-major (EXACT kernels)
    int slice_cols;
    int64_t blocks_per_slice;  // n_long + 8 * tiles_per_xcd
    const float* bias;         // optional epilogue: dst = act(sum + bias[col]); NULL = none
    int relu;                  // optional epilogue: act = max(., 0)
    int accumulate;            // general-store kernels: dst += sum instead of dst = sum (adjoint into an existing gradient)
    // binned SHORT segments (SHORT kernels; see short_list_blocks): segments of at most short_max nonzeros are taken out of
    // the tile walk and served from plan-owned lists, one lane group per segment
    const int32_t* short_list[H2GCN_MAX_HOPS];  // forward: rows whose segment of selected hop s is short, ascending;
                                                // SUM: [0] = rows whose segments of ALL selected hops are short
    int64_t short_count[H2GCN_MAX_HOPS];        // entries of each list
    int64_t short_end[H2GCN_MAX_HOPS];          // forward: running number of list workgroups up to and including hop s
    int64_t short_groups;                       // groups of 8 short-list workgroups, interleaved with the medium-list groups
    int short_max;                              // -1: no lists (the tile walk serves every segment below long_threshold)
    int short_per_wave;                         // list entries one wave serves (<= 64, a multiple of 4)
    int short_hop_major;                        // forward: 1 = all workgroups of hop 0's list, then hop 1's ...; 0 = workgroup b
                                                // serves chunk b / n_sel of hop b % n_sel (a row's hop outputs close in time)
    // ... and the MEDIUM segments (everything between short and long) of such a launch: same layout, walked one wave per
    // segment, med_per_wave consecutive entries per wave (see medium_list_blocks); SHORT kernels have no tile walk at all
    const int32_t* med_list[H2GCN_MAX_HOPS];
    int64_t med_count[H2GCN_MAX_HOPS];
    int64_t med_end[H2GCN_MAX_HOPS];
    int64_t med_groups;
    int med_per_wave;
};

template <int VEC>
struct VecT;
template <>
struct VecT<1> {
    using type = float;
};
// aligned(4): global_load/store_dwordx2/x4 on gfx9 only need dword alignment (measured: tools/unaligned_vec_probe), so
// feature rows of ANY width and any 4-byte aligned base are gathered and stored with 16-byte instructions
template <>
struct VecT<2> {
    typedef float type __attribute__((ext_vector_type(2), aligned(4)));
};
template <>
struct VecT<4> {
    typedef float type __attribute__((ext_vector_type(4), aligned(4)));
};

__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// value of `v` held by lane `src_lane` (src_lane may differ per lane): ds_bpermute_b32
__device__ __forceinline__ int lane_gather(int v, int src_lane) {
    return __builtin_amdgcn_ds_bpermute(src_lane << 2, v);
}
__device__ __forceinline__ float lane_gather(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

// sum over the lanes {l, l^32}: v_permlane32_swap (gfx950)
__device__ __forceinline__ float fold_xor32(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over the lanes {l, l^16}: v_permlane16_swap (gfx950)
__device__ __forceinline__ float fold_xor16(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over {l, l^8}: DPP row_ror:8 inside each row of 16 lanes
__device__ __forceinline__ float fold_xor8(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xf, 0xf, false);
    return v + __int_as_float(o);
}

// sum over the four lanes {l, l+4, l+8, l+12} (mod 16) of a row: after fold_xor8, one DPP row_ror:4 completes it
__device__ __forceinline__ float fold_ror4(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124 /* row_ror:4 */, 0xf, 0xf, false);
    return v + __int_as_float(o);
}

// Fold the partial sums of the G = 64/LPR lane groups; afterwards every group holds the total.
// Fixed tree: (g, g^1) first ... independent of anything but LPR.
template <int LPR>
__device__ __forceinline__ float fold_groups(float v) {
    if constexpr (LPR <= 8) v = fold_xor8(v);
    if constexpr (LPR <= 4) v = fold_ror4(v);
    if constexpr (LPR <= 16) v = fold_xor16(v);
    if constexpr (LPR <= 32) v = fold_xor32(v);
    return v;
}

// The row total out of the partials a lane holds (see Tree<LPR>): afterwards every lane group holds the total.
//   G = 1: (P0 + P1) + (P2 + P3) in registers;  G = 2: group g holds P[g], P[g+2] -> fold each across the groups, add;
//   G = 4: group g holds P[g] -> fold_groups: (g, g^1) then (g, g^2) = (P0 + P1) + (P2 + P3).
template <int VEC, int LPR, int NP>
__device__ __forceinline__ void fold_tree(const float (&acc)[NP][VEC], float (&tot)[VEC]) {
    constexpr int G = kWave / LPR;
    static_assert(NP == Tree<LPR>::NP, "partials per lane");
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        if constexpr (G == 1) tot[i] = (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
        else if constexpr (G == 2) tot[i] = fold_xor32(acc[0][i]) + fold_xor32(acc[1][i]);
        else tot[i] = fold_groups<LPR>(acc[0][i]);
    }
}

template <int VEC, int NP>
__device__ __forceinline__ void zero_acc(float (&acc)[NP][VEC]) {
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[k][i] = 0.f;
}

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type load_vec(const float* p) {
#ifdef H2GCN_NT_GATHER
    return __builtin_nontemporal_load(reinterpret_cast<const typename VecT<VEC>::type*>(p));
#else
    return *reinterpret_cast<const typename VecT<VEC>::type*>(p);
#endif
}

template <int VEC>
__device__ __forceinline__ void fma_vec(float (&acc)[VEC], float w, typename VecT<VEC>::type x) {
    if constexpr (VEC == 1) {
        acc[0] = fmaf(w, x, acc[0]);
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(w, x[i], acc[i]);
    }
}

// Gather addressing.  OFF32: every byte offset into the gather source fits 32 bits (checked on the host), so a
// load is `global_load_dwordx4 v, v_off32, s[base]` -- scalar base + one 32-bit VGPR offset; no 64-bit
// multiply per neighbour and 8 fewer address VGPRs per batch than full 64-bit per-lane pointers.
template <bool OFF32>
struct GatherAddr {
    const char* base;                                   // wave-uniform: src + hop offset + column offset
    typename std::conditional<OFF32, uint32_t, int64_t>::type lane_off;  // byte offset of this lane inside a row
    typename std::conditional<OFF32, uint32_t, int64_t>::type ld_bytes;  // row stride in bytes
    __device__ __forceinline__ const float* row(int col) const {
        if constexpr (OFF32) {
            return reinterpret_cast<const float*>(base + (uint32_t)(lane_off + (uint32_t)col * ld_bytes));
        } else {
            return reinterpret_cast<const float*>(base + (lane_off + (int64_t)col * ld_bytes));
        }
    }
};

// One batch of U gathers issued back to back, then U multiply-adds.  `t` is the first step of the batch
// (wave uniform); step t+u serves neighbour (t+u)*G + g of the current 64-wide chunk.
// The cross-lane reads of (c, v) always run with the full wave active (ds_bpermute returns 0 for a source
// lane that is masked off); only the gather itself is predicated, by `take` (false on lanes whose slot is
// padding -- they must not touch src: 0 * Inf would poison the row).
template <int VEC, int LPR, int U, bool PREDICATED, bool OFF32, int PHASE = 0, int NP>
__device__ __forceinline__ void gather_batch(int c, float v, int t, int g, const GatherAddr<OFF32>& addr, bool take,
                                             float (&acc)[NP][VEC]) {
    constexpr int G = kWave / LPR;
    typename VecT<VEC>::type x[U];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int cj;
        if constexpr (G == 1) {
            // wave-uniform neighbour: v_readlane -> SGPR, the load uses a scalar base + lane offset
            cj = __builtin_amdgcn_readlane(c, t + u);
            w[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), t + u));
        } else {
            const int idx = ((t + u) * G + g) & (kWave - 1);
            cj = lane_gather(c, idx);
            w[u] = lane_gather(v, idx);
        }
        const float* p = addr.row(cj);
        if constexpr (PREDICATED) {
            if (take) {
                x[u] = load_vec<VEC>(p);
            } else {
                w[u] = 0.f;
                if constexpr (VEC == 1) x[u] = 0.f; else x[u] = (typename VecT<VEC>::type)(0.f);
            }
        } else {
            x[u] = load_vec<VEC>(p);
        }
    }
    // step t+u serves neighbour (t+u)*G + g -> canonical partial ((t+u)*G + g) % 4 = lane-local partial (t+u) % NP;
    // the caller states t % NP as PHASE, so the partial of every step is a compile-time register choice
#pragma unroll
    for (int u = 0; u < U; ++u) fma_vec<VEC>(acc[(PHASE + u) % NP], w[u], x[u]);
}

// (column id, value) of neighbour `lane` of the 64-wide chunk starting at `base` (zero beyond the segment end)
__device__ __forceinline__ void load_chunk(const int32_t* __restrict__ colidx, const float* __restrict__ vals,
                                           int64_t base, int64_t seg_end, int lane, int& c, float& v) {
    c = 0;
    v = 0.f;
    if (base + lane < seg_end) {
        c = __builtin_nontemporal_load(colidx + base + lane);
        v = __builtin_nontemporal_load(vals + base + lane);
    }
}

// All gathers + multiply-adds of one chunk of n (<= 64) neighbours held lane-wise in (c, v).
// MAXB: deepest load batch (8 in the bandwidth kernels; 4 where register pressure matters more than the last few
// percent on long segments -- the fallback walk of the short-row kernels).  The batch depth only re-times loads: the
// accumulation order, hence the bits, do not depend on it.
template <int VEC, int LPR, bool MASKED, bool OFF32, int MAXB = H2GCN_MAIN_MAXB, int NP>
__device__ __forceinline__ void process_chunk(int c, float v, int n, int g, const GatherAddr<OFF32>& addr,
                                              bool lane_active, float (&acc)[NP][VEC]) {
    constexpr int G = kWave / LPR;
    const int full = n / G;  // steps in which every lane group has a neighbour
    int t = 0;               // batches of 8 / 4 start at t % 4 == 0: partial phase 0
    if constexpr (MAXB >= 8)
        for (; t + 8 <= full; t += 8) gather_batch<VEC, LPR, 8, MASKED, OFF32>(c, v, t, g, addr, lane_active, acc);
    if constexpr (MAXB < 8)
        for (; t + 4 <= full; t += 4) gather_batch<VEC, LPR, 4, MASKED, OFF32>(c, v, t, g, addr, lane_active, acc);
    if (t + 4 <= full) {
        gather_batch<VEC, LPR, 4, MASKED, OFF32>(c, v, t, g, addr, lane_active, acc);
        t += 4;
    }
    // the last 0..3 full steps and (G > 1) the ragged step in which only the first (n - full*G) groups still have a
    // neighbour, with the phase of every step spelled out
    const int rem = n - full * G;
    if (t + 2 <= full) {
        gather_batch<VEC, LPR, 2, MASKED, OFF32>(c, v, t, g, addr, lane_active, acc);
        t += 2;
        if (t + 1 <= full) {
            gather_batch<VEC, LPR, 1, MASKED, OFF32, 2 % NP>(c, v, t, g, addr, lane_active, acc);
            if constexpr (G > 1)
                if (rem > 0) gather_batch<VEC, LPR, 1, true, OFF32, 3 % NP>(c, v, full, g, addr, lane_active && g < rem, acc);
        } else {
            if constexpr (G > 1)
                if (rem > 0) gather_batch<VEC, LPR, 1, true, OFF32, 2 % NP>(c, v, full, g, addr, lane_active && g < rem, acc);
        }
    } else if (t + 1 <= full) {
        gather_batch<VEC, LPR, 1, MASKED, OFF32>(c, v, t, g, addr, lane_active, acc);
        if constexpr (G > 1)
            if (rem > 0) gather_batch<VEC, LPR, 1, true, OFF32, 1 % NP>(c, v, full, g, addr, lane_active && g < rem, acc);
    } else {
        if constexpr (G > 1)
            if (rem > 0) gather_batch<VEC, LPR, 1, true, OFF32>(c, v, full, g, addr, lane_active && g < rem, acc);
    }
}

// Accumulate sum_j val_j * src[col_j, :] over the nonzeros [seg_begin, seg_end) of one CSR row, taking the
// 64-wide chunks chunk0, chunk0+chunk_step, ... (regular path: all of them; long path: this wave's share).
// Each lane group accumulates its neighbours in ascending order into acc.
template <int VEC, int LPR, bool MASKED, bool OFF32, int MAXB = H2GCN_MAIN_MAXB, int NP>
__device__ __forceinline__ void accumulate_segment(const int32_t* __restrict__ colidx,
                                                   const float* __restrict__ vals, int64_t seg_begin,
                                                   int64_t seg_end, int chunk0, int chunk_step,
                                                   const GatherAddr<OFF32>& addr, int lane, bool lane_active,
                                                   float (&acc)[NP][VEC]) {
    const int g = lane / LPR;
    for (int64_t base = seg_begin + (int64_t)chunk0 * kWave; base < seg_end; base += (int64_t)chunk_step * kWave) {
        const int64_t left = seg_end - base;
        const int n = left < kWave ? (int)left : kWave;
        int c;
        float v;
        load_chunk(colidx, vals, base, seg_end, lane, c, v);
        process_chunk<VEC, LPR, MASKED, OFF32, MAXB>(c, v, n, g, addr, lane_active, acc);
    }
}

// Same sum, software-pipelined: the first chunk (c, v) was fetched by the caller while the PREVIOUS segment was
// still gathering, and every further chunk is fetched before the current one is processed -- the index fetch
// latency leaves the per-segment dependency chain (index -> gather -> fold -> store), which is what bounds
// short rows.  Loads retire in order, so waiting for the gathers implies the prefetch has landed.
template <int VEC, int LPR, bool OFF32, int MAXB = H2GCN_MAIN_MAXB, int NP>
__device__ __forceinline__ void accumulate_segment_prefetched(const int32_t* __restrict__ colidx,
                                                              const float* __restrict__ vals, int64_t seg_begin,
                                                              int64_t seg_end, int c, float v,
                                                              const GatherAddr<OFF32>& addr, int lane,
                                                              float (&acc)[NP][VEC]) {
    const int g = lane / LPR;
    for (int64_t base = seg_begin; base < seg_end; base += kWave) {
        const int64_t left = seg_end - base;
        const int n = left < kWave ? (int)left : kWave;
        int c_next = 0;
        float v_next = 0.f;
        if (left > kWave) load_chunk(colidx, vals, base + kWave, seg_end, lane, c_next, v_next);
        process_chunk<VEC, LPR, false, OFF32, MAXB>(c, v, n, g, addr, true, acc);
        c = c_next;
        v = v_next;
    }
}

// ---- one lane group per segment: the inner loop of the short class (see short_list_blocks below) ---------------------
// The G lane groups of a wave hold G DIFFERENT short segments and walk them in lock step: G segments' gathers are in
// flight together and no cross-lane fold is needed.  The arithmetic is the canonical tree of the header comment: a group
// keeps the four partials P[j mod 4] in registers and combines them as (P0 + P1) + (P2 + P3).
// Handles segments of at most LPR nonzeros (one index fetch per group): `c`, `v` hold the group's indices/values
// lane-wise (lane li of the group = neighbour li), `n_mine` its length, `n_max` the longest of the round (uniform).
template <int VEC, int LPR, bool OFF32>
__device__ __forceinline__ void accumulate_grouped(int c, float v, int n_mine, int n_max, int lane,
                                                   const GatherAddr<OFF32>& addr, float (&part)[kTreeParts][VEC]) {
    constexpr int U = kTreeParts;  // neighbours per batch: neighbour t+u (t % 4 == 0) -> canonical partial u
    const int group_base = lane & ~(LPR - 1);
    for (int t = 0; t < n_max; t += U) {
        typename VecT<VEC>::type x[U];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cj = lane_gather(c, group_base + t + u);
            w[u] = lane_gather(v, group_base + t + u);
            if (t + u < n_mine) {
                x[u] = load_vec<VEC>(addr.row(cj));
            } else {
                w[u] = 0.f;
                if constexpr (VEC == 1) x[u] = 0.f; else x[u] = (typename VecT<VEC>::type)(0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (t + u < n_mine) fma_vec<VEC>(part[u], w[u], x[u]);
    }
}

// the canonical tree applied to partials held in registers: (p0 + p1) + (p2 + p3)
template <int VEC>
__device__ __forceinline__ void combine_partials(const float (&part)[kTreeParts][VEC], float (&acc)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
}

// Optional fused epilogue of the store (reference SparseDense.call, h2gcn/models/_layers.py:45-52: `+ bias`, then the
// activation): applied to the finished sum of an output element, columns col .. col+VEC-1.
template <int VEC>
__device__ __forceinline__ void epilogue(float (&acc)[VEC], const float* __restrict__ bias, int relu, int col) {
    if (bias) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += bias[col + i];
    }
    if (relu) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaxf(acc[i], 0.f);
    }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&acc)[VEC]) {
    if constexpr (VEC == 1) {
        __builtin_nontemporal_store(acc[0], p);
    } else {
        typename VecT<VEC>::type o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = acc[i];
#if defined(H2GCN_AB_NO_STORES)   // A/B builds only (profiles/r06_ab_forward_vs_adjoint_x6.txt): every sum is computed, the output
        if (acc[0] == 1.2345678e30f)   // stream is not issued (the comparison keeps the arithmetic alive and is never true)
            __builtin_nontemporal_store(o, reinterpret_cast<typename VecT<VEC>::type*>(p));
#elif defined(H2GCN_PLAIN_STORES)   // A/B builds only (profiles/r04_ab_output_stores.txt)
        *reinterpret_cast<typename VecT<VEC>::type*>(p) = o;
#else
        __builtin_nontemporal_store(o, reinterpret_cast<typename VecT<VEC>::type*>(p));
#endif
    }
}

// Store of one lane's VEC finished sums.  The lane's nominal columns are col0 .. col0+VEC-1; its sums are those of
// columns ecol .. ecol+VEC-1 (ecol == col0 except in the TAIL lane of a width that is not a multiple of VEC, which
// gathered the row's last VEC columns, overlapping its left neighbour -- see the kernel).  `row` points at column 0 of
// the output row.  Plain kernels: one 16-byte store when the whole vector lies inside d.  General-store kernels (GEN)
// add the bias / ReLU epilogue and let the tail lane store the columns it owns element-wise -- which is how odd feature
// widths (reference: any b.shape[1], _layers.py:62-76) run on the float4 gather kernels.
template <int VEC, bool GEN, typename P>
__device__ __forceinline__ void store_out(const P& p, float* row, int col0, int ecol, float (&tot)[VEC]) {
    if (col0 + VEC <= p.d) {
        if constexpr (GEN) {
            if (p.accumulate) {
                if constexpr (VEC == 1) {
                    tot[0] += row[col0];
                } else {
                    const typename VecT<VEC>::type old = *reinterpret_cast<const typename VecT<VEC>::type*>(row + col0);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) tot[i] += old[i];
                }
            }
            epilogue<VEC>(tot, p.bias, p.relu, col0);
        }
        store_vec<VEC>(row + col0, tot);
        return;
    }
    if constexpr (GEN) {
        if (col0 < p.d) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const int c = ecol + i;
                if (c >= col0 && c < p.d) {
                    float t = tot[i];
                    if (p.accumulate) t += row[c];
                    if (p.bias) t += p.bias[c];
                    if (p.relu) t = fmaxf(t, 0.f);
                    __builtin_nontemporal_store(t, row + c);
                }
            }
        }
    }
}

// ---- binned short segments ("lane group per segment" from a plan-owned list) --------------------------------------------
// CSR-adaptive dispatch by SEGMENT CLASS: the plan bins every (row, hop) segment by its length -- short (<= kShortMax
// nonzeros), medium, long (>= long_row_threshold) -- and a launch serves each class with the walk that suits it:
//   long    one workgroup per segment, 4 waves, LDS-staged partial sums            (long_list, as before)
//   medium  one wave per segment, rows_per_wave consecutive rows per wave          (the tile walk, which skips the others)
//   short   one LANE GROUP per segment, from the list                              (here)
// A short segment walked by a whole wave is a dependent chain index -> gather -> fold -> store with a single 1-KiB load in
// flight; here the G = 64/LPR lane groups of a wave take G different segments in lock step, no cross-lane fold is needed,
// and the index fetches run H2GCN_SHORT_PREFETCH rounds ahead of the gathers.  Because the list is built from the segment
// lengths themselves, WHICH rows are short does not matter: short rows scattered among long ones (real degree sequences),
// a short hop next to a dense one (the reference's exact-1-hop vs exact-2-hop rings, _dataset.py:138-158) -- every short
// segment is served this way, and every other one by the walk of its own class, in the same launch.
// One wave serves 64 consecutive list entries: lane e fetches entry e's row id and row pointers (one coalesced list read,
// one 16-byte row-pointer read per lane and hop) and parks them in LDS; round t hands entries t*G .. t*G+G-1 to the lane
// groups.  Forward: a list workgroup belongs to one selected hop.  SUM: an entry is a row whose segments of ALL selected
// hops are short; the partials run on across the hops exactly as the wave walk's accumulators do.
// The arithmetic is the canonical tree of the header comment (a group keeps P[j mod 4] in registers, combined as
// (P0 + P1) + (P2 + P3)), so which class served a segment can never be seen in the result.
template <int VEC, int LPR, bool SUM, bool OFF32, bool EPI, int PD>
__device__ __forceinline__ void short_list_blocks(const LaunchParams& p, int64_t sblock, int lane, int wave, int64_t lane_off0,
                                                  int64_t src_col_begin, int lcol, int ecol,
                                                  uint64_t (&s_seg)[kWavesPerBlock][kShortSumHops][kWave],
                                                  int32_t (&s_row)[kWavesPerBlock][kWave]) {
    using off_t = typename std::conditional<OFF32, uint32_t, int64_t>::type;
    constexpr int G = kWave / LPR;
    constexpr uint64_t kBeginMask = (1ull << 58) - 1;
    const int li = lane % LPR, g = lane / LPR;
    const int n_sel = p.n_sel;
    int hop = 0;  // forward: the selected hop this workgroup's list belongs to
    int64_t local = sblock;
    if constexpr (!SUM) {
        if (p.short_hop_major) {
            while (hop + 1 < n_sel && sblock >= p.short_end[hop]) ++hop;
            if (hop > 0) local -= p.short_end[hop - 1];
        } else {
            local = sblock / n_sel;
            hop = (int)(sblock - local * n_sel);
        }
    }
    const int spw = p.short_per_wave;
    const int64_t entry0 = (local * kWavesPerBlock + wave) * spw;
    const int64_t left = p.short_count[hop] - entry0;
    if (left <= 0) return;
    const int n_here = left < spw ? (int)left : spw;
    const int NS = SUM ? n_sel : 1;  // segments per entry
    if (lane < n_here) {
        const int32_t row = __builtin_nontemporal_load(p.short_list[hop] + entry0 + lane);
        s_row[wave][lane] = row;
        for (int s = 0; s < NS; ++s) {
            const int64_t* rp = p.hop[SUM ? s : hop].rowptr + row;
            const int64_t b0 = rp[0], b1 = rp[1];
            s_seg[wave][s][lane] = (uint64_t)b0 | ((uint64_t)(b1 - b0) << 58);  // length <= kShortMax: 6 bits
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n_rounds = (n_here + G - 1) / G;
    const int n_steps = n_rounds * NS;  // step i = (round t, segment s), s fastest

    // index fetch of one step for this lane's group: lane li holds neighbour li of the group's segment; len -1 = no entry
    int cq[PD], lq[PD];
    float vq[PD];
    int ft = 0, fs = 0;  // (t, s) of the next step to fetch
    auto fetch = [&](int& c, float& v, int& len) {
        c = 0;
        v = 0.f;
        len = -1;
        const int e = ft * G + g;
        if (e < n_here) {
            const uint64_t packed = s_seg[wave][fs][e];
            len = (int)(packed >> 58);
            if (li < len) {
                const HopCsr& h = p.hop[SUM ? fs : hop];
                const int64_t at = (int64_t)(packed & kBeginMask) + li;
                c = __builtin_nontemporal_load(h.colidx + at);
                v = __builtin_nontemporal_load(h.vals + at);
            }
        }
        if (++fs == NS) {
            fs = 0;
            ++ft;
        }
    };
#pragma unroll
    for (int k = 0; k < PD; ++k) {
        cq[k] = 0;
        vq[k] = 0.f;
        lq[k] = -1;
        if (k < n_steps) fetch(cq[k], vq[k], lq[k]);
    }
    float part[kTreeParts][VEC];
    int t = 0, s = 0;
    for (int i = 0; i < n_steps; ++i) {
        const int c = cq[0], len = lq[0];
        const float v = vq[0];
#pragma unroll
        for (int k = 0; k + 1 < PD; ++k) {
            cq[k] = cq[k + 1];
            vq[k] = vq[k + 1];
            lq[k] = lq[k + 1];
        }
        lq[PD - 1] = -1;
        if (i + PD < n_steps) fetch(cq[PD - 1], vq[PD - 1], lq[PD - 1]);
        if (s == 0) zero_acc<VEC, kTreeParts>(part);
        int n_max = 0;
#pragma unroll
        for (int gg = 0; gg < G; ++gg) n_max = max(n_max, __builtin_amdgcn_readlane(len, gg * LPR));
        const int hs = SUM ? s : hop;
        const GatherAddr<OFF32> addr{reinterpret_cast<const char*>(p.src + p.src_hop_off[hs] + src_col_begin - VEC), (off_t)lane_off0,
                                     (off_t)(p.ld_src * 4)};
        accumulate_grouped<VEC, LPR, OFF32>(c, v, len, n_max, lane, addr, part);
        if (s == NS - 1) {
            if (len >= 0) {
                float tot[VEC];
                combine_partials<VEC>(part, tot);
                const int64_t row = s_row[wave][t * G + g];
                store_out<VEC, EPI>(p, p.dst + row * p.ld_dst + (SUM ? 0 : p.dst_hop_off[hop]), lcol, ecol, tot);
            }
            s = 0;
            ++t;
        } else {
            ++s;
        }
    }
}

// The MEDIUM class of a list-driven launch: one wave per segment, p.med_per_wave consecutive list entries per wave.
// Forward: an entry is a (row, hop) segment of the hop the workgroup's list belongs to; SUM: a row that is neither listed
// as short nor owned by the long path, all selected hops' segments accumulated into one output row.  Lane j*NS + s fetches
// the row pointers of entry j, hop s; the walk itself is accumulate_segment, i.e. the tile walk's arithmetic.
template <int VEC, int LPR, bool SUM, bool OFF32, bool EPI, int MAXB>
__device__ __forceinline__ void medium_list_blocks(const LaunchParams& p, int64_t mblock, int lane, int wave, int64_t lane_off0,
                                                   int64_t src_col_begin, int lcol, int ecol) {
    using off_t = typename std::conditional<OFF32, uint32_t, int64_t>::type;
    constexpr int NP = Tree<LPR>::NP;
    const int g = lane / LPR;
    const int n_sel = p.n_sel;
    int hop = 0;
    int64_t local = mblock;
    if constexpr (!SUM) {
        while (hop + 1 < n_sel && mblock >= p.med_end[hop]) ++hop;
        if (hop > 0) local -= p.med_end[hop - 1];
    }
    const int k = p.med_per_wave;
    const int64_t entry0 = (local * kWavesPerBlock + wave) * k;
    const int64_t left = p.med_count[hop] - entry0;
    if (left <= 0) return;
    const int n_here = left < k ? (int)left : k;
    const int NS = SUM ? n_sel : 1;
    int row = 0;
    int64_t b0 = 0, b1 = 0;
    {
        const int j = lane / NS, s = lane - j * NS;
        if (j < n_here) {
            row = __builtin_nontemporal_load(p.med_list[hop] + entry0 + j);
            const int64_t* rp = p.hop[SUM ? s : hop].rowptr + row;
            b0 = rp[0];
            b1 = rp[1];
        }
    }
    const int b0_lo = (int)(b0 & 0xffffffff), b0_hi = (int)(b0 >> 32), b1_lo = (int)(b1 & 0xffffffff), b1_hi = (int)(b1 >> 32);
    auto bounds = [&](int l, int64_t& sb, int64_t& se) {
        sb = ((int64_t)__builtin_amdgcn_readlane(b0_hi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane(b0_lo, l);
        se = ((int64_t)__builtin_amdgcn_readlane(b1_hi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane(b1_lo, l);
    };
    // (Fetching the first index chunk of segment i+1 before segment i gathers -- the tile walk's PIPE trick -- was tried here:
    // products_tail +1.5-2 %, h2gcn_like +0.5 %, but bimodal -4 to -10 %; not kept, profiles/r04_ab_short_walks.txt.)
    for (int j = 0; j < n_here; ++j) {
        const int64_t orow = __builtin_amdgcn_readlane(row, j * NS);
        float acc[NP][VEC];
        zero_acc<VEC, NP>(acc);
        for (int s = 0; s < NS; ++s) {
            int64_t sb, se;
            bounds(j * NS + s, sb, se);
            const int hs = SUM ? s : hop;
            const HopCsr& h = p.hop[hs];
            const GatherAddr<OFF32> addr{reinterpret_cast<const char*>(p.src + p.src_hop_off[hs] + src_col_begin - VEC), (off_t)lane_off0,
                                         (off_t)(p.ld_src * 4)};
            accumulate_segment<VEC, LPR, false, OFF32, MAXB>(h.colidx, h.vals, sb, se, 0, 1, addr, lane, true, acc);
        }
        float tot[VEC];
        fold_tree<VEC, LPR, NP>(acc, tot);
        if (g == 0) store_out<VEC, EPI>(p, p.dst + orow * p.ld_dst + (SUM ? 0 : p.dst_hop_off[hop]), lcol, ecol, tot);
    }
}

// ---- in-tile short-row mode (SHORT kernels): the tile walk of a wave whose rows are short ------------------------------
// Rounds of G consecutive rows (one hop) whose segments are all <= LPR nonzeros are served one lane group per segment
// (accumulate_grouped), with the next round's index fetch issued before the current round gathers; other rounds fall back to
// the wave-per-segment walk with load batches of FB.  (Instantiated in spmm_short.hip only -- see there.)
template <int VEC, int LPR, bool SUM, bool OFF32, bool EPI, int FB>
__device__ __forceinline__ void in_tile_short_rows(const LaunchParams& p, int lane, int rpw, int64_t row0, int rows_here, int rp_lo, int rp_hi,
                                                   int64_t lane_off0_, int64_t src_col_begin, int lcol, int ecol) {
    using off_t = typename std::conditional<OFF32, uint32_t, int64_t>::type;
    constexpr int NP = Tree<LPR>::NP;
    constexpr int kBias = VEC;
    const int li = lane % LPR, g = lane / LPR;
    const int n_sel = p.n_sel;
    const off_t lane_off0 = (off_t)lane_off0_;
    auto seg_bound = [&](int l) -> int64_t {
        return ((int64_t)__builtin_amdgcn_readlane(rp_hi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane(rp_lo, l);
    };
    {
        constexpr int G = kWave / LPR;
        const int short_max = min(LPR, p.long_threshold - 1);
        const int n_blocks = (rows_here + G - 1) / G;       // row blocks of G consecutive rows
        const GatherAddr<OFF32> addr0{nullptr, lane_off0, (off_t)(p.ld_src * 4)};
        // wave-per-segment walk of one row (all hops in SUM mode, hop `s_only` otherwise) -- the general path
        auto row_wave_wide = [&](int r, int s_first, int s_last) {
            float acc[NP][VEC];
            zero_acc<VEC, NP>(acc);
            if constexpr (SUM) {
                for (int s = s_first; s < s_last; ++s) {
                    const int l0 = s * (rpw + 1) + r;
                    if (seg_bound(l0 + 1) - seg_bound(l0) >= p.long_threshold) return;  // row owned by the long path
                }
            }
            for (int s = s_first; s < s_last; ++s) {
                const int l0 = s * (rpw + 1) + r;
                const int64_t sb = seg_bound(l0), se = seg_bound(l0 + 1);
                if (!SUM && se - sb >= p.long_threshold) continue;
                const HopCsr& h = p.hop[s];
                GatherAddr<OFF32> addr = addr0;
                addr.base = reinterpret_cast<const char*>(p.src + p.src_hop_off[s] + src_col_begin - kBias);
                accumulate_segment<VEC, LPR, false, OFF32, FB>(h.colidx, h.vals, sb, se, 0, 1, addr, lane, true, acc);
                if constexpr (!SUM) {
                    float tot[VEC];
                    fold_tree<VEC, LPR, NP>(acc, tot);
                    if (g == 0) store_out<VEC, EPI>(p, p.dst + (row0 + r) * p.ld_dst + p.dst_hop_off[s], lcol, ecol, tot);
                    zero_acc<VEC, NP>(acc);
                }
            }
            if constexpr (SUM) {
                float tot[VEC];
                fold_tree<VEC, LPR, NP>(acc, tot);
                if (g == 0) store_out<VEC, EPI>(p, p.dst + (row0 + r) * p.ld_dst, lcol, ecol, tot);
            }
        };
        // (begin, length) of the segment (hop s, row b*G + g) for this lane's group; length -1 for rows beyond the tile
        auto my_segment = [&](int s, int blk, int64_t& begin, int& len) {
            const int r = blk * G + g;
            const int l0 = s * (rpw + 1) + min(r, rpw - 1);
            const int64_t sb = ((int64_t)lane_gather(rp_hi, l0) << 32) | (uint32_t)lane_gather(rp_lo, l0);
            const int64_t se = ((int64_t)lane_gather(rp_hi, l0 + 1) << 32) | (uint32_t)lane_gather(rp_lo, l0 + 1);
            begin = sb;
            const int64_t n = se - sb;
            len = r < rows_here ? (int)(n > 0x7fffffff ? 0x7fffffff : n) : -1;
        };
        if constexpr (!SUM) {
            // rounds (hop s, row block blk) in order; the index fetch of round i+1 is issued before round i gathers
            // (loads retire in order, so it has landed by the time round i's gathers have) -- one memory latency per
            // round instead of two
            const int n_rounds = n_sel * n_blocks;
            int64_t begin_n = 0;
            int len_n = -1, c_n = 0;
            float v_n = 0.f;
            bool short_n = false;
            auto fetch = [&](int i) {
                const int s_ = i / n_blocks, blk_ = i - s_ * n_blocks;
                my_segment(s_, blk_, begin_n, len_n);
                short_n = __builtin_amdgcn_ballot_w64(len_n > short_max) == 0;
                c_n = 0;
                v_n = 0.f;
                if (short_n && li < len_n) {
                    c_n = __builtin_nontemporal_load(p.hop[s_].colidx + begin_n + li);
                    v_n = __builtin_nontemporal_load(p.hop[s_].vals + begin_n + li);
                }
            };
            if (n_rounds > 0) fetch(0);
            for (int i = 0; i < n_rounds; ++i) {
                const int s = i / n_blocks, blk = i - s * n_blocks;
                const int len = len_n, c = c_n;
                const float v = v_n;
                const bool is_short = short_n;
                if (i + 1 < n_rounds) fetch(i + 1);
                if (is_short) {
                    int n_max = 0;
#pragma unroll
                    for (int gg = 0; gg < G; ++gg) n_max = max(n_max, __builtin_amdgcn_readlane(len, gg * LPR));
                    float part[kTreeParts][VEC];
                    zero_acc<VEC, kTreeParts>(part);
                    GatherAddr<OFF32> addr = addr0;
                    addr.base = reinterpret_cast<const char*>(p.src + p.src_hop_off[s] + src_col_begin - kBias);
                    accumulate_grouped<VEC, LPR, OFF32>(c, v, len, n_max, lane, addr, part);
                    float tot[VEC];
                    combine_partials<VEC>(part, tot);
                    if (len >= 0) store_out<VEC, EPI>(p, p.dst + (row0 + blk * G + g) * p.ld_dst + p.dst_hop_off[s], lcol, ecol, tot);
                } else {
                    for (int gg = 0; gg < G && blk * G + gg < rows_here; ++gg) row_wave_wide(blk * G + gg, s, s + 1);
                }
            }
        } else {
            // SUM mode: a row block is served group-per-row only if the segments of ALL selected hops are short; the
            // partials run on across the hops exactly as the wave-per-segment walk's accumulators do.  Index fetches
            // are issued one segment ahead (see the forward case).
            auto block_short = [&](int blk_) {
                bool ok = true;
                for (int s_ = 0; s_ < n_sel; ++s_) {
                    int64_t b_;
                    int l_;
                    my_segment(s_, blk_, b_, l_);
                    ok = ok && __builtin_amdgcn_ballot_w64(l_ > short_max) == 0;
                }
                return ok;
            };
            int64_t begin_n = 0;
            int len_n = -1, c_n = 0;
            float v_n = 0.f;
            auto fetch = [&](int blk_, int s_, bool short_blk) {
                my_segment(s_, blk_, begin_n, len_n);
                c_n = 0;
                v_n = 0.f;
                if (short_blk && li < len_n) {
                    c_n = __builtin_nontemporal_load(p.hop[s_].colidx + begin_n + li);
                    v_n = __builtin_nontemporal_load(p.hop[s_].vals + begin_n + li);
                }
            };
            bool cur_short = n_blocks > 0 && block_short(0);
            if (n_blocks > 0) fetch(0, 0, cur_short);
            for (int blk = 0; blk < n_blocks; ++blk) {
                const bool nxt_short = blk + 1 < n_blocks && block_short(blk + 1);
                if (cur_short) {
                    float part[kTreeParts][VEC];
                    zero_acc<VEC, kTreeParts>(part);
                    bool valid = false;
                    for (int s = 0; s < n_sel; ++s) {
                        const int len = len_n, c = c_n;
                        const float v = v_n;
                        valid = len >= 0;
                        if (s + 1 < n_sel) fetch(blk, s + 1, true);
                        else if (blk + 1 < n_blocks) fetch(blk + 1, 0, nxt_short);
                        int n_max = 0;
#pragma unroll
                        for (int gg = 0; gg < G; ++gg) n_max = max(n_max, __builtin_amdgcn_readlane(len, gg * LPR));
                        GatherAddr<OFF32> addr = addr0;
                        addr.base = reinterpret_cast<const char*>(p.src + p.src_hop_off[s] + src_col_begin - kBias);
                        accumulate_grouped<VEC, LPR, OFF32>(c, v, len, n_max, lane, addr, part);
                    }
                    float tot[VEC];
                    combine_partials<VEC>(part, tot);
                    if (valid) store_out<VEC, EPI>(p, p.dst + (row0 + blk * G + g) * p.ld_dst, lcol, ecol, tot);
                } else {
                    if (blk + 1 < n_blocks) fetch(blk + 1, 0, nxt_short);
                    for (int gg = 0; gg < G && blk * G + gg < rows_here; ++gg) row_wave_wide(blk * G + gg, 0, n_sel);
                }
                cur_short = nxt_short;
            }
        }
        return;
    }
}

// Instantiations that need more than 80 VGPRs to stay out of scratch (tools/kernel_resources.py): 64-bit gather offsets
// (two address registers per load in flight) together with a general store, a 128 / 256-column slice or the index prefetch;
// the prefetching walk on 128-column slices.  They run at H2GCN_OFF64_HEAVY_MIN_WAVES (5) waves per SIMD.
template <int VEC, int LPR, bool SUM, bool OFF32, bool PIPE, bool SHORT, bool EPI>
constexpr bool heavy_registers() {
    if (VEC != 4) return false;
    if (SHORT) return (SUM && (!OFF32 || LPR >= 32)) || (!OFF32 && LPR >= 32);   // (the list-driven kernels at FB = 8)
    if (!OFF32) return EPI || LPR >= 32 || PIPE;
    return PIPE && LPR == 32;
}

// VEC    floats per lane per gathered row (4 on the fast paths)
// LPR    lanes that cover one gathered row
// EXACT  the launch covers the feature columns in n_slices = ceil(d / (VEC*LPR)) slices of VEC*LPR columns,
//        slice-major (all row tiles of slice 0, then slice 1, ...: while a slice is being processed the gather
//        working set is n_cols * slice_cols * 4 bytes, which is what the 256 MiB Infinity Cache sees).  A last
//        slice that sticks out beyond the source's valid columns costs no predication: the lanes beyond re-read the
//        last valid float4 of the row (same cache line, no extra traffic) and simply do not store.  Otherwise (!EXACT)
//        LPR == 64 and each wave loops over masked column tiles (any d, any alignment)
// SUM    adjoint mode: one output row = sum over the selected hops
// OFF32  32-bit gather offsets (see GatherAddr)
// SHORT  in-tile short-row mode, for launches whose segments are short throughout (mean < 16 nonzeros): the tile walk, but
//        rounds of G CONSECUTIVE rows whose segments of one hop are all <= LPR nonzeros are served one lane group per
//        segment (accumulate_grouped); other rounds fall back to the wave-per-segment walk.  Row pointers stay one coalesced
//        load per wave and a row's hop outputs leave the same wave -- the fastest walk when every row is short.
// LISTS  list-driven launch, CSR-adaptive by segment class, for MIXED launches (see short_list_blocks / medium_list_blocks):
//        the short segments come from the plan's binned list, one lane group per segment wherever they sit; the medium ones
//        from their list, one wave per segment; the long ones from theirs, one workgroup per segment -- no tile walk.
//        Same bits as every other walk.
// EPI    general store (see store_out): optional bias / ReLU epilogue, element-wise bounded stores for odd widths and
//        unaligned outputs (separate instantiations: the extra registers would otherwise push the 6-waves-per-SIMD
//        variants of the plain aggregation into spilling)
// FB     deepest load batch of the wave-per-segment walk inside the SHORT kernels (their fallback) and the LISTS kernels
//        (medium / long walks): 4 keeps them at 7-8 waves per SIMD (memory-resident operands dominated by short segments:
//        occupancy buys bandwidth), 8 otherwise (cache-resident operands / launches dominated by longer segments)
template <int VEC, int LPR, bool EXACT, bool SUM, bool OFF32, bool PIPE = false, bool SHORT = false, bool EPI = false, int FB = 8, bool LISTS = false>
__global__ __launch_bounds__(kBlock, EXACT ? (LISTS ? (FB == 4 ? H2GCN_SHORT_FB4_MIN_WAVES : heavy_registers<VEC, LPR, SUM, OFF32, PIPE, true, EPI>() ? H2GCN_OFF64_HEAVY_MIN_WAVES : (LPR >= 32 ? H2GCN_WIDE_MIN_WAVES : kMinWavesPerSimd))
                                                : SHORT ? (FB == 4 && !SUM ? H2GCN_SHORT_FB4_MIN_WAVES : H2GCN_SHORT_MIN_WAVES)
                                                : heavy_registers<VEC, LPR, SUM, OFF32, PIPE, false, EPI>() ? H2GCN_OFF64_HEAVY_MIN_WAVES
                                                : (LPR >= 32 ? H2GCN_WIDE_MIN_WAVES : kMinWavesPerSimd)) : 2) void spmm_hops_kernel(const LaunchParams p) {
    static_assert(!LISTS || (EXACT && !SHORT && !PIPE && (LPR == 16 || LPR == 32)), "list-driven launches exist for 64- and 128-column slices");
    static_assert(EXACT || LPR == kWave, "column-tiled path uses the whole wave per row");
    __shared__ float partial[kWavesPerBlock][kMaxTileCols];
    using off_t = typename std::conditional<OFF32, uint32_t, int64_t>::type;
    constexpr int NP = Tree<LPR>::NP;
    constexpr int kMainB = LISTS ? FB : (LPR >= 32 ? H2GCN_WIDE_MAXB : H2GCN_MAIN_MAXB);

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = wave_uniform(threadIdx.x >> 6);
    const int li = lane % LPR;  // lane inside its group
    const int g = lane / LPR;
    const int n_sel = p.n_sel;

    // slice-major block order
    const int64_t bid = blockIdx.x;
    const int slice = EXACT ? (int)(bid / p.blocks_per_slice) : 0;
    const int64_t b = EXACT ? bid - (int64_t)slice * p.blocks_per_slice : bid;
    const int col_begin = EXACT ? slice * (VEC * LPR) : 0;
    const int col_end = EXACT ? col_begin + VEC * LPR : p.d;
    const int64_t src_col_begin = EXACT ? (int64_t)slice * p.src_slice_stride : 0;  // where this slice starts in the source
    // EXACT: this lane's nominal columns are lcol .. lcol+VEC-1.  A lane whose vector would stick out beyond the readable
    // width of the source row gathers the row's LAST VEC columns instead (ecol = d_src - VEC: same cache line, no extra
    // traffic, no predication in the gather loop).  For the lane that straddles the end (width not a multiple of VEC) those
    // columns overlap its left neighbour's and include the ones it owns; lanes entirely beyond the end own nothing.
    const int lcol = col_begin + li * VEC;
    const int ecol = EXACT ? ((lcol + VEC <= p.d_src) ? lcol : p.d_src - VEC) : lcol;
    // byte offset of ecol inside the slice, biased by VEC elements so that it is never negative (the tail of a slice that
    // holds fewer than VEC columns reaches back into the previous slice); the wave-uniform base is lowered accordingly
    const off_t lane_off0 = (off_t)((ecol - col_begin + VEC) * 4);
    constexpr int kBias = EXACT ? VEC : 0;

    if (b < p.n_long) {
        // ---- long segment: the 4 waves of this workgroup share one (row, hop) [forward] / one row [SUM] ----
        const int64_t entry = p.long_list[b];
        const int64_t row = SUM ? entry : (entry >> 4);
        const int s_first = SUM ? 0 : (int)(entry & 15);
        const int s_last = SUM ? n_sel : s_first + 1;
        for (int col0 = col_begin; col0 < col_end; col0 += VEC * LPR) {
            const bool lane_active = EXACT || (col0 + li * VEC < p.d);
            float acc[NP][VEC];
            zero_acc<VEC, NP>(acc);
            for (int s = s_first; s < s_last; ++s) {
                const HopCsr& h = p.hop[s];
                const int64_t sb = h.rowptr[row], se = h.rowptr[row + 1];
                const GatherAddr<OFF32> addr{reinterpret_cast<const char*>(p.src + p.src_hop_off[s] + src_col_begin + (col0 - col_begin) - kBias),
                                             EXACT ? lane_off0 : (off_t)(li * VEC * 4), (off_t)(p.ld_src * 4)};
                accumulate_segment<VEC, LPR, !EXACT, OFF32, kMainB>(h.colidx, h.vals, sb, se, wave, kWavesPerBlock, addr, lane,
                                                            lane_active, acc);
            }
            float tot[VEC];
            fold_tree<VEC, LPR, NP>(acc, tot);
            if (g == 0) {
                // this lane's sums belong to columns ecol .. ecol+VEC-1 (tile-relative index below); a tail lane overlaps its
                // left neighbour with identical values, lanes entirely beyond the row's end contribute nothing
                const int rel = EXACT ? ecol - col_begin : li * VEC;
                if (!EXACT || lcol < p.d_src) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        if (rel + i >= 0) partial[wave][rel + i] = tot[i];
                }
            }
            __syncthreads();
            // fixed-order sum of the 4 wave totals; thread c owns column col0 + c
            const int c = threadIdx.x;
            if (c < VEC * LPR && col0 + c < p.d) {
                float t = partial[0][c];
#pragma unroll
                for (int w = 1; w < kWavesPerBlock; ++w) t += partial[w][c];
                const int64_t off = row * p.ld_dst + (SUM ? 0 : p.dst_hop_off[s_first]) + col0 + c;
                if constexpr (EPI) {
                    if (p.accumulate) t += p.dst[off];
                    if (p.bias) t += p.bias[col0 + c];
                    if (p.relu) t = fmaxf(t, 0.f);
                }
                __builtin_nontemporal_store(t, p.dst + off);
            }
            if (!EXACT) __syncthreads();
        }
        return;
    }

    // ---- regular path: XCD-aware tile map, each wave walks rows_per_wave consecutive rows ----
    const int64_t tb = b - p.n_long;
    int64_t tile;
    if constexpr (LISTS) {
        // list-driven launch: workgroups come in groups of 8 (one per XCD); short_groups of them serve the binned short
        // segments and are spread evenly among the med_groups groups that walk the medium ones, so that latency-bound
        // lane-group waves and bandwidth-bound wave-per-segment waves are resident together.  No tile walk.
        __shared__ uint64_t s_seg[kWavesPerBlock][kShortSumHops][kWave];
        __shared__ int32_t s_row[kWavesPerBlock][kWave];
        const int64_t q = tb >> 3;
        const int i8 = (int)(tb & 7);
        const int64_t n_groups = p.short_groups + p.med_groups;
        const int64_t s_before = (q * p.short_groups) / n_groups;
        if (((q + 1) * p.short_groups) / n_groups > s_before)
            short_list_blocks<VEC, LPR, SUM, OFF32, EPI, H2GCN_SHORT_PREFETCH>(p, s_before * kNumXcd + i8, lane, wave, (int64_t)lane_off0,
                                                                               src_col_begin, lcol, ecol, s_seg, s_row);
        else
            medium_list_blocks<VEC, LPR, SUM, OFF32, EPI, FB>(p, (q - s_before) * kNumXcd + i8, lane, wave, (int64_t)lane_off0,
                                                              src_col_begin, lcol, ecol);
        return;
    } else {
        tile = (tb % kNumXcd) * p.tiles_per_xcd + tb / kNumXcd;
    }
    if (tile >= p.n_tiles) return;
    const int rpw = p.rows_per_wave;
    const int64_t row0 = (tile * kWavesPerBlock + wave) * rpw;
    if (row0 >= p.n_rows) return;
    const int64_t rows_here_l = p.n_rows - row0;
    const int rows_here = rows_here_l < rpw ? (int)rows_here_l : rpw;

    // one wave-wide load fetches every row pointer this wave needs: lane l -> hop l/(rpw+1), row l%(rpw+1)
    int64_t rp = 0;
    {
        const int hs = lane / (rpw + 1), r = lane % (rpw + 1);
        if (hs < n_sel && r <= rows_here) rp = p.hop[hs].rowptr[row0 + r];
    }
    const int rp_lo = (int)(rp & 0xffffffff), rp_hi = (int)(rp >> 32);
    auto seg_bound = [&](int l) -> int64_t {
        return ((int64_t)__builtin_amdgcn_readlane(rp_hi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane(rp_lo, l);
    };

    if constexpr (SHORT && EXACT && (kWave / LPR == 2 || kWave / LPR == 4)) {
        in_tile_short_rows<VEC, LPR, SUM, OFF32, EPI, FB>(p, lane, rpw, row0, rows_here, rp_lo, rp_hi, (int64_t)lane_off0, src_col_begin, lcol, ecol);
        return;
    }
    if constexpr (PIPE && EXACT) {
        // ---- software-pipelined walk over the wave's (row, hop) segments: prefetch the next segment's first
        //      index chunk before gathering the current one ----
        const int n_seg = rows_here * n_sel;
        uint32_t skip = 0;  // bit q: segment q belongs to a workgroup of the long path
        for (int q = 0; q < n_seg; ++q) {
            const int r = q / n_sel, s_ = q - r * n_sel;
            const int l0 = s_ * (rpw + 1) + r;
            if (seg_bound(l0 + 1) - seg_bound(l0) >= p.long_threshold) skip |= SUM ? (((1u << n_sel) - 1u) << (r * n_sel)) : (1u << q);
        }
        auto seg = [&](int q, int& s_, int64_t& sb, int64_t& se) {
            const int r = q / n_sel;
            s_ = q - r * n_sel;
            const int l0 = s_ * (rpw + 1) + r;
            sb = seg_bound(l0);
            se = (skip >> q) & 1u ? sb : seg_bound(l0 + 1);
        };
        int s_cur, s_nxt = 0;
        int64_t b_cur, e_cur, b_nxt = 0, e_nxt = 0;
        int c_cur, c_nxt = 0;
        float v_cur, v_nxt = 0.f;
        seg(0, s_nxt, b_nxt, e_nxt);
        load_chunk(p.hop[s_nxt].colidx, p.hop[s_nxt].vals, b_nxt, e_nxt, lane, c_nxt, v_nxt);
        float acc[NP][VEC];
        zero_acc<VEC, NP>(acc);
        for (int q = 0; q < n_seg; ++q) {
            s_cur = s_nxt; b_cur = b_nxt; e_cur = e_nxt; c_cur = c_nxt; v_cur = v_nxt;
            if (q + 1 < n_seg) {
                seg(q + 1, s_nxt, b_nxt, e_nxt);
                load_chunk(p.hop[s_nxt].colidx, p.hop[s_nxt].vals, b_nxt, e_nxt, lane, c_nxt, v_nxt);
            }
            const int r = q / n_sel;
            const int64_t row = row0 + r;
            const HopCsr& h = p.hop[s_cur];
            const GatherAddr<OFF32> addr{reinterpret_cast<const char*>(p.src + p.src_hop_off[s_cur] + src_col_begin - kBias),
                                         lane_off0, (off_t)(p.ld_src * 4)};
            accumulate_segment_prefetched<VEC, LPR, OFF32>(h.colidx, h.vals, b_cur, e_cur, c_cur, v_cur, addr, lane, acc);
            const bool skipped = (skip >> q) & 1u;
            if (!SUM || s_cur == n_sel - 1) {
                if (!skipped) {
                    float tot[VEC];
                    fold_tree<VEC, LPR, NP>(acc, tot);
                    if (g == 0) store_out<VEC, EPI>(p, p.dst + row * p.ld_dst + (SUM ? 0 : p.dst_hop_off[s_cur]), lcol, ecol, tot);
                }
                zero_acc<VEC, NP>(acc);
            }
        }
        return;
    }

#ifdef H2GCN_AB_STORE_BEHIND_FETCH
    // A/B builds only (profiles/r06_ab_forward_vs_adjoint_x6.txt): the forward wave walk with a segment's store issued BEHIND the
    // next segment's index fetch instead of in front of it.  The wave waits for everything outstanding before it fetches a
    // segment's indices (the compiler's vmcnt(0) at the loop head), so a store issued right after the fold is a write round trip
    // the wave sits out; issued after the next index fetch it completes under that fetch and the first gather batch.
    if constexpr (EXACT && !SUM && !EPI) {
        float held[VEC];
        int64_t held_off = 0;
        bool held_valid = false;
        for (int r = 0; r < rows_here; ++r) {
            const int64_t row = row0 + r;
            for (int s = 0; s < n_sel; ++s) {
                const int l0 = s * (rpw + 1) + r;
                const int64_t sb = seg_bound(l0), se = seg_bound(l0 + 1);
                if (se - sb >= p.long_threshold) continue;  // a workgroup of the long path owns it
                const HopCsr& h = p.hop[s];
                int c;
                float v;
                load_chunk(h.colidx, h.vals, sb, se, lane, c, v);
                if (held_valid) {
                    if (g == 0) store_out<VEC, EPI>(p, p.dst + held_off, lcol, ecol, held);
                    held_valid = false;
                }
                float acc[NP][VEC];
                zero_acc<VEC, NP>(acc);
                const GatherAddr<OFF32> addr{reinterpret_cast<const char*>(p.src + p.src_hop_off[s] + src_col_begin - kBias), lane_off0, (off_t)(p.ld_src * 4)};
                accumulate_segment_prefetched<VEC, LPR, OFF32, kMainB>(h.colidx, h.vals, sb, se, c, v, addr, lane, acc);
                fold_tree<VEC, LPR, NP>(acc, held);
                held_off = row * p.ld_dst + p.dst_hop_off[s];
                held_valid = true;
            }
        }
        if (held_valid && g == 0) store_out<VEC, EPI>(p, p.dst + held_off, lcol, ecol, held);
        return;
    }
#endif

    for (int r = 0; r < rows_here; ++r) {
        const int64_t row = row0 + r;
        if constexpr (SUM) {
            bool skip_row = false;  // a workgroup of the long path owns rows with any long segment
            for (int s = 0; s < n_sel; ++s) {
                const int l0 = s * (rpw + 1) + r;
                const int64_t len = seg_bound(l0 + 1) - seg_bound(l0);
                if (len >= p.long_threshold) skip_row = true;
            }
            if (skip_row) continue;
        }
        for (int col0 = col_begin; col0 < col_end; col0 += VEC * LPR) {
            const bool lane_active = EXACT || (col0 + li * VEC < p.d);
            float acc[NP][VEC];
            zero_acc<VEC, NP>(acc);
#ifdef H2GCN_DEFER_HOP_STORES   // A/B builds only: a row's first hop total is held until the second is ready, so that the
            float held[VEC];            // two pieces of Y[row, :, slice] leave back to back (profiles/r04_ab_output_stores.txt)
            int held_s = -1;
#endif
            for (int s = 0; s < n_sel; ++s) {
                const int l0 = s * (rpw + 1) + r;
                const int64_t sb = seg_bound(l0), se = seg_bound(l0 + 1);
                if (!SUM && se - sb >= p.long_threshold) continue;  // a workgroup of the long path owns it
                const HopCsr& h = p.hop[s];
                const GatherAddr<OFF32> addr{reinterpret_cast<const char*>(p.src + p.src_hop_off[s] + src_col_begin + (col0 - col_begin) - kBias),
                                             EXACT ? lane_off0 : (off_t)(li * VEC * 4), (off_t)(p.ld_src * 4)};
                accumulate_segment<VEC, LPR, !EXACT, OFF32, kMainB>(h.colidx, h.vals, sb, se, 0, 1, addr, lane, lane_active, acc);
                if constexpr (!SUM) {
                    float tot[VEC];
                    fold_tree<VEC, LPR, NP>(acc, tot);
#ifdef H2GCN_DEFER_HOP_STORES
                    if (EXACT && n_sel == 2 && s == 0) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) held[i] = tot[i];
                        held_s = 0;
                    } else {
                        if (held_s >= 0 && g == 0 && lane_active)
                            store_out<VEC, EPI>(p, p.dst + row * p.ld_dst + p.dst_hop_off[held_s], lcol, ecol, held);
                        held_s = -1;
                        if (g == 0 && lane_active)
                            store_out<VEC, EPI>(p, p.dst + row * p.ld_dst + p.dst_hop_off[s], EXACT ? lcol : col0 + li * VEC, EXACT ? ecol : col0 + li * VEC, tot);
                    }
#else
                    if (g == 0 && lane_active)
                        store_out<VEC, EPI>(p, p.dst + row * p.ld_dst + p.dst_hop_off[s], EXACT ? lcol : col0 + li * VEC, EXACT ? ecol : col0 + li * VEC, tot);
#endif
                    zero_acc<VEC, NP>(acc);
                }
            }
#ifdef H2GCN_DEFER_HOP_STORES
            if constexpr (!SUM)
                if (held_s >= 0 && g == 0 && lane_active)
                    store_out<VEC, EPI>(p, p.dst + row * p.ld_dst + p.dst_hop_off[held_s], lcol, ecol, held);
#endif
            if constexpr (SUM) {
                float tot[VEC];
                fold_tree<VEC, LPR, NP>(acc, tot);
                if (g == 0 && lane_active)
                    store_out<VEC, EPI>(p, p.dst + row * p.ld_dst, EXACT ? lcol : col0 + li * VEC, EXACT ? ecol : col0 + li * VEC, tot);
            }
        }
    }
}

// Gather source -> slice-major scratch.  Source element (row r, hop s, column c) = x[r*ld + s*ld_hop + c]
// (forward: one "hop", the embedding X; adjoint: the stacked gradient dY[r, s, :]); scratch
// W[q][s][r][cc] with q = c / slice_cols, cc = c % slice_cols, n_slices = ceil(d / slice_cols); columns beyond d are
// written as zeros, so every block is a dense, 16-byte addressable, cache-line aligned [n_rows, slice_cols] matrix.
// Used in front of a launch
//   * when the row stride of X is a multiple of 1 KiB: gathering a 256-byte slice out of such rows leaves address
//     bits 8-9 constant during a whole slice pass and the L2 / Infinity Cache index only a quarter of their sets
//     (d = 256: 0.76 of the roofline row-major, 0.91 slice-major);
//   * when the rows of the source are not cache-line aligned and wide (d = 132, 200, 300 ...): every 256-byte block of
//     the copy is line-aligned, so a gather touches ceil(d*4/128) lines instead of one more, and the slices are
//     cache-sized;
//   * when d % 4 != 0 or the source is not 16-byte addressable (raw feature widths: Cora F = 1433, citeseer 3703;
//     reference accepts any b.shape[1], _layers.py:62-76): the float4 gather kernels then run on the padded copy
//     (VEC4 = false: element-wise reads) instead of the generic column-tiled kernel.
template <bool VEC4>
__global__ void repack_slice_major_kernel(const float* __restrict__ x, int64_t ld, int64_t ld_hop, int n_hop, int64_t n_rows,
                                          int d, int n_slices, int slice_cols, float* __restrict__ w) {
    using f4 = float __attribute__((ext_vector_type(4)));
    constexpr int E = VEC4 ? 4 : 1;             // floats per thread
    const int cu = slice_cols / E;              // units per scratch row
    const int64_t per_hop = n_rows * cu;
    const int64_t per_slice = per_hop * n_hop;
    const int64_t total = per_slice * n_slices;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i / per_slice);
        const int64_t i1 = i - q * per_slice;
        const int s = (int)(i1 / per_hop);
        const int64_t i2 = i1 - s * per_hop;
        const int64_t r = i2 / cu;
        const int c = (int)(i2 - r * cu);
        const int col = q * slice_cols + c * E;
        const float* src = x + r * ld + s * ld_hop + col;
        if constexpr (VEC4) {
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (col < d) v = *reinterpret_cast<const f4*>(src);   // d % 4 == 0 here
            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(w) + i);
        } else {
            __builtin_nontemporal_store(col < d ? *src : 0.f, w + i);
        }
    }
}

}  // namespace h2gcn
