// rings.hip -- exact-k-hop neighbourhood rings on the device: the operand construction that FEEDS the hop
// aggregation (reference `TransformSPAdj.nhoodSplit`, h2gcn/datasets/_dataset.py:138-158, and `.normalize`,
// :109-124).
//
// The reference grows reachability with a host SpGEMM, `mt <- bin(mt @ (A + I))`, and takes `mt_k - mt_{k-1}`.  Here
// a ring is computed row by row as a SET expression over CSR patterns (no values are multiplied -- the operation is
// boolean):
//
//     out[i] = ( U_{j in F[i]} A[j]  U  U_p ADD_p[i]  U  ({i} if add_diag) )  \  ( U_q SUB_q[i]  U  ({i} if sub_diag) )
//
//   ring_k  = expand frontier F = ring_{k-1} through A, subtract ring_0 .. ring_{k-1} (ring_0 = I: sub_diag)
//   a merged --adj_nhood group such as "0,1" = no expansion, ADD = the member rings.
//
// ROW WINDOWS (row-partitioned runs): the expression is evaluated for the rows [row_begin, row_begin +
// n_rows) only.  Everything indexed by the output row -- F, ADD, SUB, the result -- is then a CSR of that window (local
// row pointers), while A is the full matrix (indexed by the global ids the frontier names) and `i` in {i} is the global
// row id.  A rank of a P-way partition thus builds 1/P of every ring from the full A and its own rows of the lower
// rings: time and memory O(ring / P) instead of the build-everything-then-slice of the reference's host path
// (_dataset.py:147-157).
//
// One workgroup (256 threads; a single wave when level 0 is in global memory) owns one output row at a time (rows are handed out by an atomic ticket, longest-first is not needed:
// the work per row is bounded by its candidate count).  The row's candidate set lives in a TWO-LEVEL BITMAP:
//   level 0: one bit per column (n bits) -- in LDS when n <= kLdsBitmapCols, else in a per-workgroup slab of global
//            scratch that stays L2-resident;
//   level 1: one bit per level-0 word, always in LDS.
// Marking is an LDS/L2 atomic OR per candidate (the expansion reads A's rows coalesced, one wave per frontier node);
// subtraction clears bits; emission walks level 1, pops the set level-0 words in ascending column order (block-wide
// prefix sum of the per-thread popcounts) and clears them on the way, so the bitmap is zero again for the next row.
// Output columns therefore come out sorted -- the canonical order `tf.sparse.reorder` establishes
// (_dataset.py:535) -- and every output element is written exactly once (two passes: count -> scan -> fill).
// The kernels are integer/bit work bound by LDS atomics and the streaming reads of A: no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "capi_internal.h"
#include "h2gcn_hip.h"

namespace {

using h2gcn::fail;

constexpr int kThreadsLds = 256;     // workgroup size when level 0 lives in LDS (dense rows relative to n)
constexpr int kThreadsGlobal = 64;   // one wave per row when level 0 lives in global slabs: rows are sparse relative to n,
                                     // latency-bound, and 4x more rows in flight per CU matter more than lanes per row
constexpr int64_t kLdsBitmapCols = 1 << 17;  // level 0 in LDS up to this many columns (16 KiB: 8 workgroups per CU);
                                             // beyond that rows are sparse relative to n and occupancy matters more
                                             // than LDS atomics: level 0 moves to L2-resident global slabs
constexpr int kSlabsPerCu = 32;
constexpr size_t kMaxSlabBytes = (size_t)2 << 30;  // cap of the level-0 scratch (fewer workgroups beyond it)
constexpr int kMaxPatterns = 8;

struct Pattern {
    const int64_t* rowptr;
    const int32_t* colidx;
};

struct RingParams {
    int64_t n;            // columns (= rows of A)
    int64_t n_rows;       // rows of the window this launch evaluates
    int64_t row_begin;    // global id of the window's first row
    Pattern a;            // expansion operand A (self loops already removed); unused when frontier.rowptr == NULL
    Pattern frontier;     // F: rows whose A-rows are united (NULL rowptr = no expansion)
    Pattern add[kMaxPatterns];
    Pattern sub[kMaxPatterns];
    int n_add, n_sub;
    int add_diag, sub_diag;
    int64_t* counts;          // pass 1: counts[i] = |out[i]|
    const int64_t* out_rowptr;  // pass 2
    int32_t* out_colidx;        // pass 2
    unsigned int* ticket;     // row dispenser
    uint32_t* l0_scratch;     // level-0 slabs in global memory (gridDim.x * l0_words), zero-initialised; NULL = LDS
    int64_t l0_words;         // ceil(n / 32)
    int64_t l1_words;         // ceil(l0_words / 32)
    int only_big_rows;        // bitmap kernel: serve only the rows listed in big_rows (more than kSortCap candidates;
                              // the sorted-candidate kernel has served the others and appended these)
    unsigned int* big_count;  // number of entries of big_rows
    int32_t* big_rows;        // [n]
};

template <bool L0_LDS>
__device__ __forceinline__ void mark(uint32_t* l0, uint32_t* l1, int32_t c) {
    const uint32_t w = (uint32_t)c >> 5, bit = 1u << (c & 31);
    if constexpr (L0_LDS) {
        atomicOr(&l0[w], bit);
    } else {
        __hip_atomic_fetch_or(&l0[w], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    atomicOr(&l1[w >> 5], 1u << (w & 31));
}

template <bool L0_LDS>
__device__ __forceinline__ void unmark(uint32_t* l0, int32_t c) {
    const uint32_t w = (uint32_t)c >> 5, bit = 1u << (c & 31);
    if constexpr (L0_LDS) {
        atomicAnd(&l0[w], ~bit);
    } else {
        __hip_atomic_fetch_and(&l0[w], ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Level-0 words in global slabs are written with L2 atomics; read / reset them at agent scope as well so that no stale
// line of this CU's vector L1 is ever consulted.
template <bool L0_LDS>
__device__ __forceinline__ uint32_t l0_read(const uint32_t* l0, int64_t w) {
    if constexpr (L0_LDS) return l0[w];
    else return __hip_atomic_load(l0 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool L0_LDS>
__device__ __forceinline__ void l0_clear(uint32_t* l0, int64_t w) {
    if constexpr (L0_LDS) l0[w] = 0;
    else __hip_atomic_store(l0 + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kSortCap = 1024;   // candidates per row the sorted-candidate kernel takes (LDS: 4 KiB per wave)
constexpr int kSortWaves = 4;    // independent waves (rows) per workgroup

// Number of candidates of row i = sum over its frontier nodes of deg_A(j) + lengths of the ADD rows (+1 for the
// diagonal): what both kernels use to decide who serves the row.  Wave-uniform result.
__device__ __forceinline__ int64_t candidate_count(const RingParams& p, int64_t i, int lane) {
    int64_t c = 0;
    if (p.frontier.rowptr) {
        const int64_t fb = p.frontier.rowptr[i], fe = p.frontier.rowptr[i + 1];
        for (int64_t f = fb + lane; f < fe; f += 64) {
            const int64_t j = p.frontier.colidx[f];
            c += p.a.rowptr[j + 1] - p.a.rowptr[j];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    for (int q = 0; q < p.n_add; ++q) c += p.add[q].rowptr[i + 1] - p.add[q].rowptr[i];
    return c + (p.add_diag ? 1 : 0);
}

__device__ __forceinline__ bool pattern_row_contains(const Pattern& pat, int64_t i, int32_t c) {
    int64_t lo = pat.rowptr[i], hi = pat.rowptr[i + 1];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int32_t v = pat.colidx[mid];
        if (v == c) return true;
        if (v < c) lo = mid + 1; else hi = mid;
    }
    return false;
}

// Sorted-candidate kernel for SPARSE rows on wide graphs (n beyond the LDS bitmap): a 1M-bit set per row is the wrong
// tool when the row has ~100 candidates -- every mark becomes a random DRAM read-modify-write.  Here ONE WAVE owns a
// row: it copies the row's candidates into LDS (<= kSortCap), sorts them with a bitonic network, and emits the
// distinct values that are in none of the SUB rows (binary search; those rows are short) in ascending order with a
// ballot prefix.  Rows with more candidates are left to the bitmap kernel (p.only_big_rows there).
template <bool FILL>
__global__ __launch_bounds__(64 * kSortWaves) void ring_sorted_kernel(const RingParams p) {
    __shared__ int32_t buf_all[kSortWaves][kSortCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t* buf = buf_all[wave];
    const int64_t n_waves = (int64_t)gridDim.x * kSortWaves;
    for (int64_t i = (int64_t)blockIdx.x * kSortWaves + wave; i < p.n_rows; i += n_waves) {
        const int64_t c64 = candidate_count(p, i, lane);
        if (c64 > kSortCap) {           // the bitmap kernel serves this row
            if (lane == 0) p.big_rows[atomicAdd(p.big_count, 1u)] = (int32_t)i;
            continue;
        }
        const int c = (int)c64;
        // ---- gather the candidates: 64 frontier nodes at a time, lane l copies the A-row of its node
        int pos = 0;  // wave-uniform fill level
        if (p.frontier.rowptr) {
            const int64_t fb = p.frontier.rowptr[i], fe = p.frontier.rowptr[i + 1];
            for (int64_t f0 = fb; f0 < fe; f0 += 64) {
                int64_t ab = 0;
                int deg = 0;
                if (f0 + lane < fe) {
                    const int64_t j = p.frontier.colidx[f0 + lane];
                    ab = p.a.rowptr[j];
                    deg = (int)(p.a.rowptr[j + 1] - ab);
                }
                int incl = deg;  // inclusive prefix sum of deg over the lanes
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int t = __shfl_up(incl, off);
                    if (lane >= off) incl += t;
                }
                const int my_off = pos + incl - deg;
                for (int t = 0; t < deg; ++t) buf[my_off + t] = p.a.colidx[ab + t];
                pos += __shfl(incl, 63);
            }
        }
        for (int q = 0; q < p.n_add; ++q) {
            const int64_t b = p.add[q].rowptr[i];
            const int len = (int)(p.add[q].rowptr[i + 1] - b);
            for (int t = lane; t < len; t += 64) buf[pos + t] = p.add[q].colidx[b + t];
            pos += len;
        }
        if (p.add_diag) {
            if (lane == 0) buf[pos] = (int32_t)(p.row_begin + i);
            pos += 1;
        }
        // ---- bitonic sort of the next power of two (padding = INT32_MAX, never a valid column)
        int P = 64;
        while (P < c) P <<= 1;
        for (int t = c + lane; t < P; t += 64) buf[t] = 0x7fffffff;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = lane; t < (P >> 1); t += 64) {
                    // t-th compare-exchange of this stage: partner indices (a, a ^ j) with a's j-bit clear
                    const int a = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int b = a | j;
                    const int32_t va = buf[a], vb = buf[b];
                    const bool up = (a & k) == 0;
                    if ((va > vb) == up) {
                        buf[a] = vb;
                        buf[b] = va;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- emit: distinct values not in any SUB row, ascending
        int64_t out = FILL ? p.out_rowptr[i] : 0;
        int64_t kept = 0;
        for (int t0 = 0; t0 < c; t0 += 64) {
            const int t = t0 + lane;
            bool keep = false;
            int32_t v = 0;
            if (t < c) {
                v = buf[t];
                keep = (t == 0 || buf[t - 1] != v);
                if (keep && p.sub_diag && v == (int32_t)(p.row_begin + i)) keep = false;
                for (int q = 0; keep && q < p.n_sub; ++q) keep = !pattern_row_contains(p.sub[q], i, v);
            }
            const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
            if constexpr (FILL) {
                if (keep) p.out_colidx[out + __builtin_popcountll(m & ((1ull << lane) - 1))] = v;
                out += __builtin_popcountll(m);
            }
            kept += __builtin_popcountll(m);
        }
        if (!FILL && lane == 0) p.counts[i] = kept;
        __builtin_amdgcn_wave_barrier();
    }
}

// FILL == false: counts[i]; FILL == true: out_colidx[out_rowptr[i] ...] in ascending order.
template <bool L0_LDS, bool FILL, int kThreads>
__global__ __launch_bounds__(kThreads) void ring_kernel(const RingParams p) {
    constexpr int kWaves = kThreads / 64;
    extern __shared__ uint32_t lds[];
    uint32_t* l1 = lds;                                   // [l1_words]
    uint32_t* l0 = L0_LDS ? lds + p.l1_words : p.l0_scratch + (int64_t)blockIdx.x * p.l0_words;
    __shared__ unsigned int row_s;
    __shared__ uint32_t wave_tot[kWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // the bitmaps start zeroed (LDS part here, the global slabs by the host) and are left zeroed by every row
    for (int64_t w = tid; w < p.l1_words + (L0_LDS ? p.l0_words : 0); w += kThreads) lds[w] = 0;
    __syncthreads();

    if (tid == 0) row_s = atomicAdd(p.ticket, 1u);
    __syncthreads();
    while (true) {
        int64_t i = row_s;
        if (p.only_big_rows) {  // tickets index the list of rows the sorted-candidate kernel left over
            if (i >= (int64_t)__hip_atomic_load(p.big_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            i = p.big_rows[i];
        }
        if (i >= p.n_rows) break;

        // ---- mark: expansion of the frontier through A, one wave per frontier node, lanes over its neighbours
        if (p.frontier.rowptr) {
            const int64_t fb = p.frontier.rowptr[i], fe = p.frontier.rowptr[i + 1];
            for (int64_t f = fb + wave; f < fe; f += kWaves) {
                const int64_t j = p.frontier.colidx[f];
                const int64_t ab = p.a.rowptr[j], ae = p.a.rowptr[j + 1];
                for (int64_t e = ab + lane; e < ae; e += 64) mark<L0_LDS>(l0, l1, __builtin_nontemporal_load(p.a.colidx + e));
            }
        }
        for (int q = 0; q < p.n_add; ++q) {
            const int64_t b = p.add[q].rowptr[i], e = p.add[q].rowptr[i + 1];
            for (int64_t t = b + tid; t < e; t += kThreads) mark<L0_LDS>(l0, l1, p.add[q].colidx[t]);
        }
        if (p.add_diag && tid == 0) mark<L0_LDS>(l0, l1, (int32_t)(p.row_begin + i));
        __syncthreads();
        // ---- subtract
        for (int q = 0; q < p.n_sub; ++q) {
            const int64_t b = p.sub[q].rowptr[i], e = p.sub[q].rowptr[i + 1];
            for (int64_t t = b + tid; t < e; t += kThreads) unmark<L0_LDS>(l0, p.sub[q].colidx[t]);
        }
        if (p.sub_diag && tid == 0) unmark<L0_LDS>(l0, (int32_t)(p.row_begin + i));
        __syncthreads();

        // ---- emit: thread t owns the contiguous level-1 word range [t*per, (t+1)*per) = a contiguous column range
        const int64_t per = (p.l1_words + kThreads - 1) / kThreads;
        const int64_t w1b = min((int64_t)tid * per, p.l1_words), w1e = min(w1b + per, p.l1_words);
        uint32_t mine = 0;
        for (int64_t w1 = w1b; w1 < w1e; ++w1) {
            uint32_t m = l1[w1];
            while (m) {
                const int b = __builtin_ctz(m);
                m &= m - 1;
                mine += __builtin_popcount(l0_read<L0_LDS>(l0, w1 * 32 + b));
            }
        }
        // block-wide exclusive prefix sum of `mine` (ascending thread id = ascending column): wave scan + 4 wave totals
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const uint32_t t = wave_tot[w];
            if (w < wave) base += t;
            total += t;
        }
        const uint32_t before = base + incl - mine;
        if (!FILL && tid == 0) p.counts[i] = total;
        int64_t pos = FILL ? p.out_rowptr[i] + before : 0;
        for (int64_t w1 = w1b; w1 < w1e; ++w1) {
            uint32_t m = l1[w1];
            if (!m) continue;
            l1[w1] = 0;
            while (m) {
                const int b = __builtin_ctz(m);
                m &= m - 1;
                const int64_t w0 = w1 * 32 + b;
                uint32_t bits = l0_read<L0_LDS>(l0, w0);
                l0_clear<L0_LDS>(l0, w0);
                if constexpr (FILL) {
                    while (bits) {
                        const int c = __builtin_ctz(bits);
                        bits &= bits - 1;
                        p.out_colidx[pos++] = (int32_t)(w0 * 32 + c);
                    }
                }
            }
        }
        if (tid == 0) row_s = atomicAdd(p.ticket, 1u);  // next row (every thread has read row_s and wave_tot by now:
        __syncthreads();                                // the scan's barrier lies behind); bitmaps are zero again
    }
}

// vals[e] of a square hop matrix given as a CSR PATTERN (all stored entries are 1, as nhoodSplit produces):
//   mode 1 (SYM): fp32( (s[deg_i] * 1.0) * s[deg_j] ),  s = deg^-1/2 with inf -> 0   (_dataset.py:114-118)
//   mode 2 (RW):  fp32( s[deg_i] * 1.0 ),               s = deg^-1   with inf -> 0   (_dataset.py:119-123)
//   mode 0 (ORDINARY): 1
// deg = row sums of THIS matrix = its row lengths.  `s_table[k]` holds the fp64 scaling of a row with k entries,
// computed on the host with the reference's own numpy call (np.power) so that the fp64 products -- and the fp32
// cast sparse2Tensor applies (:528-535) -- are bit-identical; the table depends on nothing but k.
// Row windows: `rowptr` / `vals` describe n_rows rows; for SYM the length of row j of the WHOLE matrix is col_len[j]
// (NULL: the pattern is the whole square matrix, lengths come from rowptr).  A row longer than the table raises *flag.
__global__ void normalize_pattern_kernel(int64_t n_rows, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                         int mode, const double* __restrict__ s_table, int64_t s_table_len,
                                         const int64_t* __restrict__ col_len, float* __restrict__ vals, int* flag) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    bool bad = false;
    for (int64_t i = wave; i < n_rows; i += n_waves) {
        const int64_t b = rowptr[i], e = rowptr[i + 1];
        double si = 1.0;
        if (mode != 0) {
            if (e - b < s_table_len) si = s_table[e - b]; else bad = true;
        }
        for (int64_t t = b + lane; t < e; t += 64) {
            double v = si * 1.0;
            if (mode == 1) {
                const int64_t j = colidx[t];
                const int64_t dj = col_len ? col_len[j] : rowptr[j + 1] - rowptr[j];
                if (dj < s_table_len) v = v * s_table[dj]; else bad = true;
            }
            vals[t] = (float)v;
        }
    }
    if (bad) atomicOr(flag, 1);
}

int check_pattern(const char* what, const int64_t* rp, const int32_t* ci) {
    if (!rp) return fail(H2GCN_ERR_INVALID_ARGUMENT, "%s: rowptr is NULL", what);
    (void)ci;
    return H2GCN_OK;
}

}  // namespace

extern "C" {

size_t h2gcn_ring_scratch_bytes(int64_t n) {
    if (n <= 0) return 64;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t l0_words = (n + 31) / 32;
    size_t slabs = n > kLdsBitmapCols ? (size_t)cus * kSlabsPerCu * (size_t)l0_words * 4 : 0;
    if (slabs > kMaxSlabBytes) slabs = std::max<size_t>(kMaxSlabBytes / ((size_t)l0_words * 4), 1) * (size_t)l0_words * 4;
    const size_t big_list = n > kLdsBitmapCols ? ((size_t)n * 4 + 63) / 64 * 64 : 0;  // rows left to the bitmap kernel
    return 64 + big_list + slabs;
}

static int ring_pass(bool fill, int64_t n, int64_t row_begin, int64_t n_rows, const int64_t* a_rowptr, const int32_t* a_colidx, const int64_t* f_rowptr,
                     const int32_t* f_colidx, int n_add, const int64_t* const* add_rowptr, const int32_t* const* add_colidx,
                     int add_diag, int n_sub, const int64_t* const* sub_rowptr, const int32_t* const* sub_colidx, int sub_diag,
                     int64_t* out_rowptr, int32_t* out_colidx, int64_t* nnz_out, void* scratch, size_t scratch_bytes,
                     hipStream_t stream) {
    if (n < 0 || n > 0x7fffffffLL) return fail(H2GCN_ERR_INVALID_ARGUMENT, "n = %lld (column ids are int32)", (long long)n);
    if (row_begin < 0 || n_rows < 0 || row_begin + n_rows > n)
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "row window [%lld, %lld) outside 0..%lld", (long long)row_begin, (long long)(row_begin + n_rows), (long long)n);
    if (n_add < 0 || n_add > kMaxPatterns || n_sub < 0 || n_sub > kMaxPatterns)
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "at most %d add / sub patterns", kMaxPatterns);
    if (!out_rowptr) return fail(H2GCN_ERR_INVALID_ARGUMENT, "out_rowptr is NULL");
    if (f_rowptr && (!a_rowptr)) return fail(H2GCN_ERR_INVALID_ARGUMENT, "a frontier needs the expansion operand A");
    if (scratch_bytes < h2gcn_ring_scratch_bytes(n) || !scratch)
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "scratch of %zu bytes, need %zu (h2gcn_ring_scratch_bytes)", scratch_bytes,
                    h2gcn_ring_scratch_bytes(n));
    if (fill && !out_colidx) return fail(H2GCN_ERR_INVALID_ARGUMENT, "out_colidx is NULL");
    RingParams p;
    memset(&p, 0, sizeof(p));
    p.n = n;
    p.n_rows = n_rows;
    p.row_begin = row_begin;
    p.a = Pattern{a_rowptr, a_colidx};
    p.frontier = Pattern{f_rowptr, f_colidx};
    for (int q = 0; q < n_add; ++q) {
        int st = check_pattern("add", add_rowptr[q], add_colidx[q]);
        if (st != H2GCN_OK) return st;
        p.add[q] = Pattern{add_rowptr[q], add_colidx[q]};
    }
    for (int q = 0; q < n_sub; ++q) {
        int st = check_pattern("sub", sub_rowptr[q], sub_colidx[q]);
        if (st != H2GCN_OK) return st;
        p.sub[q] = Pattern{sub_rowptr[q], sub_colidx[q]};
    }
    p.n_add = n_add;
    p.n_sub = n_sub;
    p.add_diag = add_diag;
    p.sub_diag = sub_diag;
    p.l0_words = (n + 31) / 32;
    p.l1_words = (p.l0_words + 31) / 32;
    p.ticket = (unsigned int*)scratch;
    p.big_count = (unsigned int*)scratch + 1;
    if (n_rows == 0) {
        if (!fill) H2GCN_HIP_TRY(hipMemsetAsync(out_rowptr, 0, sizeof(int64_t), stream));
        if (nnz_out) *nnz_out = 0;
        return H2GCN_OK;
    }
    int dev = 0, cus = 256;
    H2GCN_HIP_TRY(hipGetDevice(&dev));
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const bool l0_lds = n <= kLdsBitmapCols;
    const size_t lds_bytes = (size_t)(p.l1_words + (l0_lds ? p.l0_words : 0)) * 4;
    if (lds_bytes > 150 * 1024)  // level 1 alone needs n / 256 bytes of LDS: ~39M columns is the limit of two levels
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "ring construction supports up to %d columns (n = %lld)", 150 * 1024 * 256, (long long)n);
    // workgroups per CU the LDS footprint allows (160 KiB per CU, keep some for the static arrays), at most 8
    int per_cu = l0_lds ? (int)std::max<size_t>(1, std::min<size_t>(8, (150 * 1024) / (lds_bytes + 2048))) : kSlabsPerCu;
    unsigned grid = (unsigned)std::min<int64_t>(n_rows, (int64_t)cus * per_cu);
    if (!l0_lds) {
        const size_t big_list = ((size_t)n * 4 + 63) / 64 * 64;
        const int64_t n_slabs = (int64_t)((h2gcn_ring_scratch_bytes(n) - 64 - big_list) / ((size_t)p.l0_words * 4));
        grid = (unsigned)std::min<int64_t>(n_rows, n_slabs);
        p.big_rows = (int32_t*)((char*)scratch + 64);
        p.l0_scratch = (uint32_t*)((char*)scratch + 64 + big_list);
        H2GCN_HIP_TRY(hipMemsetAsync(p.l0_scratch, 0, (size_t)grid * p.l0_words * 4, stream));
    }
    H2GCN_HIP_TRY(hipMemsetAsync(p.ticket, 0, 64, stream));
    const unsigned sorted_grid = (unsigned)std::min<int64_t>((n_rows + kSortWaves - 1) / kSortWaves, (int64_t)cus * 8);
    p.only_big_rows = l0_lds ? 0 : 1;   // wide graphs: sparse rows go to the sorted-candidate kernel
    if (!fill) {
        p.counts = out_rowptr + 1;  // counts land in out_rowptr[1..n_rows], the scan below turns them into row pointers
        H2GCN_HIP_TRY(hipMemsetAsync(out_rowptr, 0, sizeof(int64_t), stream));
        if (l0_lds) {
            H2GCN_HIP_TRY(hipFuncSetAttribute((const void*)ring_kernel<true, false, kThreadsLds>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            hipLaunchKernelGGL((ring_kernel<true, false, kThreadsLds>), dim3(grid), dim3(kThreadsLds), lds_bytes, stream, p);
        } else {
            hipLaunchKernelGGL((ring_sorted_kernel<false>), dim3(sorted_grid), dim3(64 * kSortWaves), 0, stream, p);
            H2GCN_HIP_TRY(hipFuncSetAttribute((const void*)ring_kernel<false, false, kThreadsGlobal>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            hipLaunchKernelGGL((ring_kernel<false, false, kThreadsGlobal>), dim3(grid), dim3(kThreadsGlobal), lds_bytes, stream, p);
        }
        H2GCN_HIP_TRY(hipGetLastError());
        // inclusive scan in place: out_rowptr[1..n]
        size_t tmp_bytes = 0;
        H2GCN_HIP_TRY(rocprim::inclusive_scan(nullptr, tmp_bytes, p.counts, p.counts, (size_t)n_rows, rocprim::plus<int64_t>(), stream));
        void* tmp = nullptr;
        H2GCN_HIP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
        hipError_t e = rocprim::inclusive_scan(tmp, tmp_bytes, p.counts, p.counts, (size_t)n_rows, rocprim::plus<int64_t>(), stream);
        int64_t total = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&total, out_rowptr + n_rows, sizeof(int64_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(tmp);
        if (e != hipSuccess) return fail(H2GCN_ERR_HIP, "row-pointer scan failed: %s", hipGetErrorString(e));
        if (nnz_out) *nnz_out = total;
    } else {
        p.out_rowptr = out_rowptr;
        p.out_colidx = out_colidx;
        if (l0_lds) {
            H2GCN_HIP_TRY(hipFuncSetAttribute((const void*)ring_kernel<true, true, kThreadsLds>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            hipLaunchKernelGGL((ring_kernel<true, true, kThreadsLds>), dim3(grid), dim3(kThreadsLds), lds_bytes, stream, p);
        } else {
            hipLaunchKernelGGL((ring_sorted_kernel<true>), dim3(sorted_grid), dim3(64 * kSortWaves), 0, stream, p);
            H2GCN_HIP_TRY(hipFuncSetAttribute((const void*)ring_kernel<false, true, kThreadsGlobal>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            hipLaunchKernelGGL((ring_kernel<false, true, kThreadsGlobal>), dim3(grid), dim3(kThreadsGlobal), lds_bytes, stream, p);
        }
        H2GCN_HIP_TRY(hipGetLastError());
    }
    return H2GCN_OK;
}

int h2gcn_ring_count_rows(int64_t n, int64_t row_begin, int64_t n_rows, const int64_t* a_rowptr, const int32_t* a_colidx,
                          const int64_t* f_rowptr, const int32_t* f_colidx, int n_add, const int64_t* const* add_rowptr,
                          const int32_t* const* add_colidx, int add_diag, int n_sub, const int64_t* const* sub_rowptr,
                          const int32_t* const* sub_colidx, int sub_diag, int64_t* out_rowptr, int64_t* nnz_out,
                          void* scratch, size_t scratch_bytes, void* stream) {
    try {
        return ring_pass(false, n, row_begin, n_rows, a_rowptr, a_colidx, f_rowptr, f_colidx, n_add, add_rowptr, add_colidx,
                         add_diag, n_sub, sub_rowptr, sub_colidx, sub_diag, out_rowptr, nullptr, nnz_out, scratch, scratch_bytes,
                         (hipStream_t)stream);
    } catch (...) {
        return fail(H2GCN_ERR_INTERNAL, "unexpected exception in ring_count");
    }
}

int h2gcn_ring_fill_rows(int64_t n, int64_t row_begin, int64_t n_rows, const int64_t* a_rowptr, const int32_t* a_colidx,
                         const int64_t* f_rowptr, const int32_t* f_colidx, int n_add, const int64_t* const* add_rowptr,
                         const int32_t* const* add_colidx, int add_diag, int n_sub, const int64_t* const* sub_rowptr,
                         const int32_t* const* sub_colidx, int sub_diag, const int64_t* out_rowptr, int32_t* out_colidx,
                         void* scratch, size_t scratch_bytes, void* stream) {
    try {
        return ring_pass(true, n, row_begin, n_rows, a_rowptr, a_colidx, f_rowptr, f_colidx, n_add, add_rowptr, add_colidx,
                         add_diag, n_sub, sub_rowptr, sub_colidx, sub_diag, const_cast<int64_t*>(out_rowptr), out_colidx, nullptr,
                         scratch, scratch_bytes, (hipStream_t)stream);
    } catch (...) {
        return fail(H2GCN_ERR_INTERNAL, "unexpected exception in ring_fill");
    }
}

int h2gcn_ring_count(int64_t n, const int64_t* a_rowptr, const int32_t* a_colidx, const int64_t* f_rowptr,
                     const int32_t* f_colidx, int n_add, const int64_t* const* add_rowptr,
                     const int32_t* const* add_colidx, int add_diag, int n_sub, const int64_t* const* sub_rowptr,
                     const int32_t* const* sub_colidx, int sub_diag, int64_t* out_rowptr, int64_t* nnz_out,
                     void* scratch, size_t scratch_bytes, void* stream) {
    return h2gcn_ring_count_rows(n, 0, n, a_rowptr, a_colidx, f_rowptr, f_colidx, n_add, add_rowptr, add_colidx, add_diag, n_sub,
                                 sub_rowptr, sub_colidx, sub_diag, out_rowptr, nnz_out, scratch, scratch_bytes, stream);
}

int h2gcn_ring_fill(int64_t n, const int64_t* a_rowptr, const int32_t* a_colidx, const int64_t* f_rowptr,
                    const int32_t* f_colidx, int n_add, const int64_t* const* add_rowptr,
                    const int32_t* const* add_colidx, int add_diag, int n_sub, const int64_t* const* sub_rowptr,
                    const int32_t* const* sub_colidx, int sub_diag, const int64_t* out_rowptr, int32_t* out_colidx,
                    void* scratch, size_t scratch_bytes, void* stream) {
    return h2gcn_ring_fill_rows(n, 0, n, a_rowptr, a_colidx, f_rowptr, f_colidx, n_add, add_rowptr, add_colidx, add_diag, n_sub,
                                sub_rowptr, sub_colidx, sub_diag, out_rowptr, out_colidx, scratch, scratch_bytes, stream);
}

int h2gcn_hop_normalize_rows(int64_t n_rows, const int64_t* rowptr, const int32_t* colidx, int mode, const double* s_table,
                             int64_t s_table_len, const int64_t* col_len, float* vals, void* stream_v) {
    if (n_rows < 0 || !rowptr || mode < 0 || mode > 2) return fail(H2GCN_ERR_INVALID_ARGUMENT, "bad arguments to hop_normalize");
    if (mode != 0 && (!s_table || s_table_len < 1)) return fail(H2GCN_ERR_INVALID_ARGUMENT, "scaling table missing");
    if (n_rows == 0) return H2GCN_OK;
    if (!vals || !colidx) return fail(H2GCN_ERR_INVALID_ARGUMENT, "vals/colidx is NULL");
    hipStream_t stream = (hipStream_t)stream_v;
    int* flag = nullptr;
    H2GCN_HIP_TRY(hipMalloc((void**)&flag, sizeof(int)));
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), stream);
    int h_flag = 0;
    if (e == hipSuccess) {
        const unsigned blocks = (unsigned)std::min<int64_t>((n_rows + 3) / 4, 256 * 32);
        hipLaunchKernelGGL(normalize_pattern_kernel, dim3(blocks), dim3(256), 0, stream, n_rows, rowptr, colidx, mode, s_table,
                           s_table_len, col_len, vals, flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_flag, flag, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(flag);
    if (e != hipSuccess) return fail(H2GCN_ERR_HIP, "hop_normalize: %s", hipGetErrorString(e));
    if (h_flag) return fail(H2GCN_ERR_INVALID_ARGUMENT, "hop_normalize: a row is longer than the scaling table (s_table_len = %lld)", (long long)s_table_len);
    return H2GCN_OK;
}

int h2gcn_hop_normalize(int64_t n, const int64_t* rowptr, const int32_t* colidx, int mode, const double* s_table,
                        int64_t s_table_len, float* vals, void* stream) {
    return h2gcn_hop_normalize_rows(n, rowptr, colidx, mode, s_table, s_table_len, nullptr, vals, stream);
}

}  // extern "C"
