// h2gcn_capi.hip -- host side of libh2gcn_hip.so: plan construction (CSR-adaptive row binning, optional
// transposed operands, one-time index validation) and the launchers behind the C ABI of include/h2gcn_hip.h.
//
// Reference behaviour this file stands in for (nothing is translated -- the reference has no native code):
//   * plan_create   <-> building `tensors.adj_hops` once per run, h2gcn/datasets/_dataset.py:559-576,528-535
//   * spmm_hops     <-> GCNLayer.call, h2gcn/models/_layers.py:78-81 (TF SparseTensorDenseMatMul, :74,76)
//   * spmm_hops_T   <-> its TF-registered gradient wrt the dense operand (h2gcn/models/H2GCN.py:66-74)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "h2gcn_hip.h"
#include "spmm_kernels.hip.h"

#include "capi_internal.h"

namespace h2gcn {

thread_local std::string g_last_error;

int fail(h2gcn_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return (int)st;
}

}  // namespace h2gcn

namespace {

using h2gcn::fail;
using h2gcn::g_last_error;

struct DeviceBuf {
    void* p = nullptr;
    ~DeviceBuf() {
        if (p) (void)hipFree(p);
    }
    DeviceBuf() = default;
    DeviceBuf(DeviceBuf&& o) noexcept : p(o.p) { o.p = nullptr; }
    DeviceBuf(const DeviceBuf&) = delete;
    DeviceBuf& operator=(const DeviceBuf&) = delete;
};

struct HopOperand {  // one CSR, device pointers
    const int64_t* rowptr = nullptr;
    const int32_t* colidx = nullptr;
    const float* vals = nullptr;
    int64_t nnz = 0;
    std::vector<int64_t> long_rows;  // rows whose segment has >= long_row_threshold nonzeros, ascending
    std::vector<uint8_t> len8;       // min(segment length, 255) per row: what the segment-class bins are built from
    int64_t n_short = 0, nnz_short = 0;  // segments of at most short_max nonzeros (empty ones included) / their nonzeros
    int64_t nnz_long = 0;                // nonzeros of the long segments
    const uint32_t* perm = nullptr;  // adjoint operands built with H2GCN_PLAN_KEEP_PERMUTATION: source entry of entry i
};

struct LongList {
    DeviceBuf dev;
    int n = 0;
    LongList() = default;
    LongList(LongList&&) noexcept = default;
};

struct ClassLists {   // rows (int32, ascending) of the short and of the medium bin of one operand / hop selection
    DeviceBuf short_dev, med_dev;   // allocated only once a launch decides to be list-driven (`built`)
    int64_t n_short = 0, nnz_short = 0, n_med = 0;
    bool built = false;
    ClassLists() = default;
    ClassLists(ClassLists&&) noexcept = default;
};

}  // namespace

struct h2gcn_plan {
    int n_hops = 0;
    int64_t n_rows = 0, n_cols = 0;
    int long_threshold = 256;
    int rows_per_wave = 4;
    int variant = 0;
    int slice_cols = 0;  // 0 = heuristic
    bool has_transpose = false;
    int device = 0;
    std::vector<HopOperand> fwd;  // A_k       [n_rows x n_cols], caller-owned arrays
    std::vector<HopOperand> adj;  // A_k^T     [n_cols x n_rows], plan-owned arrays
    std::deque<DeviceBuf> owned;  // storage behind `adj` (deque: references stay valid on growth)
    // long-segment lists are specific to a hop selection; built on first use, then cached
    mutable std::mutex mu;
    mutable std::map<uint64_t, LongList> long_cache;  // key = mask | (adjoint << 32)
    // segment-class bins besides the long lists: forward -- per hop (key = hop | 1 << 40) the rows whose segment is short /
    // medium; adjoint -- per hop selection (key = mask | 1 << 32) the rows whose segments of ALL selected hops are short /
    // the rows that are neither that nor owned by the long path
    mutable std::map<uint64_t, ClassLists> class_cache;
    int short_max = -1;      // longest segment of the short class (min(kShortMax, long_threshold - 1)); -1: no binning
    double short_min_frac = 0.05;  // a launch uses the list when at least this share of its segments is short
};

namespace h2gcn {
// one-time operand check of plan_create: any column id outside [0, n_cols) raises the flag
__global__ void check_colidx_kernel(const int32_t* __restrict__ colidx, int64_t nnz, int64_t n_cols, int* flag) {
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t c = colidx[i];
        bad |= (c < 0) | ((int64_t)c >= n_cols);
    }
    if (bad) atomicOr(flag, 1);
}

void launch_in_tile_short(bool sum, const LaunchParams& p, int slice, bool off32, bool fb4, dim3 grid, hipStream_t stream);
int transpose_csr_device(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* rowptr, const int32_t* colidx,
                         const float* vals, int64_t** t_rowptr_out, int32_t** t_colidx_out, float** t_vals_out,
                         uint32_t** perm_out_keep, hipStream_t stream, std::string* err);
void permute_values(const uint32_t* perm, const float* vals, int64_t nnz, float* t_vals, hipStream_t stream);
}

namespace {

using h2gcn::HopCsr;
using h2gcn::LaunchParams;

// Host transposition of one CSR (counting sort by column; keeps ascending row order inside each output row,
// i.e. the canonical order `tf.sparse.reorder` would give the adjoint operand).
int transpose_csr_host(int64_t n_rows, int64_t n_cols, const std::vector<int64_t>& rowptr,
                       const std::vector<int32_t>& colidx, const std::vector<float>& vals,
                       std::vector<int64_t>& t_rowptr, std::vector<int32_t>& t_colidx, std::vector<float>& t_vals) {
    const int64_t nnz = rowptr[n_rows];
    t_rowptr.assign(n_cols + 1, 0);
    t_colidx.resize(nnz);
    t_vals.resize(nnz);
    for (int64_t i = 0; i < nnz; ++i) t_rowptr[colidx[i] + 1]++;
    for (int64_t c = 0; c < n_cols; ++c) t_rowptr[c + 1] += t_rowptr[c];
    std::vector<int64_t> cursor(t_rowptr.begin(), t_rowptr.end() - 1);
    for (int64_t r = 0; r < n_rows; ++r) {
        for (int64_t i = rowptr[r]; i < rowptr[r + 1]; ++i) {
            const int64_t dst = cursor[colidx[i]]++;
            t_colidx[dst] = (int32_t)r;
            t_vals[dst] = vals[i];
        }
    }
    return 0;
}

void collect_long_rows(const std::vector<int64_t>& rowptr, int64_t n_rows, int threshold, std::vector<int64_t>& out) {
    out.clear();
    for (int64_t r = 0; r < n_rows; ++r)
        if (rowptr[r + 1] - rowptr[r] >= threshold) out.push_back(r);
}

// Segment lengths of one operand, clipped to a byte: the class of every segment (short / medium / long) follows from it.
void collect_segment_lengths(const std::vector<int64_t>& rowptr, int64_t n_rows, int short_max, int long_threshold, HopOperand& op) {
    op.len8.resize((size_t)n_rows);
    op.n_short = op.nnz_short = op.nnz_long = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t len = rowptr[r + 1] - rowptr[r];
        op.len8[(size_t)r] = (uint8_t)(len > 255 ? 255 : len);
        if (len <= short_max) {
            ++op.n_short;
            op.nnz_short += len;
        } else if (len >= long_threshold) {
            op.nnz_long += len;
        }
    }
}

// The short and medium bins of a launch (cached in the plan): forward -- hop `k`'s rows by the class of their segment;
// adjoint -- the rows whose segments of every hop in `mask` are short, and the rows that are neither that nor contain a
// long segment (those belong to the long list).  The COUNTS (what the schedule is decided from) are computed on first use;
// the device lists only when `build` is set, i.e. once a launch really is list-driven -- a products-sized operand whose
// launches stay on the tile walk never pays the 4 bytes per row and class.
// Set by the launch entry points for the duration of a call: the launch stream is being captured into a hipGraph.  The device
// lists of a hop selection are built by its FIRST launch (an allocation and a synchronous upload) -- neither may happen inside
// a capture, and a fallback to another walk would silently bake the slower schedule into the graph: such a launch is refused
// with a message that says what to do (run it once eagerly first; all-hops selections are prepared at plan creation).
thread_local bool t_stream_capturing = false;

struct CaptureScope {
    explicit CaptureScope(hipStream_t stream) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        t_stream_capturing = hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        if (!t_stream_capturing) (void)hipGetLastError();
    }
    ~CaptureScope() { t_stream_capturing = false; }
};

int refuse_build_while_capturing(const char* what) {
    return fail(H2GCN_ERR_INVALID_ARGUMENT, "the stream is being captured and this is the first launch of this hop selection: its %s "
                "has to be built first (device allocation + upload).  Run the same launch once outside the capture (a warm-up step)", what);
}

int get_class_lists(const h2gcn_plan* plan, bool adjoint, int k, uint32_t mask, bool build, const ClassLists** out) {
    *out = nullptr;
    const std::vector<HopOperand>& ops = adjoint ? plan->adj : plan->fwd;
    const int64_t n_rows = adjoint ? plan->n_cols : plan->n_rows;
    if (plan->short_max < 0 || n_rows > 0x7fffffffLL) return H2GCN_OK;
    const uint64_t key = adjoint ? ((uint64_t)mask | (1ull << 32)) : ((uint64_t)k | (1ull << 40));
    const int sm = plan->short_max;
    std::lock_guard<std::mutex> lock(plan->mu);
    auto it = plan->class_cache.find(key);
    std::vector<const uint8_t*> sel;        // adjoint: the selected operands' segment lengths
    std::vector<uint8_t> any_long;
    auto prepare_adjoint = [&]() {
        any_long.assign((size_t)n_rows, 0);
        for (int h = 0; h < plan->n_hops; ++h)
            if (mask & (1u << h)) {
                sel.push_back(ops[h].len8.data());
                for (int64_t r : ops[h].long_rows) any_long[(size_t)r] = 1;
            }
    };
    if (it == plan->class_cache.end()) {
        ClassLists fresh;
        if (!adjoint) {
            fresh.n_short = ops[k].n_short;
            fresh.nnz_short = ops[k].nnz_short;
            fresh.n_med = n_rows - ops[k].n_short - (int64_t)ops[k].long_rows.size();
        } else {
            prepare_adjoint();
            for (int64_t r = 0; r < n_rows; ++r) {
                bool all = true;
                int64_t z = 0;
                for (const uint8_t* l : sel) { all = all && l[r] <= sm; z += l[r]; }
                if (all) { ++fresh.n_short; fresh.nnz_short += z; }
                else if (!any_long[(size_t)r]) ++fresh.n_med;
            }
        }
        it = plan->class_cache.emplace(key, std::move(fresh)).first;
    }
    ClassLists& cl = it->second;
    if (build && !cl.built) {
        if (t_stream_capturing) return refuse_build_while_capturing("per-class row lists");
        std::vector<int32_t> h_short, h_med;
        h_short.reserve((size_t)cl.n_short);
        h_med.reserve((size_t)cl.n_med);
        if (!adjoint) {
            const std::vector<uint8_t>& len = ops[k].len8;
            const std::vector<int64_t>& longs = ops[k].long_rows;   // ascending
            size_t li = 0;
            for (int64_t r = 0; r < n_rows; ++r) {
                while (li < longs.size() && longs[li] < r) ++li;
                const bool is_long = li < longs.size() && longs[li] == r;
                if (len[(size_t)r] <= sm) h_short.push_back((int32_t)r);
                else if (!is_long) h_med.push_back((int32_t)r);
            }
        } else {
            if (sel.empty()) prepare_adjoint();
            for (int64_t r = 0; r < n_rows; ++r) {
                bool all = true;
                for (const uint8_t* l : sel) all = all && l[r] <= sm;
                if (all) h_short.push_back((int32_t)r);
                else if (!any_long[(size_t)r]) h_med.push_back((int32_t)r);
            }
        }
        DeviceBuf d_short, d_med;
        if (!h_short.empty()) {
            H2GCN_HIP_TRY(hipMalloc(&d_short.p, h_short.size() * sizeof(int32_t)));
            H2GCN_HIP_TRY(hipMemcpy(d_short.p, h_short.data(), h_short.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        if (!h_med.empty()) {
            H2GCN_HIP_TRY(hipMalloc(&d_med.p, h_med.size() * sizeof(int32_t)));
            H2GCN_HIP_TRY(hipMemcpy(d_med.p, h_med.data(), h_med.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        cl.short_dev.p = d_short.p;
        d_short.p = nullptr;
        cl.med_dev.p = d_med.p;
        d_med.p = nullptr;
        cl.built = true;
    }
    *out = &cl;
    return H2GCN_OK;
}

int check_rowptr(const std::vector<int64_t>& rowptr, int64_t n_rows, int hop) {
    if (rowptr[0] != 0) return fail(H2GCN_ERR_BAD_INDEX, "hop %d: rowptr[0] = %lld, expected 0", hop, (long long)rowptr[0]);
    for (int64_t r = 0; r < n_rows; ++r)
        if (rowptr[r + 1] < rowptr[r])
            return fail(H2GCN_ERR_BAD_INDEX, "hop %d: rowptr decreases at row %lld", hop, (long long)r);
    return H2GCN_OK;
}

// Long-segment list for a hop selection (cached in the plan).
int get_long_list(const h2gcn_plan* plan, bool adjoint, uint32_t mask, const int64_t** dev_out, int* n_out) {
    const uint64_t key = (uint64_t)mask | ((uint64_t)(adjoint ? 1 : 0) << 32);
    std::lock_guard<std::mutex> lock(plan->mu);
    auto it = plan->long_cache.find(key);
    if (it == plan->long_cache.end()) {
        const std::vector<HopOperand>& ops = adjoint ? plan->adj : plan->fwd;
        std::vector<int64_t> host;
        int s = 0;
        if (!adjoint) {
            // forward: one entry per long (row, selected hop) segment, encoded row << 4 | s
            for (int k = 0; k < plan->n_hops; ++k) {
                if (!(mask & (1u << k))) continue;
                for (int64_t r : ops[k].long_rows) host.push_back((r << 4) | s);
                ++s;
            }
        } else {
            // adjoint (sum over hops): a row is long if any selected hop segment is
            for (int k = 0; k < plan->n_hops; ++k) {
                if (!(mask & (1u << k))) continue;
                host.insert(host.end(), ops[k].long_rows.begin(), ops[k].long_rows.end());
            }
            std::sort(host.begin(), host.end());
            host.erase(std::unique(host.begin(), host.end()), host.end());
        }
        LongList fresh;  // inserted into the cache only once it is complete
        fresh.n = (int)host.size();
        if (fresh.n > 0 && t_stream_capturing) return refuse_build_while_capturing("long-segment list");
        if (fresh.n > 0) {
            H2GCN_HIP_TRY(hipMalloc(&fresh.dev.p, host.size() * sizeof(int64_t)));
            H2GCN_HIP_TRY(hipMemcpy(fresh.dev.p, host.data(), host.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        }
        it = plan->long_cache.emplace(key, std::move(fresh)).first;
    }
    *dev_out = (const int64_t*)it->second.dev.p;
    *n_out = it->second.n;
    return H2GCN_OK;
}

// wave-sized work items a launch should have before a wave is given more than the minimum work: ~2 per wave slot of the
// chip (256 CUs x 4 SIMDs x 6 waves)
constexpr int64_t kWavesToFill = 12288;

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Column-slice width of the EXACT kernels (float4 lanes; any d >= 4, any 4-byte aligned operands: 16-byte global loads
// and stores only need dword alignment on gfx950; n_slices = ceil(d / slice), a partial last slice is masked in the
// kernel and the lane that straddles the end of a row whose width is not a multiple of 4 overlaps its neighbour).
// Only three lane geometries exist -- 64 / 128 / 256 columns = 4 / 2 / 1 gathered rows per load instruction -- and all
// three build the canonical summation tree (spmm_kernels.hip.h):
//   * the largest width that divides d, capped so that the gather working set of one slice (n_src_rows * slice * 4 B) is
//     friendlier to the 256 MiB Infinity Cache when the whole operand is far beyond it;
//   * widths none of them divides (d = 96, 100, 200 ...): one masked slice of the next power of two up to 256 columns
//     (narrower slices of rows that do not start on a cache line fetch every boundary line twice), the widest slice
//     beyond that; 64-column slices when every row starts on a line and the operand is far beyond the caches;
//   * d < 64: ONE masked slice of 64 columns.  (Rounds 1-2 had 32- and 16-column kernels with 8 / 16 lane groups for
//     these widths; the masked 64-column kernel is faster on every one of them -- products d = 32: 4.86 vs 5.14 ms,
//     d = 48: 8.9 vs 14.7 ms, lowdeg d = 32: 3.0 vs 5.8 ms because the short-row walk becomes available,
//     profiles/r03_sweep_narrow_widths.txt -- and their wider summation tree was the one exception to the bitwise
//     guarantee, so they are gone.)
// `forced` > 0 (plan option / scratch layout) overrides the heuristic.
int pick_slice_cols(int d, int64_t n_src_rows, int forced, double avg_segment_nnz, bool rows_line_aligned = false) {
    if (forced > 0) return forced;
    if (d < 64) return 64;
    static const int widths[] = {256, 128, 64};
    for (int w : widths) {
        if (d % w != 0) continue;
        const double slice_bytes = (double)n_src_rows * w * 4.0;
        // narrower slices pay one more pass over the row pointers / indices and one more latency chain per
        // segment: only worth it when segments are long enough to amortise that, and never below 64 columns
        if (w > 64 && avg_segment_nnz >= 16.0 && slice_bytes > 768.0 * 1024 * 1024 && d % (w / 2) == 0) continue;
        return w;
    }
    if (rows_line_aligned && d % 4 == 0 && avg_segment_nnz >= 16.0 && (double)n_src_rows * d * 4.0 >= 512.0 * 1024 * 1024) {
        // every row starts on a cache line, so 64-column slices are line-aligned too (no boundary line is fetched twice):
        // the cache-sized slices of the dividing case, in place, the last one masked (d = 200: what the scratch copy buys
        // for unaligned rows -- 0.69 -> 0.78 -- without the copy: 0.80; d = 100: 0.67 -> 0.70; d = 300: 0.84).  Not for
        // widths that are not multiples of 4: the tail lane's overlapping float4 would cross into the previous slice's
        // line (d = 130: 0.62 in place vs 0.69 through the zero-padded copy)
        return 64;
    }
    if (d > 256) {
        // many masked-at-the-end slices (raw feature widths: 1433, 3703): every slice pays one pass over the row pointers
        // and indices, so take the widest one unless its working set is far beyond the caches on long segments
        // (arxiv shape, d = 1433: 256 -> 0.81, 128 -> 0.73, 64 -> 0.60 of the roofline)
        const bool beyond = avg_segment_nnz >= 16.0 && (double)n_src_rows * 256 * 4.0 > 768.0 * 1024 * 1024;
        return beyond ? 128 : 256;
    }
    int w = 64;
    while (w < d) w *= 2;
    return w;
}

// What the schedule of one launch depends on.
struct LaunchShape {
    bool adjoint;
    int n_sel;
    int64_t nnz_sel;
    int64_t n_out, n_src;   // output rows / rows of the gather source
    double avg;             // nonzeros per (row, hop) segment
    int64_t ld_src;
    int d;
    bool src_line_aligned;  // every gathered row starts on a 128-byte line: base pointer and row stride (and hop offsets)
    // binned short segments available to this launch (see get_short_list)
    const int32_t* short_list[H2GCN_MAX_HOPS];
    int64_t short_count[H2GCN_MAX_HOPS];
    const int32_t* med_list[H2GCN_MAX_HOPS];
    int64_t med_count[H2GCN_MAX_HOPS];
    int n_short_lists;      // forward: n_sel, adjoint: 1 (0: none)
    double short_frac;      // share of the launch's segments (adjoint: output rows) the lists cover
    double short_nnz_frac;  // share of its nonzeros
};

LaunchShape shape_of(const h2gcn_plan* plan, uint32_t mask, bool adjoint);
int fill_short(const h2gcn_plan* plan, uint32_t mask, bool adjoint, LaunchShape& sh, bool build_lists = false);

// Slice width of the slice-major scratch copy this launch should gather from (see repack_slice_major_kernel);
// 0 = gather from the source as it is.  Because every slice width >= 64 produces the same summation tree, the copy
// never changes a bit of the result.
int scratch_slice_cols(const h2gcn_plan* plan, const LaunchShape& sh) {
    if (plan->variant == 4) return 0;  // variant 4: never repack (A/B measurements)
    const int d = sh.d;
    if (sh.n_out == 0 || sh.n_src == 0 || sh.n_sel == 0) return 0;
    if (d <= 64) return 0;
    const double src_bytes = (double)sh.n_src * d * 4.0 * (sh.adjoint ? sh.n_sel : 1);
    if (src_bytes < 512.0 * 1024 * 1024) return 0;                  // operand must be far beyond the caches
    if ((double)sh.nnz_sel < 32.0 * (double)sh.n_src) return 0;     // ... and gathered often enough to amortise the copy
    if (!sh.adjoint && (sh.ld_src * 4) % 1024 == 0 && d % 64 == 0 && d >= 128) {
        // power-of-two-ish stride: the slice width the plain launch would use
        const int w = pick_slice_cols(d, sh.n_src, plan->slice_cols, sh.avg);
        return (w == 64 || w == 128) && d % w == 0 && w < d ? w : 0;
    }
    if ((!sh.src_line_aligned || d % 4 != 0) && d > (sh.adjoint ? 256 : 128) && sh.avg >= 16.0 && (plan->slice_cols == 0 || plan->slice_cols == 64)) {
        // rows that are not cache-line aligned and wider than one 128-column slice: line-aligned, cache-sized 64-column
        // blocks (forward, including the copy: d = 132: +4 %, 200: +8 %, 300: +30 %; d = 100 gains nothing: two blocks fetch
        // the same 512 B per edge as the unaligned row and pay a second index pass).  Adjoint: its one masked slice of 256
        // columns is already the better schedule up to d = 256 (d = 200: 0.82 plain vs 0.78 copied); beyond that the copy
        // wins as well (d = 300: 0.72 -> 0.82)
        return 64;
    }
    return 0;
}

size_t scratch_bytes(const LaunchShape& sh, int rs) {
    if (rs <= 0) return 0;
    const size_t n_slices = (size_t)((sh.d + rs - 1) / rs);
    return (size_t)sh.n_src * n_slices * (size_t)rs * 4 * (size_t)(sh.adjoint ? sh.n_sel : 1);
}

struct Schedule {
    bool pipe, scalar128, exact, shortrow, lists, fb4;
    int slice;
};

// The launch-time decisions (also reported by h2gcn_plan_schedule).  `exact_ok`: the float4 kernels can serve the
// launch (d >= 4; narrower rows take the generic column-tiled kernel).
//
// CSR-adaptive dispatch.  Long segments (>= long_row_threshold) always have their own workgroups.  For the rest:
//   * segments short throughout (mean < 16 nonzeros over the selected hops): the IN-TILE short-row mode -- tile walk, G
//     consecutive short rows per round, one lane group each (the fastest walk on such operands: lowdeg 7.0 vs 7.5-8.1 ms
//     list-driven, hbm16m 24.1 vs 25.3-27.9, profiles/r04_ab_short_walks.txt);
//   * MIXED launches (mean >= 16, but at least `short_min_frac` of the segments are short: a sparse hop next to a dense
//     one -- the reference's 1-hop / 2-hop rings --, short rows scattered among longer ones): LIST-DRIVEN by segment
//     class -- short segments from the plan's binned list, one lane group each, wherever they sit; medium ones one wave
//     each; no tile walk (bimodal shape, a third of the edges in short segments: 27.7 vs 29.6-30.3 ms on the wave walk,
//     28.9-29.2 in-tile; neutral where the longer segments carry the bytes: h2gcn_like, products_tail);
//   * otherwise the wave-per-segment tile walk.
// (Rounds 2-3 chose ONE walk per launch from the pooled mean alone.)
Schedule decide(int variant, bool exact_ok, int d, int rows_per_wave, int n_sel, int forced_slice, int64_t n_src_rows,
                double avg_segment_nnz, bool rows_line_aligned, bool gen, double short_frac, double short_nnz_frac, double short_min_frac,
                int64_t n_out_rows) {
    Schedule sc;
    // index prefetch across segments: pays on short segments (+4 % at mean degree 4), costs ~0.4 % on long ones;
    // variant 2 forces it, variant 3 forbids it (bitwise-identical results either way)
    if (variant == 4) variant = 0;
    sc.pipe = (variant == 2 || (variant == 0 && avg_segment_nnz < 16.0)) && rows_per_wave * n_sel <= 32;
    sc.scalar128 = exact_ok && variant == 1 && d == 128 && forced_slice == 0;  // variant 1 only exists for d = 128
    sc.slice = (exact_ok && !sc.scalar128) ? pick_slice_cols(d, n_src_rows, forced_slice, avg_segment_nnz, rows_line_aligned) : 0;
    sc.exact = sc.slice > 0 || sc.scalar128;
    const bool grouped_ok = sc.exact && !sc.scalar128 && !gen && (sc.slice == 64 || sc.slice == 128);   // 4 / 2 lane groups, plain stores
    // variant 5 forces the in-tile short-row mode, variant 6 the list-driven launch (when a short segment exists at all);
    // variants 2 / 3 keep the wave-per-segment walk with / without the index prefetch
    // (the in-tile mode is a THROUGHPUT device: below ~50 k output rows a launch is a few waves' dependent chains, a wave takes
    // one row -- see kWavesToFill -- and a round of the in-tile mode would use one of its G lane groups; such launches are
    // list-driven instead: citeseer epoch 0.277-0.328 ms in-tile vs 0.256 ms, profiles/r04_ab_work_per_wave.txt)
    sc.shortrow = grouped_ok && (variant == 5 || (variant == 0 && avg_segment_nnz < 16.0 && n_out_rows >= 4 * kWavesToFill));
    sc.lists = grouped_ok && !sc.shortrow && short_frac > 0.0 && (variant == 6 || (variant == 0 && short_frac >= short_min_frac));
    sc.pipe = sc.pipe && sc.exact && !sc.shortrow && !sc.lists && !gen;
    // shallow load batches (more waves per SIMD) once the gather source is far beyond the caches: in-tile mode always
    // (profiles/r02_ab_short_row_occupancy_auto.txt), list-driven launches when most nonzeros sit in short segments
    const bool beyond = (double)n_src_rows * d * 4.0 >= 512.0 * 1024 * 1024;
    sc.fb4 = beyond && (sc.shortrow || (sc.lists && short_nnz_frac > 0.5));
    return sc;
}

template <bool SUM>
int launch(LaunchParams& p, const h2gcn_plan* plan, uint32_t mask, LaunchShape& sh, bool off32, int forced_slice, hipStream_t stream) {
    using namespace h2gcn;
    // A/B measurements only (profiles/r04_ab_off64_*.txt): run the 64-bit-offset instantiations on an operand that would
    // qualify for 32-bit gather offsets
    static const bool force_off64 = getenv("H2GCN_FORCE_OFF64") != nullptr;
    if (force_off64) off32 = false;
    const int variant = plan->variant;
    const bool exact_ok = p.d >= 4;
    // general store (bias / ReLU epilogue, element-wise tail of a width that is not a multiple of 4): dedicated instantiations
    const bool gen = p.d % 4 != 0 || p.bias != nullptr || p.relu != 0 || p.accumulate != 0;
    const Schedule sc = decide(variant, exact_ok, p.d, p.rows_per_wave, p.n_sel, forced_slice, sh.n_src, sh.avg,
                               sh.src_line_aligned, gen, sh.short_frac, sh.short_nnz_frac, plan->short_min_frac, sh.n_out);
    const bool short_fb4 = sc.fb4;
    const bool pipe = sc.pipe, scalar128 = sc.scalar128, exact = sc.exact, shortrow = sc.shortrow, lists = sc.lists;
    const int slice = sc.slice;
    p.slice_cols = exact ? (slice > 0 ? slice : 128) : p.d;
    p.n_slices = exact ? (p.d + p.slice_cols - 1) / p.slice_cols : 1;
    if (p.src_slice_stride == 0) p.src_slice_stride = p.slice_cols;  // row-major source
    p.short_max = -1;
    p.short_groups = p.med_groups = 0;
    if (lists) {
        // the device lists exist from the first list-driven launch of this selection on (all-hops selections of mixed operands:
        // from plan creation, so that such launches can be captured into a hipGraph without a warm-up)
        int st = fill_short(plan, mask, SUM, sh, true);
        if (st != H2GCN_OK) return st;
        // list-driven launch: short-list workgroups serve 4 waves x 64 entries, medium-list workgroups 4 waves x
        // med_per_wave entries; forward -- one run of workgroups per selected hop and class
        int64_t n_listed = 0, n_med = 0, longest = 0;
        for (int s = 0; s < sh.n_short_lists; ++s) {
            n_listed += sh.short_count[s];
            n_med += sh.med_count[s];
            longest = std::max(longest, sh.short_count[s]);
        }
        // entries per wave follow the size of the launch (see kWavesToFill): a medium-list wave walks up to 8 segments
        // (SUM: 4 rows), a lane-group wave serves up to 64 entries (one coalesced list read, 16-32 rounds) -- fewer on small
        // operands, so that the launch still has a couple of waves per wave slot of the chip
        static const int env_mpw = getenv("H2GCN_MED_PER_WAVE") ? atoi(getenv("H2GCN_MED_PER_WAVE")) : 0;
        static const int env_spw = getenv("H2GCN_SHORT_PER_WAVE") ? atoi(getenv("H2GCN_SHORT_PER_WAVE")) : 0;
        static const int env_major = getenv("H2GCN_SHORT_HOP_MAJOR") ? atoi(getenv("H2GCN_SHORT_HOP_MAJOR")) : -1;
        const int mpw_max = SUM ? std::max(1, std::min(4, kWave / p.n_sel)) : 8;
        p.med_per_wave = (int)std::max<int64_t>(1, std::min<int64_t>(n_med / kWavesToFill, mpw_max));
        if (env_mpw >= 1 && env_mpw * p.n_sel <= kWave) p.med_per_wave = env_mpw;
        p.short_per_wave = 4;
        while (p.short_per_wave < 64 && n_listed / (2 * p.short_per_wave) >= kWavesToFill) p.short_per_wave *= 2;
        if (env_spw >= 4 && env_spw <= 64 && env_spw % 4 == 0) p.short_per_wave = env_spw;
        p.short_hop_major = env_major >= 0 ? env_major : 0;
        int64_t sblocks = 0, mblocks = 0;
        const int64_t s_per_block = (int64_t)kWavesPerBlock * p.short_per_wave;
        for (int s = 0; s < sh.n_short_lists; ++s) {
            p.short_list[s] = sh.short_list[s];
            p.short_count[s] = sh.short_count[s];
            sblocks += (sh.short_count[s] + s_per_block - 1) / s_per_block;
            p.short_end[s] = sblocks;
            p.med_list[s] = sh.med_list[s];
            p.med_count[s] = sh.med_count[s];
            const int64_t per_block = (int64_t)kWavesPerBlock * p.med_per_wave;
            mblocks += (sh.med_count[s] + per_block - 1) / per_block;
            p.med_end[s] = mblocks;
        }
        if (!SUM && !p.short_hop_major) sblocks = (longest + s_per_block - 1) / s_per_block * sh.n_short_lists;   // chunk-major: every hop padded to the longest list
        p.short_groups = (sblocks + kNumXcd - 1) / kNumXcd;
        p.med_groups = (mblocks + kNumXcd - 1) / kNumXcd;
        p.short_max = plan->short_max;
        p.blocks_per_slice = (int64_t)p.n_long + (p.short_groups + p.med_groups) * kNumXcd;
    } else {
        p.blocks_per_slice = (int64_t)p.n_long + p.tiles_per_xcd * kNumXcd;
    }
    const int64_t n_blocks = p.blocks_per_slice * p.n_slices;
    if (n_blocks <= 0) return H2GCN_OK;
    if (n_blocks > 0x7fffffffLL) return fail(H2GCN_ERR_INVALID_ARGUMENT, "grid too large (%lld blocks)", (long long)n_blocks);
    const dim3 grid((unsigned)n_blocks), block(kBlock);
#define H2GCN_LAUNCH_LISTS(VEC, LPR)                                                                                              \
    do {                                                                                                                          \
        if (off32 && short_fb4)                                                                                                   \
            hipLaunchKernelGGL((spmm_hops_kernel<VEC, LPR, true, SUM, true, false, false, false, 4, true>), grid, block, 0, stream, p);   \
        else if (off32)                                                                                                           \
            hipLaunchKernelGGL((spmm_hops_kernel<VEC, LPR, true, SUM, true, false, false, false, 8, true>), grid, block, 0, stream, p);   \
        else if (short_fb4)                                                                                                       \
            hipLaunchKernelGGL((spmm_hops_kernel<VEC, LPR, true, SUM, false, false, false, false, 4, true>), grid, block, 0, stream, p);  \
        else                                                                                                                      \
            hipLaunchKernelGGL((spmm_hops_kernel<VEC, LPR, true, SUM, false, false, false, false, 8, true>), grid, block, 0, stream, p);  \
    } while (0)
#define H2GCN_LAUNCH(VEC, LPR, EXACT)                                                                             \
    do {                                                                                                          \
        if (gen && off32)                                                                                         \
            hipLaunchKernelGGL((spmm_hops_kernel<VEC, LPR, EXACT, SUM, true, false, false, true>), grid, block, 0, stream, p);  \
        else if (gen)                                                                                             \
            hipLaunchKernelGGL((spmm_hops_kernel<VEC, LPR, EXACT, SUM, false, false, false, true>), grid, block, 0, stream, p); \
        else if (off32 && pipe && EXACT)                                                                               \
            hipLaunchKernelGGL((spmm_hops_kernel<VEC, LPR, EXACT, SUM, true, true>), grid, block, 0, stream, p);   \
        else if (off32)                                                                                           \
            hipLaunchKernelGGL((spmm_hops_kernel<VEC, LPR, EXACT, SUM, true, false>), grid, block, 0, stream, p);  \
        else if (pipe && EXACT)                                                                                   \
            hipLaunchKernelGGL((spmm_hops_kernel<VEC, LPR, EXACT, SUM, false, true>), grid, block, 0, stream, p);  \
        else                                                                                                      \
            hipLaunchKernelGGL((spmm_hops_kernel<VEC, LPR, EXACT, SUM, false, false>), grid, block, 0, stream, p); \
    } while (0)
    if (scalar128) {
        H2GCN_LAUNCH(2, 64, true);  // one neighbour per load instruction, scalar base addressing
    } else if (slice == 256) {
        H2GCN_LAUNCH(4, 64, true);
    } else if (slice == 128 && lists) {
        H2GCN_LAUNCH_LISTS(4, 32);
    } else if (slice == 64 && lists) {
        H2GCN_LAUNCH_LISTS(4, 16);
    } else if ((slice == 128 || slice == 64) && shortrow) {
        // the in-tile short-row kernels live in a translation unit of their own (spmm_short.hip): compiled next to them,
        // the tile-walk kernels of THIS file come out 2-6 VGPRs heavier and several of them spill (tools/kernel_resources.py)
        launch_in_tile_short(SUM, p, slice, off32, short_fb4, grid, stream);
    } else if (slice == 128) {
        H2GCN_LAUNCH(4, 32, true);
    } else if (slice == 64) {
        H2GCN_LAUNCH(4, 16, true);
    } else {
        H2GCN_LAUNCH(1, 64, false);  // d < 4: generic column-tiled path
    }
#undef H2GCN_LAUNCH
#undef H2GCN_LAUNCH_LISTS
    H2GCN_HIP_TRY(hipGetLastError());
    return H2GCN_OK;
}

// Copy the gather source into the slice-major scratch and point the launch at it.
void use_scratch(LaunchParams& p, const LaunchShape& sh, int rs, int64_t ld_src_hop, void* workspace, hipStream_t stream,
                 bool* off32) {
    const int n_hop = sh.adjoint ? sh.n_sel : 1;
    const int n_slices = (sh.d + rs - 1) / rs;
    const bool vec4 = sh.d % 4 == 0;
    const int64_t total = sh.n_src * (int64_t)n_slices * n_hop * (rs / (vec4 ? 4 : 1));
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 64);
    if (vec4)
        hipLaunchKernelGGL(h2gcn::repack_slice_major_kernel<true>, dim3(blocks), dim3(256), 0, stream, p.src, sh.ld_src,
                           ld_src_hop, n_hop, sh.n_src, sh.d, n_slices, rs, (float*)workspace);
    else
        hipLaunchKernelGGL(h2gcn::repack_slice_major_kernel<false>, dim3(blocks), dim3(256), 0, stream, p.src, sh.ld_src,
                           ld_src_hop, n_hop, sh.n_src, sh.d, n_slices, rs, (float*)workspace);
    p.src = (const float*)workspace;
    p.ld_src = rs;
    p.d_src = (sh.d + 3) & ~3;   // zero-padded: the lane that straddles d reads its own float4; lanes beyond re-read the last one
    p.src_slice_stride = sh.n_src * (int64_t)rs * n_hop;
    for (int s = 0; s < p.n_sel; ++s) p.src_hop_off[s] = sh.adjoint ? (int64_t)s * sh.n_src * rs : 0;
    *off32 = (double)sh.n_src * (double)n_slices * rs * n_hop * 4.0 < 4294967296.0;
}

int check_device(const h2gcn_plan* plan) {
    int cur = -1;
    H2GCN_HIP_TRY(hipGetDevice(&cur));
    if (cur != plan->device)
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "plan lives on device %d but the current HIP device is %d", plan->device, cur);
    return H2GCN_OK;
}

int resolve_mask(const h2gcn_plan* plan, uint32_t hop_mask, uint32_t* out) {
    const uint32_t all = plan->n_hops >= 32 ? 0xffffffffu : ((1u << plan->n_hops) - 1u);
    if (hop_mask == 0) hop_mask = all;
    if (hop_mask & ~all) return fail(H2GCN_ERR_INVALID_ARGUMENT, "hop_mask 0x%x selects hops beyond the plan's %d", hop_mask, plan->n_hops);
    *out = hop_mask;
    return H2GCN_OK;
}

}  // namespace

extern "C" {

int h2gcn_abi_version(void) { return H2GCN_ABI_VERSION; }

const char* h2gcn_last_error(void) { return g_last_error.c_str(); }

int h2gcn_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(H2GCN_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return n;
}

int h2gcn_plan_create(int n_hops, int64_t n_rows, int64_t n_cols, const int64_t* const* rowptr_dev,
                      const int32_t* const* colidx_dev, const float* const* vals_dev,
                      const h2gcn_plan_opts* opts, void* stream_v, h2gcn_plan_t** out_plan) {
    try {
        if (!out_plan) return fail(H2GCN_ERR_INVALID_ARGUMENT, "out_plan is NULL");
        *out_plan = nullptr;
        if (n_hops < 1 || n_hops > H2GCN_MAX_HOPS)
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "n_hops = %d, supported 1..%d", n_hops, H2GCN_MAX_HOPS);
        if (n_rows < 0 || n_cols < 0 || n_cols > 0x7fffffffLL)
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "bad shape %lld x %lld (column ids are int32)", (long long)n_rows, (long long)n_cols);
        if (n_rows >= (1LL << 58)) return fail(H2GCN_ERR_INVALID_ARGUMENT, "n_rows too large");
        if (!rowptr_dev || !colidx_dev || !vals_dev) return fail(H2GCN_ERR_INVALID_ARGUMENT, "NULL operand table");
        h2gcn_plan_opts o;
        memset(&o, 0, sizeof(o));
        if (opts) {
            if (opts->struct_size < 8 || opts->struct_size > sizeof(o))
                return fail(H2GCN_ERR_INVALID_ARGUMENT, "opts->struct_size = %u", opts->struct_size);
            memcpy(&o, opts, opts->struct_size);
        }
        if (o.slice_cols != 0 && o.slice_cols != 64 && o.slice_cols != 128 && o.slice_cols != 256)
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "slice_cols = %d, supported 0 (auto), 64, 128, 256", o.slice_cols);
        if (o.long_row_threshold < 0 || o.rows_per_wave < 0 || o.rows_per_wave > h2gcn::kMaxRowsPerWave)
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "bad tunable (long_row_threshold %d, rows_per_wave %d, max %d)",
                        o.long_row_threshold, o.rows_per_wave, h2gcn::kMaxRowsPerWave);
        hipStream_t stream = (hipStream_t)stream_v;

        std::unique_ptr<h2gcn_plan> plan(new h2gcn_plan());
        plan->n_hops = n_hops;
        plan->n_rows = n_rows;
        plan->n_cols = n_cols;
        if (o.long_row_threshold > 0) plan->long_threshold = o.long_row_threshold;
        if (o.rows_per_wave > 0) plan->rows_per_wave = o.rows_per_wave;
        else if (const char* e = getenv("H2GCN_ROWS_PER_WAVE")) plan->rows_per_wave = std::max(1, std::min(atoi(e), h2gcn::kMaxRowsPerWave));   // experiments
        else {
            // work per wave follows the size of the launch: on a graph of a few thousand rows (the reference's own datasets:
            // Cora 2 708, citeseer 3 327) a launch is a handful of workgroups and its duration is the dependent chain of ONE
            // wave, so a wave takes one row; from ~50 k rows on the chip is full either way and 4 rows per wave amortise
            // the prologue (Cora epoch 0.46 -> 0.37 ms with 1 row per wave; arxiv shape 0.193 ms with 4 vs 0.241 with 1:
            // profiles/r04_ab_work_per_wave.txt).  The row-to-wave assignment is invisible in the bits.
            plan->rows_per_wave = (int)std::max<int64_t>(1, std::min<int64_t>(n_rows / kWavesToFill, 4));
        }
        // (rows_per_wave + 1) * n_hops row pointers must fit one 64-lane load
        while ((plan->rows_per_wave + 1) * n_hops > h2gcn::kWave && plan->rows_per_wave > 1) plan->rows_per_wave--;
        plan->variant = o.variant;
        if (o.variant == 0) if (const char* e = getenv("H2GCN_VARIANT")) plan->variant = atoi(e);   // experiments through the entry points
        plan->slice_cols = o.slice_cols;
        if (o.slice_cols == 0) if (const char* e = getenv("H2GCN_SLICE_COLS")) plan->slice_cols = atoi(e);   // experiments through the entry points
        // segment classes: short <= short_max < medium < long_threshold <= long
        plan->short_max = std::min(h2gcn::kShortMax, plan->long_threshold - 1);
        if (const char* e = getenv("H2GCN_SHORT_MIN_FRAC")) plan->short_min_frac = atof(e);   // A/B measurements
        H2GCN_HIP_TRY(hipGetDevice(&plan->device));
        plan->fwd.resize(n_hops);

        DeviceBuf flag;
        H2GCN_HIP_TRY(hipMalloc(&flag.p, sizeof(int)));
        H2GCN_HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(int), stream));

        std::vector<std::vector<int64_t>> h_rowptr(n_hops);
        for (int k = 0; k < n_hops; ++k) {
            if (!rowptr_dev[k]) return fail(H2GCN_ERR_INVALID_ARGUMENT, "hop %d: rowptr is NULL", k);
            h_rowptr[k].resize(n_rows + 1);
            H2GCN_HIP_TRY(hipMemcpyAsync(h_rowptr[k].data(), rowptr_dev[k], (n_rows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
        }
        H2GCN_HIP_TRY(hipStreamSynchronize(stream));
        for (int k = 0; k < n_hops; ++k) {
            int st = check_rowptr(h_rowptr[k], n_rows, k);
            if (st != H2GCN_OK) return st;
            HopOperand& op = plan->fwd[k];
            op.rowptr = rowptr_dev[k];
            op.colidx = colidx_dev[k];
            op.vals = vals_dev[k];
            op.nnz = h_rowptr[k][n_rows];
            if (op.nnz > 0 && (!op.colidx || !op.vals)) return fail(H2GCN_ERR_INVALID_ARGUMENT, "hop %d: colidx/vals is NULL", k);
            collect_long_rows(h_rowptr[k], n_rows, plan->long_threshold, op.long_rows);
            collect_segment_lengths(h_rowptr[k], n_rows, plan->short_max, plan->long_threshold, op);
            if (!(o.flags & H2GCN_PLAN_SKIP_VALIDATION) && op.nnz > 0) {
                const int64_t want = (op.nnz + 255) / 256;
                const unsigned blocks = (unsigned)std::min<int64_t>(want, 4096);
                hipLaunchKernelGGL(h2gcn::check_colidx_kernel, dim3(blocks), dim3(256), 0, stream, op.colidx, op.nnz, n_cols, (int*)flag.p);
                H2GCN_HIP_TRY(hipGetLastError());
            }
        }
        int h_flag = 0;
        H2GCN_HIP_TRY(hipMemcpyAsync(&h_flag, flag.p, sizeof(int), hipMemcpyDeviceToHost, stream));
        H2GCN_HIP_TRY(hipStreamSynchronize(stream));
        if (h_flag) return fail(H2GCN_ERR_BAD_INDEX, "a column index lies outside [0, %lld)", (long long)n_cols);

        if (o.flags & H2GCN_PLAN_BUILD_TRANSPOSE) {
            if (n_rows > 0x7fffffffLL) return fail(H2GCN_ERR_INVALID_ARGUMENT, "transpose needs n_rows < 2^31");
            plan->adj.resize(n_hops);
            for (int k = 0; k < n_hops; ++k) {
                const int64_t nnz = plan->fwd[k].nnz;
                int64_t* t_rp = nullptr;
                int32_t* t_ci = nullptr;
                float* t_va = nullptr;
                uint32_t* t_perm = nullptr;
                const bool keep_perm = (o.flags & H2GCN_PLAN_KEEP_PERMUTATION) != 0;
                if (keep_perm && (o.flags & H2GCN_PLAN_HOST_TRANSPOSE))
                    return fail(H2GCN_ERR_INVALID_ARGUMENT, "H2GCN_PLAN_KEEP_PERMUTATION needs the device transposition");
                std::vector<int64_t> t_rowptr(n_cols + 1);
                std::string terr;
                int st = (o.flags & H2GCN_PLAN_HOST_TRANSPOSE)
                             ? -1
                             : h2gcn::transpose_csr_device(n_rows, n_cols, nnz, rowptr_dev[k], colidx_dev[k], vals_dev[k],
                                                           &t_rp, &t_ci, &t_va, keep_perm ? &t_perm : nullptr, stream, &terr);
                if (st != 0 && keep_perm && st != -3)
                    return fail(H2GCN_ERR_INVALID_ARGUMENT, "H2GCN_PLAN_KEEP_PERMUTATION: %s", terr.c_str());
                if (st == 0) {
                    plan->owned.emplace_back(); plan->owned.back().p = t_rp;
                    plan->owned.emplace_back(); plan->owned.back().p = t_ci;
                    plan->owned.emplace_back(); plan->owned.back().p = t_va;
                    plan->owned.emplace_back(); plan->owned.back().p = t_perm;
                    H2GCN_HIP_TRY(hipMemcpy(t_rowptr.data(), t_rp, (n_cols + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
                } else if (st == -3) {
                    return fail(H2GCN_ERR_OUT_OF_MEMORY, "transposition of hop %d: %s", k, terr.c_str());
                } else {
                    // host path: requested explicitly, or the device path declined (>= 2^32 nonzeros)
                    std::vector<int32_t> h_col(nnz);
                    std::vector<float> h_val(nnz);
                    if (nnz > 0) {
                        H2GCN_HIP_TRY(hipMemcpy(h_col.data(), colidx_dev[k], nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
                        H2GCN_HIP_TRY(hipMemcpy(h_val.data(), vals_dev[k], nnz * sizeof(float), hipMemcpyDeviceToHost));
                    }
                    std::vector<int32_t> t_col;
                    std::vector<float> t_val;
                    transpose_csr_host(n_rows, n_cols, h_rowptr[k], h_col, h_val, t_rowptr, t_col, t_val);
                    plan->owned.emplace_back();
                    DeviceBuf& d_rp = plan->owned.back();
                    H2GCN_HIP_TRY(hipMalloc(&d_rp.p, (n_cols + 1) * sizeof(int64_t)));
                    H2GCN_HIP_TRY(hipMemcpy(d_rp.p, t_rowptr.data(), (n_cols + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
                    plan->owned.emplace_back();
                    DeviceBuf& d_ci = plan->owned.back();
                    plan->owned.emplace_back();
                    DeviceBuf& d_va = plan->owned.back();
                    if (nnz > 0) {
                        H2GCN_HIP_TRY(hipMalloc(&d_ci.p, nnz * sizeof(int32_t)));
                        H2GCN_HIP_TRY(hipMemcpy(d_ci.p, t_col.data(), nnz * sizeof(int32_t), hipMemcpyHostToDevice));
                        H2GCN_HIP_TRY(hipMalloc(&d_va.p, nnz * sizeof(float)));
                        H2GCN_HIP_TRY(hipMemcpy(d_va.p, t_val.data(), nnz * sizeof(float), hipMemcpyHostToDevice));
                    }
                    t_rp = (int64_t*)d_rp.p;
                    t_ci = (int32_t*)d_ci.p;
                    t_va = (float*)d_va.p;
                }
                HopOperand& op = plan->adj[k];
                op.rowptr = t_rp;
                op.colidx = t_ci;
                op.vals = t_va;
                op.nnz = nnz;
                op.perm = t_perm;
                collect_long_rows(t_rowptr, n_cols, plan->long_threshold, op.long_rows);
                collect_segment_lengths(t_rowptr, n_cols, plan->short_max, plan->long_threshold, op);
            }
            plan->has_transpose = true;
        }
        // build the long-segment lists of the common selection (all hops) now, so that launches with the default
        // hop mask never allocate -- they can be issued inside a hipGraph capture without a warm-up
        {
            const uint32_t all = n_hops >= 32 ? 0xffffffffu : ((1u << n_hops) - 1u);
            const int64_t* unused_p = nullptr;
            int unused_n = 0;
            int st = get_long_list(plan.get(), false, all, &unused_p, &unused_n);
            if (st == H2GCN_OK && plan->has_transpose) st = get_long_list(plan.get(), true, all, &unused_p, &unused_n);
            if (st != H2GCN_OK) return st;
            // ... and, for MIXED operands (the all-hops launch will be list-driven at the usual widths), the segment-class lists
            // of that selection; everything else builds them on the first launch that wants them
            auto mixed = [&](const LaunchShape& sh_) {
                return plan->variant == 6 || (plan->variant == 0 && sh_.short_frac >= plan->short_min_frac &&
                                              (sh_.avg >= 16.0 || sh_.n_out < 4 * kWavesToFill));
            };
            LaunchShape sh_f = shape_of(plan.get(), all, false);
            if ((st = fill_short(plan.get(), all, false, sh_f)) != H2GCN_OK) return st;
            if (mixed(sh_f) && (st = fill_short(plan.get(), all, false, sh_f, true)) != H2GCN_OK) return st;
            if (plan->has_transpose) {
                LaunchShape sh_a = shape_of(plan.get(), all, true);
                if ((st = fill_short(plan.get(), all, true, sh_a)) != H2GCN_OK) return st;
                if (mixed(sh_a) && (st = fill_short(plan.get(), all, true, sh_a, true)) != H2GCN_OK) return st;
            }
        }
        *out_plan = plan.release();
        return H2GCN_OK;
    } catch (const std::bad_alloc&) {
        return fail(H2GCN_ERR_OUT_OF_MEMORY, "host allocation failed in plan_create");
    } catch (...) {
        return fail(H2GCN_ERR_INTERNAL, "unexpected exception in plan_create");
    }
}

void h2gcn_plan_destroy(h2gcn_plan_t* plan) { delete plan; }

int h2gcn_plan_set_values(h2gcn_plan_t* plan, int hop, const float* vals_dev, void* stream_v) {
    if (!plan) return fail(H2GCN_ERR_INVALID_ARGUMENT, "plan is NULL");
    if (hop < 0 || hop >= plan->n_hops) return fail(H2GCN_ERR_INVALID_ARGUMENT, "hop %d outside 0..%d", hop, plan->n_hops - 1);
    HopOperand& f = plan->fwd[hop];
    if (f.nnz > 0 && !vals_dev) return fail(H2GCN_ERR_INVALID_ARGUMENT, "vals is NULL");
    int st = check_device(plan);
    if (st != H2GCN_OK) return st;
    if (plan->has_transpose) {
        HopOperand& a = plan->adj[hop];
        if (a.nnz > 0 && !a.perm)
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "plan has transposed operands but was created without H2GCN_PLAN_KEEP_PERMUTATION");
        h2gcn::permute_values(a.perm, vals_dev, a.nnz, const_cast<float*>(a.vals), (hipStream_t)stream_v);
        H2GCN_HIP_TRY(hipGetLastError());
    }
    f.vals = vals_dev;
    return H2GCN_OK;
}

int h2gcn_plan_info(const h2gcn_plan_t* plan, int hop, int64_t* n_rows, int64_t* n_cols, int64_t* nnz,
                    int64_t* n_long_segments, int32_t* has_transpose) {
    if (!plan) return fail(H2GCN_ERR_INVALID_ARGUMENT, "plan is NULL");
    if (hop < 0 || hop >= plan->n_hops) return fail(H2GCN_ERR_INVALID_ARGUMENT, "hop %d outside 0..%d", hop, plan->n_hops - 1);
    if (n_rows) *n_rows = plan->n_rows;
    if (n_cols) *n_cols = plan->n_cols;
    if (nnz) *nnz = plan->fwd[hop].nnz;
    if (n_long_segments) *n_long_segments = (int64_t)plan->fwd[hop].long_rows.size();
    if (has_transpose) *has_transpose = plan->has_transpose ? 1 : 0;
    return H2GCN_OK;
}

namespace {
// hop selection -> LaunchShape (src_vec_ok / ld_src / d are filled in by the caller)
// every gathered row starts on a 128-byte cache line
bool line_aligned(const float* src, int64_t ld_src, int64_t ld_src_hop, int n_sel, bool adjoint) {
    return (reinterpret_cast<uintptr_t>(src) & 127u) == 0 && (ld_src * 4) % 128 == 0 && (!adjoint || n_sel <= 1 || (ld_src_hop * 4) % 128 == 0);
}

LaunchShape shape_of(const h2gcn_plan* plan, uint32_t mask, bool adjoint) {
    LaunchShape sh;
    memset(&sh, 0, sizeof(sh));
    sh.adjoint = adjoint;
    const std::vector<HopOperand>& ops = adjoint ? plan->adj : plan->fwd;
    for (int k = 0; k < plan->n_hops; ++k)
        if (mask & (1u << k)) { sh.nnz_sel += ops[k].nnz; ++sh.n_sel; }
    sh.n_out = adjoint ? plan->n_cols : plan->n_rows;
    sh.n_src = adjoint ? plan->n_rows : plan->n_cols;
    sh.avg = (sh.n_out > 0 && sh.n_sel > 0) ? (double)sh.nnz_sel / ((double)sh.n_out * sh.n_sel) : 0.0;
    return sh;
}

// The binned short segments this hop selection can use (lists are built on first use and cached in the plan).
// Shares of the launch's segments / nonzeros in the short class (always), and -- build_lists -- the device lists themselves.
int fill_short(const h2gcn_plan* plan, uint32_t mask, bool adjoint, LaunchShape& sh, bool build_lists) {
    sh.n_short_lists = 0;
    sh.short_frac = sh.short_nnz_frac = 0.0;
    if (plan->short_max < 0 || sh.n_out <= 0 || sh.n_sel <= 0) return H2GCN_OK;
    int64_t n = 0, nnz = 0;
    const ClassLists* cl = nullptr;
    if (!adjoint) {
        int s = 0;
        for (int k = 0; k < plan->n_hops; ++k) {
            if (!(mask & (1u << k))) continue;
            int st = get_class_lists(plan, false, k, 0, build_lists, &cl);
            if (st != H2GCN_OK) return st;
            if (!cl) return H2GCN_OK;
            sh.short_list[s] = (const int32_t*)cl->short_dev.p;
            sh.short_count[s] = cl->n_short;
            sh.med_list[s] = (const int32_t*)cl->med_dev.p;
            sh.med_count[s] = cl->n_med;
            n += cl->n_short;
            nnz += cl->nnz_short;
            ++s;
        }
        sh.n_short_lists = s;
        sh.short_frac = (double)n / ((double)sh.n_out * sh.n_sel);
    } else {
        if (sh.n_sel > h2gcn::kShortSumHops) return H2GCN_OK;
        int st = get_class_lists(plan, true, 0, mask, build_lists, &cl);
        if (st != H2GCN_OK) return st;
        if (!cl) return H2GCN_OK;
        sh.short_list[0] = (const int32_t*)cl->short_dev.p;
        sh.short_count[0] = cl->n_short;
        sh.med_list[0] = (const int32_t*)cl->med_dev.p;
        sh.med_count[0] = cl->n_med;
        n = cl->n_short;
        nnz = cl->nnz_short;
        sh.n_short_lists = 1;
        sh.short_frac = (double)n / (double)sh.n_out;
    }
    if (n == 0) sh.n_short_lists = 0;
    sh.short_nnz_frac = sh.nnz_sel > 0 ? (double)nnz / (double)sh.nnz_sel : (n > 0 ? 1.0 : 0.0);
    return H2GCN_OK;
}

}  // namespace

int h2gcn_plan_schedule(const h2gcn_plan_t* plan, uint32_t hop_mask, int adjoint, int64_t ld_src, int32_t d,
                        int32_t* slice_cols, int32_t* n_slices, int32_t* segment_walk, int32_t* scratch_copy) {
    if (!plan) return fail(H2GCN_ERR_INVALID_ARGUMENT, "plan is NULL");
    uint32_t mask;
    int st = resolve_mask(plan, hop_mask, &mask);
    if (st != H2GCN_OK) return st;
    if (d < 1 || ld_src < d) return fail(H2GCN_ERR_INVALID_ARGUMENT, "bad width %d / stride %lld", d, (long long)ld_src);
    if (adjoint && !plan->has_transpose) return fail(H2GCN_ERR_NO_TRANSPOSE, "plan was created without H2GCN_PLAN_BUILD_TRANSPOSE");
    LaunchShape sh = shape_of(plan, mask, adjoint != 0);
    if ((st = fill_short(plan, mask, adjoint != 0, sh)) != H2GCN_OK) return st;
    sh.ld_src = ld_src;
    sh.d = d;
    sh.src_line_aligned = (ld_src * 4) % 128 == 0 && (!adjoint || sh.n_sel <= 1 || (d * 4) % 128 == 0);  // aligned base assumed
    const int rs = scratch_slice_cols(plan, sh);
    const Schedule sc = decide(plan->variant, d >= 4, d, plan->rows_per_wave, sh.n_sel, rs > 0 ? rs : plan->slice_cols, sh.n_src, sh.avg,
                               rs > 0 || sh.src_line_aligned, d % 4 != 0, sh.short_frac, sh.short_nnz_frac, plan->short_min_frac, sh.n_out);
    const int w = sc.exact ? (sc.slice > 0 ? sc.slice : 128) : d;
    if (slice_cols) *slice_cols = w;
    if (n_slices) *n_slices = sc.exact ? (d + w - 1) / w : 1;
    if (segment_walk) *segment_walk = sc.pipe ? 1 : (sc.shortrow ? 2 : (sc.lists ? 3 : 0));
    if (scratch_copy) *scratch_copy = rs > 0 ? 1 : 0;
    return H2GCN_OK;
}

int h2gcn_plan_segment_classes(const h2gcn_plan_t* plan, uint32_t hop_mask, int adjoint, int64_t ld_src, int32_t d,
                               int64_t* segments, int64_t* nonzeros, int64_t* listed) {
    if (!plan) return fail(H2GCN_ERR_INVALID_ARGUMENT, "plan is NULL");
    uint32_t mask;
    int st = resolve_mask(plan, hop_mask, &mask);
    if (st != H2GCN_OK) return st;
    if (d < 1 || ld_src < d) return fail(H2GCN_ERR_INVALID_ARGUMENT, "bad width %d / stride %lld", d, (long long)ld_src);
    if (adjoint && !plan->has_transpose) return fail(H2GCN_ERR_NO_TRANSPOSE, "plan was created without H2GCN_PLAN_BUILD_TRANSPOSE");
    const std::vector<HopOperand>& ops = adjoint ? plan->adj : plan->fwd;
    const int64_t n_out = adjoint ? plan->n_cols : plan->n_rows;
    int s = 0;
    for (int k = 0; k < plan->n_hops; ++k) {
        if (!(mask & (1u << k))) continue;
        const HopOperand& op = ops[k];
        const int64_t n_long = (int64_t)op.long_rows.size();
        if (segments) {
            segments[3 * s + 0] = op.n_short;
            segments[3 * s + 1] = n_out - op.n_short - n_long;
            segments[3 * s + 2] = n_long;
        }
        if (nonzeros) {
            nonzeros[3 * s + 0] = op.nnz_short;
            nonzeros[3 * s + 1] = op.nnz - op.nnz_short - op.nnz_long;
            nonzeros[3 * s + 2] = op.nnz_long;
        }
        ++s;
    }
    if (listed) {
        LaunchShape sh = shape_of(plan, mask, adjoint != 0);
        if ((st = fill_short(plan, mask, adjoint != 0, sh)) != H2GCN_OK) return st;
        sh.ld_src = ld_src;
        sh.d = d;
        sh.src_line_aligned = (ld_src * 4) % 128 == 0 && (!adjoint || sh.n_sel <= 1 || (d * 4) % 128 == 0);
        const int rs = scratch_slice_cols(plan, sh);
        const Schedule sc = decide(plan->variant, d >= 4, d, plan->rows_per_wave, sh.n_sel, rs > 0 ? rs : plan->slice_cols, sh.n_src, sh.avg,
                                   rs > 0 || sh.src_line_aligned, d % 4 != 0, sh.short_frac, sh.short_nnz_frac, plan->short_min_frac, sh.n_out);
        *listed = 0;
        if (sc.lists)
            for (int q = 0; q < sh.n_short_lists; ++q) *listed += sh.short_count[q];
        if (sc.shortrow) *listed = -1;   // in-tile short-row mode: every round of G consecutive short rows is grouped, no list
    }
    return H2GCN_OK;
}

size_t h2gcn_spmm_workspace_bytes(const h2gcn_plan_t* plan, uint32_t hop_mask, int adjoint, const float* src_dev,
                                  int64_t ld_src, int64_t ld_src_hop, int32_t d) {
    if (!plan || d < 1 || ld_src < d) return 0;
    if (adjoint && !plan->has_transpose) return 0;
    uint32_t mask;
    if (resolve_mask(plan, hop_mask, &mask) != H2GCN_OK) return 0;
    LaunchShape sh = shape_of(plan, mask, adjoint != 0);
    sh.src_line_aligned = line_aligned(src_dev, ld_src, ld_src_hop, sh.n_sel, adjoint != 0);
    sh.ld_src = ld_src;
    sh.d = d;
    return scratch_bytes(sh, scratch_slice_cols(plan, sh));
}

int h2gcn_spmm_hops_f32(const h2gcn_plan_t* plan, uint32_t hop_mask, const float* X, int64_t ldx, int32_t d,
                        float* Y, int64_t ldy_row, int64_t ldy_hop, void* stream_v) {
    return h2gcn_spmm_hops_opts_f32(plan, hop_mask, X, ldx, d, Y, ldy_row, ldy_hop, nullptr, stream_v);
}

namespace {
int read_launch_opts(const h2gcn_launch_opts* lopts, h2gcn_launch_opts* lo) {
    memset(lo, 0, sizeof(*lo));
    if (lopts) {
        if (lopts->struct_size < 8 || lopts->struct_size > sizeof(*lo))
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "launch opts struct_size = %u", lopts->struct_size);
        memcpy(lo, lopts, lopts->struct_size);
    }
    if (lo->flags & ~(uint32_t)(H2GCN_LAUNCH_RELU | H2GCN_LAUNCH_ACCUMULATE)) return fail(H2GCN_ERR_INVALID_ARGUMENT, "unknown launch flags 0x%x", lo->flags);
    return H2GCN_OK;
}
}  // namespace

int h2gcn_spmm_hops_opts_f32(const h2gcn_plan_t* plan, uint32_t hop_mask, const float* X, int64_t ldx, int32_t d,
                             float* Y, int64_t ldy_row, int64_t ldy_hop, const h2gcn_launch_opts* lopts,
                             void* stream_v) {
    try {
        h2gcn_launch_opts lo;
        int st = read_launch_opts(lopts, &lo);
        if (st != H2GCN_OK) return st;
        if (lo.flags & ~H2GCN_LAUNCH_RELU) return fail(H2GCN_ERR_INVALID_ARGUMENT, "forward launch flags 0x%x: only H2GCN_LAUNCH_RELU applies", lo.flags);
        if (!plan) return fail(H2GCN_ERR_INVALID_ARGUMENT, "plan is NULL");
        uint32_t mask;
        st = resolve_mask(plan, hop_mask, &mask);
        if (st != H2GCN_OK) return st;
        if ((st = check_device(plan)) != H2GCN_OK) return st;
        if (d < 1) return fail(H2GCN_ERR_INVALID_ARGUMENT, "d = %d", d);
        if (plan->n_rows == 0) return H2GCN_OK;
        if (!Y) return fail(H2GCN_ERR_INVALID_ARGUMENT, "Y is NULL");
        if (!X && plan->n_cols > 0) return fail(H2GCN_ERR_INVALID_ARGUMENT, "X is NULL");
        if (ldx < d) return fail(H2GCN_ERR_INVALID_ARGUMENT, "ldx = %lld < d = %d", (long long)ldx, d);
        if (ldy_hop < 0 || ldy_row < 0) return fail(H2GCN_ERR_INVALID_ARGUMENT, "negative output stride");
        CaptureScope capture_scope((hipStream_t)stream_v);
        LaunchParams p;
        memset(&p, 0, sizeof(p));
        LaunchShape sh = shape_of(plan, mask, false);
        int s = 0;
        for (int k = 0; k < plan->n_hops; ++k) {
            if (!(mask & (1u << k))) continue;
            const HopOperand& op = plan->fwd[k];
            p.hop[s] = HopCsr{op.rowptr, op.colidx, op.vals};
            p.src_hop_off[s] = 0;
            p.dst_hop_off[s] = (int64_t)s * ldy_hop;
            ++s;
        }
        p.n_sel = s;
        if (s > 1 && ldy_hop < d) return fail(H2GCN_ERR_INVALID_ARGUMENT, "ldy_hop = %lld < d = %d: hop outputs would overlap", (long long)ldy_hop, d);
        p.d = d;
        p.d_src = d;
        p.n_rows = plan->n_rows;
        p.src = X;
        p.ld_src = ldx;
        p.dst = Y;
        p.ld_dst = ldy_row;
        p.bias = lo.bias_dev;
        p.relu = (lo.flags & H2GCN_LAUNCH_RELU) ? 1 : 0;
        st = get_long_list(plan, false, mask, &p.long_list, &p.n_long);
        if (st != H2GCN_OK) return st;
        p.long_threshold = plan->long_threshold;
        p.rows_per_wave = plan->rows_per_wave;
        const int64_t rows_per_tile = (int64_t)p.rows_per_wave * h2gcn::kWavesPerBlock;
        p.n_tiles = (p.n_rows + rows_per_tile - 1) / rows_per_tile;
        p.tiles_per_xcd = (p.n_tiles + h2gcn::kNumXcd - 1) / h2gcn::kNumXcd;
        // 32-bit gather offsets when the farthest byte of X is below 4 GiB
        bool off32 = ((double)(plan->n_cols > 0 ? plan->n_cols - 1 : 0) * (double)ldx + d) * 4.0 < 4294967296.0;
        int forced_slice = plan->slice_cols;
        sh.ld_src = ldx;
        sh.d = d;
        sh.src_line_aligned = line_aligned(X, ldx, 0, s, false);
        const int rs = (lo.workspace_dev && aligned16(lo.workspace_dev)) ? scratch_slice_cols(plan, sh) : 0;
        if (rs > 0 && lo.workspace_bytes >= scratch_bytes(sh, rs)) {
            use_scratch(p, sh, rs, 0, lo.workspace_dev, (hipStream_t)stream_v, &off32);
            H2GCN_HIP_TRY(hipGetLastError());
            forced_slice = rs;
        }
        if ((st = fill_short(plan, mask, false, sh)) != H2GCN_OK) return st;
        return launch<false>(p, plan, mask, sh, off32, forced_slice, (hipStream_t)stream_v);
    } catch (...) {
        return fail(H2GCN_ERR_INTERNAL, "unexpected exception in spmm_hops_opts_f32");
    }
}

int h2gcn_spmm_hops_T_f32(const h2gcn_plan_t* plan, uint32_t hop_mask, const float* dY, int64_t ldg_row,
                          int64_t ldg_hop, int32_t d, float* dX, int64_t ldx, void* stream_v) {
    return h2gcn_spmm_hops_T_opts_f32(plan, hop_mask, dY, ldg_row, ldg_hop, d, dX, ldx, nullptr, stream_v);
}

int h2gcn_spmm_hops_T_opts_f32(const h2gcn_plan_t* plan, uint32_t hop_mask, const float* dY, int64_t ldg_row,
                               int64_t ldg_hop, int32_t d, float* dX, int64_t ldx, const h2gcn_launch_opts* lopts,
                               void* stream_v) {
    try {
        h2gcn_launch_opts lo;
        int st = read_launch_opts(lopts, &lo);
        if (st != H2GCN_OK) return st;
        if (lo.bias_dev || (lo.flags & ~H2GCN_LAUNCH_ACCUMULATE))
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "the adjoint launch has no bias / activation epilogue (only H2GCN_LAUNCH_ACCUMULATE is accepted)");
        if (!plan) return fail(H2GCN_ERR_INVALID_ARGUMENT, "plan is NULL");
        if (!plan->has_transpose)
            return fail(H2GCN_ERR_NO_TRANSPOSE, "plan was created without H2GCN_PLAN_BUILD_TRANSPOSE");
        uint32_t mask;
        st = resolve_mask(plan, hop_mask, &mask);
        if (st != H2GCN_OK) return st;
        if ((st = check_device(plan)) != H2GCN_OK) return st;
        if (d < 1) return fail(H2GCN_ERR_INVALID_ARGUMENT, "d = %d", d);
        if (plan->n_cols == 0) return H2GCN_OK;
        if (!dX) return fail(H2GCN_ERR_INVALID_ARGUMENT, "dX is NULL");
        if (!dY && plan->n_rows > 0) return fail(H2GCN_ERR_INVALID_ARGUMENT, "dY is NULL");
        if (ldx < d) return fail(H2GCN_ERR_INVALID_ARGUMENT, "ldx = %lld < d = %d", (long long)ldx, d);
        if (ldg_row < d || ldg_hop < 0) return fail(H2GCN_ERR_INVALID_ARGUMENT, "bad gradient strides");
        CaptureScope capture_scope((hipStream_t)stream_v);
        LaunchParams p;
        memset(&p, 0, sizeof(p));
        LaunchShape sh = shape_of(plan, mask, true);
        int s = 0;
        for (int k = 0; k < plan->n_hops; ++k) {
            if (!(mask & (1u << k))) continue;
            const HopOperand& op = plan->adj[k];
            p.hop[s] = HopCsr{op.rowptr, op.colidx, op.vals};
            p.src_hop_off[s] = (int64_t)s * ldg_hop;
            p.dst_hop_off[s] = 0;
            ++s;
        }
        p.n_sel = s;
        p.d = d;
        p.d_src = d;
        p.n_rows = plan->n_cols;
        p.src = dY;
        p.ld_src = ldg_row;
        p.dst = dX;
        p.ld_dst = ldx;
        p.accumulate = (lo.flags & H2GCN_LAUNCH_ACCUMULATE) ? 1 : 0;
        st = get_long_list(plan, true, mask, &p.long_list, &p.n_long);
        if (st != H2GCN_OK) return st;
        p.long_threshold = plan->long_threshold;
        p.rows_per_wave = plan->rows_per_wave;
        const int64_t rows_per_tile = (int64_t)p.rows_per_wave * h2gcn::kWavesPerBlock;
        p.n_tiles = (p.n_rows + rows_per_tile - 1) / rows_per_tile;
        p.tiles_per_xcd = (p.n_tiles + h2gcn::kNumXcd - 1) / h2gcn::kNumXcd;
        bool off32 = ((double)(plan->n_rows > 0 ? plan->n_rows - 1 : 0) * (double)ldg_row + (double)(s - 1) * (double)ldg_hop + d) * 4.0 < 4294967296.0;
        int forced_slice = plan->slice_cols;
        sh.ld_src = ldg_row;
        sh.d = d;
        sh.src_line_aligned = line_aligned(dY, ldg_row, ldg_hop, s, true);
        const int rs = (lo.workspace_dev && aligned16(lo.workspace_dev) && plan->n_rows > 0) ? scratch_slice_cols(plan, sh) : 0;
        if (rs > 0 && lo.workspace_bytes >= scratch_bytes(sh, rs)) {
            use_scratch(p, sh, rs, ldg_hop, lo.workspace_dev, (hipStream_t)stream_v, &off32);
            H2GCN_HIP_TRY(hipGetLastError());
            forced_slice = rs;
        }
        if ((st = fill_short(plan, mask, true, sh)) != H2GCN_OK) return st;
        return launch<true>(p, plan, mask, sh, off32, forced_slice, (hipStream_t)stream_v);
    } catch (...) {
        return fail(H2GCN_ERR_INTERNAL, "unexpected exception in spmm_hops_T_f32");
    }
}

}  // extern "C"
