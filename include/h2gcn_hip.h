/*
 * h2gcn_hip.h -- C ABI of libh2gcn_hip.so: the MI355X (gfx950) hop-aggregation path of H2GCN.
 *
 * The library replaces exactly one thing in the reference (GemsLab/H2GCN): the sparse aggregation that
 * `GCNLayer.call` performs through TensorFlow,
 *
 *     tf.stack([tf.sparse.sparse_dense_matmul(A_k, X) for A_k in adjhops], axis=-2)      -> [N, H, d]
 *
 * (reference h2gcn/models/_layers.py:54-81; the op underneath is TensorFlow's SparseTensorDenseMatMul,
 * call sites _layers.py:74,76) and its gradient wrt X (`dX = sum_k A_k^T dY[:,k,:]`, reached from
 * tape.gradient at h2gcn/models/H2GCN.py:66-74).  There is no native boundary in the reference to copy:
 * the reference's native boundary is TensorFlow's op registry.  Each entry point below cites the reference
 * behaviour it stands in for.
 *
 * Conventions
 *   - plain C linkage, POD arguments, no C++/torch types, no exceptions across the boundary;
 *   - every pointer named *_dev is a DEVICE pointer of the current HIP device, owned by the caller and
 *     required to outlive the plan / the launch that uses it;
 *   - every function returning int returns H2GCN_OK (0) or a negative h2gcn_status; the message for the last
 *     failure on the calling thread is h2gcn_last_error();
 *   - launches are asynchronous on the given stream (a hipStream_t passed as void*; NULL = default stream);
 *   - a plan is immutable after creation, so concurrent launches that share a plan are safe.
 *
 * Operand layout (what `sparse2Tensor` + `tf.sparse.reorder` produce in the reference,
 * h2gcn/datasets/_dataset.py:528-535, re-expressed as CSR):
 *   rowptr  int64 [n_rows+1]   ascending, rowptr[0] == 0
 *   colidx  int32 [nnz]        ascending inside each row (row-major canonical order), 0 <= col < n_cols
 *   vals    fp32  [nnz]        the normalised adjacency values (SYM: D^-1/2 A_k D^-1/2, RW: D^-1 A_k;
 *                              _dataset.py:109-124) -- the kernel is agnostic to how they were normalised.
 */
#ifndef H2GCN_HIP_H_
#define H2GCN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H2GCN_ABI_VERSION 4
#define H2GCN_MAX_HOPS 8

typedef enum h2gcn_status {
    H2GCN_OK = 0,
    H2GCN_ERR_INVALID_ARGUMENT = -1, /* NULL pointer, bad size/stride/alignment, bad hop mask            */
    H2GCN_ERR_HIP = -2,              /* a HIP runtime call failed (message carries hipGetErrorString)     */
    H2GCN_ERR_OUT_OF_MEMORY = -3,    /* host or device allocation failed                                  */
    H2GCN_ERR_BAD_INDEX = -4,        /* rowptr not monotone / colidx out of range (TF: InvalidArgument)   */
    H2GCN_ERR_NO_TRANSPOSE = -5,     /* adjoint launch on a plan created without H2GCN_PLAN_BUILD_TRANSPOSE */
    H2GCN_ERR_INTERNAL = -6,
    H2GCN_ERR_EXCHANGE_TIMEOUT = -7  /* a peer's shard did not arrive within the exchange's time limit      */
} h2gcn_status;

/* flags for h2gcn_plan_opts.flags */
#define H2GCN_PLAN_BUILD_TRANSPOSE 0x1u /* also build A_k^T (needed by h2gcn_spmm_hops_T_f32)               */
#define H2GCN_PLAN_SKIP_VALIDATION 0x2u /* skip the one-time column-range check on the device               */
#define H2GCN_PLAN_HOST_TRANSPOSE  0x4u /* build A_k^T with the host counting sort instead of the device radix
                                           sort (debug / cross-check; same result)                            */

#define H2GCN_PLAN_KEEP_PERMUTATION 0x8u /* with BUILD_TRANSPOSE: remember which forward entry every transposed entry
                                           came from, so that h2gcn_plan_set_values can refresh A_k^T (+4 B/nonzero) */

/* Tunables of the CSR-adaptive schedule.  Zero in a field means "library default". */
typedef struct h2gcn_plan_opts {
    uint32_t struct_size;        /* = sizeof(h2gcn_plan_opts), for forward compatibility                   */
    uint32_t flags;              /* H2GCN_PLAN_*                                                            */
    int32_t long_row_threshold;  /* (row,hop) segments with >= this many nonzeros are split across the
                                    waves of one workgroup (LDS-staged partial sums); default 256          */
    int32_t rows_per_wave;       /* consecutive rows a wave walks in the regular path; default 4           */
    int32_t variant;             /* segment walks of the kernel: 0 = default, CSR-adaptive -- long segments
                                    (>= long_row_threshold): one workgroup each, always; launches of >= ~50 k rows whose segments
                                    average < 16 nonzeros: in-tile short-row mode (G consecutive short rows per
                                    round, one lane group each) when the slice is 64 or 128 columns, else the wave
                                    walk with index prefetch across segments; MIXED launches (mean >= 16 but at
                                    least 5 % of the segments have <= 16 nonzeros, slice 64 / 128): list-driven by
                                    segment class -- short segments from the plan's binned list, one lane group
                                    each, medium ones one wave each; otherwise one wave per segment.
                                    1 = scalar-addressed float2 gathers at d=128; 2 = wave walk, always prefetch;
                                    3 = plain wave walk; 4 = default, but never use the slice-major scratch copy
                                    (A/B measurements); 5 = force the in-tile short-row mode; 6 = force the
                                    list-driven launch (whenever a short segment exists).  Variants 0, 2, 3, 4, 5,
                                    6 give identical bits.                                                      */
    int32_t slice_cols;          /* feature columns per slice of the slice-major schedule: 64 / 128 / 256,
                                    0 = heuristic (narrower slices when X is far beyond the Infinity Cache).
                                    All three give identical bits (one canonical summation tree); widths
                                    below 64 columns run as one masked 64-column slice                            */
    int32_t reserved[2];
} h2gcn_plan_opts;

/* Opaque: segment-class bins (long-segment lists, binned short-segment lists), optional transposed CSR, launch geometry. */
typedef struct h2gcn_plan h2gcn_plan_t;

/* ABI version of the loaded library (== H2GCN_ABI_VERSION it was built with). */
int h2gcn_abi_version(void);

/* Message of the last failure on this thread ("" if none).  Never NULL; valid until the next failing call
 * on the same thread. */
const char* h2gcn_last_error(void);

/* Number of HIP devices visible; negative h2gcn_status when the runtime cannot be initialised. */
int h2gcn_device_count(void);

/*
 * Build the launch plan for a list of hop matrices A_0..A_{H-1} that share one row space.
 * Stands in for: the once-per-run construction of `tensors.adj_hops`
 * (reference h2gcn/datasets/_dataset.py:559-576) as far as the device side is concerned -- the normalised
 * values themselves are computed by the caller.
 *
 *   n_hops            1..H2GCN_MAX_HOPS  (H2GCN uses 2: exact-1-hop and exact-2-hop neighbourhoods; the reference takes
 *                     as many groups as --adj_nhood lists).  Every hop count gives the same per-row summation tree.
 *                     Scheduling above 4 SELECTED hops: the forward keeps its per-hop short-segment lists at any
 *                     count; the adjoint (SUM mode: one list of the rows that are short in every selected hop, staged
 *                     through LDS per hop) keeps its list for selections of up to 4 hops and walks the row tiles for
 *                     5..8 -- a scheduling difference only, the bits do not change.  rows_per_wave is at most 7, so
 *                     (rows_per_wave + 1) * n_hops never exceeds the 64 row pointers one wave-wide load holds.
 *   n_rows, n_cols    matrix shape; n_rows != n_cols is allowed (row-partitioned shards: n_rows = N/P)
 *   rowptr_dev[k], colidx_dev[k], vals_dev[k]   CSR of hop k, device pointers (see layout above)
 *   opts              NULL for defaults
 *   stream            stream used for the one-time validation / transposition work; the call returns after
 *                     that work has completed (it synchronises the stream)
 *
 * Index validity (monotone rowptr, 0 <= col < n_cols) is checked here, once -- not per launch.  TensorFlow's
 * CPU kernel reports out-of-range indices as InvalidArgument at run time; this returns H2GCN_ERR_BAD_INDEX.
 */
int h2gcn_plan_create(int n_hops, int64_t n_rows, int64_t n_cols,
                      const int64_t* const* rowptr_dev, const int32_t* const* colidx_dev,
                      const float* const* vals_dev, const h2gcn_plan_opts* opts, void* stream,
                      h2gcn_plan_t** out_plan);

/* Release the plan and everything it owns (never the caller's CSR arrays).  NULL is a no-op. */
void h2gcn_plan_destroy(h2gcn_plan_t* plan);

/*
 * Replace the VALUES of hop `hop` (same sparsity pattern: rowptr/colidx unchanged).  Stands in for what
 * SparseDropout does to the sparse feature operand every training step (reference h2gcn/models/_layers.py:7-19:
 * a fresh Bernoulli mask on `input.values`, survivors scaled by 1/keep_prob): the caller writes the new values
 * (dropped entries as explicit zeros) and points the plan at them; transposed operands are refreshed on `stream`
 * (needs H2GCN_PLAN_KEEP_PERMUTATION).  `vals_dev` must stay alive like the original array.  This MUTATES the plan:
 * it must not run concurrently with launches of the same plan on other streams.
 */
int h2gcn_plan_set_values(h2gcn_plan_t* plan, int hop, const float* vals_dev, void* stream);

/* Introspection (for reports and tests).  Any out pointer may be NULL. */
int h2gcn_plan_info(const h2gcn_plan_t* plan, int hop, int64_t* n_rows, int64_t* n_cols, int64_t* nnz,
                    int64_t* n_long_segments, int32_t* has_transpose);

/* The schedule a launch of this plan would use for feature width d and source row stride ld_src (reports, tests):
 * columns per slice of the slice-major schedule, number of slices, segment walk (0 = wave per segment, 1 = the same
 * with index prefetch across segments, 2 = in-tile short-row mode: rounds of consecutive short rows one lane group per
 * segment, 3 = list-driven by segment class: short segments from the binned list, one lane group each, medium ones one wave
 * each -- see h2gcn_plan_segment_classes), whether a launch that is given
 * scratch would gather from a slice-major copy.  adjoint != 0 asks about h2gcn_spmm_hops_T_f32 (ld_src = ldg_row,
 * hop stride d).  Any out pointer may be NULL. */
int h2gcn_plan_schedule(const h2gcn_plan_t* plan, uint32_t hop_mask, int adjoint, int64_t ld_src, int32_t d,
                        int32_t* slice_cols, int32_t* n_slices, int32_t* segment_walk, int32_t* scratch_copy);

/* CSR-adaptive dispatch, introspection: the plan bins every (row, hop) segment by length -- short (<= 16 nonzeros, empty
 * ones included), medium, long (>= long_row_threshold) -- and a launch serves each class with its own walk (see
 * h2gcn_plan_opts.variant).  For the s-th selected hop, segments[3 s + {0, 1, 2}] / nonzeros[3 s + {0, 1, 2}] receive the
 * number of short / medium / long segments and the nonzeros they hold (adjoint != 0: of A_k^T); *listed receives how many
 * segments (adjoint: output rows, i.e. rows whose segments of ALL selected hops are short) a launch at width d / source
 * stride ld_src serves from the binned short list -- 0 when that launch leaves the short class to the wave walk, -1 when it
 * runs in the in-tile short-row mode (rounds of consecutive short rows are grouped on the fly, no list).  Any out
 * pointer may be NULL; segments / nonzeros need 3 * (number of selected hops) entries. */
int h2gcn_plan_segment_classes(const h2gcn_plan_t* plan, uint32_t hop_mask, int adjoint, int64_t ld_src, int32_t d,
                               int64_t* segments, int64_t* nonzeros, int64_t* listed);

/*
 * Fused multi-hop aggregation, forward:
 *
 *     Y[i*ldy_row + s*ldy_hop + c] = sum_j A_k[i,j] * X[j*ldx + c]      0<=i<n_rows, 0<=c<d
 *
 * for every selected hop k (bit k of hop_mask set; s = rank of k among the selected hops, ascending).
 * Stands in for GCNLayer.call (reference h2gcn/models/_layers.py:78-81): with ldy_hop = d and
 * ldy_row = H_sel*d the output is the stacked [n_rows, H_sel, d] tensor, already flattened the way the
 * following `V` (Flatten) layer wants it (h2gcn/models/H2GCN.py:271-272); other strides let the caller land
 * the hops directly inside a wider concat buffer (ConcatLayer, _layers.py:90-96).
 * `hop_mask` reproduces GCNLayer(hops=...) (_layers.py:57-59,80-81); 0 means "all hops of the plan".
 * Rows without nonzeros produce zeros (TF zero-initialises the output).  Offsets are 64-bit throughout:
 * the reference's column-split workaround for nnz*d > 2^31 (_layers.py:65-74) has no counterpart here.
 *
 *   X_dev   fp32, n_cols rows of d values, row stride ldx >= d (elements)
 *   Y_dev   fp32, must not alias X
 *   d       feature width >= 1, any value, any 4-byte aligned X / Y and any strides (the reference accepts any
 *           b.shape[1], _layers.py:62-76; raw feature widths such as Cora's 1433 occur).  Every d >= 4 runs on the
 *           float4 gather kernels, in place: 16-byte global loads and stores need only dword alignment on gfx950, and
 *           in a row whose width is not a multiple of 4 the lane that straddles the end gathers the row's last four
 *           columns (overlapping its neighbour) and stores the ones it owns.  d < 4 takes a generic column-tiled
 *           kernel.  Fastest when rows are cache-line aligned (ldx a multiple of 32 floats).
 *
 * Floating point: fp32 multiply-add per nonzero in ONE canonical summation tree per output element -- neighbour j of
 * the row (ascending column order, the reference's order after tf.sparse.reorder, _dataset.py:535) is added into
 * partial P[j mod 4], the element is (P0 + P1) + (P2 + P3) -- for every kernel, slice width, feature chunking,
 * scratch copy and segment walk (rows with >= long_row_threshold nonzeros: the same tree per wave over the wave's
 * 64-neighbour chunks, wave totals added in order).  The bits of Y depend only on the row's nonzeros, X and
 * long_row_threshold: a row-partitioned multi-GPU run equals the single-GPU run bit-for-bit.
 *
 * hipGraph capture: a launch may be issued on a capturing stream, with one proviso.  The device-side row lists of a hop
 * selection (its long-segment list; its per-class lists once a launch is list-driven) are built by the FIRST launch of
 * that selection -- a device allocation and a synchronous upload under the plan's lock, neither of which may happen inside
 * a capture.  The all-hops selection is prepared by h2gcn_plan_create; for any other hop_mask (and for the adjoint, per
 * hop_mask) run the launch once eagerly at the width you are going to capture -- a warm-up step -- before capturing it.
 * A launch that would have to build a list while its stream is capturing returns H2GCN_ERR_INVALID_ARGUMENT with that
 * advice (it never falls back to another walk silently: the graph would replay the slower schedule for ever).  The same
 * first launch is also where an eager caller pays the one-off host synchronisation.
 */
int h2gcn_spmm_hops_f32(const h2gcn_plan_t* plan, uint32_t hop_mask, const float* X_dev, int64_t ldx,
                        int32_t d, float* Y_dev, int64_t ldy_row, int64_t ldy_hop, void* stream);

/*
 * Same launch with options.
 *
 * workspace / workspace_bytes: caller-provided scratch for a slice-major copy of the gather source (one streaming pass,
 *   a few % of the launch), from which the launch then gathers.  h2gcn_spmm_workspace_bytes() says how much scratch a
 *   launch wants (0 = the plain launch is already the fastest); NULL / too small simply selects the plain launch.
 *   The copy is a pure performance device, used (a) when the row stride of X is a multiple of 1 KiB (e.g. a contiguous
 *   [N, 256] embedding) and the operand is far beyond the caches: gathering column slices straight out of such rows
 *   wastes three quarters of the cache sets; (b) when the rows are wide and do not start on cache lines (base or ld*4 not
 *   a multiple of 128; forward d > 128, adjoint d > 256) on such an operand: the copy's 64-column blocks are aligned
 *   (zero-padded when d is not a multiple of 4).  Rows that DO start on cache lines are sliced 64 columns wide in place.  Results are bit-identical with and without scratch (canonical summation tree).
 *   The scratch is only used by this launch (on `stream`); launches that may run concurrently need separate scratch.
 * bias / H2GCN_LAUNCH_RELU: fused epilogue of the store, Y = act(A X + bias[c]) -- what SparseDense.call applies
 *   after its sparse product (reference h2gcn/models/_layers.py:45-52: `+ self.bias`, then `self.activation`), so
 *   that the feature embedding needs no second pass over its output.  bias: d floats (device) or NULL.  Forward only.
 *
 * h2gcn_spmm_workspace_bytes: adjoint != 0 asks about h2gcn_spmm_hops_T_opts_f32; src_dev / ld_src / ld_src_hop
 *   describe the gather source of that launch (forward: X_dev, ldx, 0; adjoint: dY_dev, ldg_row, ldg_hop); the pointer
 *   is only inspected (do its rows start on 128-byte cache lines?), never dereferenced.
 */
#define H2GCN_LAUNCH_RELU 0x1u
#define H2GCN_LAUNCH_ACCUMULATE 0x2u   /* adjoint only: dX += A^T dY instead of dX = A^T dY (see h2gcn_spmm_hops_T_opts_f32) */
typedef struct h2gcn_launch_opts {
    uint32_t struct_size;      /* = sizeof(h2gcn_launch_opts)                                               */
    uint32_t flags;            /* H2GCN_LAUNCH_*                                                             */
    void* workspace_dev;       /* scratch or NULL                                                            */
    size_t workspace_bytes;
    const float* bias_dev;     /* d floats or NULL                                                           */
} h2gcn_launch_opts;
size_t h2gcn_spmm_workspace_bytes(const h2gcn_plan_t* plan, uint32_t hop_mask, int adjoint, const float* src_dev,
                                  int64_t ld_src, int64_t ld_src_hop, int32_t d);
int h2gcn_spmm_hops_opts_f32(const h2gcn_plan_t* plan, uint32_t hop_mask, const float* X_dev, int64_t ldx,
                             int32_t d, float* Y_dev, int64_t ldy_row, int64_t ldy_hop,
                             const h2gcn_launch_opts* opts, void* stream);

/*
 * Adjoint (backward wrt X):
 *
 *     dX[j*ldx + c] = sum_s sum_i A_k[i,j] * dY[i*ldg_row + s*ldg_hop + c]   0<=j<n_cols
 *
 * Stands in for the TF-registered gradient of SparseTensorDenseMatMul wrt its dense operand
 * (adjoint_a=True SpMM), summed over the hops that `tf.stack` fanned out, as reached from
 * tape.gradient (reference h2gcn/models/H2GCN.py:66-74).  The gradient wrt the adjacency values, which TF
 * also computes and the reference discards, is not computed.  Requires H2GCN_PLAN_BUILD_TRANSPOSE.
 * dX is overwritten (not accumulated into).
 */
int h2gcn_spmm_hops_T_f32(const h2gcn_plan_t* plan, uint32_t hop_mask, const float* dY_dev, int64_t ldg_row,
                          int64_t ldg_hop, int32_t d, float* dX_dev, int64_t ldx, void* stream);
/* The adjoint with options: workspace_dev / workspace_bytes and the flag H2GCN_LAUNCH_ACCUMULATE (bias_dev must be NULL).  With
 * scratch the stacked gradient is copied slice-major per hop when its rows are wide (d > 256) and not cache-line
 * aligned on an operand far beyond the caches -- same bits as the plain launch.
 * H2GCN_LAUNCH_ACCUMULATE: dX[j, c] += sum ... -- the adjoint's result is ADDED to what dX holds, each element as
 * `old + sum` with `sum` the canonical-tree value of the plain launch.  This is the backward of the concat-free propagation:
 * the gradient slot of r_{k-1} inside the [N, 448] gradient of the representation already holds the classifier's
 * contribution, and the round's adjoint lands on top of it without a separate `+=` pass (h2gcn/models/_layers.py:90-96 is the
 * concat whose gradient this adds up).  dX must not overlap dY.  Runs on the general-store kernels (no short-row / index
 * prefetch variants): meant for the wave-per-segment regime (see h2gcn_plan_schedule). */
int h2gcn_spmm_hops_T_opts_f32(const h2gcn_plan_t* plan, uint32_t hop_mask, const float* dY_dev, int64_t ldg_row,
                               int64_t ldg_hop, int32_t d, float* dX_dev, int64_t ldx,
                               const h2gcn_launch_opts* opts, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Operand construction on the device: exact-k-hop neighbourhood rings and their normalisation -- the step that
 * FEEDS the aggregation.  Stands in for TransformSPAdj.nhoodSplit (reference h2gcn/datasets/_dataset.py:138-158:
 * `mt <- bin(mt @ (A + I))`, ring_k = mt_k - mt_{k-1}, a host SpGEMM in scipy) and TransformSPAdj.normalize
 * (:109-124).  A ring is a SET expression over CSR patterns, evaluated row by row in a two-level LDS bitmap
 * (h2gcn_amd/csrc/rings.hip):
 *
 *     out[i] = ( U_{j in F[i]} A[j]  U  U_p ADD_p[i]  U  ({i} if add_diag) )  \  ( U_q SUB_q[i]  U  ({i} if sub_diag) )
 *
 *   ring_k (k >= 1):  F = ring_{k-1} (ring_0 = I, i.e. F[i] = {i}; for k = 1 simply pass F = I or use ring_1 = A),
 *                     SUB = ring_1 .. ring_{k-1}, sub_diag = 1
 *   merged group "0,1" of --adj_nhood (getTensors, :560-572):  no F, ADD = {ring_1}, add_diag = 1
 *
 * All patterns are n x n CSR (int64 rowptr, int32 colidx ascending, no values); outputs have ascending columns.
 * Two passes: h2gcn_ring_count writes out_rowptr[0..n] (row pointers of the result) and returns the number of
 * nonzeros (it synchronises the stream -- the caller has to allocate out_colidx); h2gcn_ring_fill, called with the
 * SAME inputs, writes the columns.  `scratch` is h2gcn_ring_scratch_bytes(n) bytes of device memory.
 * F == NULL (f_rowptr NULL) means "no expansion"; a_* may then be NULL as well.
 */
size_t h2gcn_ring_scratch_bytes(int64_t n);
int h2gcn_ring_count(int64_t n, const int64_t* a_rowptr_dev, const int32_t* a_colidx_dev,
                     const int64_t* f_rowptr_dev, const int32_t* f_colidx_dev,
                     int n_add, const int64_t* const* add_rowptr_dev, const int32_t* const* add_colidx_dev, int add_diag,
                     int n_sub, const int64_t* const* sub_rowptr_dev, const int32_t* const* sub_colidx_dev, int sub_diag,
                     int64_t* out_rowptr_dev, int64_t* nnz_out, void* scratch_dev, size_t scratch_bytes, void* stream);
int h2gcn_ring_fill(int64_t n, const int64_t* a_rowptr_dev, const int32_t* a_colidx_dev,
                    const int64_t* f_rowptr_dev, const int32_t* f_colidx_dev,
                    int n_add, const int64_t* const* add_rowptr_dev, const int32_t* const* add_colidx_dev, int add_diag,
                    int n_sub, const int64_t* const* sub_rowptr_dev, const int32_t* const* sub_colidx_dev, int sub_diag,
                    const int64_t* out_rowptr_dev, int32_t* out_colidx_dev, void* scratch_dev, size_t scratch_bytes,
                    void* stream);
/*
 * The same for a ROW WINDOW [row_begin, row_begin + n_rows) -- what one rank of a row-partitioned run needs (SURVEY.md
 * 8(e)): F, ADD, SUB and the result are CSRs of that window (n_rows + 1 local row pointers each), A is the whole n x n
 * matrix (the frontier names global ids), {i} is the global row id.  Ring k of the window needs the whole A and the
 * window's rows of the lower rings only, so each rank builds 1/P of every ring (the reference builds everything on one
 * host, _dataset.py:147-157).  h2gcn_ring_count / _fill are the window [0, n).
 */
int h2gcn_ring_count_rows(int64_t n, int64_t row_begin, int64_t n_rows, const int64_t* a_rowptr_dev, const int32_t* a_colidx_dev,
                          const int64_t* f_rowptr_dev, const int32_t* f_colidx_dev,
                          int n_add, const int64_t* const* add_rowptr_dev, const int32_t* const* add_colidx_dev, int add_diag,
                          int n_sub, const int64_t* const* sub_rowptr_dev, const int32_t* const* sub_colidx_dev, int sub_diag,
                          int64_t* out_rowptr_dev, int64_t* nnz_out, void* scratch_dev, size_t scratch_bytes, void* stream);
int h2gcn_ring_fill_rows(int64_t n, int64_t row_begin, int64_t n_rows, const int64_t* a_rowptr_dev, const int32_t* a_colidx_dev,
                         const int64_t* f_rowptr_dev, const int32_t* f_colidx_dev,
                         int n_add, const int64_t* const* add_rowptr_dev, const int32_t* const* add_colidx_dev, int add_diag,
                         int n_sub, const int64_t* const* sub_rowptr_dev, const int32_t* const* sub_colidx_dev, int sub_diag,
                         const int64_t* out_rowptr_dev, int32_t* out_colidx_dev, void* scratch_dev, size_t scratch_bytes,
                         void* stream);
/*
 * Values of a hop matrix given as a square CSR PATTERN (every stored entry is 1, as nhoodSplit produces):
 *   mode 0 ORDINARY: 1;  mode 1 SYM: fp32((s[deg_i] * 1.0) * s[deg_j]);  mode 2 RW: fp32(s[deg_i] * 1.0)
 * with deg = the row lengths of THIS matrix (reference: D = rowsum(A_k) of that hop matrix, :115-123) and
 * s_table[k] = the fp64 scaling of a row with k entries (k^-1/2 resp. k^-1, inf -> 0), supplied by the caller -- the
 * Python front end computes it with the reference's own numpy call, which keeps the fp64 products and the fp32 cast
 * of sparse2Tensor (:528-535) bit-identical to the reference.  s_table_len > max row length (checked).
 */
int h2gcn_hop_normalize(int64_t n, const int64_t* rowptr_dev, const int32_t* colidx_dev, int mode,
                        const double* s_table_dev, int64_t s_table_len, float* vals_dev, void* stream);
/* Row-window form: the pattern holds n_rows rows of the matrix; col_len_dev[j] = length of row j of the WHOLE matrix
 * (needed by SYM for s[deg_j]; NULL = the pattern is the whole square matrix).  Both calls check every row length
 * against s_table_len on the device and return H2GCN_ERR_INVALID_ARGUMENT when the table is too short (they
 * synchronise the stream for that). */
int h2gcn_hop_normalize_rows(int64_t n_rows, const int64_t* rowptr_dev, const int32_t* colidx_dev, int mode,
                             const double* s_table_dev, int64_t s_table_len, const int64_t* col_len_dev,
                             float* vals_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * The classifier side of a training step: keras `Dropout(rate)` followed by the output `Dense` -- `D0.5-MO` of the
 * network-setup DSL (reference h2gcn/models/H2GCN.py:235-257 builds the two layers, :308-325 calls them in order) -- as
 * ONE pass over the [N, K] concat buffer per direction, on the fp32 matrix cores (h2gcn_amd/csrc/classifier.hip):
 *
 *     forward    Z[n, c]  = sum_k D[n, k] * W[k, c] + bias[c]          D[n, k] = keep(n, k) ? X[n, k] / keep_prob : 0
 *     backward   dX[n, k] = keep(n, k) ? (sum_c G[n, c] * W[k, c]) / keep_prob : 0
 *                dW[k, c] = sum_n D[n, k] * G[n, c]                     (db = column sums of G: left to the caller)
 *
 * The dropout mask is COUNTER-BASED, recomputed wherever it is needed instead of stored.  One keyed hash per aligned group
 * of four columns of a row:
 *     gid = n * ceil(K / 4) + k / 4  (64-bit),
 *     h = lo32(gid) ^ rotl16(hi32(gid)) ^ key0;  h ^= h >> 16;  h *= 0x7FEB352D;  h ^= key1;  h ^= h >> 15;  h *= 0x846CA68B;
 *     w0 = h ^ (h >> 16)                                                                              (all mod 2^32)
 *     key0 = mix(lo32(seed) ^ mix(lo32(step) + 0x9E3779B9)),   key1 = mix(hi32(seed) ^ hi32(step) ^ key0 ^ 0x85EBCA6B),
 *     mix(h): h ^= h >> 16; h *= 0x7FEB352D; h ^= h >> 15; h *= 0x846CA68B; h ^= h >> 16
 *   T = floor(keep_prob * 65536).  T a multiple of 256 (keep_prob a multiple of 1/256: 0.5, 0.75, 0.875, ...):
 *     keep(n, k)  <=>  byte (k % 4) of w0  <  T / 256                                                  (8-bit fields)
 *   otherwise  w1 = mix(w0 ^ 0x85EBCA6B),  field = (w0 & 0xffff, w0 >> 16, w1 & 0xffff, w1 >> 16),
 *     keep(n, k)  <=>  field[k % 4] < T                                                                (16-bit fields)
 * The survivors' 1 / keep_prob scale is applied to the finished sums (Z, dW) or at the store (dX).
 * `step` is read from DEVICE memory (*step_dev; NULL = 0) so that a captured hipGraph draws a fresh mask on every replay
 * (the caller bumps the counter with a stream-ordered op); the backward call must see the value its forward saw.
 * keep_prob = 1 switches the mask off (evaluation).  TensorFlow's stateful RNG stream cannot be reproduced by any other
 * implementation, so parity with the reference is statistical here (keep rate) and exact for everything that is a
 * function of the mask (oracle/classifier.py restates the generator bit for bit).
 *
 *   X        fp32 [n_rows, K], row stride ldx (any 4-byte aligned base / stride)      W   fp32 [K, C] contiguous, C <= 64
 *   G        fp32 [n_rows, C], row stride ldg                                         dX  fp32 [n_rows, K], row stride lddx, or NULL
 *   dW       fp32 [K, C] contiguous (overwritten), or NULL; partial sums are added in a fixed order (deterministic)
 *   workspace  h2gcn_dropout_dense_workspace_bytes(n_rows, K, C) bytes, 16-byte aligned, used by this call on `stream`
 * Arithmetic: v_mfma_f32_16x16x4_f32, i.e. exact fp32 multiply-adds (no reduced precision); the summation order over k
 * (forward), c (dX) and rows (dW) is fixed by the shapes.
 */
/* Operands of at most this many rows (default 12288) with C <= 16 classes are latency-bound and take three plain kernels instead
 * of the matrix-core ones (same mask, same scale placement, deterministic; Cora: ~18 -> ~5 us per pass).  Sets the bound when
 * rows >= 0 (0 = never; a process-wide tunable, meant for tests and measurements) and returns the previous one. */
int64_t h2gcn_dropout_dense_small_rows(int64_t rows);
size_t h2gcn_dropout_dense_workspace_bytes(int64_t n_rows, int32_t k, int32_t c);
int h2gcn_dropout_dense_f32(const float* X_dev, int64_t ldx, int64_t n_rows, int32_t k, const float* W_dev, int32_t c,
                            const float* bias_dev, float keep_prob, uint64_t seed, const int64_t* step_dev,
                            float* Z_dev, int64_t ldz, void* workspace_dev, size_t workspace_bytes, void* stream);
int h2gcn_dropout_dense_backward_f32(const float* X_dev, int64_t ldx, int64_t n_rows, int32_t k, const float* W_dev, int32_t c,
                                     const float* G_dev, int64_t ldg, float keep_prob, uint64_t seed, const int64_t* step_dev,
                                     float* dX_dev, int64_t lddx, float* dW_dev, void* workspace_dev, size_t workspace_bytes,
                                     void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Masked softmax cross-entropy and masked accuracy (reference h2gcn/models/_metrics.py:8-25; called per mask by train_step /
 * test_step, h2gcn/models/H2GCN.py:66-74, 77-107) in ONE pass over the logits (h2gcn_amd/csrc/metrics.hip).  A "set" m is a
 * label matrix Y_m [n_rows, C] (one-hot or all-zero rows; any non-negative rows work) plus a row-weight vector w_m [n_rows]
 * -- the reference's `mask / mean(mask) / n_rows`, i.e. mask / sum(mask); a row-partitioned run divides by the GLOBAL sum:
 *
 *     loss[m] = sum_n w_m[n] * ( - sum_c Y_m[n, c] * log_softmax(Z[n])[c] )
 *     acc[m]  = sum_n w_m[n] * [ argmax_c Z[n, c] == argmax_c Y_m[n, c] ]          (first maximum on ties)
 *
 * Rows whose weight is zero in every set are not read; a label matrix is read only at the rows of non-zero weight.  Per-row
 * terms are fp32, the sums over rows run in fp64 per workgroup and are combined in a fixed order (deterministic).
 * DIVERGENCE from the reference on non-finite logits: the reference MULTIPLIES every row's term by its mask weight
 * (_metrics.py:12-14, 22-24), so a NaN / Inf logit in a row the mask excludes still yields NaN there (0 * NaN); here such a
 * row is skipped and the result stays finite.  Rows the mask includes propagate NaN / Inf exactly as the reference does, so a
 * diverged model is still visible through any set that covers the affected rows.
 * h2gcn_masked_ce_backward_f32 is the gradient of loss (one set) with respect to Z, times the scalar *gscale_dev (NULL = 1):
 *     dZ[n, c] = g * w[n] * ( softmax(Z[n])[c] * sum_c' Y[n, c'] - Y[n, c] )     (rows of weight 0: zeros).
 *   Z  fp32 [n_rows, C] row stride ldz, C <= 64      Y, ldy, w  HOST arrays of n_sets (<= H2GCN_METRICS_MAX_SETS) device pointers /
 *   strides      loss_out, acc_out  DEVICE fp32 [n_sets] (acc_out may be NULL)      workspace  h2gcn_masked_metrics_workspace_bytes(n_rows)
 */
#define H2GCN_METRICS_MAX_SETS 4
size_t h2gcn_masked_metrics_workspace_bytes(int64_t n_rows);
int h2gcn_masked_metrics_f32(const float* Z_dev, int64_t ldz, int64_t n_rows, int32_t c, int32_t n_sets, const float* const* Y_dev,
                             const int64_t* ldy, const float* const* w_dev, float* loss_out_dev, float* acc_out_dev,
                             void* workspace_dev, size_t workspace_bytes, void* stream);
int h2gcn_masked_ce_backward_f32(const float* Z_dev, int64_t ldz, int64_t n_rows, int32_t c, const float* Y_dev, int64_t ldy,
                                 const float* w_dev, const float* gscale_dev, float* dZ_dev, int64_t lddz, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * The reference's optimizer step: Keras Adam as `optimizer.apply_gradients` runs it (reference h2gcn/models/H2GCN.py:62-63, 73;
 * TensorFlow's ApplyAdam, a third-party kernel absent from the reference tree), for ALL parameter tensors in one launch:
 *     alpha = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t)                          t = 1-based step, all arithmetic fp32
 *     m += (g - m) * (1 - beta_1);  v += (g*g - v) * (1 - beta_2);  param -= (m * alpha) / (sqrt(v) + epsilon)
 * (epsilon joins the UNCORRECTED sqrt(v): Keras / TensorFlow semantics, not torch.optim.Adam's).  params / grads / m / v / sizes
 * are HOST arrays of n_tensors device pointers / element counts (contiguous fp32 tensors; m and v zero before step 1).
 * t is read from *step_dev when step_dev != NULL (device memory: the caller bumps it with a stream-ordered op, so a captured
 * hipGraph advances it on every replay), else from `step`.
 */
#define H2GCN_ADAM_MAX_TENSORS 16   /* tensors per launch; longer lists are processed in groups */
int h2gcn_adam_keras_f32(int32_t n_tensors, float* const* params_dev, const float* const* grads_dev, float* const* m_dev,
                         float* const* v_dev, const int64_t* sizes, float lr, float beta1, float beta2, float epsilon,
                         const int64_t* step_dev, int64_t step, void* stream);

/* The same step with the keras `regularizers.l2(w)` of the dense kernels (reference h2gcn/models/H2GCN.py:239-240, 247-248) folded
 * in: l2 is a HOST array of n_tensors coefficients (0 = tensor not regularised; NULL = none) and the update uses
 * g + 2 * l2 * param as the gradient -- with the roundings of autograd's separate multiply and add, so a run that keeps the
 * penalty out of the autograd graph and passes it here follows the one that does not bit for bit.  h2gcn_l2_penalty_f32 is the
 * VALUE of that penalty, sum_k l2[k] * sum(param_k^2) (fp64 inside a tensor, fp32 across tensors in order), for the loss a step
 * reports (H2GCN.py:363-367): one launch for up to H2GCN_ADAM_MAX_TENSORS tensors.  workspace: h2gcn_l2_penalty_workspace_bytes()
 * bytes of device memory, 8-byte aligned, ZERO before its first use (the kernel re-arms it); out_dev: one float. */
int h2gcn_adam_keras_l2_f32(int32_t n_tensors, float* const* params_dev, const float* const* grads_dev, float* const* m_dev,
                            float* const* v_dev, const int64_t* sizes, const float* l2, float lr, float beta_1, float beta_2,
                            float epsilon, const int64_t* step_dev, int64_t step, void* stream);
size_t h2gcn_l2_penalty_workspace_bytes(void);
int h2gcn_l2_penalty_f32(int32_t n_tensors, const float* const* params_dev, const int64_t* sizes, const float* l2,
                         float* out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Row-shard exchange between the GPUs of one node (no counterpart in the reference: it is single-process,
 * single-device -- SURVEY.md 8(e) adds the row partition).  Before a hop aggregation every rank needs the whole
 * embedding X[N, d] while it owns only X[rows_p, :]; this object performs that all-gather WITHOUT a collective
 * library: every rank stages its shard into a buffer it has exported with hipIpcGetMemHandle, tells its peers
 * (a sequence number stored into their flag words over xGMI), and pulls the peers' shards straight out of their
 * exported buffers -- either with copy-engine transfers (hipMemcpyAsync device-to-device, one stream per peer:
 * no CU is used, every point-to-point xGMI link carries its own transfer) or with one small copy kernel.
 * It is an alternative to ncclAllGather behind the same Python interface (h2gcn_amd/partition.py); which is
 * faster is a property of the node and is measured, not assumed (bench.py).
 *
 * Life cycle (one object per rank, all ranks make the same calls in the same order):
 *   create -> export (blob) -> [blobs of all ranks are exchanged by the caller, e.g. torch.distributed
 *   all_gather_object] -> connect -> { allgather_begin(channel) ... allgather_end(channel) }* -> [caller
 *   barrier] -> destroy.
 * A "channel" is an independent double-buffered send slot with its own sequence counter (the feature-chunk
 * pipeline uses one channel per chunk).  Nothing here blocks the host; a peer that never posts makes the waiting
 * GPU give up after `timeout_ms` and the failure is reported by h2gcn_xchg_status().
 */
typedef struct h2gcn_xchg h2gcn_xchg_t;

#define H2GCN_XCHG_BLOB_BYTES 192          /* size of the opaque blob h2gcn_xchg_export writes                  */
#define H2GCN_XCHG_COPY_ENGINE 0           /* mode: pulls are hipMemcpyAsync D2D on one stream per peer (SDMA)  */
#define H2GCN_XCHG_COPY_KERNEL 1           /* mode: pulls are one copy kernel reading the peers' memory         */

/*   world, rank      number of ranks / this rank (one rank per process; peers must be other processes)
 *   n_channels       independent slots (1..64)
 *   slot_bytes       capacity of one slot = the largest shard (rows_per_rank * width * 4) posted on a channel
 *   mode             H2GCN_XCHG_COPY_ENGINE or H2GCN_XCHG_COPY_KERNEL
 *   timeout_ms       how long a GPU waits for a peer's shard before giving up (0 = default 10000)            */
int h2gcn_xchg_create(int world, int rank, int n_channels, size_t slot_bytes, int mode, int timeout_ms,
                      h2gcn_xchg_t** out);
/* Writes H2GCN_XCHG_BLOB_BYTES bytes describing this rank's exported buffers. */
int h2gcn_xchg_export(const h2gcn_xchg_t* x, void* blob_out);
/* `blobs` = the world blobs in rank order (world * H2GCN_XCHG_BLOB_BYTES bytes); opens every peer's buffers. */
int h2gcn_xchg_connect(h2gcn_xchg_t* x, const void* blobs);
/*
 * Start the all-gather of one shard on `channel`:
 *   src_dev   this rank's rows, fp32 [rows, width] with row stride ld_src (elements); rows <= rows_per_rank
 *   full_dev  destination, fp32 [world * rows_per_rank, width] contiguous; rank q's shard lands in rows
 *             [q*rows_per_rank, (q+1)*rows_per_rank) (rows beyond a short last shard are zero)
 *   stream    the stream that produced src_dev; the staging copy and the notification are ordered on it, and
 *             the pulls (internal streams) start only after everything enqueued on it so far -- so a kernel that
 *             is still reading full_dev from the previous use is safe.
 * Returns immediately.  Exactly one allgather_end must follow before the next begin on the same channel. */
int h2gcn_xchg_allgather_begin(h2gcn_xchg_t* x, int channel, const float* src_dev, int64_t ld_src, int64_t rows,
                               int64_t rows_per_rank, int32_t width, float* full_dev, void* stream);
/* The same in two halves, for pipelines over several channels: `post` stages and announces (ordered on `stream`),
 * `pull` issues the pulls (same rows_per_rank / width / full_dev).  Posting ALL channels before pulling any keeps a
 * later channel's announcement from queueing behind an earlier channel's wait when streams share a hardware queue.
 * begin == post immediately followed by pull. */
int h2gcn_xchg_allgather_post(h2gcn_xchg_t* x, int channel, const float* src_dev, int64_t ld_src, int64_t rows,
                              int64_t rows_per_rank, int32_t width, float* full_dev, void* stream);
int h2gcn_xchg_allgather_pull(h2gcn_xchg_t* x, int channel, int64_t rows_per_rank, int32_t width, float* full_dev);
/* Halo form of the pull (copy-kernel mode, width % 4 == 0): from peer q only the rows rows_dev[q][0 .. counts[q]) (ascending LOCAL
 * row ids of q's shard, int32, device memory; rows_dev / counts are HOST arrays of `world` entries, the own rank's is ignored) -- the
 * rows of the embedding this rank's hop matrices really name.  Rows not listed stay untouched in full_dev.  Same protocol, same
 * allgather_end.  Dense pulls remain the right thing whenever (nearly) every row is named: the synthetic shapes, products-like 2-hop rings. */
int h2gcn_xchg_allgather_pull_rows(h2gcn_xchg_t* x, int channel, int64_t rows_per_rank, int32_t width, float* full_dev,
                                   const int32_t* const* rows_dev, const int64_t* counts);
/* Make `stream` wait (device-side, no host block) until every shard of `channel` has landed in full_dev. */
int h2gcn_xchg_allgather_end(h2gcn_xchg_t* x, int channel, void* stream);
/*
 * The mirror image for the backward pass: every rank holds a full-height contribution src_dev
 * [world * rows_per_rank, width] (contiguous) -- the shard adjoint A_k[rows_p, :]^T dY_p -- and needs the sum over
 * ranks of ITS row block: out_dev[r, :] = sum_q src_q[rank * rows_per_rank + r, :].  Same protocol: the matrix is
 * staged into the exported slot (slot_bytes must cover the whole matrix), announced, and every rank pulls its block
 * out of every peer's slot; `end` sums the blocks in ascending rank order (deterministic) on `stream`.
 * Shares the channel's sequence counter with the all-gather: every rank must issue the same sequence of calls.
 */
int h2gcn_xchg_reduce_scatter_begin(h2gcn_xchg_t* x, int channel, const float* src_dev, int64_t rows_per_rank,
                                    int32_t width, float* out_dev, void* stream);
int h2gcn_xchg_reduce_scatter_end(h2gcn_xchg_t* x, int channel, void* stream);
/*
 * hipGraph capture (H2GCN_XCHG_COPY_KERNEL only): every call above may be issued on a capturing stream -- sequence
 * numbers and slot parity live in device memory, so a replayed graph advances the protocol exactly like an eager call
 * (all ranks must replay the same graphs in the same order).  Call h2gcn_xchg_reset_dependencies() right before a
 * capture begins, with the device idle: the object then forgets the events of earlier (un-captured or other-capture)
 * steps, which a capturing stream must not wait on; whole graphs are ordered by the stream they are replayed on.
 * Copy-engine mode cannot be captured (hipMemcpyAsync needs the slot address on the host).
 */
int h2gcn_xchg_reset_dependencies(h2gcn_xchg_t* x);
/* H2GCN_OK, or H2GCN_ERR_EXCHANGE_TIMEOUT if any wait on this object has ever given up (results are then
 * undefined).  Does not synchronise: call after the streams involved have been synchronised. */
int h2gcn_xchg_status(const h2gcn_xchg_t* x);
/* Release everything.  The caller must make sure (barrier) that no peer is still pulling from this rank. */
void h2gcn_xchg_destroy(h2gcn_xchg_t* x);

#ifdef __cplusplus
}
#endif
#endif /* H2GCN_HIP_H_ */
