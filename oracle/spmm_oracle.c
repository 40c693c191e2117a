/*
 * spmm_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the arithmetic on H2GCN's hop-aggregation
 * path, used as the parity checker by tests/, __graft_entry__.smoke() and the `cpu_baseline` leg of bench.py.
 * Nothing under h2gcn_amd/ may import, link or call this file.
 *
 * What it restates
 *   (a3) TensorFlow's CPU kernel for `tf.sparse.sparse_dense_matmul` (op SparseTensorDenseMatMul), which is
 *        where the reference's arithmetic for this path lives: call sites h2gcn/models/_layers.py:47,74,76.
 *        TensorFlow is a third-party dependency of the reference, NOT vendored under /root/reference and not
 *        installable here (README.md:34 "TensorFlow >= 2.0 (tested on 2.2)", no lock file).  Its published
 *        CPU algorithm [upstream, recalled]: `out` is zero-initialised; for each stored nonzero, in stored
 *        order, `out[m, :] += a_val * b[k, :]`, in fp32, single-threaded.  The reference stores nonzeros in
 *        row-major canonical order (`tf.sparse.reorder`, h2gcn/datasets/_dataset.py:535), so each output row
 *        accumulates its terms in ascending column order.  -> oracle_spmm_coo_f32 / oracle_spmm_csr_f32.
 *   (a1) GCNLayer.call, h2gcn/models/_layers.py:78-81: one SpMM per selected hop, `tf.stack(axis=-2)`
 *        -> oracle_gcn_layer_f32 writes Y[n_rows, H, d].
 *   (a4) the gradient wrt the dense operand, dX = sum_k A_k^T dY[:, k, :] (adjoint_a SpMM, reached from
 *        h2gcn/models/H2GCN.py:66-74) -> oracle_gcn_layer_grad_f32.
 *
 * PARITY PINNING: the reference has no tests and TensorFlow cannot run here, so the SpMM arithmetic itself is
 * "parity unpinned" against TensorFlow; it is pinned instead against scipy's csr @ dense in fp32/fp64
 * (tests/test_oracle.py) and the OPERANDS are pinned against the reference's own preprocessing code imported
 * in the build container (tests/golden/, made by tests/golden/make_golden.py).
 *
 * Build: `make -C oracle` (gcc, -O2, -ffp-contract=off so every term is a separately rounded mul and add;
 * the *_fma variant uses fmaf to bracket the other possible TF build).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* (a3) COO in stored order -- the literal loop nest of the upstream kernel. */
void oracle_spmm_coo_f32(int64_t nnz, const int64_t* rows, const int64_t* cols, const float* vals,
                         const float* b, int64_t ldb, int64_t d, float* out, int64_t ldo, int64_t n_out_rows) {
    for (int64_t i = 0; i < n_out_rows; ++i) memset(out + i * ldo, 0, (size_t)d * sizeof(float));
    for (int64_t e = 0; e < nnz; ++e) {
        const float a = vals[e];
        const float* br = b + cols[e] * ldb;
        float* o = out + rows[e] * ldo;
        for (int64_t c = 0; c < d; ++c) o[c] = o[c] + a * br[c];
    }
}

/* Same arithmetic from CSR (row-major canonical COO == CSR order). */
void oracle_spmm_csr_f32(int64_t n_rows, const int64_t* rowptr, const int32_t* colidx, const float* vals,
                         const float* b, int64_t ldb, int64_t d, float* out, int64_t ldo) {
    for (int64_t i = 0; i < n_rows; ++i) {
        float* o = out + i * ldo;
        memset(o, 0, (size_t)d * sizeof(float));
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const float a = vals[e];
            const float* br = b + (int64_t)colidx[e] * ldb;
            for (int64_t c = 0; c < d; ++c) o[c] = o[c] + a * br[c];
        }
    }
}

/* fused-multiply-add variant (a TF build with FMA enabled would round like this) */
void oracle_spmm_csr_f32_fma(int64_t n_rows, const int64_t* rowptr, const int32_t* colidx, const float* vals,
                             const float* b, int64_t ldb, int64_t d, float* out, int64_t ldo) {
    for (int64_t i = 0; i < n_rows; ++i) {
        float* o = out + i * ldo;
        memset(o, 0, (size_t)d * sizeof(float));
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const float a = vals[e];
            const float* br = b + (int64_t)colidx[e] * ldb;
            for (int64_t c = 0; c < d; ++c) o[c] = fmaf(a, br[c], o[c]);
        }
    }
}

/* fp64 accumulation of the fp32 operands: the "exact" answer used to bound rounding error. */
void oracle_spmm_csr_f64acc(int64_t n_rows, const int64_t* rowptr, const int32_t* colidx, const float* vals,
                            const float* b, int64_t ldb, int64_t d, double* out, int64_t ldo) {
    for (int64_t i = 0; i < n_rows; ++i) {
        double* o = out + i * ldo;
        for (int64_t c = 0; c < d; ++c) o[c] = 0.0;
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const double a = (double)vals[e];
            const float* br = b + (int64_t)colidx[e] * ldb;
            for (int64_t c = 0; c < d; ++c) o[c] += a * (double)br[c];
        }
    }
}

/* (a1) GCNLayer.call: Y[i, k, :] = (A_k @ X)[i, :] for the H given hops, stacked on axis -2. */
void oracle_gcn_layer_f32(int n_hops, int64_t n_rows, const int64_t* const* rowptr, const int32_t* const* colidx,
                          const float* const* vals, const float* x, int64_t ldx, int64_t d, float* y) {
    for (int k = 0; k < n_hops; ++k)
        oracle_spmm_csr_f32(n_rows, rowptr[k], colidx[k], vals[k], x, ldx, d, y + (int64_t)k * d, (int64_t)n_hops * d);
}

/* The same layer with the output rows spread over OpenMP threads (per-row arithmetic unchanged: identical bits).  Only
 * for the "all host cores" CPU baseline of bench.py -- TensorFlow's own CPU kernel is single-threaded. */
void oracle_gcn_layer_f32_mt(int n_hops, int64_t n_rows, const int64_t* const* rowptr, const int32_t* const* colidx,
                             const float* const* vals, const float* x, int64_t ldx, int64_t d, float* y) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n_rows; ++i)
        for (int k = 0; k < n_hops; ++k) {
            float* o = y + (i * n_hops + k) * d;
            memset(o, 0, (size_t)d * sizeof(float));
            for (int64_t e = rowptr[k][i]; e < rowptr[k][i + 1]; ++e) {
                const float a = vals[k][e];
                const float* br = x + (int64_t)colidx[k][e] * ldx;
                for (int64_t c = 0; c < d; ++c) o[c] = o[c] + a * br[c];
            }
        }
}

/* (a4) dX[j, :] = sum_k sum_i A_k[i, j] * dY[i, k, :]; accumulation order: hop-major, then row-major stored
 * order (what unstacking + one adjoint SpMM per hop + add_n gives). */
void oracle_gcn_layer_grad_f32(int n_hops, int64_t n_rows, int64_t n_cols, const int64_t* const* rowptr,
                               const int32_t* const* colidx, const float* const* vals, const float* dy, int64_t d,
                               float* dx) {
    float* tmp = (float*)malloc((size_t)n_cols * (size_t)d * sizeof(float));
    for (int64_t i = 0; i < n_cols * d; ++i) dx[i] = 0.f;
    for (int k = 0; k < n_hops; ++k) {
        for (int64_t i = 0; i < n_cols * d; ++i) tmp[i] = 0.f;
        for (int64_t i = 0; i < n_rows; ++i) {
            const float* g = dy + (i * n_hops + k) * d;
            for (int64_t e = rowptr[k][i]; e < rowptr[k][i + 1]; ++e) {
                const float a = vals[k][e];
                float* o = tmp + (int64_t)colidx[k][e] * d;
                for (int64_t c = 0; c < d; ++c) o[c] = o[c] + a * g[c];
            }
        }
        for (int64_t i = 0; i < n_cols * d; ++i) dx[i] = dx[i] + tmp[i];
    }
    free(tmp);
}

/* ---------------------------------------------------------------------------------------------------------------
 * Kernel-order restatement (TEST INFRASTRUCTURE, like everything in this file).  The reference's arithmetic per
 * output element is a plain sequential fp32 sum in ascending column order (above).  The HIP library documents ONE
 * canonical regrouping of exactly those terms (include/h2gcn_hip.h, "Floating point"): neighbour j of the row goes,
 * with one fused multiply-add, into partial P[j mod 4]; the element is (P0 + P1) + (P2 + P3); a segment with >=
 * long_threshold nonzeros is cut into 64-neighbour chunks dealt round-robin to 4 "waves", each wave builds that tree
 * over its chunks, and the 4 wave totals are added in order.  These functions restate that documented order on the
 * CPU so that tests can demand BIT-EXACT agreement of every kernel variant / slice width / feature chunking / row
 * partition with one fixed function of the inputs -- which is how SURVEY.md 8(e)'s "P-GPU == 1-GPU bit for bit" is
 * pinned.  They are not a second reference: parity with the reference's order is what the 1e-5 tests above check.
 */
static void tree_chunks(const int32_t* ci, const float* va, int64_t sb, int64_t se, int chunk0, int chunk_step,
                        const float* src, int64_t ld, int64_t d, float* P /* [4][d] */) {
    for (int64_t base = sb + (int64_t)chunk0 * 64; base < se; base += (int64_t)chunk_step * 64) {
        const int64_t n = se - base < 64 ? se - base : 64;
        for (int64_t jj = 0; jj < n; ++jj) {
            float* p = P + (jj & 3) * d;
            const float a = va[base + jj];
            const float* row = src + (int64_t)ci[base + jj] * ld;
            for (int64_t c = 0; c < d; ++c) p[c] = fmaf(a, row[c], p[c]);
        }
    }
}

static void tree_fold(const float* P, int64_t d, float* out) {
    for (int64_t c = 0; c < d; ++c) out[c] = (P[c] + P[d + c]) + (P[2 * d + c] + P[3 * d + c]);
}

/* One output row of the canonical tree (P, T: caller-provided scratch of 4 * d floats each).
 * mode_sum == 0: Y[i, s, :] per hop (forward, y row stride ldy_row, hop stride ldy_hop);
 * mode_sum == 1: out[i, :] = sum over the hops of A_s[i, :] @ src[:, s, :] (adjoint on transposed operands: src row
 *                stride ld_src, hop stride ld_src_hop; the partials run on across the hops). */
static void tree_row(int64_t i, int n_hops, const int64_t* const* rowptr, const int32_t* const* colidx,
                     const float* const* vals, const float* src, int64_t ld_src, int64_t ld_src_hop, int64_t d,
                     int long_threshold, int mode_sum, float* y, int64_t ldy_row, int64_t ldy_hop, float* P, float* T) {
    if (!mode_sum) {
        for (int s = 0; s < n_hops; ++s) {
            const int64_t sb = rowptr[s][i], se = rowptr[s][i + 1];
            float* o = y + i * ldy_row + s * ldy_hop;
            if (se - sb >= long_threshold) {
                for (int w = 0; w < 4; ++w) {
                    memset(P, 0, (size_t)4 * d * sizeof(float));
                    tree_chunks(colidx[s], vals[s], sb, se, w, 4, src, ld_src, d, P);
                    tree_fold(P, d, T + w * d);
                }
                for (int64_t c = 0; c < d; ++c) o[c] = ((T[c] + T[d + c]) + T[2 * d + c]) + T[3 * d + c];
            } else {
                memset(P, 0, (size_t)4 * d * sizeof(float));
                tree_chunks(colidx[s], vals[s], sb, se, 0, 1, src, ld_src, d, P);
                tree_fold(P, d, o);
            }
        }
    } else {
        int is_long = 0;
        for (int s = 0; s < n_hops; ++s) is_long |= (rowptr[s][i + 1] - rowptr[s][i] >= long_threshold);
        float* o = y + i * ldy_row;
        if (is_long) {
            for (int w = 0; w < 4; ++w) {
                memset(P, 0, (size_t)4 * d * sizeof(float));
                for (int s = 0; s < n_hops; ++s)
                    tree_chunks(colidx[s], vals[s], rowptr[s][i], rowptr[s][i + 1], w, 4, src + s * ld_src_hop, ld_src, d, P);
                tree_fold(P, d, T + w * d);
            }
            for (int64_t c = 0; c < d; ++c) o[c] = ((T[c] + T[d + c]) + T[2 * d + c]) + T[3 * d + c];
        } else {
            memset(P, 0, (size_t)4 * d * sizeof(float));
            for (int s = 0; s < n_hops; ++s)
                tree_chunks(colidx[s], vals[s], rowptr[s][i], rowptr[s][i + 1], 0, 1, src + s * ld_src_hop, ld_src, d, P);
            tree_fold(P, d, o);
        }
    }
}

void oracle_spmm_tree_f32(int n_hops, int64_t n_rows, const int64_t* const* rowptr, const int32_t* const* colidx,
                          const float* const* vals, const float* src, int64_t ld_src, int64_t ld_src_hop, int64_t d,
                          int long_threshold, int mode_sum, float* y, int64_t ldy_row, int64_t ldy_hop) {
    float* P = (float*)malloc((size_t)4 * d * sizeof(float));
    float* T = (float*)malloc((size_t)4 * d * sizeof(float));
    for (int64_t i = 0; i < n_rows; ++i)
        tree_row(i, n_hops, rowptr, colidx, vals, src, ld_src, ld_src_hop, d, long_threshold, mode_sum, y, ldy_row, ldy_hop, P, T);
    free(P);
    free(T);
}

/* The same function with the output rows spread over OpenMP threads: per-row arithmetic unchanged (tree_row), so the
 * bits are those of oracle_spmm_tree_f32.  Exists so that the BASELINE shapes can be checked at FULL size -- every
 * row of Y, not a sample -- in seconds on the test box's host cores (tests/test_fullsize_parity_gpu.py). */
void oracle_spmm_tree_f32_mt(int n_hops, int64_t n_rows, const int64_t* const* rowptr, const int32_t* const* colidx,
                             const float* const* vals, const float* src, int64_t ld_src, int64_t ld_src_hop, int64_t d,
                             int long_threshold, int mode_sum, float* y, int64_t ldy_row, int64_t ldy_hop) {
#pragma omp parallel
    {
        float* P = (float*)malloc((size_t)4 * d * sizeof(float));
        float* T = (float*)malloc((size_t)4 * d * sizeof(float));
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n_rows; ++i)
            tree_row(i, n_hops, rowptr, colidx, vals, src, ld_src, ld_src_hop, d, long_threshold, mode_sum, y, ldy_row, ldy_hop, P, T);
        free(P);
        free(T);
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * Whole-array comparisons for the full-size checks (OpenMP; plain reductions over integers / maxima, so the result
 * does not depend on the thread count). */

/* Order-independent fingerprint: the sum of the fp32 BIT PATTERNS (as int32) in int64 -- what bench.py computes
 * on the device as y.view(int32).to(int64).sum(). */
int64_t oracle_bits_checksum_f32(const float* y, int64_t n) {
    int64_t total = 0;
#pragma omp parallel for reduction(+ : total) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        int32_t b;
        memcpy(&b, y + i, 4);
        total += (int64_t)b;
    }
    return total;
}

/* number of elements whose bit patterns differ; *first = index of the first one (or -1) */
int64_t oracle_count_bit_mismatches_f32(const float* a, const float* b, int64_t n, int64_t* first) {
    int64_t bad = 0, lo = -1;
#pragma omp parallel
    {
        int64_t my_lo = -1;
#pragma omp for reduction(+ : bad) schedule(static)
        for (int64_t i = 0; i < n; ++i)
            if (memcmp(a + i, b + i, 4) != 0) {
                ++bad;
                if (my_lo < 0) my_lo = i;
            }
#pragma omp critical
        if (my_lo >= 0 && (lo < 0 || my_lo < lo)) lo = my_lo;
    }
    if (first) *first = lo;
    return bad;
}

/* max |a - b| (NaN anywhere -> +inf) */
double oracle_max_abs_diff_f32(const float* a, const float* b, int64_t n) {
    double m = 0.0;
#pragma omp parallel for reduction(max : m) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double df = fabs((double)a[i] - (double)b[i]);
        const double v = (df == df) ? df : INFINITY;
        if (v > m) m = v;
    }
    return m;
}

/* ---------------------------------------------------------------------------------------------------------------
 * The synthetic operands of the BASELINE shapes, restated in C so that the full-size checks can
 * rebuild EVERY row of them on the host in seconds.  The definition is h2gcn_amd/synth.py (not reference code --
 * the reference has no benchmark inputs): edge e (global running index of the hop's raw degree sequence) has column
 * (splitmix64(e + key) >> 1) mod n_cols; per row the ids are sorted and de-duplicated; values 1/deg in fp32;
 * features U[-1, 1) on a 2^-23 grid.  Pinned bit-for-bit against the numpy generator in tests/test_synth.py. */
static uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int cmp_i32(const void* a, const void* b) {
    const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x > y) - (x < y);
}

/* rows [r0, r1) of one hop matrix.  raw_ptr: prefix sums of the raw degrees, [n_rows + 1] (global).  col_out / val_out need
 * room for raw_ptr[r1] - raw_ptr[r0] entries; rowptr_out [r1 - r0 + 1] is local (starts at 0); returns the nonzero count. */
int64_t oracle_synth_hop_rows(const int64_t* raw_ptr, int64_t r0, int64_t r1, uint64_t key, int64_t n_cols,
                              int64_t* rowptr_out, int32_t* col_out, float* val_out) {
    const int64_t n = r1 - r0, e_base = raw_ptr[r0];
    int32_t* tmp = (int32_t*)malloc((size_t)(raw_ptr[r1] - e_base + 1) * sizeof(int32_t));
    rowptr_out[0] = 0;
#pragma omp parallel for schedule(dynamic, 512)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t e0 = raw_ptr[r0 + i], e1 = raw_ptr[r0 + i + 1];
        int32_t* t = tmp + (e0 - e_base);
        for (int64_t e = e0; e < e1; ++e) t[e - e0] = (int32_t)((splitmix64((uint64_t)e + key) >> 1) % (uint64_t)n_cols);
        qsort(t, (size_t)(e1 - e0), sizeof(int32_t), cmp_i32);
        int64_t u = 0;
        for (int64_t j = 0; j < e1 - e0; ++j)
            if (j == 0 || t[j] != t[j - 1]) t[u++] = t[j];
        rowptr_out[i + 1] = u;
    }
    for (int64_t i = 0; i < n; ++i) rowptr_out[i + 1] += rowptr_out[i];
#pragma omp parallel for schedule(dynamic, 512)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t cnt = rowptr_out[i + 1] - rowptr_out[i];
        const int32_t* t = tmp + (raw_ptr[r0 + i] - e_base);
        const float inv = 1.0f / (float)cnt;
        for (int64_t j = 0; j < cnt; ++j) {
            col_out[rowptr_out[i] + j] = t[j];
            val_out[rowptr_out[i] + j] = inv;
        }
    }
    free(tmp);
    return rowptr_out[n];
}

/* features of rows [r0, r1), out [r1 - r0, d] */
void oracle_synth_features(int64_t d, uint64_t key, int64_t r0, int64_t r1, float* out) {
    const int64_t i0 = r0 * d, n = (r1 - r0) * d;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t k = splitmix64((uint64_t)(i0 + i) + key) >> 40;      /* 24 random bits: exact in fp32 */
        out[i] = (float)k * 0x1p-23f - 1.0f;
    }
}
