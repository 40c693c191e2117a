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

/* mode_sum == 0: Y[i, s, :] per hop (forward, y row stride ldy_row, hop stride ldy_hop);
 * mode_sum == 1: out[i, :] = sum over the hops of A_s[i, :] @ src[:, s, :] (adjoint on transposed operands: src row
 *                stride ld_src, hop stride ld_src_hop; the partials run on across the hops). */
void oracle_spmm_tree_f32(int n_hops, int64_t n_rows, const int64_t* const* rowptr, const int32_t* const* colidx,
                          const float* const* vals, const float* src, int64_t ld_src, int64_t ld_src_hop, int64_t d,
                          int long_threshold, int mode_sum, float* y, int64_t ldy_row, int64_t ldy_hop) {
    float* P = (float*)malloc((size_t)4 * d * sizeof(float));
    float* T = (float*)malloc((size_t)4 * d * sizeof(float));
    for (int64_t i = 0; i < n_rows; ++i) {
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
    free(P);
    free(T);
}
