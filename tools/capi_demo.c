/* capi_demo.c -- the C ABI used from plain C, no Python / torch anywhere: what a native host (or a cgo / JNI /
 * ctypes binding) does.  Builds a 4x4 two-hop CSR by hand, runs the fused forward launch and the adjoint, rebuilds
 * the 2-hop ring with the ring kernels (whole and as a row window), runs a fused bias+ReLU launch and the dropout+dense
 * classifier kernel, and checks everything against values worked out by hand.
 *   hipcc -x c ... is not needed: compile as C with gcc, link libamdhip64 + libh2gcn_hip:
 *   gcc -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/capi_demo.c -o capi_demo \
 *       -L h2gcn_amd/csrc -lh2gcn_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../h2gcn_amd/csrc' -lm
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "h2gcn_hip.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define H2(x) do { int s_ = (x); if (s_ < 0) { fprintf(stderr, "%s: %s\n", #x, h2gcn_last_error()); return 3; } } while (0)

static void* to_device(const void* src, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) return NULL;
    if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(void) {
    enum { N = 4, D = 4, H = 2 };
    /* hop 0: path graph 0-1-2-3, row-normalised; hop 1: exact 2-hop ring of the same graph (row 1 <-> 3, 0 <-> 2) */
    const int64_t rp0[N + 1] = {0, 1, 3, 5, 6};
    const int32_t ci0[6] = {1, 0, 2, 1, 3, 2};
    const float va0[6] = {1.f, .5f, .5f, .5f, .5f, 1.f};
    const int64_t rp1[N + 1] = {0, 1, 2, 3, 4};
    const int32_t ci1[4] = {2, 3, 0, 1};
    const float va1[4] = {1.f, 1.f, 1.f, 1.f};
    float x[N * D];
    for (int i = 0; i < N * D; ++i) x[i] = (float)(i + 1);

    if (h2gcn_abi_version() != H2GCN_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    if (h2gcn_device_count() < 1) { fprintf(stderr, "no GPU: %s\n", h2gcn_last_error()); return 1; }

    const int64_t* rowptr[H] = {to_device(rp0, sizeof rp0), to_device(rp1, sizeof rp1)};
    const int32_t* colidx[H] = {to_device(ci0, sizeof ci0), to_device(ci1, sizeof ci1)};
    const float* vals[H] = {to_device(va0, sizeof va0), to_device(va1, sizeof va1)};
    float* dx = to_device(x, sizeof x);
    float* dy = NULL;
    float* dg = NULL;
    HIP(hipMalloc((void**)&dy, sizeof(float) * N * H * D));
    HIP(hipMalloc((void**)&dg, sizeof(float) * N * D));

    h2gcn_plan_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.struct_size = sizeof opts;
    opts.flags = H2GCN_PLAN_BUILD_TRANSPOSE;
    h2gcn_plan_t* plan = NULL;
    H2(h2gcn_plan_create(H, N, N, rowptr, colidx, vals, &opts, NULL, &plan));
    H2(h2gcn_spmm_hops_f32(plan, 0, dx, D, D, dy, H * D, D, NULL));
    H2(h2gcn_spmm_hops_T_f32(plan, 0, dy, H * D, D, D, dg, D, NULL)); /* dX for dY := Y */
    HIP(hipDeviceSynchronize());
    {   /* CSR-adaptive dispatch, introspection: every segment of this 4-node graph has 1-2 nonzeros -> all "short" */
        int64_t seg[3 * H], nz[3 * H], listed = 0;
        H2(h2gcn_plan_segment_classes(plan, 0, 0, D, D, seg, nz, &listed));
        if (seg[0] != N || seg[1] != 0 || seg[2] != 0 || seg[3] != N || nz[0] != 6 || nz[3] != 4 || listed != 2 * N) {
            fprintf(stderr, "segment classes: short %lld/%lld medium %lld long %lld, nonzeros %lld/%lld, listed %lld\n", (long long)seg[0],
                    (long long)seg[3], (long long)seg[1], (long long)seg[2], (long long)nz[0], (long long)nz[3], (long long)listed);
            return 4;
        }
    }

    float y[N * H * D], g[N * D];
    HIP(hipMemcpy(y, dy, sizeof y, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(g, dg, sizeof g, hipMemcpyDeviceToHost));

    /* host check: dense 4x4 matrices */
    const float A0[N][N] = {{0, 1, 0, 0}, {.5f, 0, .5f, 0}, {0, .5f, 0, .5f}, {0, 0, 1, 0}};
    const float A1[N][N] = {{0, 0, 1, 0}, {0, 0, 0, 1}, {1, 0, 0, 0}, {0, 1, 0, 0}};
    double worst = 0;
    for (int i = 0; i < N; ++i)
        for (int c = 0; c < D; ++c) {
            double w0 = 0, w1 = 0;
            for (int j = 0; j < N; ++j) { w0 += A0[i][j] * x[j * D + c]; w1 += A1[i][j] * x[j * D + c]; }
            worst = fmax(worst, fabs(w0 - y[(i * H + 0) * D + c]));
            worst = fmax(worst, fabs(w1 - y[(i * H + 1) * D + c]));
        }
    for (int j = 0; j < N; ++j)
        for (int c = 0; c < D; ++c) {
            double w = 0;
            for (int i = 0; i < N; ++i) w += A0[i][j] * y[(i * H + 0) * D + c] + A1[i][j] * y[(i * H + 1) * D + c];
            worst = fmax(worst, fabs(w - g[j * D + c]));
        }
    /* ---- operand construction from C: exact 2-hop ring of the path graph 0-1-2-3 with the ring kernels, RW values --- */
    {
        const int64_t arp[N + 1] = {0, 1, 3, 5, 6};                 /* A = the path graph's pattern (= hop 0's pattern) */
        const int64_t* d_arp = to_device(arp, sizeof arp);
        const int32_t* d_aci = colidx[0];
        const int64_t* sub_rp[1] = {d_arp};
        const int32_t* sub_ci[1] = {d_aci};
        size_t sb = h2gcn_ring_scratch_bytes(N);
        void* scratch = NULL;
        int64_t* ring_rp = NULL;
        int64_t nnz2 = -1;
        HIP(hipMalloc(&scratch, sb));
        HIP(hipMalloc((void**)&ring_rp, sizeof(int64_t) * (N + 1)));
        /* ring_2 = (expansion of the frontier ring_1 = A through A) minus ring_1 minus the diagonal */
        H2(h2gcn_ring_count(N, d_arp, d_aci, d_arp, d_aci, 0, NULL, NULL, 0, 1, sub_rp, sub_ci, 1, ring_rp, &nnz2, scratch, sb, NULL));
        int32_t* ring_ci = NULL;
        float* ring_va = NULL;
        HIP(hipMalloc((void**)&ring_ci, sizeof(int32_t) * (nnz2 > 0 ? nnz2 : 1)));
        HIP(hipMalloc((void**)&ring_va, sizeof(float) * (nnz2 > 0 ? nnz2 : 1)));
        H2(h2gcn_ring_fill(N, d_arp, d_aci, d_arp, d_aci, 0, NULL, NULL, 0, 1, sub_rp, sub_ci, 1, ring_rp, ring_ci, scratch, sb, NULL));
        const double s_rw[3] = {0.0, 1.0, 0.5};                      /* k^-1 with inf -> 0 */
        const double* d_tab = to_device(s_rw, sizeof s_rw);
        H2(h2gcn_hop_normalize(N, ring_rp, ring_ci, 2 /* RW */, d_tab, 3, ring_va, NULL));
        HIP(hipDeviceSynchronize());
        int32_t h_ci[4];
        float h_va[4];
        if (nnz2 != 4) { fprintf(stderr, "2-hop ring has %lld nonzeros, expected 4\n", (long long)nnz2); return 6; }
        HIP(hipMemcpy(h_ci, ring_ci, sizeof h_ci, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(h_va, ring_va, sizeof h_va, hipMemcpyDeviceToHost));
        for (int i = 0; i < 4; ++i)
            if (h_ci[i] != ci1[i] || h_va[i] != 1.f) { fprintf(stderr, "ring entry %d: col %d val %g\n", i, h_ci[i], h_va[i]); return 6; }
    }
    /* ---- launch options: Y = relu(A X + b) in one launch (SparseDense.call) ---------------------------------------- */
    {
        const float bias[D] = {-3.f, -2.5f, 100.f, -1000.f};
        const float* d_bias = to_device(bias, sizeof bias);
        h2gcn_launch_opts lo;
        memset(&lo, 0, sizeof lo);
        lo.struct_size = sizeof lo;
        lo.flags = H2GCN_LAUNCH_RELU;
        lo.bias_dev = d_bias;
        H2(h2gcn_spmm_hops_opts_f32(plan, 0x1, dx, D, D, dy, D, D, &lo, NULL));    /* hop 0 only -> [N, 1, D] */
        HIP(hipDeviceSynchronize());
        float y0[N * D];
        HIP(hipMemcpy(y0, dy, sizeof y0, hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i)
            for (int c = 0; c < D; ++c) {
                double w = bias[c];
                for (int j = 0; j < N; ++j) w += A0[i][j] * x[j * D + c];
                worst = fmax(worst, fabs(fmax(w, 0.0) - y0[i * D + c]));
            }
    }
    /* ---- a row WINDOW of the 2-hop ring (what one rank of a row partition builds): rows [2, 4) only ----------------------- */
    {
        const int64_t arp[N + 1] = {0, 1, 3, 5, 6};
        const int64_t* d_arp = to_device(arp, sizeof arp);
        const int64_t win_rp[3] = {0, 2, 3};                        /* rows 2, 3 of A as a window CSR: {1, 3}, {2} */
        const int32_t win_ci[3] = {1, 3, 2};
        const int64_t* d_wrp = to_device(win_rp, sizeof win_rp);
        const int32_t* d_wci = to_device(win_ci, sizeof win_ci);
        const int64_t* sub_rp[1] = {d_wrp};
        const int32_t* sub_ci[1] = {d_wci};
        size_t sb = h2gcn_ring_scratch_bytes(N);
        void* scratch = NULL;
        int64_t* ring_rp = NULL;
        int32_t* ring_ci = NULL;
        int64_t nnz2 = -1;
        HIP(hipMalloc(&scratch, sb));
        HIP(hipMalloc((void**)&ring_rp, sizeof(int64_t) * 3));
        HIP(hipMalloc((void**)&ring_ci, sizeof(int32_t) * 4));
        /* frontier = the window's rows of ring_1, A = the whole pattern, minus the window's ring_1 rows, minus the diagonal */
        H2(h2gcn_ring_count_rows(N, 2, 2, d_arp, colidx[0], d_wrp, d_wci, 0, NULL, NULL, 0, 1, sub_rp, sub_ci, 1, ring_rp, &nnz2, scratch, sb, NULL));
        H2(h2gcn_ring_fill_rows(N, 2, 2, d_arp, colidx[0], d_wrp, d_wci, 0, NULL, NULL, 0, 1, sub_rp, sub_ci, 1, ring_rp, ring_ci, scratch, sb, NULL));
        HIP(hipDeviceSynchronize());
        int32_t h_ci[2];
        HIP(hipMemcpy(h_ci, ring_ci, sizeof h_ci, hipMemcpyDeviceToHost));
        if (nnz2 != 2 || h_ci[0] != 0 || h_ci[1] != 1) { fprintf(stderr, "ring window: nnz %lld cols %d %d\n", (long long)nnz2, h_ci[0], h_ci[1]); return 7; }
    }
    /* ---- the classifier side: Z = X W + b through the dropout+dense kernel with the mask off (keep_prob = 1), then with
     *      keep_prob = 0.5: every output must be a sum over a SUBSET of the terms, each scaled by 2 -------------------------- */
    {
        enum { C = 3 };
        float w[D * C], b[C] = {0.5f, -1.f, 2.f}, z[N * C], z2[N * C];
        for (int i = 0; i < D * C; ++i) w[i] = 0.25f * (float)((i % 5) - 2);
        const float* d_w = to_device(w, sizeof w);
        const float* d_b = to_device(b, sizeof b);
        float* d_z = NULL;
        void* ws = NULL;
        size_t wsb = h2gcn_dropout_dense_workspace_bytes(N, D, C);
        HIP(hipMalloc((void**)&d_z, sizeof z));
        HIP(hipMalloc(&ws, wsb));
        H2(h2gcn_dropout_dense_f32(dx, D, N, D, d_w, C, d_b, 1.0f, 42u, NULL, d_z, C, ws, wsb, NULL));
        HIP(hipDeviceSynchronize());
        HIP(hipMemcpy(z, d_z, sizeof z, hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i)
            for (int c = 0; c < C; ++c) {
                double want = b[c];
                for (int k = 0; k < D; ++k) want += x[i * D + k] * w[k * C + c];
                worst = fmax(worst, fabs(want - z[i * C + c]) / 16.0);
            }
        H2(h2gcn_dropout_dense_f32(dx, D, N, D, d_w, C, NULL, 0.5f, 42u, NULL, d_z, C, ws, wsb, NULL));
        HIP(hipDeviceSynchronize());
        HIP(hipMemcpy(z2, d_z, sizeof z2, hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i) {                                /* some subset of {0..D-1} must explain the whole row */
            int found = 0;
            for (int m = 0; m < (1 << D) && !found; ++m) {
                int ok = 1;
                for (int c = 0; c < C && ok; ++c) {
                    double want = 0;
                    for (int k = 0; k < D; ++k) if (m & (1 << k)) want += 2.0 * x[i * D + k] * w[k * C + c];
                    ok = fabs(want - z2[i * C + c]) <= 1e-4;
                }
                found = ok;
            }
            if (!found) { fprintf(stderr, "dropout row %d is not a masked, rescaled product\n", i); return 8; }
        }
    }
    /* ---- masked cross-entropy / accuracy of logits Z = x (first 3 columns) against one-hot labels, one set, then its gradient --- */
    {
        enum { C = 3 };
        float zl[N * C], yl[N * C], wl[N], out[2], dz[N * C];
        double want_loss = 0, want_acc = 0;
        for (int i = 0; i < N; ++i) {
            int lab = i % C, arg = 0;
            double se = 0;
            for (int c = 0; c < C; ++c) { zl[i * C + c] = x[i * D + c] * 3.f; yl[i * C + c] = c == lab ? 1.f : 0.f; }
            for (int c = 0; c < C; ++c) { se += exp((double)zl[i * C + c]); if (zl[i * C + c] > zl[i * C + arg]) arg = c; }
            wl[i] = i == 1 ? 0.f : 1.f / (float)(N - 1);             /* row 1 is outside the mask */
            want_loss += wl[i] * (log(se) - zl[i * C + lab]);
            want_acc += wl[i] * (arg == lab);
        }
        const float* d_zl = to_device(zl, sizeof zl);
        const float* d_yl = to_device(yl, sizeof yl);
        const float* d_wl = to_device(wl, sizeof wl);
        float *d_out = NULL, *d_dz = NULL;
        void* ws = NULL;
        size_t wsb = h2gcn_masked_metrics_workspace_bytes(N);
        HIP(hipMalloc((void**)&d_out, sizeof out));
        HIP(hipMalloc((void**)&d_dz, sizeof dz));
        HIP(hipMalloc(&ws, wsb));
        const float* ys[1] = {d_yl};
        const float* ws_rows[1] = {d_wl};
        const int64_t ldys[1] = {C};
        H2(h2gcn_masked_metrics_f32(d_zl, C, N, C, 1, ys, ldys, ws_rows, d_out, d_out + 1, ws, wsb, NULL));
        H2(h2gcn_masked_ce_backward_f32(d_zl, C, N, C, d_yl, C, d_wl, NULL, d_dz, C, NULL));
        HIP(hipDeviceSynchronize());
        HIP(hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(dz, d_dz, sizeof dz, hipMemcpyDeviceToHost));
        worst = fmax(worst, fabs(out[0] - want_loss));
        worst = fmax(worst, fabs(out[1] - want_acc));
        for (int i = 0; i < N; ++i) {                                 /* dZ = w (softmax - y); rows sum to zero; masked-out row is zero */
            double rs = 0;
            for (int c = 0; c < C; ++c) rs += dz[i * C + c];
            worst = fmax(worst, fabs(rs));
            if (i == 1 && (dz[C] != 0.f || dz[C + 1] != 0.f || dz[C + 2] != 0.f)) { fprintf(stderr, "gradient of a row outside the mask is not zero\n"); return 9; }
        }
    }
    /* error channel: a hop mask beyond the plan must fail cleanly */
    int st = h2gcn_spmm_hops_f32(plan, 0x8, dx, D, D, dy, H * D, D, NULL);
    h2gcn_plan_destroy(plan);
    if (st != H2GCN_ERR_INVALID_ARGUMENT) { fprintf(stderr, "expected invalid-argument, got %d\n", st); return 4; }
    printf("capi_demo: max |diff| = %.3g (%s); bad-mask error: \"%s\"\n", worst, worst <= 1e-5 ? "ok" : "FAIL", h2gcn_last_error());
    return worst <= 1e-5 ? 0 : 5;
}
