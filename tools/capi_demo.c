/* capi_demo.c -- the C ABI used from plain C, no Python / torch anywhere: what a native host (or a cgo / JNI /
 * ctypes binding) does.  Builds a 4x4 two-hop CSR by hand, runs the fused forward launch and the adjoint, and
 * checks both against values worked out by hand.
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
    /* error channel: a hop mask beyond the plan must fail cleanly */
    int st = h2gcn_spmm_hops_f32(plan, 0x8, dx, D, D, dy, H * D, D, NULL);
    h2gcn_plan_destroy(plan);
    if (st != H2GCN_ERR_INVALID_ARGUMENT) { fprintf(stderr, "expected invalid-argument, got %d\n", st); return 4; }
    printf("capi_demo: max |diff| = %.3g (%s); bad-mask error: \"%s\"\n", worst, worst <= 1e-5 ? "ok" : "FAIL", h2gcn_last_error());
    return worst <= 1e-5 ? 0 : 5;
}
