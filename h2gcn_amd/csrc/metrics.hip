// metrics.hip -- masked softmax cross-entropy and masked accuracy of an H2GCN step in ONE pass over the logits.
//
// Reference: h2gcn/models/_metrics.py:8-25 (`masked_softmax_cross_entropy`: per-row softmax cross-entropy times mask / mean(mask),
// then the mean; `masked_accuracy`: the same weighting of argmax agreement), called once per mask by `train_step` and `test_step`
// (h2gcn/models/H2GCN.py:66-74, 77-107: train loss; train / val / test accuracy, val / test loss).  With stock element-wise
// kernels that is ~12 passes over the [N, C] logits and label matrices per evaluation (log-softmax, product with the labels,
// row sums, weighting, argmax, comparisons ...): ~1.4 ms of a 60 ms training step and more of an evaluation at the products
// shape (N = 2.4 M, C = 47).  Here:
//   * h2gcn_masked_metrics_f32: one read of Z and -- only for rows whose weight is non-zero -- of each label matrix; per row the
//     max, log-sum-exp and argmax are shared by all (label matrix, row weight) sets; per-workgroup partials in fp64, summed in a
//     fixed order (deterministic);
//   * h2gcn_masked_ce_backward_f32: dZ = g * w[n] * (softmax(Z[n]) * sum_c Y[n, c] - Y[n]) in one pass (rows of weight 0: zeros,
//     without reading them).
// Geometry: 16 lanes per row (4 classes per lane as one 16-byte load, C <= 64), 4 rows per wave; row reductions are xor
// butterflies inside the 16-lane group.  HBM-bound: the algorithmic bytes are N * C * 4 (+ labels of the weighted rows).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "capi_internal.h"
#include "h2gcn_hip.h"

namespace {

using h2gcn::fail;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kMaxSets = H2GCN_METRICS_MAX_SETS;
constexpr int kThreads = 256;
constexpr int kRowsPerBlock = 16;   // 4 waves x 4 rows

struct Sets {
    const float* y[kMaxSets];
    int64_t ldy[kMaxSets];
    const float* w[kMaxSets];
    int n;
};

// columns c0 .. c0+3 of a row (zero beyond C); `ok[i]` tells which are real
__device__ __forceinline__ f4u load_quad(const float* __restrict__ row, int c0, int C) {
    f4u v = {0.f, 0.f, 0.f, 0.f};
    if (c0 + 4 <= C) {
        v = *reinterpret_cast<const f4u*>(row + c0);
    } else if (c0 < C) {
        v[0] = row[c0];
        if (c0 + 1 < C) v[1] = row[c0 + 1];
        if (c0 + 2 < C) v[2] = row[c0 + 2];
    }
    return v;
}
__device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
    for (int k = 1; k < 16; k <<= 1) v += __shfl_xor(v, k, 16);
    return v;
}
__device__ __forceinline__ float grp_max(float v) {
#pragma unroll
    for (int k = 1; k < 16; k <<= 1) v = fmaxf(v, __shfl_xor(v, k, 16));
    return v;
}
// index of the first maximum of the row (ties -> lowest column, as numpy / torch argmax)
__device__ __forceinline__ int grp_argmax(const f4u& v, int c0, int C) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (c0 + i < C && (v[i] > bv || bi == 0x7fffffff)) {
            bv = v[i];
            bi = c0 + i;
        }
#pragma unroll
    for (int k = 1; k < 16; k <<= 1) {
        const float ov = __shfl_xor(bv, k, 16);
        const int oi = __shfl_xor(bi, k, 16);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) {
            bv = ov;
            bi = oi;
        }
    }
    return bi;
}
// max and log-sum-exp of a row
__device__ __forceinline__ void row_lse(const f4u& z, int c0, int C, float& mx, float& lse) {
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (c0 + i < C) m = fmaxf(m, z[i]);
    mx = grp_max(m);
    float e = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (c0 + i < C) e += expf(z[i] - mx);
    lse = mx + logf(grp_sum(e));
}

__global__ __launch_bounds__(kThreads) void masked_metrics_kernel(const float* __restrict__ Z, int64_t ldz, int64_t n_rows, int C, Sets s,
                                                                  double* __restrict__ partial) {
    __shared__ double red[kThreads / 16][2 * kMaxSets];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, rg = lane >> 4;
    const int c0 = 4 * sub;
    double loss_acc[kMaxSets], hit_acc[kMaxSets];
#pragma unroll
    for (int m = 0; m < kMaxSets; ++m) loss_acc[m] = hit_acc[m] = 0.0;
    // Three rows of this lane group are in flight: the weights of row i+2, the logits / label rows of row i+1 (requested with
    // weights fetched an iteration earlier) and the reductions of row i -- otherwise every row pays weights -> data -> reduce as
    // three serial memory latencies.  (Products shape, three sets covering every row: 0.75 -> 0.63 ms; what remains is the
    // ~200 VALU instructions of the per-row reductions, exp / log and fp64 accumulation rather than memory.)
    const int64_t stride = (int64_t)gridDim.x * kRowsPerBlock;
    auto load_w = [&](int64_t row, float (&w)[kMaxSets]) -> bool {
        bool any = false;
#pragma unroll
        for (int m = 0; m < kMaxSets; ++m) {
            w[m] = (m < s.n && row < n_rows) ? s.w[m][row] : 0.f;
            any |= w[m] != 0.f;
        }
        return any;   // uniform inside the 16-lane group: the butterflies below only pair lanes of one group
    };
    auto load_zy = [&](int64_t row, const float (&w)[kMaxSets], bool any, f4u& z, f4u (&ys)[kMaxSets]) {
        if (!any) return;
        z = load_quad(Z + row * ldz, c0, C);
#pragma unroll
        for (int m = 0; m < kMaxSets; ++m)
            if (m < s.n && w[m] != 0.f) ys[m] = load_quad(s.y[m] + row * s.ldy[m], c0, C);   // zero beyond C
    };
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 4 + rg;
    float w_c[kMaxSets], w_b[kMaxSets], w_a[kMaxSets];
    f4u z_c = {0.f, 0.f, 0.f, 0.f}, z_b = z_c, ys_c[kMaxSets], ys_b[kMaxSets];
    bool any_c = load_w(row0, w_c);
    load_zy(row0, w_c, any_c, z_c, ys_c);
    bool any_b = load_w(row0 + stride, w_b);
    for (int64_t row = row0; row < n_rows; row += stride) {
        load_zy(row + stride, w_b, any_b, z_b, ys_b);
        const bool any_a = load_w(row + 2 * stride, w_a);
        if (any_c) {
            float mx, lse;
            row_lse(z_c, c0, C, mx, lse);
            const int zi = grp_argmax(z_c, c0, C);
#pragma unroll
            for (int m = 0; m < kMaxSets; ++m) {
                if (m >= s.n || w_c[m] == 0.f) continue;
                const f4u y = ys_c[m];
                const float ydot = grp_sum(y[0] * z_c[0] + y[1] * z_c[1] + y[2] * z_c[2] + y[3] * z_c[3]);
                const float ysum = grp_sum((y[0] + y[1]) + (y[2] + y[3]));
                const int yi = grp_argmax(y, c0, C);
                loss_acc[m] += (double)w_c[m] * (double)(ysum * lse - ydot);   // - sum_c y_c (z_c - lse)
                hit_acc[m] += zi == yi ? (double)w_c[m] : 0.0;
            }
        }
        z_c = z_b;
        any_c = any_b;
        any_b = any_a;
#pragma unroll
        for (int m = 0; m < kMaxSets; ++m) {
            ys_c[m] = ys_b[m];
            w_c[m] = w_b[m];
            w_b[m] = w_a[m];
        }
    }
    if (sub == 0) {
#pragma unroll
        for (int m = 0; m < kMaxSets; ++m) {
            red[wave * 4 + rg][m] = loss_acc[m];
            red[wave * 4 + rg][kMaxSets + m] = hit_acc[m];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * kMaxSets) {
        double t = 0.0;
        for (int g = 0; g < kThreads / 16; ++g) t += red[g][threadIdx.x];
        partial[(int64_t)blockIdx.x * 2 * kMaxSets + threadIdx.x] = t;
    }
}

// out[q] = sum over workgroups of partial[.][q], q = (kind, set): 32 strided sub-sums per quantity, combined in order
__global__ __launch_bounds__(kThreads) void metrics_finish_kernel(const double* __restrict__ partial, int n_blocks, int n_sets,
                                                                  float* __restrict__ loss_out, float* __restrict__ acc_out) {
    __shared__ double sub[kThreads / (2 * kMaxSets)][2 * kMaxSets];
    const int q = threadIdx.x % (2 * kMaxSets), chunk = threadIdx.x / (2 * kMaxSets);
    constexpr int n_chunks = kThreads / (2 * kMaxSets);
    double t = 0.0;
    for (int b = chunk; b < n_blocks; b += n_chunks) t += partial[(int64_t)b * 2 * kMaxSets + q];
    sub[chunk][q] = t;
    __syncthreads();
    if (chunk == 0) {
        double tot = 0.0;
        for (int c = 0; c < n_chunks; ++c) tot += sub[c][q];
        const int m = q % kMaxSets;
        if (m < n_sets) {
            if (q < kMaxSets) loss_out[m] = (float)tot;
            else if (acc_out) acc_out[m] = (float)tot;
        }
    }
}

__global__ __launch_bounds__(kThreads) void masked_ce_backward_kernel(const float* __restrict__ Z, int64_t ldz, int64_t n_rows, int C,
                                                                      const float* __restrict__ Y, int64_t ldy, const float* __restrict__ w,
                                                                      const float* __restrict__ gscale, float* __restrict__ dZ, int64_t lddz) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, rg = lane >> 4;
    const int c0 = 4 * sub;
    const float g = gscale ? *gscale : 1.f;
    for (int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 4 + rg; row < n_rows; row += (int64_t)gridDim.x * kRowsPerBlock) {
        const float wr = w[row];
        f4u o = {0.f, 0.f, 0.f, 0.f};
        if (wr != 0.f) {
            const f4u z = load_quad(Z + row * ldz, c0, C);
            const f4u y = load_quad(Y + row * ldy, c0, C);
            float mx, lse;
            row_lse(z, c0, C, mx, lse);
            const float ysum = grp_sum((y[0] + y[1]) + (y[2] + y[3]));
            const float gw = g * wr;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = gw * (expf(z[i] - lse) * ysum - y[i]);
        }
        float* dst = dZ + row * lddz + c0;
        if (c0 + 4 <= C) {
            *reinterpret_cast<f4u*>(dst) = o;
        } else if (c0 < C) {
            dst[0] = o[0];
            if (c0 + 1 < C) dst[1] = o[1];
            if (c0 + 2 < C) dst[2] = o[2];
        }
    }
}

int cu_count() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
}
int64_t grid_for(int64_t n_rows) {
    return std::max<int64_t>(1, std::min<int64_t>((n_rows + kRowsPerBlock - 1) / kRowsPerBlock, (int64_t)cu_count() * 8));
}

}  // namespace

extern "C" {

size_t h2gcn_masked_metrics_workspace_bytes(int64_t n_rows) {
    if (n_rows < 0) return 0;
    return (size_t)grid_for(n_rows) * 2 * kMaxSets * sizeof(double);
}

int h2gcn_masked_metrics_f32(const float* Z, int64_t ldz, int64_t n_rows, int32_t C, int32_t n_sets, const float* const* Y,
                             const int64_t* ldy, const float* const* w, float* loss_out, float* acc_out, void* workspace,
                             size_t workspace_bytes, void* stream_v) {
    if (n_rows < 0 || C < 1 || C > 64 || n_sets < 1 || n_sets > kMaxSets)
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "masked_metrics: n_rows %lld, C %d (<= 64), n_sets %d (1..%d)", (long long)n_rows, C, n_sets, kMaxSets);
    if ((!Z && n_rows > 0) || ldz < C || !Y || !ldy || !w || !loss_out) return fail(H2GCN_ERR_INVALID_ARGUMENT, "masked_metrics: NULL operand or ldz < C");
    Sets s{};
    s.n = n_sets;
    for (int m = 0; m < n_sets; ++m) {
        if ((!Y[m] || !w[m]) && n_rows > 0) return fail(H2GCN_ERR_INVALID_ARGUMENT, "masked_metrics: set %d has a NULL label matrix / weight vector", m);
        if (ldy[m] < C) return fail(H2GCN_ERR_INVALID_ARGUMENT, "masked_metrics: set %d label row stride %lld < C", m, (long long)ldy[m]);
        s.y[m] = Y[m];
        s.ldy[m] = ldy[m];
        s.w[m] = w[m];
    }
    const int64_t grid = grid_for(n_rows);
    if (!workspace || workspace_bytes < (size_t)grid * 2 * kMaxSets * sizeof(double) || ((uintptr_t)workspace & 7u))
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "masked_metrics: workspace of %zu bytes (8-byte aligned) needed, got %zu",
                    (size_t)grid * 2 * kMaxSets * sizeof(double), workspace_bytes);
    hipStream_t stream = (hipStream_t)stream_v;
    double* partial = (double*)workspace;
    hipLaunchKernelGGL(masked_metrics_kernel, dim3((unsigned)grid), dim3(kThreads), 0, stream, Z, ldz, n_rows, (int)C, s, partial);
    H2GCN_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(metrics_finish_kernel, dim3(1), dim3(kThreads), 0, stream, (const double*)partial, (int)grid, (int)n_sets, loss_out, acc_out);
    H2GCN_HIP_TRY(hipGetLastError());
    return H2GCN_OK;
}

int h2gcn_masked_ce_backward_f32(const float* Z, int64_t ldz, int64_t n_rows, int32_t C, const float* Y, int64_t ldy, const float* w,
                                 const float* gscale_dev, float* dZ, int64_t lddz, void* stream_v) {
    if (n_rows < 0 || C < 1 || C > 64) return fail(H2GCN_ERR_INVALID_ARGUMENT, "masked_ce_backward: n_rows %lld, C %d (<= 64)", (long long)n_rows, C);
    if (n_rows == 0) return H2GCN_OK;
    if (!Z || !Y || !w || !dZ || ldz < C || ldy < C || lddz < C) return fail(H2GCN_ERR_INVALID_ARGUMENT, "masked_ce_backward: NULL operand or a row stride < C");
    hipLaunchKernelGGL(masked_ce_backward_kernel, dim3((unsigned)grid_for(n_rows)), dim3(kThreads), 0, (hipStream_t)stream_v, Z, ldz, n_rows, (int)C, Y,
                       ldy, w, gscale_dev, dZ, lddz);
    H2GCN_HIP_TRY(hipGetLastError());
    return H2GCN_OK;
}

}  // extern "C"
