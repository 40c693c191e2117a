// optimizer.hip -- the Adam update of an H2GCN training step with the arithmetic of the reference's optimizer, one launch for
// all parameters.
//
// Reference: `keras.optimizers.get(optimizer).from_config({"lr": lr})` (h2gcn/models/H2GCN.py:62-63; default "adam",
// lr 0.01, :18-21) applied by `optimizer.apply_gradients` (:73).  Keras' Adam (beta_1 0.9, beta_2 0.999, epsilon 1e-7, no
// amsgrad) runs TensorFlow's ApplyAdam kernel -- third-party, absent from /root/reference (TensorFlow >= 2.0, README.md:34;
// core/kernels/training_ops.cc) -- whose published update is, with t the 1-based step:
//     alpha = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t)
//     m += (g - m) * (1 - beta_1);   v += (g*g - v) * (1 - beta_2);   var -= (m * alpha) / (sqrt(v) + epsilon)
// i.e. epsilon is added to the UNCORRECTED sqrt(v) ("epsilon hat" of the Adam paper, section 2), not to the bias-corrected one
// as torch.optim.Adam does: the two differ by a factor sqrt(1 - beta_2^t) on epsilon (31.6x at t = 1).  H2GCN-2 has four
// parameter tensors (~95 k elements on Cora): stock multi-tensor Adam is ~10 launches per step, this is one, and the step
// counter is read from device memory so that a replayed hipGraph advances it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "capi_internal.h"
#include "h2gcn_hip.h"

namespace {

using h2gcn::fail;
constexpr int kMaxTensors = H2GCN_ADAM_MAX_TENSORS;

struct AdamSet {
    float* p[kMaxTensors];
    const float* g[kMaxTensors];
    float* m[kMaxTensors];
    float* v[kMaxTensors];
    int64_t n[kMaxTensors];
    float l2x2[kMaxTensors];   // 2 * (L2 coefficient of the tensor); 0 = not regularised
};

// a + b * c with TWO roundings (what two separate element-wise kernels produce); hipcc contracts `a + b * c` -- and the
// __fmul_rn / __fadd_rn spellings, which are plain operators in the HIP headers -- into one fma unless told not to
__device__ __forceinline__ float mul_then_add(float a, float b, float c) {
#pragma clang fp contract(off)
    const float prod = b * c;
    return a + prod;
}

__global__ __launch_bounds__(256) void adam_keras_kernel(AdamSet s, float lr, float beta1, float beta2, float eps,
                                                         const int64_t* __restrict__ step_dev, int64_t step_host) {
    const int k = blockIdx.y;
    const int64_t n = s.n[k];
    const int64_t t = step_dev ? *step_dev : step_host;
    // fp32 like the reference (Keras forms beta^t and alpha as fp32 tensors)
    const float b1p = powf(beta1, (float)t), b2p = powf(beta2, (float)t);
    const float alpha = lr * sqrtf(1.f - b2p) / (1.f - b1p);
    const float c1 = 1.f - beta1, c2 = 1.f - beta2;
    float* __restrict__ p = s.p[k];
    const float* __restrict__ g = s.g[k];
    float* __restrict__ m = s.m[k];
    float* __restrict__ v = s.v[k];
    const float l2x2 = s.l2x2[k];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        // gradient of the keras l2 regulariser, l2 * sum(w^2) (reference h2gcn/models/H2GCN.py:239-240, 247-248): 2 * l2 * w,
        // added with the roundings autograd's separate multiply and add would make (no fused multiply-add)
        if (l2x2 != 0.f) gi = mul_then_add(gi, p[i], l2x2);
        float mi = m[i], vi = v[i];
        mi += (gi - mi) * c1;
        vi += (gi * gi - vi) * c2;
        m[i] = mi;
        v[i] = vi;
        p[i] -= (mi * alpha) / (sqrtf(vi) + eps);
    }
}

}  // namespace

extern "C" int h2gcn_adam_keras_f32(int32_t n_tensors, float* const* params, const float* const* grads, float* const* m,
                                    float* const* v, const int64_t* sizes, float lr, float beta1, float beta2, float eps,
                                    const int64_t* step_dev, int64_t step, void* stream_v) {
    return h2gcn_adam_keras_l2_f32(n_tensors, params, grads, m, v, sizes, nullptr, lr, beta1, beta2, eps, step_dev, step, stream_v);
}

extern "C" int h2gcn_adam_keras_l2_f32(int32_t n_tensors, float* const* params, const float* const* grads, float* const* m,
                                       float* const* v, const int64_t* sizes, const float* l2, float lr, float beta1, float beta2,
                                       float eps, const int64_t* step_dev, int64_t step, void* stream_v) {
    if (n_tensors < 0 || (n_tensors > 0 && (!params || !grads || !m || !v || !sizes)))
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "adam: n_tensors %d or a NULL table", n_tensors);
    if (!(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f))
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "adam: lr %g, beta_1 %g, beta_2 %g, epsilon %g", (double)lr, (double)beta1, (double)beta2, (double)eps);
    if (!step_dev && step < 1) return fail(H2GCN_ERR_INVALID_ARGUMENT, "adam: step %lld (1-based) and no device counter", (long long)step);
    for (int base = 0; base < n_tensors; base += kMaxTensors) {
        AdamSet s{};
        const int count = std::min(kMaxTensors, n_tensors - base);
        int64_t longest = 0;
        for (int k = 0; k < count; ++k) {
            const int q = base + k;
            if (sizes[q] < 0 || (sizes[q] > 0 && (!params[q] || !grads[q] || !m[q] || !v[q])))
                return fail(H2GCN_ERR_INVALID_ARGUMENT, "adam: tensor %d has size %lld or a NULL pointer", q, (long long)sizes[q]);
            s.p[k] = params[q];
            s.g[k] = grads[q];
            s.m[k] = m[q];
            s.v[k] = v[q];
            s.n[k] = sizes[q];
            s.l2x2[k] = l2 ? 2.f * l2[q] : 0.f;
            if (l2 && !(l2[q] >= 0.f)) return fail(H2GCN_ERR_INVALID_ARGUMENT, "adam: l2[%d] = %g", q, (double)l2[q]);
            longest = std::max(longest, sizes[q]);
        }
        if (longest == 0) continue;
        const unsigned gx = (unsigned)std::min<int64_t>((longest + 255) / 256, 2048);
        hipLaunchKernelGGL(adam_keras_kernel, dim3(gx, (unsigned)count), dim3(256), 0, (hipStream_t)stream_v, s, lr, beta1, beta2, eps, step_dev, step);
        H2GCN_HIP_TRY(hipGetLastError());
    }
    return H2GCN_OK;
}

// ---- the value of the L2 penalty: sum_k l2_k * sum(w_k^2) over all regularised kernels in ONE launch ------------------------
// (reference: keras regularizers.l2 on every dense kernel, h2gcn/models/H2GCN.py:239-240, 247-248, 363-367: the loss a step
// reports is cross-entropy + these terms.  With stock element-wise ops this is a pow, a reduction, a multiply and an add PER
// kernel per loss -- on Cora a third of an epoch's launches, profiles/r04_cora_epoch_kernels.txt.)
namespace {
constexpr int kPenaltyBlocks = 32;  // workgroups per tensor (weight matrices are small, F x hidden: the launch is one latency chain)

struct PenaltySet {
    const float* p[kMaxTensors];
    int64_t n[kMaxTensors];
    float l2[kMaxTensors];
    int count;
};

// workspace: double partial[kMaxTensors][kPenaltyBlocks], then one unsigned ticket (zero before the first use; re-armed here)
__global__ __launch_bounds__(256) void l2_penalty_kernel(PenaltySet s, double* partial, unsigned int* ticket, float* out) {
    const int k = blockIdx.y, b = blockIdx.x;
    const float* __restrict__ w = s.p[k];
    const int64_t n = s.n[k];
    double acc = 0.0, acc2 = 0.0;
    const int64_t stride = (int64_t)kPenaltyBlocks * blockDim.x;
    int64_t i = (int64_t)b * blockDim.x + threadIdx.x;
    for (; i + stride < n; i += 2 * stride) {   // two independent loads / chains per iteration
        const double x = (double)w[i], y = (double)w[i + stride];
        acc += x * x;
        acc2 += y * y;
    }
    if (i < n) {
        const double x = (double)w[i];
        acc += x * x;
    }
    acc += acc2;
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {          // fixed tree: deterministic
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    __shared__ int last;
    if (threadIdx.x == 0) {
        partial[k * kPenaltyBlocks + b] = red[0];
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x * gridDim.y - 1 ? 1 : 0;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        float total = 0.f;   // fp32, tensor by tensor, like the reference's running `loss += l2 * sum(w^2)`
        for (int q = 0; q < s.count; ++q) {
            double sq = 0.0;
            for (int j = 0; j < kPenaltyBlocks; ++j) sq += __hip_atomic_load(&partial[q * kPenaltyBlocks + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            total = mul_then_add(total, s.l2[q], (float)sq);
        }
        *out = total;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
}  // namespace

extern "C" size_t h2gcn_l2_penalty_workspace_bytes(void) { return sizeof(double) * kMaxTensors * kPenaltyBlocks + 16; }

extern "C" int h2gcn_l2_penalty_f32(int32_t n_tensors, const float* const* params, const int64_t* sizes, const float* l2,
                                    float* out_dev, void* workspace_dev, size_t workspace_bytes, void* stream_v) {
    if (n_tensors < 0 || n_tensors > kMaxTensors || (n_tensors > 0 && (!params || !sizes || !l2)) || !out_dev)
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "l2 penalty: n_tensors %d (max %d) or a NULL argument", n_tensors, kMaxTensors);
    if (!workspace_dev || workspace_bytes < h2gcn_l2_penalty_workspace_bytes() || (reinterpret_cast<uintptr_t>(workspace_dev) & 7u))
        return fail(H2GCN_ERR_INVALID_ARGUMENT, "l2 penalty: workspace of %zu bytes (8-byte aligned, zero before its first use) needed",
                    h2gcn_l2_penalty_workspace_bytes());
    hipStream_t stream = (hipStream_t)stream_v;
    if (n_tensors == 0) {
        H2GCN_HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(float), stream));
        return H2GCN_OK;
    }
    PenaltySet s{};
    s.count = n_tensors;
    for (int k = 0; k < n_tensors; ++k) {
        if (sizes[k] < 0 || (sizes[k] > 0 && !params[k]) || !(l2[k] >= 0.f))
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "l2 penalty: tensor %d has size %lld, a NULL pointer or l2 %g", k, (long long)sizes[k], (double)l2[k]);
        s.p[k] = params[k];
        s.n[k] = sizes[k];
        s.l2[k] = l2[k];
    }
    double* partial = (double*)workspace_dev;
    unsigned int* ticket = (unsigned int*)((char*)workspace_dev + sizeof(double) * kMaxTensors * kPenaltyBlocks);
    hipLaunchKernelGGL(l2_penalty_kernel, dim3(kPenaltyBlocks, (unsigned)n_tensors), dim3(256), 0, stream, s, partial, ticket, out_dev);
    H2GCN_HIP_TRY(hipGetLastError());
    return H2GCN_OK;
}
