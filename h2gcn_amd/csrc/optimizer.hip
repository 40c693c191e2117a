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
};

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
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
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
            longest = std::max(longest, sizes[q]);
        }
        if (longest == 0) continue;
        const unsigned gx = (unsigned)std::min<int64_t>((longest + 255) / 256, 2048);
        hipLaunchKernelGGL(adam_keras_kernel, dim3(gx, (unsigned)count), dim3(256), 0, (hipStream_t)stream_v, s, lr, beta1, beta2, eps, step_dev, step);
        H2GCN_HIP_TRY(hipGetLastError());
    }
    return H2GCN_OK;
}
