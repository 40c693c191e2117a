// unaligned_vec_probe.hip -- do 16-byte global loads / non-temporal stores work, and how fast are they, at addresses that
// are only 4-byte aligned (gfx950)?  Decides whether odd feature widths can be gathered in place by the float4 kernels.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/unaligned_vec_probe tools/unaligned_vec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ void k(const float* __restrict__ x, float* __restrict__ y, long n4, int off) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f4u v = *reinterpret_cast<const f4u*>(x + 4 * i + off);
        f4u o = v * 2.f;
        __builtin_nontemporal_store(o, reinterpret_cast<f4u*>(y + 4 * i + off));
    }
}
int main() {
    const long n4 = 1L << 26;  // 1 GiB per buffer
    float *x, *y;
    hipMalloc(&x, (n4 * 4 + 8) * 4);
    hipMalloc(&y, (n4 * 4 + 8) * 4);
    std::vector<float> h(4096 + 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    hipMemset(x, 0, (n4 * 4 + 8) * 4);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int off = 0; off < 4; ++off) {
        hipMemset(y, 0, (n4 * 4 + 8) * 4);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        k<<<4096, 256>>>(x, y, n4, off);
        hipEventRecord(a);
        for (int r = 0; r < 5; ++r) k<<<4096, 256>>>(x, y, n4, off);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<float> o(4096 + 8);
        hipMemcpy(o.data(), y, o.size() * 4, hipMemcpyDeviceToHost);
        for (int i = off; i < 4096 + off; ++i) bad += (o[i] != 2.f * h[i]);
        for (int i = 0; i < off; ++i) bad += (o[i] != 0.f);
        printf("offset %d floats: %s, copy %.1f GB/s (read+write)\n", off, bad ? "WRONG" : "ok", 5.0 * 2 * n4 * 16 / ms / 1e6);
    }
    return bad != 0;
}
