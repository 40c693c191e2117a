// placement_probe.hip -- micro-benchmark (not part of the product): does the random-row gather rate of a DRAM-resident table
// depend on HOW / WHERE the table was allocated?  (profiles/r05_products_x6_process_to_process.txt: the same binary's
// DRAM-resident launch moves by +-6 % between processes and by +-0.1 % within one.)  Random 256-byte row gathers (the SpMM's
// 64-column slice) out of an 8 GiB table, allocated several times per method: plain hipMalloc, hipMalloc after churning the
// allocator with odd-sized blocks, hipExtMallocWithFlags(hipDeviceMallocContiguous).
//   hipcc --offload-arch=gfx950 -O3 -o placement_probe tools/placement_probe.hip && ./placement_probe [GiB] [trials]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline unsigned long long mix(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

constexpr int LPR = 16;   // 16 lanes x 16 B = 256-byte rows, 4 rows per load instruction
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ table, long n_rows, int gathers_per_wave, float4* out) {
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, li = lane % LPR;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int t = 0; t < gathers_per_wave; t += 8 * G) {
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned long long r = mix((unsigned long long)wave * 1000003ull + t + u * G + g) % (unsigned long long)n_rows;
            x[u] = table[r * LPR + li];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
    if (g == 0) __builtin_nontemporal_store(acc.x + acc.y + acc.z + acc.w, (float*)out + wave * LPR + li);
}

double gather_rate(const float4* table, long n_rows, float4* out) {
    const long n_waves = 1 << 18; const int gpw = 512;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gather_kernel, dim3(n_waves / 4), dim3(256), 0, 0, table, n_rows, gpw, out);
    CHECK(hipEventRecord(a));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gather_kernel, dim3(n_waves / 4), dim3(256), 0, 0, table, n_rows, gpw, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return (double)n_waves * gpw * LPR * 16.0 * reps / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const int trials = argc > 2 ? atoi(argv[2]) : 4;
    const size_t bytes = (size_t)(gib * (1ull << 30)) / 256 * 256;
    const long n_rows = (long)(bytes / 256);
    float4* out; CHECK(hipMalloc(&out, (size_t)(1 << 18) * LPR * 16));
    const char* names[3] = {"hipMalloc", "hipMalloc after allocator churn", "hipExtMallocWithFlags(hipDeviceMallocContiguous)"};
    for (int method = 0; method < 3; ++method) {
        printf("%s, %.1f GiB table, 256-byte rows:", names[method], gib);
        for (int t = 0; t < trials; ++t) {
            std::vector<void*> churn;
            if (method == 1) {   // fragment the free space: odd-sized blocks, every other one freed before the table is allocated
                for (int i = 0; i < 64; ++i) { void* p; CHECK(hipMalloc(&p, (size_t)(37 + 11 * i + 5 * t) << 20)); churn.push_back(p); }
                for (size_t i = 0; i < churn.size(); i += 2) { CHECK(hipFree(churn[i])); churn[i] = nullptr; }
            }
            float4* table = nullptr;
            hipError_t e = method == 2 ? hipExtMallocWithFlags((void**)&table, bytes, hipDeviceMallocContiguous) : hipMalloc((void**)&table, bytes);
            if (e != hipSuccess) { (void)hipGetLastError(); printf(" [alloc failed: %s]", hipGetErrorString(e)); break; }
            CHECK(hipMemset(table, 1, bytes));
            printf(" %.0f", gather_rate(table, n_rows, out));
            fflush(stdout);
            CHECK(hipFree(table));
            for (void* p : churn) if (p) CHECK(hipFree(p));
        }
        printf(" GB/s\n");
    }
    return 0;
}
