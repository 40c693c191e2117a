// gather_probe.hip -- micro-benchmark (not part of the product): bandwidth of random ROW gathers on MI355X as a
// function of table size (L2 / Infinity Cache / HBM resident) and row width.  It bounds what the SpMM's gather
// stream can reach: each wave reads `rows_per_wave` pseudo-random rows of `row_bytes` (16 B per lane, LPR lanes
// per row, G = 64/LPR rows per load instruction, 8 loads in flight), sums them, writes one row.
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe tools/gather_probe.hip && ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline unsigned long long mix(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

template <int LPR>
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

template <int LPR>
double run(const float4* table, long n_rows, long n_waves, int gpw, float4* out) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gather_kernel<LPR>, dim3(n_waves / 4), dim3(256), 0, 0, table, n_rows, gpw, out);
    CHECK(hipEventRecord(a));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gather_kernel<LPR>, dim3(n_waves / 4), dim3(256), 0, 0, table, n_rows, gpw, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return (double)n_waves * gpw * LPR * 16.0 * reps / (ms * 1e-3) / 1e9;
}

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const float __attribute__((ext_vector_type(4)))*>(src) + i),
                                    reinterpret_cast<float __attribute__((ext_vector_type(4)))*>(dst) + i);
}

using f4v = float __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read_kernel(const f4v* __restrict__ src, size_t n, float* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f4v a = {0, 0, 0, 0}, b = a, c = a, e = a;
    for (; i + 3 * stride < n; i += 4 * stride) {
        a += __builtin_nontemporal_load(src + i);
        b += __builtin_nontemporal_load(src + i + stride);
        c += __builtin_nontemporal_load(src + i + 2 * stride);
        e += __builtin_nontemporal_load(src + i + 3 * stride);
    }
    for (; i < n; i += stride) a += __builtin_nontemporal_load(src + i);
    a += b + c + e;
    const float t = a.x + a.y + a.z + a.w;
    if (t == 12345.678f) *sink = t;  // never true for the memset pattern; keeps the loads alive
}

// read-only stream of `bytes`: what HBM delivers to a pure reader
double run_read(size_t bytes) {
    f4v* a; float* sink; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&sink, 4)); CHECK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t n = bytes / 16;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(read_kernel, dim3(256 * 16), dim3(256), 0, 0, a, n, sink);
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(read_kernel, dim3(256 * 16), dim3(256), 0, 0, a, n, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipFree(a)); CHECK(hipFree(sink));
    return (double)bytes * reps / (ms * 1e-3) / 1e9;
}

// plain stream copy of `bytes` (read + write counted): the achievable HBM rate of this box
double run_copy(size_t bytes) {
    float4 *a, *b; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t n = bytes / 16;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(copy_kernel, dim3(256 * 32), dim3(256), 0, 0, a, b, n);
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copy_kernel, dim3(256 * 32), dim3(256), 0, 0, a, b, n);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipFree(a)); CHECK(hipFree(b));
    return 2.0 * bytes * reps / (ms * 1e-3) / 1e9;
}

// `gather_probe --point <table_MiB> <row_bytes>`: one gather point + the copy ceiling, as one JSON line (bench.py)
int point(size_t table_mib, int row_bytes) {
    const size_t bytes = table_mib << 20;
    float4* table; CHECK(hipMalloc(&table, bytes)); CHECK(hipMemset(table, 0, bytes));
    const long n_waves = 1 << 18; const int gpw = 512;
    float4* out; CHECK(hipMalloc(&out, n_waves * 64 * sizeof(float4)));
    double g = 0;
    switch (row_bytes) {
        case 64: g = run<4>(table, bytes / 64, n_waves, gpw, out); break;
        case 128: g = run<8>(table, bytes / 128, n_waves, gpw, out); break;
        case 256: g = run<16>(table, bytes / 256, n_waves, gpw, out); break;
        case 512: g = run<32>(table, bytes / 512, n_waves, gpw, out); break;
        default: g = run<64>(table, bytes / 1024, n_waves, gpw, out); row_bytes = 1024; break;
    }
    CHECK(hipFree(table)); CHECK(hipFree(out));
    const double c = run_copy(2ull << 30);
    const double r = run_read(4ull << 30);
    // what the box says about its memory system ("confirm on the box"): hipDeviceProp_t's memory clock and bus width
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\"memory_clock_khz\": %d, \"memory_bus_width_bits\": %d, \"device_name\": \"%s\", \"gcn_arch\": \"%s\", \"total_global_mem_GiB\": %.1f, ",
           prop.memoryClockRate, prop.memoryBusWidth, prop.name, prop.gcnArchName, (double)prop.totalGlobalMem / (double)(1ull << 30));
    printf("\"table_MiB\": %zu, \"row_bytes\": %d, \"gather_GBps\": %.1f, \"copy_GBps\": %.1f, \"stream_read_GBps\": %.1f, "
           "\"what\": \"tools/gather_probe: random row gathers out of a table of this size (8 x 16-B loads in flight per lane); "
           "copy = nontemporal stream copy of 2 GiB, read+write bytes; stream_read = read-only pass over 4 GiB\"}\n", table_mib, row_bytes, g, c, r);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 4 && !strcmp(argv[1], "--point")) return point((size_t)atol(argv[2]), atoi(argv[3]));
    const size_t max_bytes = 8ull << 30;
    float4* table; CHECK(hipMalloc(&table, max_bytes)); CHECK(hipMemset(table, 0, max_bytes));
    const long n_waves = 1 << 18; const int gpw = 512;
    float4* out; CHECK(hipMalloc(&out, n_waves * 64 * sizeof(float4)));
    printf("%12s %10s %12s\n", "table_MiB", "row_bytes", "gather_GB/s");
    const size_t sizes[] = {16ull << 20, 128ull << 20, 512ull << 20, 1229ull << 20, 4096ull << 20, 8192ull << 20};
    for (size_t sz : sizes) {
        printf("%12zu %10d %12.0f\n", sz >> 20, 128, run<8>(table, sz / 128, n_waves, gpw, out));
        printf("%12zu %10d %12.0f\n", sz >> 20, 256, run<16>(table, sz / 256, n_waves, gpw, out));
        printf("%12zu %10d %12.0f\n", sz >> 20, 512, run<32>(table, sz / 512, n_waves, gpw, out));
        printf("%12zu %10d %12.0f\n", sz >> 20, 1024, run<64>(table, sz / 1024, n_waves, gpw, out));
    }
    return 0;
}
