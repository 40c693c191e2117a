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

using f4v = float __attribute__((ext_vector_type(4)));

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

// ---- `gather_probe --mixed <table_MiB>` (round 6): what does an OUTPUT stream of ~2 % of the bytes cost a saturated random-gather
// stream, and does it matter how the writes are bunched?  Every wave gathers `seg` random 256-byte rows per "segment" (8 loads in
// flight, like the SpMM's wave walk) and owes 256 B of output per segment; it pays after every `bunch` segments with ONE contiguous
// non-temporal write of bunch x 256 B (bunch = 0: never writes).  Neighbouring waves own neighbouring output regions.
template <int BUNCH>
__global__ __launch_bounds__(256) void mixed_kernel(const float4* __restrict__ table, long n_rows, int segments, int seg, float4* out) {
    constexpr int LPR = 16, G = 4;
    const int lane = threadIdx.x & 63, g = lane / LPR, li = lane % LPR;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float4* my_out = out + wave * (long)segments * LPR;          // segments x 256 B, contiguous per wave
    float4 held = make_float4(0, 0, 0, 0);                       // BUNCH <= 4: lane group g holds the piece of segment (s % BUNCH) == g
    for (int s = 0; s < segments; ++s) {
        float4 acc = make_float4(0, 0, 0, 0);
        for (int t = 0; t < seg; t += 8 * G) {
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned long long r = mix(((unsigned long long)wave * 4099ull + s) * 1000003ull + t + u * G + g) % (unsigned long long)n_rows;
                x[u] = table[r * LPR + li];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
        }
        // fold the 4 lane groups (every group ends up with the segment's piece)
        acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
        acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
        if constexpr (BUNCH == 1) {
            if (g == 0) __builtin_nontemporal_store(*reinterpret_cast<f4v*>(&acc), reinterpret_cast<f4v*>(my_out + (long)s * LPR) + li);
        } else if constexpr (BUNCH == 4) {
            if (g == (s & 3)) held = acc;
            if ((s & 3) == 3)      // 64 lanes x 16 B = 1 KiB contiguous: the pieces of segments s-3 .. s
                __builtin_nontemporal_store(*reinterpret_cast<f4v*>(&held), reinterpret_cast<f4v*>(my_out + (long)(s - 3) * LPR) + lane);
        } else if constexpr (BUNCH == 16) {
            __shared__ float4 stage[4][16][16];
            if (g == 0) stage[threadIdx.x >> 6][s & 15][li] = acc;
            if ((s & 15) == 15) {  // 4 x 1 KiB = 4 KiB contiguous
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 v = stage[threadIdx.x >> 6][k * 4 + g][li];
                    __builtin_nontemporal_store(*reinterpret_cast<const f4v*>(&v), reinterpret_cast<f4v*>(my_out + (long)(s - 15 + k * 4) * LPR) + lane);
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            if (acc.x == 12345.678f) my_out[li] = acc;   // never true; keeps the arithmetic alive
        }
    }
}

template <int BUNCH>
double run_mixed(const float4* table, long n_rows, long n_waves, int segments, int seg, float4* out) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(mixed_kernel<BUNCH>, dim3(n_waves / 4), dim3(256), 0, 0, table, n_rows, segments, seg, out);
    CHECK(hipEventRecord(a));
    const int reps = 3;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(mixed_kernel<BUNCH>, dim3(n_waves / 4), dim3(256), 0, 0, table, n_rows, segments, seg, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return (double)n_waves * segments * seg * 256.0 * reps / (ms * 1e-3) / 1e9;    // GATHERED bytes per second
}

int mixed(size_t table_mib) {
    const size_t bytes = table_mib << 20;
    float4* table; CHECK(hipMalloc(&table, bytes)); CHECK(hipMemset(table, 0, bytes));
    const long n_waves = 1 << 17; const int segments = 256, seg = 64;       // 64 gathers of 256 B per 256 B written: 1.6 % writes
    float4* out; CHECK(hipMalloc(&out, (size_t)n_waves * segments * 256));  // 8 GiB of output
    printf("# random 256-B gathers out of %zu MiB, %ld waves x %d segments x %d gathers; 256 B owed per segment (%.1f %% of the bytes)\n",
           table_mib, n_waves, segments, seg, 100.0 / seg);
    for (int rep = 0; rep < 2; ++rep) {
        printf("round %d: no writes %.0f GB/s | 256 B per segment %.0f | 1 KiB per 4 segments %.0f | 4 KiB per 16 segments %.0f\n", rep,
               run_mixed<0>(table, bytes / 256, n_waves, segments, seg, out), run_mixed<1>(table, bytes / 256, n_waves, segments, seg, out),
               run_mixed<4>(table, bytes / 256, n_waves, segments, seg, out), run_mixed<16>(table, bytes / 256, n_waves, segments, seg, out));
    }
    CHECK(hipFree(table)); CHECK(hipFree(out));
    return 0;
}

// ---- `gather_probe --short <table_MiB>` (round 6): the ceiling of the SHORT-ROW regimes (lowdeg: ~4 neighbours per segment, hbm16m: ~8):
// every LANE GROUP owns a stream of output pieces; per piece it gathers K random rows of LPR*16 bytes (8 loads in flight per lane,
// i.e. 8/K pieces per batch), sums them and writes one piece of the same width (non-temporal).  Bytes counted: gathered + written.
template <int LPR, int K, int U = 8>
__global__ __launch_bounds__(256) void short_kernel(const float4* __restrict__ table, long n_rows, int batches, float4* out) {
    constexpr int G = 64 / LPR, P = U / K;       // lane groups per wave, pieces per batch and group (U loads in flight per lane)
    const int lane = threadIdx.x & 63, g = lane / LPR, li = lane % LPR;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float4* my_out = out + ((wave * G + g) * (long)batches * P) * LPR;      // this group's pieces, contiguous
    for (int b = 0; b < batches; ++b) {
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long r = mix(((unsigned long long)(wave * G + g) * 4099ull + b) * 1000003ull + u) % (unsigned long long)n_rows;
            x[u] = table[r * LPR + li];
        }
#pragma unroll
        for (int q = 0; q < P; ++q) {
            float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < K; ++u) { acc.x += x[q * K + u].x; acc.y += x[q * K + u].y; acc.z += x[q * K + u].z; acc.w += x[q * K + u].w; }
            __builtin_nontemporal_store(*reinterpret_cast<f4v*>(&acc), reinterpret_cast<f4v*>(my_out + ((long)b * P + q) * LPR) + li);
        }
    }
}

template <int LPR, int K, int U = 8>
double run_short(const float4* table, long n_rows, long n_waves, int batches, float4* out) {
    batches = batches * 8 / U;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((short_kernel<LPR, K, U>), dim3(n_waves / 4), dim3(256), 0, 0, table, n_rows, batches, out);
    CHECK(hipEventRecord(a));
    const int reps = 3;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((short_kernel<LPR, K, U>), dim3(n_waves / 4), dim3(256), 0, 0, table, n_rows, batches, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double gathered = (double)n_waves * 64 * (double)U * batches * 16.0, written = gathered / K;
    return (gathered + written) * reps / (ms * 1e-3) / 1e9;
}

int short_rows(size_t table_mib) {
    const size_t bytes = table_mib << 20;
    float4* table; CHECK(hipMalloc(&table, bytes)); CHECK(hipMemset(table, 0, bytes));
    const long n_waves = 1 << 17; const int batches = 64;
    float4* out; CHECK(hipMalloc(&out, (size_t)n_waves * 64 * 8 * batches * 16 / 4));   // K = 4: a quarter of the gathered bytes
    printf("# short rows: every lane group gathers K random rows per output piece out of %zu MiB; gathered + written GB/s\n", table_mib);
    for (int rep = 0; rep < 2; ++rep)
        printf("round %d: 512-B rows K=4 (lowdeg d=128) %.0f GB/s | 512-B rows K=8 (hbm16m d=128) %.0f | 256-B rows K=4 (lowdeg d=64) %.0f | 256-B rows K=8 %.0f\n", rep,
               run_short<32, 4>(table, bytes / 512, n_waves, batches, out), run_short<32, 8>(table, bytes / 512, n_waves, batches, out),
               run_short<16, 4>(table, bytes / 256, n_waves, batches, out), run_short<16, 8>(table, bytes / 256, n_waves, batches, out));
    for (int rep = 0; rep < 2; ++rep)
        printf("round %d, 16 loads in flight: 512-B rows K=4 %.0f GB/s | 512-B rows K=8 %.0f | 256-B rows K=4 %.0f | 4 loads in flight: 512-B rows K=4 %.0f\n", rep,
               run_short<32, 4, 16>(table, bytes / 512, n_waves, batches, out), run_short<32, 8, 16>(table, bytes / 512, n_waves, batches, out),
               run_short<16, 4, 16>(table, bytes / 256, n_waves, batches, out), run_short<32, 4, 4>(table, bytes / 512, n_waves, batches, out));
    CHECK(hipFree(table)); CHECK(hipFree(out));
    return 0;
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
    if (argc == 3 && !strcmp(argv[1], "--mixed")) return mixed((size_t)atol(argv[2]));
    if (argc == 3 && !strcmp(argv[1], "--short")) return short_rows((size_t)atol(argv[2]));
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
