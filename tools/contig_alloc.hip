// contig_alloc.hip -- experiment (not product): a torch pluggable allocator that takes its blocks from
// hipExtMallocWithFlags(hipDeviceMallocContiguous) (physically contiguous VRAM) or, H2GCN_ALLOC_PLAIN=1, from plain hipMalloc.
//   hipcc -shared -fPIC --offload-arch=gfx950 -o build/contig_alloc.so tools/contig_alloc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
extern "C" void* contig_alloc(size_t size, int device, hipStream_t stream) {
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, size, hipDeviceMallocContiguous);
    if (e != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "[contig_alloc] %zu bytes: %s -- plain hipMalloc\n", size, hipGetErrorString(e)); if (hipMalloc(&p, size) != hipSuccess) return nullptr; }
    return p;
}
extern "C" void contig_free(void* ptr, size_t size, int device, hipStream_t stream) { (void)hipFree(ptr); }
