// spmm_short.hip -- the in-tile short-row instantiations of h2gcn::spmm_hops_kernel (SHORT = true; see spmm_kernels.hip.h)
// and their launcher.  A translation unit of its own on purpose: compiled in the same unit as the tile-walk kernels, their
// presence makes the compiler allocate 2-6 more VGPRs to THOSE kernels and pushes several of them into scratch
// (tools/kernel_resources.py; round 3 shipped that way).  Nothing else lives here.
#include <hip/hip_runtime.h>

#include "spmm_kernels.hip.h"

namespace h2gcn {

template <bool SUM, int LPR>
static void launch_one(const LaunchParams& p, bool off32, bool fb4, dim3 grid, hipStream_t stream) {
    const dim3 block(kBlock);
    if (off32 && fb4)
        hipLaunchKernelGGL((spmm_hops_kernel<4, LPR, true, SUM, true, false, true, false, 4>), grid, block, 0, stream, p);
    else if (off32)
        hipLaunchKernelGGL((spmm_hops_kernel<4, LPR, true, SUM, true, false, true, false, 8>), grid, block, 0, stream, p);
    else if (fb4)
        hipLaunchKernelGGL((spmm_hops_kernel<4, LPR, true, SUM, false, false, true, false, 4>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((spmm_hops_kernel<4, LPR, true, SUM, false, false, true, false, 8>), grid, block, 0, stream, p);
}

// slice: 64 (4 lane groups per wave) or 128 (2) feature columns; the caller checks hipGetLastError
void launch_in_tile_short(bool sum, const LaunchParams& p, int slice, bool off32, bool fb4, dim3 grid, hipStream_t stream) {
    if (sum) {
        if (slice == 128) launch_one<true, 32>(p, off32, fb4, grid, stream);
        else launch_one<true, 16>(p, off32, fb4, grid, stream);
    } else {
        if (slice == 128) launch_one<false, 32>(p, off32, fb4, grid, stream);
        else launch_one<false, 16>(p, off32, fb4, grid, stream);
    }
}

}  // namespace h2gcn
