// capi_internal.h -- error plumbing shared by the translation units of libh2gcn_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "h2gcn_hip.h"

namespace h2gcn {

// Records the message behind h2gcn_last_error() for the calling thread and returns `st` as int.
int fail(h2gcn_status st, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

}  // namespace h2gcn

#define H2GCN_HIP_TRY(expr)                                                                          \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return ::h2gcn::fail(_e == hipErrorOutOfMemory ? H2GCN_ERR_OUT_OF_MEMORY : H2GCN_ERR_HIP, \
                                 "%s failed: %s", #expr, hipGetErrorString(_e));                     \
    } while (0)
