// transpose.hip -- device-side construction of A^T in CSR (the operand of the adjoint launch).
//
// The adjoint of GCNLayer (reference: TF's gradient of SparseTensorDenseMatMul wrt the dense operand, reached
// from h2gcn/models/H2GCN.py:66-74) needs every hop matrix transposed, in the same canonical order the forward
// operand has (row-major, ascending column -- what `tf.sparse.reorder` would give the adjoint COO).  At products
// scale (1.2e8 nonzeros per hop) a host transposition costs tens of seconds of PCIe + CPU time; on the device it
// is one stable radix sort:
//   1. perm = stable_sort_by_key(colidx, iota)      (rocPRIM radix sort over ceil(log2 n_cols) bits;
//                                                    stability keeps ascending source-row order per column)
//   2. t_colidx[i] = row_of_edge(perm[i]),  t_vals[i] = vals[perm[i]]   (row found by binary search in rowptr)
//   3. t_rowptr[c] = lower_bound(sorted colidx, c)
// Everything is deterministic (no atomics).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>

#include <rocprim/device/device_radix_sort.hpp>

namespace h2gcn {

namespace {

__global__ void iota_kernel(uint32_t* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (uint32_t)i;
}

// largest r with rowptr[r] <= e  (rowptr ascending, rowptr[0] = 0, e < rowptr[n_rows])
__device__ __forceinline__ int64_t row_of_edge(const int64_t* __restrict__ rowptr, int64_t n_rows, int64_t e) {
    int64_t lo = 0, hi = n_rows;  // invariant: rowptr[lo] <= e < rowptr[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (rowptr[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void gather_transposed_kernel(const uint32_t* __restrict__ perm, const int64_t* __restrict__ rowptr,
                                         int64_t n_rows, const float* __restrict__ vals, int64_t nnz,
                                         int32_t* __restrict__ t_colidx, float* __restrict__ t_vals) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = perm[i];
        t_vals[i] = vals[e];
        t_colidx[i] = (int32_t)row_of_edge(rowptr, n_rows, e);
    }
}

__global__ void rowptr_from_sorted_keys_kernel(const int32_t* __restrict__ keys, int64_t nnz, int64_t n_out_rows,
                                               int64_t* __restrict__ t_rowptr) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c <= n_out_rows; c += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = nnz;  // first index with keys[idx] >= c
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys[mid] < c) lo = mid + 1; else hi = mid;
        }
        t_rowptr[c] = lo;
    }
}

struct Scratch {
    void* p = nullptr;
    ~Scratch() { if (p) (void)hipFree(p); }
};

#define T_TRY(expr)                                                                   \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            *err = std::string(#expr) + " failed: " + hipGetErrorString(_e);          \
            return _e == hipErrorOutOfMemory ? -3 : -2;                               \
        }                                                                             \
    } while (0)

}  // namespace

// Returns 0 on success, a negative h2gcn_status otherwise (message in *err).  On success the three output
// arrays are fresh hipMalloc allocations owned by the caller (t_colidx / t_vals are NULL when nnz == 0).
int transpose_csr_device(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* rowptr, const int32_t* colidx,
                         const float* vals, int64_t** t_rowptr_out, int32_t** t_colidx_out, float** t_vals_out,
                         uint32_t** perm_out_keep, hipStream_t stream, std::string* err) {
    if (perm_out_keep) *perm_out_keep = nullptr;
    *t_rowptr_out = nullptr;
    *t_colidx_out = nullptr;
    *t_vals_out = nullptr;
    if (nnz >= (1LL << 32)) {
        *err = "device transposition supports < 2^32 nonzeros per hop";
        return -1;
    }
    Scratch d_rowptr, d_col, d_val, keys_out, perm_in, perm_out, temp;
    T_TRY(hipMalloc(&d_rowptr.p, (size_t)(n_cols + 1) * sizeof(int64_t)));
    const dim3 block(256);
    auto blocks_for = [](int64_t n) { int64_t b = (n + 255) / 256; return dim3((unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b))); };
    if (nnz > 0) {
        T_TRY(hipMalloc(&d_col.p, (size_t)nnz * sizeof(int32_t)));
        T_TRY(hipMalloc(&d_val.p, (size_t)nnz * sizeof(float)));
        T_TRY(hipMalloc(&keys_out.p, (size_t)nnz * sizeof(int32_t)));
        T_TRY(hipMalloc(&perm_in.p, (size_t)nnz * sizeof(uint32_t)));
        T_TRY(hipMalloc(&perm_out.p, (size_t)nnz * sizeof(uint32_t)));
        hipLaunchKernelGGL(iota_kernel, blocks_for(nnz), block, 0, stream, (uint32_t*)perm_in.p, nnz);
        unsigned end_bit = 1;
        while (end_bit < 31 && (1LL << end_bit) < n_cols) ++end_bit;
        size_t temp_bytes = 0;
        T_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, colidx, (int32_t*)keys_out.p, (uint32_t*)perm_in.p,
                                        (uint32_t*)perm_out.p, (size_t)nnz, 0u, end_bit, stream));
        T_TRY(hipMalloc(&temp.p, temp_bytes ? temp_bytes : 16));
        T_TRY(rocprim::radix_sort_pairs(temp.p, temp_bytes, colidx, (int32_t*)keys_out.p, (uint32_t*)perm_in.p,
                                        (uint32_t*)perm_out.p, (size_t)nnz, 0u, end_bit, stream));
        hipLaunchKernelGGL(gather_transposed_kernel, blocks_for(nnz), block, 0, stream, (const uint32_t*)perm_out.p, rowptr,
                           n_rows, vals, nnz, (int32_t*)d_col.p, (float*)d_val.p);
    }
    hipLaunchKernelGGL(rowptr_from_sorted_keys_kernel, blocks_for(n_cols + 1), block, 0, stream,
                       (const int32_t*)keys_out.p, nnz, n_cols, (int64_t*)d_rowptr.p);
    T_TRY(hipGetLastError());
    T_TRY(hipStreamSynchronize(stream));
    *t_rowptr_out = (int64_t*)d_rowptr.p; d_rowptr.p = nullptr;
    *t_colidx_out = (int32_t*)d_col.p; d_col.p = nullptr;
    *t_vals_out = (float*)d_val.p; d_val.p = nullptr;
    if (perm_out_keep) {  // transposed entry i came from forward entry perm[i]: lets the caller refresh the values later
        *perm_out_keep = (uint32_t*)perm_out.p;
        perm_out.p = nullptr;
    }
    return 0;
}

// t_vals[i] = vals[perm[i]]: new values for an already transposed operand (same sparsity pattern)
__global__ void permute_values_kernel(const uint32_t* __restrict__ perm, const float* __restrict__ vals, int64_t nnz,
                                      float* __restrict__ t_vals) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x)
        t_vals[i] = vals[perm[i]];
}

void permute_values(const uint32_t* perm, const float* vals, int64_t nnz, float* t_vals, hipStream_t stream) {
    if (nnz <= 0) return;
    const int64_t b = (nnz + 255) / 256;
    hipLaunchKernelGGL(permute_values_kernel, dim3((unsigned)(b > 65536 ? 65536 : b)), dim3(256), 0, stream, perm, vals, nnz, t_vals);
}

}  // namespace h2gcn
