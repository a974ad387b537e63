// Target->reference edge list (COO, any order) -> CSR by reference node, on the device.
//
// Feeds null_score_kernel (score_null.hip), which sums a reference node's edges in float64 IN ROW ORDER: the CSR
// must therefore keep the caller's edge order inside a row (a STABLE sort by reference node), or the sums -- and
// with them the comparisons `permuted score >= observed score` -- would depend on the order the hardware happened
// to place the edges in.  The sort is rocPRIM's LSD radix sort of (reference node, edge position) pairs, which is
// stable; only the bits a node id needs are sorted.  Everything else (keys, row pointers, gather) is HBM streaming:
// 250M edges (5M cells x k=50, BASELINE.json configs[4]) take a fraction of a second here against ~25 s for the
// same argsort on one host core.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>

namespace nabo {

__global__ __launch_bounds__(256) void csr_keys_kernel(const int64_t *__restrict__ edge_r, int64_t E, int64_t n_ref,
                                                       uint32_t *__restrict__ keys, uint32_t *__restrict__ pos,
                                                       unsigned int *__restrict__ flag)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    int64_t r = edge_r[e];
    if (r < 0 || r >= n_ref) {
        atomicOr(flag, 1u);
        r = 0;
    }
    keys[e] = (uint32_t)r;
    pos[e] = (uint32_t)e;
}

// row_ptr[r] = number of edges whose reference node is < r = first sorted position with key >= r
__global__ __launch_bounds__(256) void csr_rowptr_kernel(const uint32_t *__restrict__ keys, int64_t E, int64_t n_ref,
                                                         int64_t *__restrict__ row_ptr)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > E) return;
    const int64_t prev = i == 0 ? -1 : (int64_t)keys[i - 1];
    const int64_t cur = i == E ? n_ref : (int64_t)keys[i];
    for (int64_t r = prev + 1; r <= cur; ++r) row_ptr[r] = i;
}

__global__ __launch_bounds__(256) void csr_gather_kernel(const uint32_t *__restrict__ pos, const int64_t *__restrict__ edge_t,
                                                         const double *__restrict__ edge_w, int64_t E, int64_t n_t,
                                                         int64_t *__restrict__ out_t, double *__restrict__ out_w,
                                                         unsigned int *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= E) return;
    const uint32_t e = pos[i];
    int64_t t = edge_t[e];
    if (t < 0 || t >= n_t) {
        atomicOr(flag, 2u);
        t = 0;
    }
    out_t[i] = t;
    out_w[i] = edge_w[e];
}

static int key_bits_for(int64_t n_ref)
{
    int b = 1;
    while (b < 32 && ((int64_t)1 << b) < n_ref) ++b;
    return b;
}

hipError_t csr_sort_temp_bytes(int64_t E, int64_t n_ref, size_t *bytes)
{
    *bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                     (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)E, 0u,
                                     (unsigned)key_bits_for(n_ref), (hipStream_t) nullptr);
}

// keys_a/pos_a, keys_b/pos_b: [E] u32 scratch; flag: 1 = a reference id out of range, 2 = a target id out of range.
hipError_t csr_build_launch(const int64_t *edge_r, const int64_t *edge_t, const double *edge_w, int64_t E, int64_t n_ref,
                            int64_t n_t, uint32_t *keys_a, uint32_t *pos_a, uint32_t *keys_b, uint32_t *pos_b, void *temp,
                            size_t temp_bytes, int64_t *row_ptr, int64_t *out_t, double *out_w, unsigned int *flag,
                            hipStream_t st)
{
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(unsigned int), st);
    if (e != hipSuccess) return e;
    const unsigned gE = (unsigned)((E + 255) / 256), gE1 = (unsigned)((E + 256) / 256);
    if (E > 0) {
        hipLaunchKernelGGL(csr_keys_kernel, dim3(gE), dim3(256), 0, st, edge_r, E, n_ref, keys_a, pos_a, flag);
        e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t *)keys_a, keys_b, (const uint32_t *)pos_a, pos_b,
                                      (size_t)E, 0u, (unsigned)key_bits_for(n_ref), st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(csr_rowptr_kernel, dim3(gE1), dim3(256), 0, st, keys_b, E, n_ref, row_ptr);
    if (E > 0)
        hipLaunchKernelGGL(csr_gather_kernel, dim3(gE), dim3(256), 0, st, pos_b, edge_t, edge_w, E, n_t, out_t, out_w, flag);
    return hipGetLastError();
}

}  // namespace nabo
