// order.hip -- locality order of the cells the Euclidean filter streams (l2q_topk.hip): pack-time permutation of the
// reference and target rows + the tile every wave starts its stream at.
//
// Why.  A streamed top-L makes L ln(n / L) list updates per row when the references arrive in random order (246 at
// n = 1M, L = 23; 111 on a 125k-reference shard with L = 12) -- the whole hit path of the filter, a third of a shard's
// kernel time.  Nothing in the algorithm fixes the order of the stream: if a row meets its
// neighbourhood FIRST, its threshold is near its final value after a few tiles and the rest of the stream produces
// almost no hits.  The inputs are PCA embeddings (nabo/_dataset.py:985-1033): the leading components carry the most
// variance, cells of one cluster share the signs of their leading centred components.  So
//   key(cell)  = the sign bits of its first NB centred components, component 0 most significant (NB = min(g, 16));
//   references = sorted by key (stable LSD radix sort: rocPRIM, a library call off the hot path) and packed in that
//                order; targets likewise, so the 128 rows of a wave share a neighbourhood;
//   wave home  = the tile holding the first reference whose key is >= the key of the wave's middle row; the wave
//                visits a few tiles around it first, then the stream proper from its first tile (l2q_topk.hip: tmap).
// Simulated on the bench's synthetic embeddings (n = 250k, L = 23, 128 rows per home): 238 updates per row in caller
// order, 149 streaming cyclically from home, 155 with 8 home tiles and then the key-ordered stream from its start.
// It is an ORDER only: every reference is still visited, results are the same bits (the refine step maps
// positions back to caller indices before it sorts by (distance, index)).
//
// MEASURED (round 3, 1M x 1M x 50, k = 15, one MI355X, profiles/r3_order_experiment.txt) -- and therefore OFF by default
// (NABO_L2Q_ORDER=7 switches it on; parity-tested either way):
//   list counters (-DNABO_LISTS_PROF): episodes 88.4M -> 25.2M, staged records 193M -> 110M, list entries written
//   244M -> 172M, drains 3.0M -> 1.7M -- as simulated; kernel 239 ms -> 287 ms.  Piece by piece on one box: references in
//   key order alone 329 ms (a row approaches its own cluster through ever closer cells: an adversarial order), targets in
//   key order alone 258 ms with the references untouched -- the hit counts of a random stream do not depend on the
//   target order at all, so those 8 % are not list work: rows of a wave that resemble each other make the 16 columns of
//   every MFMA switch in step, and the part, which is power-limited under this kernel, holds a lower
//   clock on such operands.  Both orders + the home pre-pass: 287 ms; a cyclic start per wave instead of the pre-pass:
//   284 ms (every wave then streams its own window of the references: no sharing in L2).  What the lists save, the
//   clock takes back twice over.  Caller order -- cells in the random order of a count matrix -- is the fast order.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>

namespace nabo {

__global__ __launch_bounds__(256) void loc_key_kernel(const double *__restrict__ V, int64_t n, int g,
                                                      const double *__restrict__ centre, int nb,
                                                      uint32_t *__restrict__ keys, uint32_t *__restrict__ pos)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t k = 0;
    for (int j = 0; j < nb; ++j) k = (k << 1) | (V[i * g + j] > centre[j] ? 1u : 0u);      // (NaN compares false: bit 0)
    keys[i] = k;
    pos[i] = (uint32_t)i;
}

// start[w] = tile of the first sorted reference whose key is >= the key of the middle row of wave w's rows
// [w * rows_per_wave, (w + 1) * rows_per_wave) of the sorted targets
__global__ __launch_bounds__(256) void wave_start_kernel(const uint32_t *__restrict__ tkeys, int64_t m, int rows_per_wave,
                                                         const uint32_t *__restrict__ rkeys, int64_t n, int64_t n_waves,
                                                         int32_t *__restrict__ start)
{
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_waves) return;
    int64_t r = w * rows_per_wave + rows_per_wave / 2;
    if (r >= m) r = m - 1;
    const uint32_t key = tkeys[r];
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (rkeys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    if (lo >= n) lo = n - 1;
    start[w] = (int32_t)(lo / 32);
}

int loc_key_bits(int g) { return g < 16 ? g : 16; }

hipError_t loc_sort_temp_bytes(int64_t n, int nb, size_t *bytes)
{
    *bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                     (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0u, (unsigned)nb,
                                     (hipStream_t) nullptr);
}

// keys_a / pos_a: scratch [n]; keys_sorted / perm: outputs [n] (perm[i] = caller row at sorted position i)
hipError_t loc_order_launch(const double *V, int64_t n, int g, const double *centre, uint32_t *keys_a, uint32_t *pos_a,
                            uint32_t *keys_sorted, uint32_t *perm, void *temp, size_t temp_bytes, hipStream_t st)
{
    const int nb = loc_key_bits(g);
    hipLaunchKernelGGL(loc_key_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, V, n, g, centre, nb, keys_a, pos_a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t *)keys_a, keys_sorted, (const uint32_t *)pos_a, perm,
                                     (size_t)n, 0u, (unsigned)nb, st);
}

hipError_t wave_start_launch(const uint32_t *tkeys, int64_t m, int rows_per_wave, const uint32_t *rkeys, int64_t n,
                             int64_t n_waves, int32_t *start, hipStream_t st)
{
    hipLaunchKernelGGL(wave_start_kernel, dim3((unsigned)((n_waves + 255) / 256)), dim3(256), 0, st, tkeys, m, rows_per_wave,
                       rkeys, n, n_waves, start);
    return hipGetLastError();
}

}  // namespace nabo
