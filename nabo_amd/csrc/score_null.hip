// score_null.hip -- permutation null for mapping scores (gfx950).  EXTENSION: the reference has a mapping
// score (Graph.get_mapping_score, nabo/_graph.py:555-697: weighted degree of a target sample's edges on every
// reference node, x score_multiplier / n_target_nodes) but no permutation test; BASELINE.json's configs[4]
// asks for "1000-permutation null-model mapping scores".  Definition used here:
//
//   pooled target cells t = 0..n_t-1 with edges (t, r, w) onto reference nodes and a group flag (1 = the
//   sample of interest, n_A cells).   S_obs[r] = mult * sum_{e: r_e = r, group[t_e]} w_e / n_A.
//   Permutation p relabels the pooled cells: key(t,p) = top key_bits of splitmix64(seed, p, t),
//   T_p = the n_A-th smallest key, label_p[t] = (key <= T_p), n_p = #labelled (= n_A unless keys tie at T_p),
//   S_p[r] = mult * sum_{e: r_e = r, label_p[t_e]} w_e / n_p.
//   Outputs per reference node: S_obs, n_ge = #{p: S_p >= S_obs}, mean and sd of S_p over p.
//
// Integer results (thresholds, labels, n_p, n_ge) are bit-exact against the oracle; edge sums run in float64 in
// CSR order like the oracle's.  HBM-bound byte work: a label BIT-matrix [n_t][ceil(P/32)] is built once
// (n_t * P hashes), the reduction then reads one 4-byte word per (edge, 32 permutations).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nabo {

__device__ __forceinline__ uint64_t null_key(uint64_t seed, uint32_t p, uint64_t t, int key_bits)
{
    uint64_t z = seed + (uint64_t)(p + 1u) * 0x9E3779B97F4A7C15ull + t * 0xD1B54A32D192ED03ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z >> (64 - key_bits);
}

// One radix-select pass: histogram of the next 8 key bits among the cells whose key starts with prefix[p].
// grid = (blocks over cells, P); hist [P][256] must be zeroed by the caller.
__global__ __launch_bounds__(256) void null_hist_kernel(int64_t n_t, uint64_t seed, int key_bits,
                                                        const uint64_t *__restrict__ prefix, int done_bits,
                                                        unsigned int *__restrict__ hist)
{
    __shared__ unsigned int h[256];
    const uint32_t p = blockIdx.y;
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t pre = prefix[p];
    const int shift = key_bits - done_bits - 8;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_t; t += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = null_key(seed, p, (uint64_t)t, key_bits);
        if (done_bits == 0 || (k >> (key_bits - done_bits)) == pre) atomicAdd(&h[(k >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[(size_t)p * 256 + threadIdx.x], h[threadIdx.x]);
}

// Label bits: word w of cell t holds permutations 32w .. 32w+31; slot P (the first bit after the
// permutations) carries the observed grouping.
__global__ __launch_bounds__(256) void null_label_kernel(int64_t n_t, int P, int W, uint64_t seed, int key_bits,
                                                         const uint64_t *__restrict__ thr,
                                                         const uint8_t *__restrict__ group,
                                                         uint32_t *__restrict__ bits)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_t * W) return;
    const int64_t t = e / W;
    const int w = (int)(e - t * W);
    uint32_t word = 0;
    for (int b = 0; b < 32; ++b) {
        const int p = w * 32 + b;
        bool on = false;
        if (p < P) on = null_key(seed, (uint32_t)p, (uint64_t)t, key_bits) <= thr[p];
        else if (p == P) on = group[t] != 0;
        if (on) word |= 1u << b;
    }
    bits[e] = word;
}

// One workgroup per reference node (CSR row): thread q accumulates permutations q, q+256, ... (and slot P,
// the observed grouping) over the row's edges in order, then the row's statistics are reduced.
template <int NACC>
__global__ __launch_bounds__(256) void null_score_kernel(const int64_t *__restrict__ row_ptr,
                                                         const int64_t *__restrict__ edge_t,
                                                         const double *__restrict__ edge_w, int P, int W,
                                                         const uint32_t *__restrict__ bits,
                                                         const int64_t *__restrict__ n_lab, int64_t n_a,
                                                         double mult, double *__restrict__ out_obs,
                                                         int64_t *__restrict__ out_nge, double *__restrict__ out_mean,
                                                         double *__restrict__ out_sd)
{
    __shared__ double s_sum[256], s_sq[256];
    __shared__ unsigned int s_ge;
    __shared__ double s_obs;
    const int64_t r = blockIdx.x;
    const int q = threadIdx.x;
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    if (q == 0) s_ge = 0;
    for (int64_t e = row_ptr[r]; e < row_ptr[r + 1]; ++e) {
        const double w = edge_w[e];
        const uint32_t *bw = bits + edge_t[e] * W;
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const int p = q + 256 * i;
            if (p <= P) acc[i] = __dadd_rn(acc[i], ((bw[p >> 5] >> (p & 31)) & 1u) ? w : 0.0);
        }
    }
    // observed score: slot P
    if (q == (P & 255)) s_obs = __ddiv_rn(__dmul_rn(mult, acc[P >> 8]), (double)n_a);
    __syncthreads();
    const double obs = s_obs;
    double sum = 0.0, sq = 0.0;
    unsigned int ge = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        const int p = q + 256 * i;
        if (p < P) {
            const double s = __ddiv_rn(__dmul_rn(mult, acc[i]), (double)n_lab[p]);
            sum += s;
            sq += s * s;
            ge += (s >= obs) ? 1u : 0u;
        }
    }
    s_sum[q] = sum;
    s_sq[q] = sq;
    if (ge) atomicAdd(&s_ge, ge);
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (q < o) { s_sum[q] += s_sum[q + o]; s_sq[q] += s_sq[q + o]; }
        __syncthreads();
    }
    if (q == 0) {
        const double mean = s_sum[0] / (double)P;
        double var = s_sq[0] / (double)P - mean * mean;
        if (var < 0.0) var = 0.0;
        out_obs[r] = obs;
        out_nge[r] = (int64_t)s_ge;
        out_mean[r] = mean;
        out_sd[r] = sqrt(var);
    }
}

hipError_t null_hist_launch(int64_t n_t, int P, uint64_t seed, int key_bits, const uint64_t *prefix, int done_bits,
                            unsigned int *hist, hipStream_t st)
{
    int64_t bx = (n_t + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(null_hist_kernel, dim3((unsigned)bx, (unsigned)P), dim3(256), 0, st, n_t, seed, key_bits, prefix,
                       done_bits, hist);
    return hipGetLastError();
}

hipError_t null_label_launch(int64_t n_t, int P, int W, uint64_t seed, int key_bits, const uint64_t *thr,
                             const uint8_t *group, uint32_t *bits, hipStream_t st)
{
    const int64_t tot = n_t * W;
    hipLaunchKernelGGL(null_label_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, n_t, P, W, seed,
                       key_bits, thr, group, bits);
    return hipGetLastError();
}

hipError_t null_score_launch(int64_t n_ref, const int64_t *row_ptr, const int64_t *edge_t, const double *edge_w, int P,
                             int W, const uint32_t *bits, const int64_t *n_lab, int64_t n_a, double mult,
                             double *out_obs, int64_t *out_nge, double *out_mean, double *out_sd, hipStream_t st)
{
    const int nacc = (P + 1 + 255) / 256;
#define NABO_NS(N)                                                                                                 \
    hipLaunchKernelGGL((null_score_kernel<N>), dim3((unsigned)n_ref), dim3(256), 0, st, row_ptr, edge_t, edge_w, P, W, \
                       bits, n_lab, n_a, mult, out_obs, out_nge, out_mean, out_sd)
    if (nacc <= 1) NABO_NS(1);
    else if (nacc <= 2) NABO_NS(2);
    else if (nacc <= 4) NABO_NS(4);
    else if (nacc <= 8) NABO_NS(8);
    else if (nacc <= 17) NABO_NS(17);
    else return hipErrorInvalidValue;
#undef NABO_NS
    return hipGetLastError();
}

}  // namespace nabo
