// canberra.hip -- exact modified-Canberra distance + fused top-L selection, list merge and
// SNN counting kernels (gfx950).
//
// _mod_canberra_dist (nabo/_mapping.py:29-45) is element-wise abs / compare / divide -- not a
// contraction, so there is nothing for MFMA here: the kernel is float64 VALU-bound (one IEEE
// divide per in-window dimension).  It evaluates the reference's exact float64 expression
// for every (target, reference) pair and keeps, per target, the L smallest under the
// canonical (distance, index) order -- mod-Canberra produces exact ties routinely (every
// out-of-window dimension adds exactly 1), so the selection is tie-exact by construction:
// references are visited in ascending index order and a later equal distance never displaces
// an earlier one.
//
// Layout: lane = reference (64 per chunk, coalesced from the transposed pack Yt[chunk][k][64]),
// target row = wave-uniform; each wave owns T targets and their candidate lists in LDS.
#include "knn_common.h"

namespace nabo {

__global__ void transpose_ref_kernel(const double *__restrict__ Y, int64_t n, int g, double *__restrict__ Yt)
{
    const int64_t chunk = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t j = chunk * 64 + lane;
    for (int k = threadIdx.x >> 6; k < g; k += (blockDim.x >> 6))
        Yt[(chunk * g + k) * 64 + lane] = j < n ? Y[j * g + k] : 0.0;
}

template <int EPL>
__device__ __forceinline__ double compact_row_f64(double *kb, uint32_t *ib, int count,
                                                  double (&key)[EPL], uint32_t (&val)[EPL])
{
    constexpr int L = 32 * EPL;
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const int e = r * 64 + lane;
        key[r] = __builtin_inf();
        val[r] = 0xFFFFFFFFu;
        if (e < count) { key[r] = kb[e]; val[r] = ib[e]; }
    }
    wave_bitonic_sort<EPL, double>(key, val);
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const int e = r * 64 + lane;
        if (e < L) { kb[e] = key[r]; ib[e] = val[r]; }
    }
    return __shfl(key[(L - 1) >> 6], (L - 1) & 63, 64);
}

constexpr int CANB_T = 16;   // targets per wave

// grid.x = ceil(m / (4*T)), grid.y = S splits of `chunks_per_split` 64-reference chunks.
template <int EPL>
__global__ __launch_bounds__(256) void canberra_topk_kernel(const double *__restrict__ X, int64_t m,
                                                            const double *__restrict__ Yt, int64_t n, int g,
                                                            double f, const uint8_t *__restrict__ mask,
                                                            int64_t n_chunks, int64_t chunks_per_split,
                                                            double *__restrict__ cand_d,
                                                            uint32_t *__restrict__ cand_i)
{
    constexpr int CAP = 64 * EPL, L = 32 * EPL, T = CANB_T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // per wave: keys [T][CAP] f64 | idx [T][CAP] u32 | tau [T] f64 | cnt [T] i32
    constexpr size_t WAVE_BYTES = (size_t)T * CAP * 12 + T * 8 + T * 4 + 8;
    unsigned char *wb = smem_raw + (size_t)wave * ((WAVE_BYTES + 15) & ~(size_t)15);
    double *keys = reinterpret_cast<double *>(wb);
    double *tau = keys + T * CAP;
    uint32_t *idxs = reinterpret_cast<uint32_t *>(tau + T);
    int *cnt = reinterpret_cast<int *>(idxs + T * CAP);

    const int S = gridDim.y;
    const int split = blockIdx.y;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * T;
    if (lane < T) { tau[lane] = __builtin_inf(); cnt[lane] = 0; }
    // (wave-private LDS: DS operations of one wave execute in order, no barrier needed)

    const int64_t c_begin = split * chunks_per_split;
    int64_t c_end = c_begin + chunks_per_split;
    if (c_end > n_chunks) c_end = n_chunks;
    for (int64_t chunk = c_begin; chunk < c_end; ++chunk) {
        const int64_t j = chunk * 64 + lane;
        const bool valid = (j < n) && !(mask && mask[j]);
        const double *yt = Yt + chunk * g * 64 + lane;
        for (int t = 0; t < T; ++t) {
            const int64_t row = row0 + t;
            if (row >= m) break;
            const double *x = X + row * g;
            const double tau_t = tau[t];
            // Pass 1 (3 float64 ops per dimension): count the in-window dimensions with the reference's
            // exact test.  Every out-of-window dimension adds exactly 1 and every in-window term is >= 0,
            // so dist >= g - n_in rigorously (adding ones is exact, rounding is monotone): a pair whose
            // bound already reaches the row threshold cannot enter the list and skips the divisions.
            int n_in = 0;
            for (int k = 0; k < g; ++k) {
                const double xv = x[k], yv = yt[(int64_t)k * 64];
                n_in += (fabs(__dsub_rn(xv, yv)) < __dmul_rn(f, fabs(xv))) ? 1 : 0;
            }
            const bool maybe = valid && ((double)(g - n_in) < tau_t);
            double dist = __builtin_inf();
            if (__builtin_amdgcn_ballot_w64(maybe) != 0) {
                // Pass 2: the reference expression, term by term in its order (nabo/_mapping.py:36-44)
                dist = 0.0;
                for (int k = 0; k < g; ++k) {
                    const double xv = x[k], yv = yt[(int64_t)k * 64];
                    const double absx = fabs(xv);
                    const double num = fabs(__dsub_rn(xv, yv));
                    if (num < __dmul_rn(f, absx)) {
                        const double den = __dadd_rn(__dadd_rn(absx, fabs(yv)), 0.01);
                        dist = __dadd_rn(dist, __ddiv_rn(num, den));
                    } else {
                        dist = __dadd_rn(dist, 1.0);
                    }
                }
            }
            bool pend = maybe && (dist < tau_t);
            uint64_t pm = __builtin_amdgcn_ballot_w64(pend);
            while (pm != 0) {
                const int c = cnt[t];
                const int room = CAP - c;
                if (room == 0) {
                    double key[EPL];
                    uint32_t val[EPL];
                    const double nt = compact_row_f64<EPL>(keys + t * CAP, idxs + t * CAP, c, key, val);
                    if (lane == 0) { tau[t] = nt; cnt[t] = L; }
                    pend = pend && (dist < nt);
                } else {
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                    const bool take = pend && rank < room;
                    if (take) {
                        keys[t * CAP + c + rank] = dist;
                        idxs[t * CAP + c + rank] = (uint32_t)j;
                    }
                    const int np = __popcll(pm);
                    if (lane == 0) cnt[t] = c + (np < room ? np : room);
                    pend = pend && !take;
                }
                pm = __builtin_amdgcn_ballot_w64(pend);
            }
        }
    }
    // flush: sorted L smallest per target
    for (int t = 0; t < T; ++t) {
        const int64_t row = row0 + t;
        if (row >= m) break;
        double key[EPL];
        uint32_t val[EPL];
        compact_row_f64<EPL>(keys + t * CAP, idxs + t * CAP, cnt[t], key, val);
        const int64_t o = (row * S + split) * (int64_t)L;
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const int e = r * 64 + lane;
            if (e < L) { cand_d[o + e] = key[r]; cand_i[o + e] = val[r]; }
        }
    }
}

// ---- list merge: P exact (distance, index) candidates per row -> first k after the drop ---
// IDX64: indices are int64 (global, entries < 0 absent) -- the multi-GPU shard merge;
// otherwise uint32 local indices (0xFFFFFFFF absent) + base.
template <int NCL, bool IDX64>
__global__ __launch_bounds__(256) void merge_kernel(const double *__restrict__ pd, const void *__restrict__ pi_,
                                                    int64_t m, int P, int64_t part_stride /*0: [m][P] contiguous*/,
                                                    int n_parts, int kp, int k, int drop, int64_t base,
                                                    int64_t *__restrict__ out_idx, double *__restrict__ out_dist,
                                                    int *__restrict__ n_found)
{
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    double key[NCL];
    uint32_t val[NCL];
#pragma unroll
    for (int r = 0; r < NCL; ++r) {
        const int e = r * 64 + lane;
        key[r] = __builtin_inf();
        val[r] = 0xFFFFFFFFu;
        if (e < P) {
            int64_t off;
            if (IDX64) {        // parts laid out [n_parts][m][kp]
                const int part = e / kp, s = e - part * kp;
                off = (int64_t)part * part_stride + row * kp + s;
                const int64_t j = reinterpret_cast<const int64_t *>(pi_)[off];
                if (j >= 0) { val[r] = (uint32_t)j; key[r] = pd[off]; }
            } else {
                off = row * P + e;
                const uint32_t j = reinterpret_cast<const uint32_t *>(pi_)[off];
                if (j != 0xFFFFFFFFu) { val[r] = j; key[r] = pd[off]; }
            }
        }
    }
    wave_bitonic_sort<NCL, double>(key, val);
    int nreal = 0;
#pragma unroll
    for (int r = 0; r < NCL; ++r) nreal += __popcll(__builtin_amdgcn_ballot_w64(val[r] != 0xFFFFFFFFu));
#pragma unroll
    for (int r = 0; r < NCL; ++r) {
        const int e = r * 64 + lane;
        const int o = e - drop;
        if (o >= 0 && o < k) {
            if (e < nreal) { out_idx[row * k + o] = base + (int64_t)val[r]; out_dist[row * k + o] = key[r]; }
            else { out_idx[row * k + o] = -1; out_dist[row * k + o] = __builtin_nan(""); }
        }
    }
    if (n_found && lane == 0) n_found[row] = nreal;
}

// SNN shared-neighbour counts (nabo/_mapping.py:190-193): thread per (target, slot).
__global__ void snn_counts_kernel(const int64_t *__restrict__ t_idx, int64_t m, const int64_t *__restrict__ r_idx,
                                  int64_t n, int k, int32_t *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * k) return;
    const int64_t t = e / k;
    const int64_t j = t_idx[e];
    int snn = 0;
    if (j >= 0 && j < n) {
        const int64_t *a = t_idx + t * k, *b = r_idx + j * k;
        for (int p = 0; p < k; ++p) {
            const int64_t ap = a[p];
            for (int q = 0; q < k; ++q)
                if (ap == b[q]) { ++snn; break; }
        }
    }
    out[e] = snn;
}

hipError_t transpose_ref_launch(const double *Y, int64_t n, int g, double *Yt, hipStream_t st)
{
    const int64_t chunks = (n + 63) / 64;
    hipLaunchKernelGGL(transpose_ref_kernel, dim3((unsigned)chunks), dim3(256), 0, st, Y, n, g, Yt);
    return hipGetLastError();
}

size_t canberra_lds_bytes(int epl)
{
    const size_t wave = ((size_t)CANB_T * (64 * epl) * 12 + CANB_T * 8 + CANB_T * 4 + 8 + 15) & ~(size_t)15;
    return wave * 4;
}

hipError_t canberra_topk_launch(int epl, const double *X, int64_t m, const double *Yt, int64_t n, int g, double f,
                                const uint8_t *mask, int S, double *cand_d, uint32_t *cand_i, hipStream_t st)
{
    const int64_t n_chunks = (n + 63) / 64;
    const int64_t cps = (n_chunks + S - 1) / S;
    dim3 grid((unsigned)((m + 4 * CANB_T - 1) / (4 * CANB_T)), S), block(256);
    const size_t lds = canberra_lds_bytes(epl);
    hipError_t e;
    if (epl == 1) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&canberra_topk_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((canberra_topk_kernel<1>), grid, block, lds, st, X, m, Yt, n, g, f, mask, n_chunks, cps,
                           cand_d, cand_i);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&canberra_topk_kernel<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((canberra_topk_kernel<2>), grid, block, lds, st, X, m, Yt, n, g, f, mask, n_chunks, cps,
                           cand_d, cand_i);
    }
    return hipGetLastError();
}

template <bool IDX64>
static hipError_t merge_dispatch(const double *pd, const void *pi, int64_t m, int P, int64_t part_stride,
                                 int n_parts, int kp, int k, int drop, int64_t base, int64_t *out_idx,
                                 double *out_dist, int *n_found, hipStream_t st)
{
    const int ncl = (P + 63) / 64;
    dim3 grid((unsigned)((m + 3) / 4)), block(256);
#define NABO_MG(N)                                                                                              \
    hipLaunchKernelGGL((merge_kernel<N, IDX64>), grid, block, 0, st, pd, pi, m, P, part_stride, n_parts, kp, k,  \
                       drop, base, out_idx, out_dist, n_found)
    if (ncl <= 1) NABO_MG(1);
    else if (ncl <= 2) NABO_MG(2);
    else if (ncl <= 4) NABO_MG(4);
    else if (ncl <= 8) NABO_MG(8);
    else if (ncl <= 16) NABO_MG(16);
    else return hipErrorInvalidValue;
#undef NABO_MG
    return hipGetLastError();
}

hipError_t merge_local_launch(const double *cand_d, const uint32_t *cand_i, int64_t m, int P, int k, int drop,
                              int64_t base, int64_t *out_idx, double *out_dist, int *n_found, hipStream_t st)
{
    return merge_dispatch<false>(cand_d, cand_i, m, P, 0, 1, P, k, drop, base, out_idx, out_dist, n_found, st);
}

hipError_t merge_parts_launch(const double *parts_d, const int64_t *parts_i, int n_parts, int64_t m, int kp, int k,
                              int drop, int64_t *out_idx, double *out_dist, hipStream_t st)
{
    return merge_dispatch<true>(parts_d, parts_i, m, n_parts * kp, m * (int64_t)kp, n_parts, kp, k, drop, 0, out_idx,
                                out_dist, nullptr, st);
}

hipError_t snn_counts_launch(const int64_t *t_idx, int64_t m, const int64_t *r_idx, int64_t n, int k, int32_t *out,
                             hipStream_t st)
{
    const int64_t tot = m * k;
    if (tot == 0) return hipSuccess;
    hipLaunchKernelGGL(snn_counts_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, t_idx, m, r_idx, n, k,
                       out);
    return hipGetLastError();
}

}  // namespace nabo
