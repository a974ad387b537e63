// refine.hip -- float64 re-evaluation, certification and exact fallback (gfx950).
//
// This is where results become the reference's: every candidate distance is recomputed
// exactly as nabo/_mapping.py:20-26 does it -- float64, components in ascending order,
// rounded multiply then rounded add (no FMA: __dmul_rn/__dadd_rn), correctly rounded sqrt --
// and rows are ordered by (distance, index), the canonical form of the np.argsort at
// nabo/_mapping.py:139-145 (masked refs last, positional `[1:]` drop for intra_ref).
//
// Certification (per row): the fp32 filter guarantees every NON-candidate j has score
// a_j >= tau.  With  |a_j + ||x~||^2 - d_j^2| <= E  (rounding-error bound of the fp32 fma
// chain + fp32 input rounding), a row is certified when
//       tau_min + ||x~||^2 - E  >  d_(k')^2          (k' = k + drop_first, exact value)
// i.e. nothing outside the candidate set can enter or tie with the first k'.  Rows that fail
// (duplicates, pathological gaps) are re-solved by the exact kernels below, which brute-force the
// whole reference set in float64 on the GPU.  There is no CPU path.
#include "knn_common.h"

namespace nabo {

// Exact reference distance (nabo/_mapping.py:20-26).
__device__ __forceinline__ double euclid_exact(const double *__restrict__ x, const double *__restrict__ y, int g)
{
    double td = 0.0;
    for (int k = 0; k < g; ++k) {
        const double t = __dsub_rn(x[k], y[k]);
        td = __dadd_rn(td, __dmul_rn(t, t));
    }
    return __dsqrt_rn(td);
}

// Exact modified Canberra (nabo/_mapping.py:33-45); x = target, y = reference.
__device__ __forceinline__ double canberra_exact(const double *__restrict__ x, const double *__restrict__ y, int g,
                                                 double f)
{
    double dist = 0.0;
    for (int k = 0; k < g; ++k) {
        const double xv = x[k], yv = y[k];
        const double absx = fabs(xv);
        const double num = fabs(__dsub_rn(xv, yv));
        if (num < __dmul_rn(f, absx)) {
            const double den = __dadd_rn(__dadd_rn(absx, fabs(yv)), 0.01);
            dist = __dadd_rn(dist, __ddiv_rn(num, den));
        } else {
            dist = __dadd_rn(dist, 1.0);
        }
    }
    return dist;
}

// EXTENSION (not in the reference, parity unpinned -- oracle/nabo_oracle.c cosine_pair): cosine distance
// 1 - <x,y> / (sqrt<x,x> * sqrt<y,y>), sums in ascending k, rounded multiply then rounded add; a zero
// vector is at distance 1 from everything.
__device__ __forceinline__ double cosine_exact(const double *__restrict__ x, const double *__restrict__ y, int g)
{
    double dot = 0.0, nx = 0.0, ny = 0.0;
    for (int k = 0; k < g; ++k) {
        const double xv = x[k], yv = y[k];
        dot = __dadd_rn(dot, __dmul_rn(xv, yv));
        nx = __dadd_rn(nx, __dmul_rn(xv, xv));
        ny = __dadd_rn(ny, __dmul_rn(yv, yv));
    }
    if (nx == 0.0 || ny == 0.0) return 1.0;
    return __dsub_rn(1.0, __ddiv_rn(dot, __dmul_rn(__dsqrt_rn(nx), __dsqrt_rn(ny))));
}

// metric: 0 Euclidean, 1 modified Canberra, 2 cosine
__device__ __forceinline__ double exact_dist(int metric, const double *__restrict__ x, const double *__restrict__ y,
                                             int g, double f)
{
    return metric == 0 ? euclid_exact(x, y, g) : metric == 1 ? canberra_exact(x, y, g, f) : cosine_exact(x, y, g);
}

// Rows scaled to unit length in float64 (zero rows stay zero): the cosine metric runs the Euclidean
// filter on these, since ||x^ - y^||^2 = 2 * (1 - cos).
__global__ void normalise_rows_kernel(const double *__restrict__ X, int64_t m, int g, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double *x = X + i * g;
    double nx = 0.0;
    for (int k = 0; k < g; ++k) nx = __dadd_rn(nx, __dmul_rn(x[k], x[k]));
    const double s = __dsqrt_rn(nx);
    for (int k = 0; k < g; ++k) out[i * g + k] = nx == 0.0 ? 0.0 : __ddiv_rn(x[k], s);
}

hipError_t normalise_rows_launch(const double *X, int64_t m, int g, double *out, hipStream_t st)
{
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(normalise_rows_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, X, m, g, out);
    return hipGetLastError();
}

// ---- fine seam: dense D[m,n] (a1 / a2 literal) -------------------------------------------
__global__ void pairwise_kernel(const double *__restrict__ X, int64_t m, const double *__restrict__ Y, int64_t n,
                                int g, int metric, double f, double *__restrict__ D)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= n || i >= m) return;
    const double *x = X + i * g, *y = Y + j * g;
    D[i * n + j] = exact_dist(metric, x, y, g, f);
}

// Rows with fewer valid (unmasked) references than k': the order row continues with the
// masked references (NaN-filled by numpy.ma -> sorted last), by ascending index, carrying
// their true distances.  masked_list holds the first masked indices in ascending order.
__device__ void emit_masked_tail(const double *__restrict__ x, const double *__restrict__ Y, int g, int metric,
                                 double f, const uint32_t *__restrict__ masked_list, int n_masked_list,
                                 int n_valid, int k, int drop, int64_t base, int64_t *__restrict__ oi,
                                 double *__restrict__ od)
{
    // positions p (in the full order row) n_valid .. k+drop-1 come from the masked list
    for (int p = n_valid + (int)threadIdx.x % 64; p < k + drop; p += 64) {
        const int q = p - n_valid;
        const int o = p - drop;
        if (o < 0) continue;
        if (q < n_masked_list) {
            const uint32_t j = masked_list[q];
            oi[o] = base + j;
            od[o] = exact_dist(metric, x, Y + (int64_t)j * g, g, f);
        } else {
            oi[o] = -1;
            od[o] = __builtin_nan("");
        }
    }
}

// One wave per target row; NCL = ceil(S*L/64) candidates per lane.
// MET = 0: Euclidean candidates from the fp32 / f16x3 score filter, certified with the
//               rounding-error bound E (header comment).
// MET = 1: modified-Canberra candidates from the fp32 LOWER-BOUND filter (canberra_f32.hip):
//               every non-candidate has exact distance >= tau, so the row is certified when
//               tau > d_(k') strictly, or when tau is the all-dimensions-out-of-window plateau (then
//               every non-candidate is at distance exactly g and carries a larger index than the last
//               entry of its list) AND the k'-th entry is either closer than g or precedes all of those.
// MET = 2: cosine candidates from the Euclidean filter run on unit-length rows x^, y^ (float64, then
//               packed like any other input).  The Euclidean certificate bounds ||x^-y^||^2 of every
//               non-candidate from below by B; ||x^-y^||^2 = 2(1-cos) up to the float64 rounding of the
//               normalisation and of cosine_exact itself (< 1e-13 for g <= 128), so the row is certified
//               when  B/2 - 2e-13 > c_(k')  (exact cosine value of the k'-th candidate).
// STAGE (NCL == 1, g <= 64): candidate rows are fetched by the whole wave, one coalesced g*8-byte read per candidate,
// into a wave-private LDS block and each lane then evaluates ITS candidate from LDS, in the reference's component
// order -- a lane walking its own row in global memory moved 8 bytes per 64-byte request (1.15 TB/s at 1M rows).
// The staged gather of refine_kernel / refine_cand_kernel: the candidates' rows (lane `src` of `live` holds candidate myj),
// g * 8 coalesced bytes each, into rows 0, 1, .. of the wave's LDS block -- NABO_STAGE_BATCH rows requested before the first one is
// stored.  (One load, its wait, its store per candidate made a row's refine 23 memory latencies long: 7 ms for the 1M rows of
// the headline where the bytes moved need ~2.5.)
#ifndef NABO_STAGE_BATCH
#define NABO_STAGE_BATCH 8
#endif
__device__ __forceinline__ void stage_rows_to_lds(const double *__restrict__ Y, int g, int gp, double *stg, int stage_rows,
                                                  uint32_t myj, uint64_t live)
{
    const int lane = lane_id();
    constexpr int NB = NABO_STAGE_BATCH;
    for (int c0 = 0; live != 0; c0 += NB) {
        double v[NB];
        bool ok[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            ok[u] = live != 0 && c0 + u < stage_rows;           // (wave-uniform)
            uint32_t j = 0;
            if (live != 0) {
                const int src = __builtin_ctzll(live);
                live &= live - 1;
                j = (uint32_t)__builtin_amdgcn_readlane((int)myj, src);
            }
            v[u] = (ok[u] && lane < g) ? Y[(int64_t)j * g + lane] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (ok[u] && lane < g) stg[(c0 + u) * gp + lane] = v[u];
    }
}

template <int NCL, int MET, bool STAGE>
__global__ __launch_bounds__(256) void refine_kernel(const double *__restrict__ X, int64_t row0, int64_t m,
                                                     const double *__restrict__ Y, int g,
                                                     const uint32_t *__restrict__ cand_idx,
                                                     const float *__restrict__ cand_tau, int S, int L,
                                                     const double *__restrict__ xnorm, double err_coef,
                                                     double ymax_sqrt, double tau_scale, double cb_f, float cb_plateau,
                                                     int k, int drop, int64_t base, int64_t n_valid_total,
                                                     const uint32_t *__restrict__ masked_list, int n_masked_list,
                                                     int64_t *__restrict__ out_idx, double *__restrict__ out_dist,
                                                     uint32_t *__restrict__ fail_rows,
                                                     unsigned int *__restrict__ fail_count, int stage_rows,
                                                     const uint32_t *__restrict__ rperm,
                                                     const uint32_t *__restrict__ tperm, float *__restrict__ fail_seed)
{
    // rows [row0, m) of the filter's row order; candidate arrays are indexed by the row LOCAL to this launch.
    // Locality order (order.hip): the filter worked on permuted rows -- position prow holds target row tperm[prow]
    // (xnorm is indexed by position), candidate value c is reference row rperm[c]; both are mapped back HERE, before
    // the sort by (distance, index), so order rows and indices are the caller's.
    const int lane = lane_id();
    const int64_t lrow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t prow = row0 + lrow;
    if (prow >= m) return;
    const int64_t row = tperm ? (int64_t)tperm[prow] : prow;
    const double *x = X + row * g;
    const int ncand = S * L;
    double key[NCL];
    uint32_t val[NCL];
    if (STAGE) {
        extern __shared__ __attribute__((aligned(16))) unsigned char refine_smem[];
        const int gp = g | 1;                                  // odd row stride (in doubles): lanes spread over the banks
        double *stg = reinterpret_cast<double *>(refine_smem) + (size_t)(threadIdx.x >> 6) * stage_rows * gp;
        uint32_t myj = lane < ncand ? cand_idx[lrow * ncand + lane] : 0xFFFFFFFFu;
        if (rperm && myj != 0xFFFFFFFFu) myj = rperm[myj];
        // valid candidates are packed into the first stage_rows LDS rows (the lists hold <= lkeep of their L entries:
        // sizing the block for all S * L kept the kernel at three waves per SIMD)
        uint64_t live = __builtin_amdgcn_ballot_w64(myj != 0xFFFFFFFFu);
        const int slot = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
        stage_rows_to_lds(Y, g, gp, stg, stage_rows, myj, live);
        key[0] = __builtin_inf();
        val[0] = 0xFFFFFFFFu;
        if (myj != 0xFFFFFFFFu) {
            val[0] = myj;
            key[0] = slot < stage_rows ? exact_dist(MET, x, stg + slot * gp, g, cb_f)
                                       : exact_dist(MET, x, Y + (int64_t)myj * g, g, cb_f);
        }
    } else {
#pragma unroll
        for (int r = 0; r < NCL; ++r) {
            const int e = r * 64 + lane;
            key[r] = __builtin_inf();
            val[r] = 0xFFFFFFFFu;
            if (e < ncand) {
                uint32_t j = cand_idx[lrow * ncand + e];
                if (j != 0xFFFFFFFFu) {
                    if (rperm) j = rperm[j];
                    val[r] = j;
                    key[r] = exact_dist(MET, x, Y + (int64_t)j * g, g, cb_f);
                }
            }
        }
    }
    wave_bitonic_sort<NCL, double>(key, val);
    // number of real candidates
    int nreal = 0;
#pragma unroll
    for (int r = 0; r < NCL; ++r) nreal += __popcll(__builtin_amdgcn_ballot_w64(val[r] != 0xFFFFFFFFu));
    const int kk = k + drop;
    // threshold below which no reference was discarded by the filter (min over splits)
    float tmin = __builtin_inff();
    for (int s = lane; s < S; s += 64) tmin = fminf(tmin, cand_tau[lrow * S + s]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tmin = fminf(tmin, __shfl_xor(tmin, o, 64));

    bool certified = true;
    // A row that fails its certificate hands the next pass a SEED (fail_seed, Euclidean / cosine): the filter threshold, in
    // the filter's score units, below which every reference that can still enter or tie with the first k' must fall --
    // the k'-th exact distance among THESE candidates bounds the true one from above.  A pass that starts from it
    // (l2c_topk.hip: tau_init) only collects those few references instead of building its lists from +inf.
    float seed = __builtin_inff();
    if (nreal >= kk) {
        // exact k'-th distance
        const int e = kk - 1;
        double dk = 0.0;
#pragma unroll
        for (int r = 0; r < NCL; ++r)
            if ((e >> 6) == r) dk = __shfl(key[r], e & 63, 64);
        if (tmin == __builtin_inff()) {
            // no list ever overflowed, i.e. the filter claims it dropped nothing: then EVERY unmasked reference
            // must be among the candidates.  (References whose fp32 score is NaN -- a target that dwarfs the
            // references by more than the fp32 range -- never pass `score < tau` and vanish without a trace.)
            certified = (int64_t)nreal >= n_valid_total;
        } else if (MET == 1) {
            {
                certified = (double)tmin > dk;
                if (!certified && tmin == cb_plateau) {
                    // Every reference some list dropped sits at distance exactly g and carries a larger index than
                    // the LAST entry of that list (the lists arrive sorted by (key, index); with tau == plateau
                    // that last entry is the largest kept plateau index).  A CANDIDATE at distance g may have had a
                    // key below the plateau (a dimension exactly on the window edge is not "provably out") and any
                    // index, so the k'-th entry is final only if it is closer than g or precedes every dropped one.
                    uint32_t ie = 0;
#pragma unroll
                    for (int r = 0; r < NCL; ++r)
                        if ((e >> 6) == r) ie = (uint32_t)__shfl((int)val[r], e & 63, 64);
                    uint32_t pmin = 0xFFFFFFFFu;
                    for (int s2 = lane; s2 < S; s2 += 64)
                        if (cand_tau[lrow * S + s2] == cb_plateau) {
                            const uint32_t pl = cand_idx[lrow * ncand + (int64_t)s2 * L + (L - 1)];
                            pmin = pl < pmin ? pl : pmin;
                        }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const uint32_t other = (uint32_t)__shfl_xor((int)pmin, o, 64);
                        pmin = other < pmin ? other : pmin;
                    }
                    certified = (dk < (double)g) || (ie <= pmin);
                }
            }
        } else {
            const double sx = sqrt(xnorm[prow]);
            const double E = err_coef * (sx + ymax_sqrt) * (sx + ymax_sqrt);
            const double bound = ((double)tmin * tau_scale + xnorm[prow] - E) * (1.0 - 1e-12);   // tau in score units
            certified = MET == 2 ? (0.5 * bound - 2e-13 > dk) : (bound > dk * dk * (1.0 + 1e-12));
            if (!certified) {
                // threshold T with  (T tau_scale + ||x||^2 - E)(1 - 1e-12) > d_k'^2 (1 + 1e-12)  (cosine: the same bound on
                // ||x^ - y^||^2 = 2 c), rounded UP to fp32: a pass that drops nothing below T certifies the row
                const double t2 = MET == 2 ? 2.0 * (dk + 2e-13) * (1.0 + 4e-12) + 4e-13 : dk * dk * (1.0 + 4e-12);
                const double v = (t2 - xnorm[prow] + E) / tau_scale;
                float sf = (float)v;
                if ((double)sf < v)            // next float up (sf is finite here, or NaN and dropped below)
                    sf = sf > 0.0f ? __uint_as_float(__float_as_uint(sf) + 1u) : sf < 0.0f ? __uint_as_float(__float_as_uint(sf) - 1u) : 1.0e-45f;
                if (sf == sf) seed = sf;
            }
        }
    } else {
        // fewer candidates than k': only legitimate when EVERY unmasked reference is a candidate
        // (tiny reference sets).  Anything else (e.g. non-finite fp32 scores) goes to the exact path.
        certified = (int64_t)nreal >= n_valid_total;
    }
    if (!certified) {
        if (lane == 0) {
            const unsigned int slot = atomicAdd(fail_count, 1u);
            fail_rows[slot] = (uint32_t)row;
            if (fail_seed) fail_seed[slot] = seed;
        }
        return;
    }
    int64_t *oi = out_idx + row * k;
    double *od = out_dist + row * k;
#pragma unroll
    for (int r = 0; r < NCL; ++r) {
        const int e = r * 64 + lane;
        const int o = e - drop;
        if (o >= 0 && o < k && e < nreal) {
            oi[o] = base + val[r];
            od[o] = key[r];
        }
    }
    if (nreal < kk)
        emit_masked_tail(x, Y, g, MET, cb_f, masked_list, n_masked_list, nreal, k, drop, base, oi, od);
}

// Shard mode (reference rows sharded over GPUs, global certification -- sharded.hip): no local
// verdict.  Emits the row's candidates in exact float64 order (first `kout`, absent = idx -1 / +inf) and
// `bound`, a lower bound on the exact SQUARED distance of every reference of this shard that is NOT among
// the emitted candidates: min over splits of (tau * scale + ||x~||^2 - E), and additionally the squared
// distance of the first candidate that did not fit into `kout`.  +inf when nothing was dropped, -inf when
// the filter saw non-finite scores (forces the exact second phase).
template <int NCL, int MET, bool STAGE>
__global__ __launch_bounds__(256) void refine_cand_kernel(const double *__restrict__ X, int64_t row0, int64_t m,
                                                          const double *__restrict__ Y, int g,
                                                          const uint32_t *__restrict__ cand_idx,
                                                          const float *__restrict__ cand_tau, int S, int L,
                                                          const double *__restrict__ xnorm, double err_coef,
                                                          double ymax_sqrt, double tau_scale, int kout, int64_t base,
                                                          int64_t n_valid_total, int64_t *__restrict__ out_idx,
                                                          double *__restrict__ out_dist, double *__restrict__ out_bound,
                                                          int stage_rows, const uint32_t *__restrict__ rperm,
                                                          const uint32_t *__restrict__ tperm)
{
    const int lane = lane_id();
    const int64_t lrow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t prow = row0 + lrow;                          // position in the filter's row order (refine_kernel)
    if (prow >= m) return;
    const int64_t row = tperm ? (int64_t)tperm[prow] : prow;
    const double *x = X + row * g;
    const int ncand = S * L;
    double key[NCL];
    uint32_t val[NCL];
    if (STAGE) {                                               // coalesced candidate rows through LDS (refine_kernel)
        extern __shared__ __attribute__((aligned(16))) unsigned char refine_smem[];
        const int gp = g | 1;
        double *stg = reinterpret_cast<double *>(refine_smem) + (size_t)(threadIdx.x >> 6) * stage_rows * gp;
        uint32_t myj = lane < ncand ? cand_idx[lrow * ncand + lane] : 0xFFFFFFFFu;
        if (rperm && myj != 0xFFFFFFFFu) myj = rperm[myj];
        // valid candidates are packed into the first stage_rows LDS rows (the lists hold <= lkeep of their L entries:
        // sizing the block for all S * L kept the kernel at three waves per SIMD)
        uint64_t live = __builtin_amdgcn_ballot_w64(myj != 0xFFFFFFFFu);
        const int slot = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
        stage_rows_to_lds(Y, g, gp, stg, stage_rows, myj, live);
        key[0] = __builtin_inf();
        val[0] = 0xFFFFFFFFu;
        if (myj != 0xFFFFFFFFu) {
            val[0] = myj;
            key[0] = slot < stage_rows ? exact_dist(MET, x, stg + slot * gp, g, 0.0)
                                       : exact_dist(MET, x, Y + (int64_t)myj * g, g, 0.0);
        }
    } else {
#pragma unroll
        for (int r = 0; r < NCL; ++r) {
            const int e = r * 64 + lane;
            key[r] = __builtin_inf();
            val[r] = 0xFFFFFFFFu;
            if (e < ncand) {
                uint32_t j = cand_idx[lrow * ncand + e];
                if (j != 0xFFFFFFFFu) {
                    if (rperm) j = rperm[j];
                    val[r] = j;
                    key[r] = exact_dist(MET, x, Y + (int64_t)j * g, g, 0.0);
                }
            }
        }
    }
    wave_bitonic_sort<NCL, double>(key, val);
    int nreal = 0;
#pragma unroll
    for (int r = 0; r < NCL; ++r) nreal += __popcll(__builtin_amdgcn_ballot_w64(val[r] != 0xFFFFFFFFu));
    float tmin = __builtin_inff();
    for (int s = lane; s < S; s += 64) tmin = fminf(tmin, cand_tau[lrow * S + s]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tmin = fminf(tmin, __shfl_xor(tmin, o, 64));
    // `bound` is always compared as  d^2 (1 + 1e-12) < bound  by the caller.  Cosine (MET = 2): every reference
    // not emitted has cosine distance >= B/2 - 2e-13 (refine_kernel's certificate), so the bound is the square of
    // that value (cosine distances are >= -1e-15, squaring keeps the order once the value is positive).
    double bound = __builtin_inf();
    if (tmin != __builtin_inff()) {
        const double sx = sqrt(xnorm[prow]);
        const double E = err_coef * (sx + ymax_sqrt) * (sx + ymax_sqrt);
        bound = ((double)tmin * tau_scale + xnorm[prow] - E) * (1.0 - 1e-12);
        if (MET == 2) {
            const double c = 0.5 * bound - 2e-13;
            bound = c > 0.0 ? c * c * (1.0 - 1e-12) : -__builtin_inf();
        }
    } else if ((int64_t)nreal < n_valid_total) {
        bound = -__builtin_inf();          // nothing "dropped" yet references are missing: non-finite scores
    }
    if (nreal > kout) {                    // candidates beyond kout are not reported either
        double dn = 0.0;
#pragma unroll
        for (int r = 0; r < NCL; ++r)
            if ((kout >> 6) == r) dn = __shfl(key[r], kout & 63, 64);
        bound = fmin(bound, dn * dn * (1.0 - 1e-12));
    }
    if (!(xnorm[prow] == xnorm[prow])) bound = -__builtin_inf();      // target outside the filter's range (pack.hip)
#pragma unroll
    for (int r = 0; r < NCL; ++r) {
        const int e = r * 64 + lane;
        if (e < kout) {
            const bool real = e < nreal;
            out_idx[row * kout + e] = real ? base + (int64_t)val[r] : -1;
            out_dist[row * kout + e] = real ? key[r] : __builtin_inf();
        }
    }
    if (lane == 0) out_bound[row] = bound;
}

// Exact brute force for flagged rows, two phases so that a handful of rows does not serialise on one CU each:
//   1. exact_dist_rows_kernel: D[b][j] = exact float64 distance of flagged row b to EVERY reference (NaN for
//      masked ones), grid = (references / 256, rows of the batch / 8) -- the whole chip works on every row;
//   2. exact_select_rows_kernel: one 256-thread block per row, k' selection passes over D[b] (each pass picks
//      the smallest (d, j) strictly after the previous pick -- the canonical order, ties included).
// RB flagged rows per thread: a thread owns one reference j, reads Y[j][k] once per component and advances the
// RB rows' accumulators in the reference's own order (k ascending, rounded multiply then rounded add), so the
// strided reference reads are shared by RB rows.
constexpr int EXACT_RB = 8;

__global__ __launch_bounds__(256) void exact_dist_rows_kernel(const double *__restrict__ X, const double *__restrict__ Y,
                                                              int64_t n, int g, int metric, double f,
                                                              const uint8_t *__restrict__ mask,
                                                              const uint32_t *__restrict__ rows, int nrows,
                                                              double *__restrict__ D)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int b0 = blockIdx.y * EXACT_RB;
    const double *xr[EXACT_RB];
#pragma unroll
    for (int r = 0; r < EXACT_RB; ++r) xr[r] = X + (int64_t)rows[b0 + r < nrows ? b0 + r : nrows - 1] * g;
    const bool masked = mask && mask[j];
    const double *y = Y + j * g;
    double a0[EXACT_RB], a1[EXACT_RB], a2 = 0.0;
#pragma unroll
    for (int r = 0; r < EXACT_RB; ++r) { a0[r] = 0.0; a1[r] = 0.0; }
    if (!masked) {
        for (int k = 0; k < g; ++k) {
            const double yv = y[k];
            if (metric == 0) {
#pragma unroll
                for (int r = 0; r < EXACT_RB; ++r) {
                    const double t = __dsub_rn(xr[r][k], yv);
                    a0[r] = __dadd_rn(a0[r], __dmul_rn(t, t));
                }
            } else if (metric == 1) {
#pragma unroll
                for (int r = 0; r < EXACT_RB; ++r) {
                    const double xv = xr[r][k];
                    const double absx = fabs(xv);
                    const double num = fabs(__dsub_rn(xv, yv));
                    if (num < __dmul_rn(f, absx))
                        a0[r] = __dadd_rn(a0[r], __ddiv_rn(num, __dadd_rn(__dadd_rn(absx, fabs(yv)), 0.01)));
                    else
                        a0[r] = __dadd_rn(a0[r], 1.0);
                }
            } else {
                a2 = __dadd_rn(a2, __dmul_rn(yv, yv));
#pragma unroll
                for (int r = 0; r < EXACT_RB; ++r) {
                    const double xv = xr[r][k];
                    a0[r] = __dadd_rn(a0[r], __dmul_rn(xv, yv));
                    a1[r] = __dadd_rn(a1[r], __dmul_rn(xv, xv));
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < EXACT_RB; ++r) {
        if (b0 + r >= nrows) break;
        double d = __builtin_nan("");                                       // masked reference
        if (!masked) {
            if (metric == 0) d = __dsqrt_rn(a0[r]);
            else if (metric == 1) d = a0[r];
            else d = (a1[r] == 0.0 || a2 == 0.0) ? 1.0
                                                 : __dsub_rn(1.0, __ddiv_rn(a0[r], __dmul_rn(__dsqrt_rn(a1[r]), __dsqrt_rn(a2))));
        }
        D[(int64_t)(b0 + r) * n + j] = d;
    }
}

__global__ __launch_bounds__(256) void exact_select_rows_kernel(const double *__restrict__ X, const double *__restrict__ Y,
                                                                int64_t n, int g, int metric, double f,
                                                                const double *__restrict__ D,
                                                                const uint32_t *__restrict__ rows,
                                                                int k, int drop, int64_t base,
                                                                const uint32_t *__restrict__ masked_list, int n_masked_list,
                                                                int64_t *__restrict__ out_idx, double *__restrict__ out_dist)
{
    __shared__ double s_d[256];
    __shared__ uint32_t s_j[256];
    __shared__ double prev_d;
    __shared__ uint32_t prev_j;
    __shared__ int have_prev;
    const int64_t row = rows[blockIdx.x];
    const double *Dr = D + (int64_t)blockIdx.x * n;
    const double *x = X + row * g;
    const int kk = k + drop;
    if (threadIdx.x == 0) { have_prev = 0; prev_d = 0.0; prev_j = 0; }
    __syncthreads();
    int found = 0;
    for (int p = 0; p < kk; ++p) {
        double bd = __builtin_inf();
        uint32_t bj = 0xFFFFFFFFu;
        const bool hp = have_prev != 0;
        const double pd = prev_d;
        const uint32_t pj = prev_j;
        for (int64_t j = threadIdx.x; j < n; j += 256) {
            const double d = Dr[j];
            if (d != d) continue;                                           // masked reference
            if (hp && !kv_less<double>(pd, pj, d, (uint32_t)j)) continue;   // not after previous pick
            if (kv_less<double>(d, (uint32_t)j, bd, bj)) { bd = d; bj = (uint32_t)j; }
        }
        s_d[threadIdx.x] = bd;
        s_j[threadIdx.x] = bj;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) {
                if (kv_less<double>(s_d[threadIdx.x + w], s_j[threadIdx.x + w], s_d[threadIdx.x], s_j[threadIdx.x])) {
                    s_d[threadIdx.x] = s_d[threadIdx.x + w];
                    s_j[threadIdx.x] = s_j[threadIdx.x + w];
                }
            }
            __syncthreads();
        }
        const uint32_t wj = s_j[0];
        const double wd = s_d[0];
        __syncthreads();
        if (wj == 0xFFFFFFFFu) break;       // ran out of valid references
        if (threadIdx.x == 0) {
            prev_d = wd; prev_j = wj; have_prev = 1;
            const int o = p - drop;
            if (o >= 0) { out_idx[row * k + o] = base + wj; out_dist[row * k + o] = wd; }
        }
        ++found;
        __syncthreads();
    }
    if (found < kk && threadIdx.x < 64)
        emit_masked_tail(x, Y, g, metric, f, masked_list, n_masked_list, found, k, drop, base, out_idx + row * k,
                         out_dist + row * k);
}

// Order-row tail for every row when the index holds fewer valid references than k'.
__global__ __launch_bounds__(256) void masked_tail_kernel(const double *__restrict__ X, int64_t m,
                                                          const double *__restrict__ Y, int g, int metric, double f,
                                                          const uint32_t *__restrict__ masked_list, int n_masked_list,
                                                          int n_valid, int k, int drop, int64_t base,
                                                          int64_t *__restrict__ out_idx, double *__restrict__ out_dist)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    emit_masked_tail(X + row * g, Y, g, metric, f, masked_list, n_masked_list, n_valid, k, drop, base,
                     out_idx + row * k, out_dist + row * k);
}

hipError_t masked_tail_launch(const double *X, int64_t m, const double *Y, int g, int metric, double f,
                              const uint32_t *masked_list, int n_masked_list, int n_valid, int k, int drop,
                              int64_t base, int64_t *out_idx, double *out_dist, hipStream_t st)
{
    hipLaunchKernelGGL(masked_tail_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, X, m, Y, g, metric, f,
                       masked_list, n_masked_list, n_valid, k, drop, base, out_idx, out_dist);
    return hipGetLastError();
}

// Uncertified rows are re-solved as one dense batch: gather their float64 rows, scatter the answers.
__global__ void gather_rows_kernel(const double *__restrict__ X, const uint32_t *__restrict__ rows, int64_t nrows, int g,
                                   double *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nrows * g) return;
    out[e] = X[(int64_t)rows[e / g] * g + e % g];
}

__global__ void scatter_rows_kernel(const int64_t *__restrict__ si, const double *__restrict__ sd,
                                    const uint32_t *__restrict__ rows, int64_t nrows, int k,
                                    int64_t *__restrict__ out_idx, double *__restrict__ out_dist)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nrows * k) return;
    const int64_t o = (int64_t)rows[e / k] * k + e % k;
    out_idx[o] = si[e];
    out_dist[o] = sd[e];
}

__global__ void iota_kernel(uint32_t *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}

// ---- several lists per row merged by their FILTER keys ---------------------------------------------------------------------
// A row whose references were streamed in S pieces (reference splits, the pieces of a cut launch, the tail round) arrives
// with S lists of up to lkeep entries each, and the exact float64 re-evaluation of all S lkeep candidates was what the
// split cost (100k x 100k: refine 0.37 ms with one list per row, 1.07 with two).  One streamed list would have kept the
// lkeep smallest SCORES of the union, so that is what goes on: the union is sorted by (score, index), the first lkeep stay,
// and the merged threshold is  min(every list's threshold, the first score cut here)  -- each reference that is not in
// the merged list was either dropped by its own list (score >= that list's threshold) or cut here (score >= the first
// one cut).  refine.hip's certificate reads nothing else.  One wave per row.
template <int NCL>
__global__ __launch_bounds__(256) void merge_lists_kernel(const uint32_t *__restrict__ cand_idx, const float *__restrict__ cand_key,
                                                          const float *__restrict__ cand_tau, int64_t rows, int S, int L,
                                                          int lkeep, int Lout, uint32_t *__restrict__ out_idx,
                                                          float *__restrict__ out_tau)
{
    const int lane = lane_id();
    const int64_t lrow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lrow >= rows) return;
    const int ncand = S * L;
    float key[NCL];
    uint32_t val[NCL];
#pragma unroll
    for (int r = 0; r < NCL; ++r) {
        const int e = r * 64 + lane;
        key[r] = __builtin_inff();
        val[r] = 0xFFFFFFFFu;
        if (e < ncand) {
            const uint32_t j = cand_idx[lrow * ncand + e];
            if (j != 0xFFFFFFFFu) { val[r] = j; key[r] = cand_key[lrow * ncand + e]; }
        }
    }
    wave_sort_f32<NCL>(key, val);
    float tmin = __builtin_inff();
    for (int s2 = lane; s2 < S; s2 += 64) tmin = fminf(tmin, cand_tau[lrow * S + s2]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tmin = fminf(tmin, __shfl_xor(tmin, o, 64));
    if (lkeep < 64 * NCL) {                                 // the first entry cut (+inf: a sentinel, nothing was cut)
        float cut = __builtin_inff();
#pragma unroll
        for (int r = 0; r < NCL; ++r)
            if ((lkeep >> 6) == r) cut = __shfl(key[r], lkeep & 63, 64);
        tmin = fminf(tmin, cut);
    }
#pragma unroll
    for (int r = 0; r < (NCL < 2 ? NCL : 2); ++r) {              // (Lout <= 128: the first two registers)
        const int e = r * 64 + lane;
        if (e < Lout) out_idx[lrow * Lout + e] = e < lkeep ? val[r] : 0xFFFFFFFFu;
    }
    if (lane == 0) out_tau[lrow] = tmin;
}

// lkeep entries stay, in rows of Lout (<= 128) -- Lout = L and lkeep = the lists' own length for a first pass; a seeded pass keeps
// up to 128 of its S x 32 (api.hip)
hipError_t merge_lists_launch(const uint32_t *cand_idx, const float *cand_key, const float *cand_tau, int64_t rows, int S, int L,
                              int lkeep, int Lout, uint32_t *out_idx, float *out_tau, hipStream_t st)
{
    if (rows <= 0) return hipSuccess;
    const int ncl = (S * L + 63) / 64;
    if (Lout > 128 || Lout > 64 * ncl || lkeep > Lout || lkeep < 1 || S < 1 || L < 1) return hipErrorInvalidValue;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define NABO_ML(N) hipLaunchKernelGGL((merge_lists_kernel<N>), grid, block, 0, st, cand_idx, cand_key, cand_tau, rows, S, L, lkeep, Lout, out_idx, out_tau)
    if (ncl <= 1) NABO_ML(1);
    else if (ncl <= 2) NABO_ML(2);
    else if (ncl <= 4) NABO_ML(4);
    else if (ncl <= 8) NABO_ML(8);
    else if (ncl <= 16) NABO_ML(16);
    else return hipErrorInvalidValue;
#undef NABO_ML
    return hipGetLastError();
}

hipError_t iota_launch(uint32_t *out, int64_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n);
    return hipGetLastError();
}

hipError_t gather_rows_launch(const double *X, const uint32_t *rows, int64_t nrows, int g, double *out, hipStream_t st)
{
    if (nrows == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((nrows * g + 255) / 256)), dim3(256), 0, st, X, rows, nrows, g, out);
    return hipGetLastError();
}

hipError_t scatter_rows_launch(const int64_t *si, const double *sd, const uint32_t *rows, int64_t nrows, int k,
                               int64_t *out_idx, double *out_dist, hipStream_t st)
{
    if (nrows == 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((nrows * k + 255) / 256)), dim3(256), 0, st, si, sd, rows, nrows,
                       k, out_idx, out_dist);
    return hipGetLastError();
}

hipError_t pairwise_launch(const double *X, int64_t m, const double *Y, int64_t n, int g, int metric, double f,
                           double *D, hipStream_t st)
{
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)m);
    hipLaunchKernelGGL(pairwise_kernel, grid, dim3(256), 0, st, X, m, Y, n, g, metric, f, D);
    return hipGetLastError();
}

hipError_t refine_launch(const double *X, int64_t row0, int64_t m, const double *Y, int g, const uint32_t *cand_idx,
                         const float *cand_tau, int S, int L, const double *xnorm, double err_coef,
                         double ymax_sqrt, double tau_scale, int k, int drop, int64_t base, int64_t n_valid_total,
                         const uint32_t *masked_list, int n_masked_list, int64_t *out_idx, double *out_dist,
                         uint32_t *fail_rows, unsigned int *fail_count, hipStream_t st, int metric = 0,
                         double cb_f = 0.0, float cb_plateau = 0.0f, int lvalid = 0, const uint32_t *rperm = nullptr,
                         const uint32_t *tperm = nullptr, float *fail_seed = nullptr)
{
    const int ncl = (S * L + 63) / 64;
    if (m <= row0) return hipSuccess;
    dim3 grid((unsigned)((m - row0 + 3) / 4)), block(256);
    // staged gather: one candidate per lane, rows of <= 64 components, 4 waves x S*L rows within 64 KB of LDS
    // lvalid: entries of a list that can be valid (lkeep; 0 = all L)
    const int stage_rows = S * ((lvalid > 0 && lvalid < L) ? lvalid : L);
    const size_t stage_bytes = (size_t)4 * stage_rows * (g | 1) * sizeof(double);
    const bool stage = ncl <= 1 && g <= 64 && stage_bytes <= 65536;
#define NABO_RF2(N, MV, SV)                                                                                          \
    hipLaunchKernelGGL((refine_kernel<N, MV, SV>), grid, block, SV ? stage_bytes : 0, st, X, row0, m, Y, g, cand_idx,   \
                       cand_tau, S, L, xnorm, err_coef, ymax_sqrt, tau_scale, cb_f, cb_plateau, k, drop, base,       \
                       n_valid_total, masked_list, n_masked_list, out_idx, out_dist, fail_rows, fail_count, stage_rows, rperm, tperm, \
                       fail_seed)
#define NABO_RF(N)                                                                                               \
    do {                                                                                                         \
        if (metric == 1) NABO_RF2(N, 1, false);                                                                  \
        else if (metric == 2) NABO_RF2(N, 2, false);                                                             \
        else NABO_RF2(N, 0, false);                                                                              \
    } while (0)
    if (stage) {
        if (metric == 1) NABO_RF2(1, 1, true);
        else if (metric == 2) NABO_RF2(1, 2, true);
        else NABO_RF2(1, 0, true);
    } else
    if (ncl <= 1) NABO_RF(1);
    else if (ncl <= 2) NABO_RF(2);
    else if (ncl <= 4) NABO_RF(4);
    else if (ncl <= 8) NABO_RF(8);
    else if (ncl <= 16) NABO_RF(16);
    else return hipErrorInvalidValue;
#undef NABO_RF2
#undef NABO_RF
    return hipGetLastError();
}

hipError_t refine_cand_launch(const double *X, int64_t row0, int64_t m, const double *Y, int g, const uint32_t *cand_idx,
                              const float *cand_tau, int S, int L, const double *xnorm, double err_coef,
                              double ymax_sqrt, double tau_scale, int kout, int64_t base, int64_t n_valid_total,
                              int64_t *out_idx, double *out_dist, double *out_bound, hipStream_t st, int metric = 0,
                              int lvalid = 0, const uint32_t *rperm = nullptr, const uint32_t *tperm = nullptr)
{
    const int ncl = (S * L + 63) / 64;
    if (m <= row0) return hipSuccess;
    dim3 grid((unsigned)((m - row0 + 3) / 4)), block(256);
    const int stage_rows = S * ((lvalid > 0 && lvalid < L) ? lvalid : L);
    const size_t stage_bytes = (size_t)4 * stage_rows * (g | 1) * sizeof(double);
    const bool stage = ncl <= 1 && g <= 64 && stage_bytes <= 65536;
#define NABO_RC2(N, MV, SV)                                                                                             \
    hipLaunchKernelGGL((refine_cand_kernel<N, MV, SV>), grid, block, SV ? stage_bytes : 0, st, X, row0, m, Y, g, cand_idx, \
                       cand_tau, S, L, xnorm, err_coef, ymax_sqrt, tau_scale, kout, base, n_valid_total, out_idx,       \
                       out_dist, out_bound, stage_rows, rperm, tperm)
#define NABO_RC(N)                                                                                                  \
    do {                                                                                                            \
        if (metric == 2) NABO_RC2(N, 2, false);                                                                     \
        else NABO_RC2(N, 0, false);                                                                                 \
    } while (0)
    if (stage) {
        if (metric == 2) NABO_RC2(1, 2, true);
        else NABO_RC2(1, 0, true);
    } else
    if (ncl <= 1) NABO_RC(1);
    else if (ncl <= 2) NABO_RC(2);
    else if (ncl <= 4) NABO_RC(4);
    else if (ncl <= 8) NABO_RC(8);
    else if (ncl <= 16) NABO_RC(16);
    else return hipErrorInvalidValue;
#undef NABO_RC2
#undef NABO_RC
    return hipGetLastError();
}

// D: workspace for d_rows x n float64 distances; the flagged rows are processed d_rows at a time.
hipError_t exact_rows_launch(const double *X, const double *Y, int64_t n, int g, int metric, double f,
                             const uint8_t *mask, const uint32_t *rows, unsigned int nrows, int k, int drop,
                             int64_t base, const uint32_t *masked_list, int n_masked_list, int64_t *out_idx,
                             double *out_dist, double *D, unsigned int d_rows, hipStream_t st)
{
    if (nrows == 0) return hipSuccess;
    if (!D || d_rows == 0) return hipErrorInvalidValue;
    for (unsigned int r0 = 0; r0 < nrows; r0 += d_rows) {
        const unsigned int nb = nrows - r0 < d_rows ? nrows - r0 : d_rows;
        hipLaunchKernelGGL(exact_dist_rows_kernel, dim3((unsigned)((n + 255) / 256), (nb + EXACT_RB - 1) / EXACT_RB), dim3(256),
                           0, st, X, Y, n, g, metric, f, mask, rows + r0, (int)nb, D);
        hipLaunchKernelGGL(exact_select_rows_kernel, dim3(nb), dim3(256), 0, st, X, Y, n, g, metric, f, D, rows + r0, k,
                           drop, base, masked_list, n_masked_list, out_idx, out_dist);
    }
    return hipGetLastError();
}

}  // namespace nabo
