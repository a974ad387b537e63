// canberra_f32.hip -- fp32 LOWER-BOUND filter for the modified Canberra distance (gfx950).
//
// _mod_canberra_dist (nabo/_mapping.py:29-45) is abs / compare / divide per dimension -- VALU work,
// nothing for MFMA.  Evaluating the reference's float64 expression for every pair costs ~45 float64
// instructions per dimension (the IEEE divide expands); this kernel instead computes, in ~12 fp32
// instructions per dimension, a value lb that is PROVABLY <= the reference's float64 distance, keeps
// the 32 (64) references with the smallest lb per target, and hands them to refine.hip, which
// evaluates the exact float64 expression for those candidates and certifies the row: every
// non-candidate has lb >= tau, hence exact distance >= tau.  Uncertified rows (ties at the k'-th
// place) are re-solved by the exact kernel of canberra.hip.  Results are therefore the same bits.
//
// Per dimension, with x32 = fl32(x), y32 = fl32(y) (relative input error 2^-24):
//   s   = |x32| + |y32|
//   nlb = max(|x32 - y32| - s * 2^-22, 0)                        <= |x - y|            (exact value)
//   thr = fl32up(f) * (|x32| * (1 + 2^-22) + 2^-124) + 1 ulp      >= fl64(f * |x|)      (reference threshold)
//   nlb >= thr  =>  the reference's test `num < f*|x|` is false  =>  both add exactly 1
//   otherwise the reference adds either q = num/den (< 1 for f <= 1 ... any f: we take min(.,1)) or 1, and
//   qlb = nlb * rcp(fma(s, 1 + 2^-21 + 2^-22, 0.01 * (1 + 2^-21)))  <= fl64(num / den)   (< 1 always)
//   term = (nlb >= thr) ? 1 : qlb                                 <= the reference's term
// lb = fl32 sum of the terms - g*(g+2)*2^-24 (fp32 summation error of g terms <= 1).  A reference that is
// "definitely out of window" in EVERY dimension (counted, not inferred from the rounded sum) has float64
// distance exactly g: it gets the key `plateau` = g - slack; every other pair gets a key strictly below
// it, so refine can recognise exact ties at distance g.  Inputs beyond the fp32 range are flagged by the
// pack kernels and the caller falls back to the exact kernel.
//
// Layout: lane = reference (64 per chunk, values resident in VGPRs for the whole chunk), target row =
// wave-uniform (its (x32, thr) pairs arrive through the scalar cache); each wave owns T targets and their
// candidate lists (fp32 key + u32 index) in LDS.
#include "knn_common.h"

namespace nabo {

constexpr int CBF_T = 8;     // targets per wave

// targets: xq[row][k] = (x32, thr); references: ycf[chunk][k][64] = y32
// *flag is set when a value does not fit fp32 comfortably (|v| > 1e37 or non-finite): the bound's
// derivation assumes normal fp32 arithmetic, the caller then uses the exact kernel.
// Both arrays are padded to GP components with NEUTRAL elements (x = y = 0, thr = +inf: the padded term is
// min(0 * rcp(0.01), 1) = 0 and never counts as out-of-window), so the kernel's inner loop has no guards.
__global__ void cbf_pack_targets_kernel(const double *__restrict__ X, int64_t m, int g, int gp, double f,
                                        float2 *__restrict__ xq, unsigned int *__restrict__ flag)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * gp) return;
    const int64_t row = e / gp;
    const int k = (int)(e - row * gp);
    if (k >= g) { xq[e] = make_float2(0.0f, __builtin_inff()); return; }
    const float x32 = (float)X[row * g + k];
    if (!(fabsf(x32) <= 1e37f)) atomicOr(flag, 1u);
    // fl32(f) rounded up; (1 + 2^-22) absorbs the three fp32 roundings; 2^-124 covers |x| below the fp32
    // normal range; one more ulp for the final multiply
    float f32 = (float)f;
    if ((double)f32 < f) f32 = __uint_as_float(__float_as_uint(f32) + 1);
    const float thr = f32 * (fabsf(x32) * (1.0f + 2.384185791015625e-07f) + 4.70197740328915e-38f);
    xq[e] = make_float2(x32, __uint_as_float(__float_as_uint(thr) + 1u));
}

__global__ void cbf_pack_refs_kernel(const double *__restrict__ Y, int64_t n, int g, int gp, float *__restrict__ ycf,
                                     unsigned int *__restrict__ flag)
{
    const int64_t chunk = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t j = chunk * 64 + lane;
    for (int k = threadIdx.x >> 6; k < gp; k += (blockDim.x >> 6)) {
        const float v = (j < n && k < g) ? (float)Y[j * g + k] : 0.0f;
        if (!(fabsf(v) <= 1e37f)) atomicOr(flag, 1u);
        ycf[(chunk * gp + k) * 64 + lane] = v;
    }
}

template <int EPL>
__device__ __forceinline__ float cbf_compact(float *kb, uint32_t *ib, int count, float (&key)[EPL], uint32_t (&val)[EPL])
{
    constexpr int L = 32 * EPL;
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const int e = r * 64 + lane;
        key[r] = __builtin_inff();
        val[r] = 0xFFFFFFFFu;
        if (e < count) { key[r] = kb[e]; val[r] = ib[e]; }
    }
    wave_bitonic_sort<EPL, float>(key, val);
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const int e = r * 64 + lane;
        if (e < L) { kb[e] = key[r]; ib[e] = val[r]; }
    }
    return __shfl(key[(L - 1) >> 6], (L - 1) & 63, 64);
}

// grid.x = ceil(m / (4*T)), grid.y = S splits of `chunks_per_split` 64-reference chunks.
// GP >= g: padded dimensionality (register array for one chunk of references; the wave's T target rows
// sit in LDS for the whole kernel and are read back as broadcasts).
template <int GP, int EPL>
__global__ __launch_bounds__(256) void cbf_filter_kernel(const float2 *__restrict__ xq, int64_t m,
                                                         const float *__restrict__ ycf, int64_t n, int g,
                                                         const uint8_t *__restrict__ mask, int64_t n_chunks,
                                                         int64_t chunks_per_split, float slack, float plateau,
                                                         uint32_t *__restrict__ cand_idx, float *__restrict__ cand_tau)
{
    constexpr int CAP = 64 * EPL, L = 32 * EPL, T = CBF_T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // per wave: xs [T][GP] float2 | keys [T][CAP] f32 | idx [T][CAP] u32 | tau [T] f32 | cnt [T] i32
    constexpr size_t WAVE_BYTES = (size_t)T * GP * 8 + (size_t)T * CAP * 8 + T * 8;
    unsigned char *wb = smem_raw + (size_t)wave * WAVE_BYTES;
    float2 *xs = reinterpret_cast<float2 *>(wb);
    float *keys = reinterpret_cast<float *>(xs + T * GP);
    uint32_t *idxs = reinterpret_cast<uint32_t *>(keys + T * CAP);
    float *tau = reinterpret_cast<float *>(idxs + T * CAP);
    int *cnt = reinterpret_cast<int *>(tau + T);

    const int S = gridDim.y;
    const int split = blockIdx.y;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * T;
    if (lane < T) { tau[lane] = __builtin_inff(); cnt[lane] = 0; }
    for (int e = lane; e < T * GP; e += 64) {
        const int64_t row = row0 + e / GP;
        xs[e] = row < m ? xq[row * GP + e % GP] : make_float2(0.0f, __builtin_inff());
    }
    // (wave-private LDS: DS operations of one wave execute in order, no barrier needed)
    const float below_plateau = __uint_as_float(__float_as_uint(plateau) - 1u);     // plateau > 0

    const int64_t c_begin = split * chunks_per_split;
    int64_t c_end = c_begin + chunks_per_split;
    if (c_end > n_chunks) c_end = n_chunks;
    for (int64_t chunk = c_begin; chunk < c_end; ++chunk) {
        const int64_t j = chunk * 64 + lane;
        const bool valid = (j < n) && !(mask && mask[j]);
        float yv[GP];
#pragma unroll
        for (int k = 0; k < GP; ++k) yv[k] = ycf[(chunk * GP + k) * 64 + lane];
        for (int t = 0; t < T; ++t) {
            const int64_t row = row0 + t;
            if (row >= m) break;
            const float2 *xr = xs + t * GP;               // same address in every lane: LDS broadcast
            float lb = 0.0f;
            int n_out = 0;
#pragma unroll
            for (int k = 0; k < GP; ++k) {
                const float2 xt = xr[k];
                const float s = fabsf(xt.x) + fabsf(yv[k]);
                const float nlb = fmaxf(__builtin_fmaf(s, -2.5e-07f, fabsf(xt.x - yv[k])), 0.0f);   // 2.5e-7 > 2^-22
                // den >= (|x|+|y|+0.01) * (1 + 2^-21): the surplus pays for v_rcp_f32 (1 ulp) and the multiply;
                // nlb <= s < den, so q < 1 without a clamp
                const float den = __builtin_fmaf(s, 1.00000072f, 0.01000002f);
                const float q = nlb * __builtin_amdgcn_rcpf(den);
                const bool out = nlb >= xt.y;
                n_out += out ? 1 : 0;
                lb += out ? 1.0f : q;
            }
            const float key = (n_out == g) ? plateau : fminf(lb - slack, below_plateau);
            bool pend = valid && (key < tau[t]);
            uint64_t pm = __builtin_amdgcn_ballot_w64(pend);
            while (pm != 0) {
                const int c = cnt[t];
                const int room = CAP - c;
                if (room == 0) {
                    float kr[EPL];
                    uint32_t vr[EPL];
                    const float nt = cbf_compact<EPL>(keys + t * CAP, idxs + t * CAP, c, kr, vr);
                    if (lane == 0) { tau[t] = nt; cnt[t] = L; }
                    pend = pend && (key < nt);
                } else {
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                    const bool take = pend && rank < room;
                    if (take) {
                        keys[t * CAP + c + rank] = key;
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
    // flush: the L smallest (key, index) per target; tau = L-th key if anything was ever dropped
    for (int t = 0; t < T; ++t) {
        const int64_t row = row0 + t;
        if (row >= m) break;
        float kr[EPL];
        uint32_t vr[EPL];
        const int c = cnt[t];
        const float nt = cbf_compact<EPL>(keys + t * CAP, idxs + t * CAP, c, kr, vr);
        float t_row = tau[t];
        if (c > L) t_row = nt;
        const int64_t o = (row * S + split) * (int64_t)L;
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const int e = r * 64 + lane;
            if (e < L) cand_idx[o + e] = vr[r];
        }
        if (lane == 0) cand_tau[row * S + split] = t_row;
    }
}

hipError_t cbf_pack_targets_launch(const double *X, int64_t m, int g, int gp, double f, float *xq, unsigned int *flag,
                                   hipStream_t st)
{
    const int64_t tot = m * gp;
    hipLaunchKernelGGL(cbf_pack_targets_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, X, m, g, gp, f,
                       reinterpret_cast<float2 *>(xq), flag);
    return hipGetLastError();
}

hipError_t cbf_pack_refs_launch(const double *Y, int64_t n, int g, int gp, float *ycf, unsigned int *flag,
                                hipStream_t st)
{
    const int64_t chunks = (n + 63) / 64;
    hipLaunchKernelGGL(cbf_pack_refs_kernel, dim3((unsigned)chunks), dim3(256), 0, st, Y, n, g, gp, ycf, flag);
    return hipGetLastError();
}

// instantiated padded dimensionalities (multiples of 8 up to 64, then of 16)
int cbf_pick_gp(int g)
{
    const int inst[] = {8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128};
    for (int v : inst)
        if (g <= v) return v;
    return -1;
}

// slack / plateau of the fp32 bound for g dimensions (host and device must agree: computed once, here)
void cbf_constants(int g, float *slack, float *plateau)
{
    const float s = (float)g * ((float)g + 2.0f) * 5.9604644775390625e-08f;      // g (g+2) 2^-24
    *slack = s;
    *plateau = (float)g - s;
}

template <int GP, int EPL>
static hipError_t cbf_launch_one(const float *xq, int64_t m, const float *ycf, int64_t n, int g, const uint8_t *mask,
                                 int S, uint32_t *cand_idx, float *cand_tau, hipStream_t st)
{
    const int64_t n_chunks = (n + 63) / 64;
    const int64_t cps = (n_chunks + S - 1) / S;
    float slack, plateau;
    cbf_constants(g, &slack, &plateau);
    const size_t lds = 4 * ((size_t)CBF_T * GP * 8 + (size_t)CBF_T * (64 * EPL) * 8 + CBF_T * 8);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&cbf_filter_kernel<GP, EPL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)((m + 4 * CBF_T - 1) / (4 * CBF_T)), S), block(256);
    hipLaunchKernelGGL((cbf_filter_kernel<GP, EPL>), grid, block, lds, st, reinterpret_cast<const float2 *>(xq), m, ycf, n,
                       g, mask, n_chunks, cps, slack, plateau, cand_idx, cand_tau);
    return hipGetLastError();
}

hipError_t cbf_filter_launch(int gp, int epl, const float *xq, int64_t m, const float *ycf, int64_t n, int g,
                             const uint8_t *mask, int S, uint32_t *cand_idx, float *cand_tau, hipStream_t st)
{
#define NABO_CBF(GPV)                                                                                          \
    case GPV:                                                                                                  \
        return epl == 1 ? cbf_launch_one<GPV, 1>(xq, m, ycf, n, g, mask, S, cand_idx, cand_tau, st)            \
                        : cbf_launch_one<GPV, 2>(xq, m, ycf, n, g, mask, S, cand_idx, cand_tau, st);
    switch (gp) {
        NABO_CBF(8)
        NABO_CBF(16)
        NABO_CBF(24)
        NABO_CBF(32)
        NABO_CBF(40)
        NABO_CBF(48)
        NABO_CBF(56)
        NABO_CBF(64)
        NABO_CBF(80)
        NABO_CBF(96)
        NABO_CBF(112)
        NABO_CBF(128)
    default:
        return hipErrorInvalidValue;
    }
#undef NABO_CBF
}

}  // namespace nabo
