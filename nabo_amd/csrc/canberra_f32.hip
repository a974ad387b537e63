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
//   (the kernel tests the slightly stronger, s-free condition |fl32(x32 - y32)| >= thr2, see the pack kernel,
//    so that a 3-instruction counting pass can decide "definitely out" on its own)
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
// wave-uniform (its packed operands are LDS broadcasts); each wave owns T targets and their candidate lists
// (fp32 key + u32 index) in LDS.  See cbf_filter_kernel for the count / compact / bound structure.
#include <cstdlib>
#include "knn_common.h"

namespace nabo {


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
    // thr2: |fl32(x32 - y32)| >= thr2 implies nlb >= thr WITHOUT looking at s = |x32|+|y32| (header):
    // s <= (2|x32| + |d|)(1 + 2^-23), so nlb >= |d|(1 - 2.6e-7) - 5.2e-7 |x32|; the factors below are generous
    const float thr2 = (__uint_as_float(__float_as_uint(thr) + 1u) + 6e-07f * fabsf(x32)) * 1.000001f;
    xq[e] = make_float2(x32, __uint_as_float(__float_as_uint(thr2) + 1u));
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

// ---- operands of the counting pass: 7-bit integers, four dimensions per 32-bit word ---------------------------------
// The counting pass only has to PROVE "this dimension is out of window" for as many dimensions as it can; what it
// cannot prove merely counts as in-window.  That tolerates coarse arithmetic: every dimension k is quantised on the
// references' own range, q(y) = floor((y - min_k) / step_k) in [0, 127] (step_k = (max_k - min_k) / 127), and the
// reference's test  |x - y| >= f |x|  is bracketed by two integers per (target, dimension):
//   T+  = fl64(f |x|) (1 + 2^-52) (1 + 1e-12)      the reference's own threshold, padded for its fl64 |x - y|
//   qlo = floor((x - T+ - min_k) / step_k - margin)   q(y) <  qlo  =>  y <  min_k + qlo step_k <= x - T+   =>  out
//   qhi = ceil ((x + T+ - min_k) / step_k + margin) - 1   q(y) > qhi  =>  y >= min_k + (qhi + 1) step_k >= x + T+  =>  out
// (margin = 1e-6 (1 + |.|) buckets: far more than the float64 rounding of these expressions and of q(y) itself), both
// clamped to [0, 127] -- a clamp only ever turns "out" into "cannot tell".  In-window-possible <=> qlo <= q(y) <= qhi.
// Four dimensions share a word, one byte each with bit 7 as a guard:
//   references   yg = q | 0x80,   yc = (127 - q) | 0x80        (two planes per chunk)
//   targets      lo = qlo,        hc = 127 - qhi
//   A = yg - lo   has bit 7 of a byte set  <=>  q >= qlo       (128 + q - qlo >= 1: no borrow crosses a byte)
//   B = yc - hc   has bit 7 set            <=>  q <= qhi
//   in-window dimensions of the word = popcount(A & B & 0x80808080)
// i.e. v_sub_u32, v_sub_u32, v_bitop3_b32, v_bcnt_u32_b32 (accumulating) per FOUR dimensions and 64 references:
// 1.0 vector instruction per pair and dimension, exact integer arithmetic, where the packed-f16 form of the first
// version (v_pk_add, v_pk_fma clamp, v_dot2c per TWO dimensions) needed 1.5.  Padding dimensions are always in-window
// (q = 0, lo = 0, hc = 0).
__device__ __forceinline__ unsigned int cbf_ord(float v)                      // unsigned int that orders like the float
{
    const unsigned int b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// per-dimension min and max of the references as ordered float bits: colmm[k] = min, colmm[g + k] = max (fp32,
// widened to the enclosing float64 interval by the host).  A block reduces 256 rows in LDS first.
constexpr int CBF_COLMAX_ROWS = 256;
__global__ __launch_bounds__(256) void cbf_colminmax_kernel(const double *__restrict__ Y, int64_t n, int g,
                                                            unsigned int *__restrict__ colmm)
{
    extern __shared__ unsigned int smm[];                  // [g] min | [g] max
    for (int k = threadIdx.x; k < g; k += blockDim.x) { smm[k] = 0xFFFFFFFFu; smm[g + k] = 0u; }
    __syncthreads();
    const int64_t e0 = (int64_t)blockIdx.x * CBF_COLMAX_ROWS * g;
    int64_t cnt = (n - (int64_t)blockIdx.x * CBF_COLMAX_ROWS) * g;
    if (cnt > (int64_t)CBF_COLMAX_ROWS * g) cnt = (int64_t)CBF_COLMAX_ROWS * g;
    for (int64_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        const double y = Y[e0 + i];
        float lo = (float)y, hi = lo;                      // enclose y: round-to-nearest may land on either side
        if ((double)lo > y) lo = nextafterf(lo, -__builtin_inff());
        if ((double)hi < y) hi = nextafterf(hi, __builtin_inff());
        if (lo == lo && hi == hi) {
            atomicMin(&smm[(e0 + i) % g], cbf_ord(lo));
            atomicMax(&smm[g + (e0 + i) % g], cbf_ord(hi));
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < g; k += blockDim.x) {
        atomicMin(&colmm[k], smm[k]);
        atomicMax(&colmm[g + k], smm[g + k]);
    }
}

// refs: ych[(chunk * W + w) * 2 + plane][64], W = gp / 4 words; quant[k] = min_k, quant[g + k] = 1 / step_k (0: constant column)
__global__ void cbf_pack_refs8_kernel(const double *__restrict__ Y, int64_t n, int g, int gp,
                                      const double *__restrict__ quant, uint32_t *__restrict__ ych)
{
    const int64_t chunk = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t j = chunk * 64 + lane;
    const int W = gp / 4;
    for (int w = threadIdx.x >> 6; w < W; w += (blockDim.x >> 6)) {
        uint32_t yg = 0x80808080u, yc = 0x80808080u;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = 4 * w + b;
            int q = 0;
            if (j < n && k < g) {
                double u = (Y[j * g + k] - quant[k]) * quant[g + k];
                u = u < 0.0 ? 0.0 : (u > 127.0 ? 127.0 : u);          // (NaN stays out of range checks: flagged elsewhere)
                q = (int)u;
                q = q < 0 ? 0 : (q > 127 ? 127 : q);
            }
            yg |= (uint32_t)q << (8 * b);
            yc |= (uint32_t)((k < g ? 127 : 0) - (k < g ? q : 0)) << (8 * b);
        }
        ych[((chunk * W + w) * 2 + 0) * 64 + lane] = yg;
        ych[((chunk * W + w) * 2 + 1) * 64 + lane] = yc;
    }
}

// targets: xh[row][w] = (lo word, hc word)
__global__ void cbf_pack_targets8_kernel(const double *__restrict__ X, int64_t m, int g, int gp, double f,
                                         const double *__restrict__ quant, uint2 *__restrict__ xh)
{
    const int W = gp / 4;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * W) return;
    const int64_t row = e / W;
    const int w = (int)(e - row * W);
    uint32_t lw = 0, hw = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int k = 4 * w + b;
        int lo = 0, hc = 0;                                   // padding / "cannot tell": always in-window
        if (k < g) {
            const double x = X[row * g + k], inv = quant[g + k];
            const double tp = (f * fabs(x)) * (1.0 + 2.3e-16) * (1.0 + 1e-12);             // T+
            if (inv > 0.0 && tp == tp && tp < 1e300) {
                const double u = (x - tp - quant[k]) * inv, v = (x + tp - quant[k]) * inv;
                double ql = floor(u - 1e-6 * (1.0 + fabs(u)));
                double qh = ceil(v + 1e-6 * (1.0 + fabs(v))) - 1.0;
                ql = ql < 0.0 ? 0.0 : (ql > 127.0 ? 127.0 : ql);
                qh = qh < 0.0 ? 0.0 : (qh > 127.0 ? 127.0 : qh);
                if (ql == ql && qh == qh) { lo = (int)ql; hc = 127 - (int)qh; }
            }
        }
        lw |= (uint32_t)lo << (8 * b);
        hw |= (uint32_t)hc << (8 * b);
    }
    xh[e] = make_uint2(lw, hw);
}

// row-major fp32 copy of the references ([n][gp], zero padded) for the bound pass: a lane that evaluates one
// (target, reference) pair reads both rows with 16-byte loads instead of one 4-byte gather per dimension
__global__ void cbf_pack_refs_rows_kernel(const double *__restrict__ Y, int64_t n, int g, int gp, float *__restrict__ yrow)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * gp) return;
    const int64_t j = e / gp;
    const int k = (int)(e - j * gp);
    yrow[e] = k < g ? (float)Y[j * g + k] : 0.0f;
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
    wave_sort_f32<EPL>(key, val);
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const int e = r * 64 + lane;
        if (e < L) { kb[e] = key[r]; ib[e] = val[r]; }
    }
    return __shfl(key[(L - 1) >> 6], (L - 1) & 63, 64);
}

// grid.x = ceil(m / T), grid.y = S splits of `chunks_per_split` 64-reference chunks; one wave per workgroup.
// Each WAVE owns T target rows and their candidate lists (LDS, wave-private: no barriers anywhere) and
// streams the split's references past them: NCH chunks of 64 references (lane = reference) are resident in
// VGPRs per step as packed f16 pairs.  The targets' packed (x', thr') pairs sit in the wave's LDS and are read
// back as broadcasts (one ds_read_b64 per two dimensions and NCH chunks).
//
// Two passes:
//   1. COUNT, 7-bit integers, four dimensions per word (the operands described above the pack kernels): v_sub_u32,
//      v_sub_u32, v_bitop3_b32, v_bcnt_u32_b32 per word and chunk = 1.0 vector instruction per pair and dimension.
//      inw = dimensions that MAY be in the window; the others are PROVEN out, each adds exactly 1 to the reference's
//      distance and the rest add >= 0, so distance >= (counted dimensions - inw); a pair whose bound already reaches the
//      row's threshold is dropped here (all but ~1e-3 of the pairs once the lists have warmed up).  GWD = 1 drops the last
//      word from the count when it holds padding only (g <= GP - 4: g = 50 counts 13 words, not GP / 4 = 14).
//   2. BOUND (the fp32 ~15-slot divide-and-accumulate expression) only for the survivors, which are compacted
//      through a wave-private work list so that all 64 lanes of a batch carry a live pair: lane p takes pair
//      (t_p, j_p) and gathers x and y from the packed fp32 arrays (16-byte loads from the row-major fp32 copies; rare).
template <int GP, int EPL, int T, int NCH, int GWD>
__global__ __launch_bounds__(64, 2)
void cbf_filter_kernel(const float2 *__restrict__ xq, const uint2 *__restrict__ xh, int64_t m,
                       const float *__restrict__ yrow, const uint32_t *__restrict__ ych, int64_t n, int g,
                       const uint8_t *__restrict__ mask, int64_t n_chunks, int64_t chunks_per_split, float slack,
                       float plateau, uint32_t *__restrict__ cand_idx, float *__restrict__ cand_tau, int dbg)
{
    constexpr int L = 32 * EPL, CAP = L + 16 * EPL;     // kept + pending entries per list
    constexpr int GH = GP / 4;           // packed words: four dimensions each
    constexpr int GW = GH - GWD;         // words the counting pass looks at
    constexpr int GHS = GH + 1;          // LDS row stride (uint2): spreads the T rows over the banks
    constexpr int WLN = 512;             // work-list ring (entries); >= 63 + 64 * NCH
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = lane_id();
    // one wave per workgroup (nothing is shared between waves; small workgroups pack the CU's LDS better):
    //   xs [T][GHS] uint2 (f16 count operands) | xf [T][GPS] float2 (fp32 bound operands) |
    //   keys [T][CAP] f32 | idx [T][CAP] u32 | tau [T] f32 | cnt [T] i32 | wl_j [WLN] u32 | wl_t [WLN] u8
    constexpr int GPS = GP + 1;
    unsigned char *wb = smem_raw;
    uint2 *xs = reinterpret_cast<uint2 *>(wb);
    float2 *xf = reinterpret_cast<float2 *>(xs + T * GHS);
    float *keys = reinterpret_cast<float *>(xf + T * GPS);
    uint32_t *idxs = reinterpret_cast<uint32_t *>(keys + T * CAP);
    float *tau = reinterpret_cast<float *>(idxs + T * CAP);
    int *cnt = reinterpret_cast<int *>(tau + T);
    uint32_t *thr_l = reinterpret_cast<uint32_t *>(cnt + T);        // survivor threshold of the counting pass, per row
    uint32_t *wl = reinterpret_cast<uint32_t *>(thr_l + T);
    unsigned char *wl_t = reinterpret_cast<unsigned char *>(wl + WLN);

    const int S = gridDim.y;
    const int split = blockIdx.y;
    const int64_t row0 = (int64_t)blockIdx.x * T;
    if (row0 >= m) return;                                   // (no barriers in this kernel)
    // Survivor test of the counting pass, ONE compare per chunk.  n_out = 4 GW - inw dimensions are PROVEN out of window,
    // each adds exactly 1 to the reference's distance and the others add >= 0: distance >= n_out.  A pair is dropped when
    // n_out >= t1 = tau + slack (+2e-5, rounded up), i.e. when inw <= 4 GW - t1; the list key of a pair,
    //     (all g dimensions out) ? plateau : min(lower bound - slack, below_plateau),
    // is then >= tau as well (key >= n_out - slack for the first form, and the second only arises for tau <=
    // plateau).  Pairs on the boundary survive needlessly; pass 2 evaluates their key and applies `key < tau`.
    // The integer threshold is kept next to tau and recomputed only when tau changes (it cost 9 vector instructions
    // per row and chunk group inside the counting loop).
    auto count_threshold = [&](float tau_t) -> uint32_t {
        float t1 = tau_t + (2e-5f + slack);
        t1 = __uint_as_float(__float_as_uint(t1) + (t1 < __builtin_inff() ? 1u : 0u));           // next float up (tau_t > 0)
        const float need = (float)(4 * GW) - t1;               // survivors have inw > need (4 GW dimensions are counted)
        return need < 0.0f ? 0u : (uint32_t)(int)floorf(need) + 1u;                              // (t1 = +inf: need = -inf)
    };
    for (int e = lane; e < T; e += 64) { tau[e] = __builtin_inff(); cnt[e] = 0; thr_l[e] = 0u; }
    for (int e = lane; e < T * GH; e += 64) {
        const int64_t row = row0 + e / GH;
        xs[(e / GH) * GHS + e % GH] = row < m ? xh[row * GH + e % GH] : make_uint2(0u, 0u);
    }
    for (int e = lane; e < T * GP; e += 64) {
        const int64_t row = row0 + e / GP;
        xf[(e / GP) * GPS + e % GP] = row < m ? xq[row * GP + e % GP] : make_float2(0.0f, __builtin_inff());
    }
    // (wave-private LDS: DS operations of one wave execute in order, no barrier needed)
    const float below_plateau = __uint_as_float(__float_as_uint(plateau) - 1u);     // plateau > 0
    int t_cnt = T;                                            // live target rows of this wave
    if (row0 + T > m) t_cnt = (int)(m - row0);

    int wl_head = 0, wl_n = 0;                               // wave-uniform ring state

    // Pass 2 for `nb` (<= 64) work-list entries starting at wl_head: lane p evaluates the bound of pair
    // (t_p, j_p); x from the wave's fp32 LDS rows, y from the row-major fp32 references (one burst of 16-byte loads).
    auto drain = [&](int nb) {
        const bool act = lane < nb;
        const int slot = (wl_head + lane) & (WLN - 1);
        const uint32_t j = act ? wl[slot] : 0u;
        const uint32_t e = act ? (uint32_t)wl_t[slot] : 0u;
        wl_head = (wl_head + nb) & (WLN - 1);
        wl_n -= nb;
        if ((dbg & 4) && lane == 0) {                       // experiments: survivors / batches of the whole launch
            unsigned int *ctr = reinterpret_cast<unsigned int *>(cand_tau + m * gridDim.y);
            atomicAdd(ctr, (unsigned int)nb);
            atomicAdd(ctr + 1, 1u);
        }
        const int t_p = (int)(e & 0xFFu);
        const float2 *xp = xf + t_p * GPS;                                               // LDS, lane-varying row
        const float4 *yp = reinterpret_cast<const float4 *>(yrow + (int64_t)j * GP);     // 4 dimensions per 16 bytes
        float lb = 0.0f;
        int no_p = 0;
#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf) {                     // two bursts of GP/8 sixteen-byte loads (register budget)
            float4 yy[GP / 8];
#pragma unroll
            for (int q = 0; q < GP / 8; ++q) yy[q] = yp[hf * (GP / 8) + q];
            const float2 *xph = xp + hf * (GP / 2);
#pragma unroll
            for (int k = 0; k < GP / 2; ++k) {
                const float2 xt = xph[k];
                const float4 y4 = yy[k >> 2];
                const float y = (k & 3) == 0 ? y4.x : (k & 3) == 1 ? y4.y : (k & 3) == 2 ? y4.z : y4.w;
                const float s = fabsf(xt.x) + fabsf(y);
                const float ad = fabsf(xt.x - y);
                const float nlb = fmaxf(__builtin_fmaf(s, -2.5e-07f, ad), 0.0f);   // 2.5e-7 > 2^-22
                // den >= (|x|+|y|+0.01) * (1 + 2^-21): the surplus pays for v_rcp_f32 (1 ulp) and the multiply;
                // nlb <= s < den, so q < 1 without a clamp
                const float den = __builtin_fmaf(s, 1.00000072f, 0.01000002f);
                const float q = nlb * __builtin_amdgcn_rcpf(den);
                const bool out = ad >= xt.y;
                no_p += out ? 1 : 0;
                lb += out ? 1.0f : q;
            }
        }
        const float key = (no_p == g) ? plateau : fminf(lb - slack, below_plateau);
        const bool hit = act && (key < tau[t_p]);
        if (__builtin_amdgcn_ballot_w64(hit) == 0) return;
        for (int t2 = 0; t2 < t_cnt; ++t2) {
            bool pend = hit && (t_p == t2);
            uint64_t pm = __builtin_amdgcn_ballot_w64(pend);
            while (pm != 0) {
                const int c = cnt[t2];
                const int room = CAP - c;
                if (room == 0) {
                    float kr[EPL];
                    uint32_t vr[EPL];
                    const float nt = cbf_compact<EPL>(keys + t2 * CAP, idxs + t2 * CAP, c, kr, vr);
                    if (lane == 0) { tau[t2] = nt; cnt[t2] = L; thr_l[t2] = count_threshold(nt); }
                    pend = pend && (key < nt);
                } else {
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                    const bool take = pend && rank < room;
                    if (take) {
                        keys[t2 * CAP + c + rank] = key;
                        idxs[t2 * CAP + c + rank] = j;
                    }
                    const int np = __popcll(pm);
                    if (lane == 0) cnt[t2] = c + (np < room ? np : room);
                    pend = pend && !take;
                }
                pm = __builtin_amdgcn_ballot_w64(pend);
            }
        }
    };

    const int64_t c_begin = split * chunks_per_split;
    int64_t c_end = c_begin + chunks_per_split;
    if (c_end > n_chunks) c_end = n_chunks;
    for (int64_t chunk0 = c_begin; chunk0 < c_end; chunk0 += NCH) {
        uint32_t yv[NCH][2 * GW];                               // (yg, yc) words of NCH chunks, lane = reference
        uint64_t vmask[NCH];                                    // live (in range, not ignored) references of each chunk
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int64_t chunk = chunk0 + c;
            const int64_t j = chunk * 64 + lane;
            vmask[c] = __builtin_amdgcn_ballot_w64((chunk < c_end) && (j < n) && !(mask && mask[j]));
#pragma unroll
            for (int p = 0; p < 2 * GW; ++p)
                yv[c][p] = (chunk < c_end) ? ych[(chunk * 2 * GH + p) * 64 + lane] : 0x80808080u;
        }

        for (int t = 0; t < t_cnt; ++t) {
            const uint2 *xr = xs + t * GHS;               // same address in every lane: LDS broadcast
            uint32_t inw[NCH];                            // dimensions that may be in-window (padding included)
#pragma unroll
            for (int c = 0; c < NCH; ++c) inw[c] = 0u;
#pragma unroll
            for (int p = 0; p < GW; ++p) {
                const uint2 xt = xr[p];                   // (lo word, hc word)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const uint32_t a = yv[c][2 * p] - xt.x, b = yv[c][2 * p + 1] - xt.y;
                    inw[c] += (uint32_t)__builtin_popcount(__builtin_amdgcn_bitop3_b32(a, b, 0x80808080u, 0x80));
                }
            }
            const uint32_t thr_in = thr_l[t];                  // survivors have inw >= thr_in (count_threshold above)
            uint64_t sm[NCH];
            uint64_t any = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                sm[c] = __builtin_amdgcn_ballot_w64(inw[c] >= thr_in) & vmask[c];
                any |= sm[c];
            }
            if (any != 0 && !(dbg & 1)) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const uint64_t bm = sm[c];
                    if (bm != 0) {
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                        if ((bm >> lane) & 1ull) {
                            const int slot = (wl_head + wl_n + rank) & (WLN - 1);
                            wl[slot] = (uint32_t)((chunk0 + c) * 64 + lane);
                            wl_t[slot] = (unsigned char)t;
                        }
                        wl_n += __popcll(bm);
                    }
                }
                while (wl_n >= 64) drain(64);
            }
        }
    }
    while (wl_n > 0) drain(wl_n < 64 ? wl_n : 64);
    // flush: the L smallest (key, index) per target; tau = L-th key if anything was ever dropped
    for (int t = 0; t < t_cnt; ++t) {
        const int64_t row = row0 + t;
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

// lists written per (target row, split): the caller sizes cand_idx / cand_tau and tells refine
int cbf_lists_per_split() { return 1; }

constexpr int cbf_t_rows(int epl) { return epl == 1 ? 16 : 8; }     // target rows per wave

template <int GP, int EPL>
static hipError_t cbf_launch_one(const float *xq, const void *xh, int64_t m, const float *ycf, const void *ych, int64_t n,
                                 int g, const uint8_t *mask, int S, uint32_t *cand_idx, float *cand_tau, hipStream_t st)
{
    constexpr int T = cbf_t_rows(EPL);
    constexpr int NCH = GP <= 64 ? 4 : GP <= 96 ? 2 : 1;      // reference chunks resident in VGPRs (GP/2 registers each)
    constexpr int L = 32 * EPL, CAP = L + 16 * EPL;
    const int64_t n_chunks = (n + 63) / 64;
    const int64_t cps = (n_chunks + S - 1) / S;
    float slack, plateau;
    cbf_constants(g, &slack, &plateau);
    const size_t lds = (size_t)T * (GP / 4 + 1) * 8 + (size_t)T * (GP + 1) * 8 + (size_t)T * CAP * 8 + (size_t)T * 12 + 512 * 5;
    // the last count word holds padding only: leave it out (g = 50: 13 of 14 words, 7 % of the counting pass)
    // (not at GP = 8: LLVM's iterative-ilp scheduler crashes on the one-word instantiation)
    constexpr int GWD = GP >= 16 ? 1 : 0;
    const bool trim = GWD == 1 && g <= GP - 4;
    auto kern = trim ? &cbf_filter_kernel<GP, EPL, T, NCH, GWD> : &cbf_filter_kernel<GP, EPL, T, NCH, 0>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)((m + T - 1) / T), S), block(64);
    hipLaunchKernelGGL(kern, grid, block, lds, st, reinterpret_cast<const float2 *>(xq),
                       reinterpret_cast<const uint2 *>(xh), m, ycf, reinterpret_cast<const uint32_t *>(ych), n, g, mask,
                       n_chunks, cps, slack, plateau, cand_idx, cand_tau, debug_ablate());
    return hipGetLastError();
}

// target rows per workgroup (the caller's split heuristic needs it)
int cbf_rows_per_wg(int epl) { return epl == 1 ? cbf_t_rows(1) : cbf_t_rows(2); }

hipError_t cbf_colminmax_launch(const double *Y, int64_t n, int g, unsigned int *colmm, hipStream_t st)
{
    const int64_t blocks = (n + CBF_COLMAX_ROWS - 1) / CBF_COLMAX_ROWS;
    hipLaunchKernelGGL(cbf_colminmax_kernel, dim3((unsigned)blocks), dim3(256), (size_t)2 * g * sizeof(unsigned int), st, Y, n, g, colmm);
    return hipGetLastError();
}

hipError_t cbf_pack_refs_rows_launch(const double *Y, int64_t n, int g, int gp, float *yrow, hipStream_t st)
{
    const int64_t tot = n * gp;
    hipLaunchKernelGGL(cbf_pack_refs_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, Y, n, g, gp, yrow);
    return hipGetLastError();
}

hipError_t cbf_pack_refs8_launch(const double *Y, int64_t n, int g, int gp, const double *quant, void *ych, hipStream_t st)
{
    const int64_t chunks = (n + 63) / 64;
    hipLaunchKernelGGL(cbf_pack_refs8_kernel, dim3((unsigned)chunks), dim3(256), 0, st, Y, n, g, gp, quant,
                       reinterpret_cast<uint32_t *>(ych));
    return hipGetLastError();
}

hipError_t cbf_pack_targets8_launch(const double *X, int64_t m, int g, int gp, double f, const double *quant, void *xh,
                                    hipStream_t st)
{
    const int64_t tot = m * (gp / 4);
    hipLaunchKernelGGL(cbf_pack_targets8_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, X, m, g, gp, f,
                       quant, reinterpret_cast<uint2 *>(xh));
    return hipGetLastError();
}

hipError_t cbf_filter_launch(int gp, int epl, const float *xq, const void *xh, int64_t m, const float *ycf,
                             const void *ych, int64_t n, int g, const uint8_t *mask, int S, uint32_t *cand_idx,
                             float *cand_tau, hipStream_t st)
{
#define NABO_CBF(GPV)                                                                                          \
    case GPV:                                                                                                  \
        return epl == 1 ? cbf_launch_one<GPV, 1>(xq, xh, m, ycf, ych, n, g, mask, S, cand_idx, cand_tau, st)  \
                        : cbf_launch_one<GPV, 2>(xq, xh, m, ycf, ych, n, g, mask, S, cand_idx, cand_tau, st);
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
