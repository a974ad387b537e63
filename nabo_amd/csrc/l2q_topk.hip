// l2q_topk.hip -- the f16x3 Euclidean filter of l2h_topk.hip on the 16x16x32 MFMA shape (gfx950).
//
// Same arithmetic, same K-concatenated f16 operands (hi/lo split, norm slot; l2h_topk.hip's header), same staged
// candidate lists -- but the contraction runs on v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16.  Both
// shapes deliver 1024 flop per cycle and SIMD; what differs is the CLOCK the chip holds under them.  The f16 kernels
// are power-limited: with the matrix pipe 93 % busy the part runs the 32x32x16 stream at 1.75 GHz, and every stall
// removed from the loop comes back as a lower clock (GRBM_GUI_ACTIVE per ablation).  tools/
// mfma_clock_lab.hip (random operands, registers only, one wave per SIMD): 32x32x16 holds 1.66 GHz = 1657 TFLOP/s,
// 16x16x32 holds 2.05 GHz = 2008 TFLOP/s at the same cycles per flop (MI355X_MICROARCH.md, "DVFS give-back" item 7).
//
// Layout: a wave owns 128 target rows as EIGHT row-blocks of 16; a reference tile is 32 cells = two halves of 16.
//   v_mfma_f32_16x16x32_f16: A = 16 references x 32 slots, B = 32 slots x 16 targets, lane l supplies cell l & 15,
//   slots 8 (l >> 4) .. +7 of the step; D: lane l holds target l & 15, references 4 (l >> 4) + i, i < 4.
// A (row-block, tile) pair is 2 x KS MFMAs (KS = KC / 2 steps of 32 slots) into two accumulators (reference halves):
// lane l ends with 8 scores of ONE target row, references jb + (i & 3) + 16 (i >> 2), jb = 32 tile + 4 (l >> 4) --
// topk_lists.h's records hold 8 scores here (16 for the 32x32 kernels), four lanes share a row instead of two.
// Row-blocks are processed in pairs (4 accumulators, one filter branch per 20 MFMAs = 320 matrix-pipe cycles as in the
// 32x32 kernel).
#include <cstdio>
#include <cstdlib>

#include <hip/hip_fp16.h>

#include "knn_common.h"
#include "topk_lists.h"

namespace nabo {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int L2Q_NREC = 64;        // staging records (8 scores each) per wave
constexpr int L2Q_ROW = 33;         // list entries per row (odd)
#ifndef L2Q_HOME
#define L2Q_HOME 8                   // tiles of the home pre-pass (locality order)
#endif

struct qacc { f32x4 v[2][2]; };     // [row-block of the pair][reference half]

// One pair of row-blocks against one reference tile.  a[h][s]: reference half h, step s.  RELOAD: refill the tile's
// registers in place with tile `next`, behind their last reader.
template <int KS, bool RELOAD>
__device__ __forceinline__ qacc qchain(f16x8 (&a)[2][KS], const f16x8 (&b0)[KS], const f16x8 (&b1)[KS],
                                       const unsigned char *__restrict__ next, int lane)
{
    qacc acc;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc.v[r][h] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // The four accumulators one after another (chains of KS dependent MFMAs), not round-robin: 1.7 % faster at 1M x 1M,
    // as in tools/mfma_clock_lab.hip (the part holds a higher clock on this order).  In the refill chain a[0][*] was last
    // read by the third accumulator's chain, a[1][s-1] one MFMA ago.
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int s = 0; s < KS + ((RELOAD && c == 3) ? 1 : 0); ++s) {
            if (s < KS)
                acc.v[c >> 1][c & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c & 1][s], (c >> 1) ? b1[s] : b0[s], acc.v[c >> 1][c & 1], 0, 0, 0);
#ifndef NABO_L2H_NORELOAD
            if (RELOAD && c == 3) {
                __builtin_amdgcn_sched_barrier(0);
                if (s < KS) a[0][s] = reinterpret_cast<const f16x8 *>(next)[s * 64 + lane];
                if (s >= 1) a[1][s - 1] = reinterpret_cast<const f16x8 *>(next)[(KS + s - 1) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
    }
    return acc;
}

// Filter of a pair, in two parts.  EVAL (no control flow: hipcc schedules it between the MFMAs of the chain that is
// issued next -- placed after a refill chain, behind that chain's scheduling fences, it ran with the matrix pipe idle):
// lane minimum of the 8 scores of each row-block against its threshold.  STAGE: ONE wave-uniform branch on "any hit".
struct qverdict { float m0, m1; uint64_t any; };

template <int NB>
__device__ __forceinline__ qverdict qfilter_eval(const qacc &acc, int rb0, const float (&tauv)[NB])
{
    qverdict v;
    v.m0 = acc.v[0][0][0];
    v.m1 = acc.v[1][0][0];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        v.m0 = fminf(v.m0, acc.v[0][i >> 2][i & 3]);
        v.m1 = fminf(v.m1, acc.v[1][i >> 2][i & 3]);
    }
    v.any = __builtin_amdgcn_ballot_w64((v.m0 < tauv[rb0]) || (v.m1 < tauv[rb0 + 1]));
    return v;
}

template <typename C, int EPL, int NB, int NREC>
__device__ __forceinline__ void qfilter_stage(const qacc &acc, const qverdict &v, int rb0, uint32_t jb, unsigned char *w,
                                              uint32_t &scnt, int lkeep, float (&tauv)[NB])
{
    if (__builtin_expect(v.any != 0, 0)) {          // (unlikely: the staging code goes out of line, the common path falls through)
        NABO_PROF_T0();
        float s0[8], s1[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s0[i] = acc.v[0][0][i]; s0[4 + i] = acc.v[0][1][i];
            s1[i] = acc.v[1][0][i]; s1[4 + i] = acc.v[1][1][i];
        }
        stage_hits2<C, EPL, NB, NREC>(s0, v.m0, s1, v.m1, rb0, jb, w, scnt, lkeep, tauv);
        NABO_PROF_ADD(w, 0, 1);
        NABO_PROF_ADD(w, 1, NABO_PROF_DT() >> 4);
    }
}

template <typename C, int EPL, int NB, int NREC>
__device__ __forceinline__ void qfilter(const qacc &acc, int rb0, uint32_t jb, unsigned char *w, uint32_t &scnt, int lkeep,
                                        float (&tauv)[NB])
{
    const qverdict v = qfilter_eval<NB>(acc, rb0, tauv);
    qfilter_stage<C, EPL, NB, NREC>(acc, v, rb0, jb, w, scnt, lkeep, tauv);
}

// Grid: x = target super-blocks (4 waves x 128 rows), y = reference splits.
// HOMEP: the locality-ordered form (order.hip; off by default) -- compiled separately so that the default kernel carries
// none of its per-tile index arithmetic.
template <int KC, int EPL, int ROWN, bool HOMEP = false>
__global__ __launch_bounds__(256, 1) void l2q_topk_kernel(const unsigned char *__restrict__ Xpk,
                                                          const unsigned char *__restrict__ Ypk,
                                                          int tiles_per_split, int64_t tile_off, int lkeep,
                                                          uint32_t *__restrict__ cand_idx,
                                                          float *__restrict__ cand_key,
                                                          float *__restrict__ cand_tau, int64_t pad_tile, int dbg,
                                                          const int32_t *__restrict__ wave_start)
{
    constexpr int KS = KC / 2;                         // steps of 32 slots
    constexpr int NB = 8;                              // row-blocks of 16 targets per wave
    constexpr int NP = NB / 2;                         // pairs
    constexpr int NREC = L2Q_NREC;
    using C = ListCfg<EPL, ROWN, NB, NREC, 16>;
    constexpr int TB = KC * 1024;                      // bytes per packed 32-cell tile (targets and references alike)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int lane = lane_id();
    const int lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int split = blockIdx.y;
    const int S = gridDim.y;
    const int64_t ltile0 = ((int64_t)blockIdx.x * 4 + wave) * (NB / 2);      // in 32-row tiles
    const int64_t ttile0 = tile_off + ltile0;

    f16x8 xb[NB][KS];
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) {
        const f16x8 *p = reinterpret_cast<const f16x8 *>(Xpk + (ttile0 + (rb >> 1)) * TB);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xb[rb][s] = p[((rb & 1) * KS + s) * 64 + lane];
            asm volatile("" : "+a"(xb[rb][s]));        // pinned in AGPRs (see l2h_topk.hip)
        }
    }
    unsigned char *wl = smem_raw + (size_t)wave * C::BYTES;          // this wave's lists (topk_lists.h)
    float tauv[NB];
    const float tau0 = (dbg & 1) ? -__builtin_inff() : __builtin_inff();
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) tauv[rb] = tau0;
    uint32_t scnt = 0;
    lists_init<C>(wl, lkeep, tau0, (uint32_t)split * (uint32_t)tiles_per_split * 32u);

    const int t_begin = split * tiles_per_split;
    const int t_end = t_begin + tiles_per_split;
    // Locality order (order.hip): before the common stream a wave visits L2Q_HOME tiles around the tile that holds its
    // rows' neighbourhood (wave_start, per 128 target rows): its thresholds are near their final values when the stream
    // proper begins, and that stream still runs from the split's first tile in step with every other wave of the XCD
    // (one copy of the stream in L2 -- a cyclic start per wave gave 4x fewer episodes and a 14 % SLOWER kernel: every wave
    // then streams its own window).  Step ts of the loops below is tile tmap(ts): the home tiles, then the split's tiles
    // in order without them.
    int h0 = t_begin, H = 0;
    if (HOMEP && wave_start && tiles_per_split >= 4 * L2Q_HOME) {
        H = L2Q_HOME;
        h0 = __builtin_amdgcn_readfirstlane(wave_start[ttile0 / (NB / 2)]) - L2Q_HOME / 2;
        h0 = h0 < t_begin ? t_begin : (h0 > t_end - L2Q_HOME ? t_end - L2Q_HOME : h0);
    }
    auto tmap = [&](int ts) {
        if (!HOMEP) return ts;
        if (ts - t_begin < H) return h0 + (ts - t_begin);
        const int j = ts - H;
        return j < h0 ? j : j + H;
    };
    // past the split's last tile: an all-padding tile (+inf norms, nothing passes) -- the loop always runs two steps
    auto tile_ptr = [&](int ts) {
        const int t = tmap(ts);
        const int64_t tc = ts < t_end ? (int64_t)t : pad_tile;
        // dbg & 2 / dbg & 4 (timing experiments, garbage results): the stream wraps inside a window of 128 tiles (stays in
        // the XCD's L2) / of 2 tiles (stays in the CU's vector L1)
        return Ypk + ((dbg & 2) ? (int64_t)(t_begin + ((t - t_begin) & 127)) : (dbg & 4) ? (int64_t)(t_begin + ((t - t_begin) & 1)) : tc) * TB;
    };

    f16x8 a0[2][KS], a1[2][KS];
    {
        const f16x8 *p0 = reinterpret_cast<const f16x8 *>(tile_ptr(t_begin)), *p1 = reinterpret_cast<const f16x8 *>(tile_ptr(t_begin + 1));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < KS; ++s) { a0[h][s] = p0[(h * KS + s) * 64 + lane]; a1[h][s] = p1[(h * KS + s) * 64 + lane]; }
    }

    qacc accP;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) accP.v[r][h] = f32x4{__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};

    // all NP pair-chains of tile t on register set `a`; the last one refills `a` with tile t+2.  The previous pair's
    // verdict is computed "before" the chain (see qfilter_eval) and acted on after it.
    auto tile_step = [&](f16x8(&a)[2][KS], int t) {
        const unsigned char *next2 = tile_ptr(t + 2);
        qacc accA;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int prev = (p + NP - 1) % NP;
            const uint32_t jbp = (uint32_t)(tmap(p == 0 ? t - 1 : t) * 32 + 4 * lq);     // (a padding step stages nothing)
            if (p & 1) {
#ifndef NABO_L2H_NOFILTER
                const qverdict v = qfilter_eval<NB>(accA, 2 * prev, tauv);
#endif
                if (p == NP - 1) accP = qchain<KS, true>(a, xb[2 * p], xb[2 * p + 1], next2, lane);
                else accP = qchain<KS, false>(a, xb[2 * p], xb[2 * p + 1], next2, lane);
#ifndef NABO_L2H_NOFILTER
                qfilter_stage<C, EPL, NB, NREC>(accA, v, 2 * prev, jbp, wl, scnt, lkeep, tauv);
#else
                asm volatile("" ::"v"(accA.v[0][0]), "v"(accA.v[0][1]), "v"(accA.v[1][0]), "v"(accA.v[1][1]));
#endif
            } else {
#ifndef NABO_L2H_NOFILTER
                const qverdict v = qfilter_eval<NB>(accP, 2 * prev, tauv);
#endif
                if (p == NP - 1) accA = qchain<KS, true>(a, xb[2 * p], xb[2 * p + 1], next2, lane);
                else accA = qchain<KS, false>(a, xb[2 * p], xb[2 * p + 1], next2, lane);
#ifndef NABO_L2H_NOFILTER
                qfilter_stage<C, EPL, NB, NREC>(accP, v, 2 * prev, jbp, wl, scnt, lkeep, tauv);
#else
                asm volatile("" ::"v"(accP.v[0][0]), "v"(accP.v[0][1]), "v"(accP.v[1][0]), "v"(accP.v[1][1]));
#endif
            }
        }
    };
    // Peeled first pair of steps + padding tile: the wait hipcc places at the loop head must be vmcnt(19..10) (one set's
    // refills done, the other's in flight), see l2h_topk.hip.
    tile_step(a0, t_begin);
    tile_step(a1, t_begin + 1);
    for (int t = t_begin + 2; t < t_end; t += 2) {
        tile_step(a0, t);
        tile_step(a1, t + 1);
    }
    {
        const int tl = t_begin + ((tiles_per_split + 1) & ~1) - 1;          // the last step run (t_end - 1 or the padding step)
        qfilter<C, EPL, NB, NREC>(accP, NB - 2, (uint32_t)(tmap(tl) * 32 + 4 * lq), wl, scnt, lkeep, tauv);
    }

    lists_flush<C, EPL, NB>(wl, scnt, ltile0 * 32, split, S, lkeep, tauv, cand_idx, cand_key, cand_tau);
}

template <int KC, int EPL, int ROWN, bool HOMEP>
static hipError_t qlaunch_k(const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S, int gx,
                            int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                            int64_t pad_tile, hipStream_t st, const int32_t *wave_start)
{
    const int dbg = debug_ablate();
    constexpr size_t lds = (size_t)4 * ListCfg<EPL, ROWN, 8, L2Q_NREC, 16>::BYTES;
    static_assert(lds <= 163840, "LDS budget");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&l2q_topk_kernel<KC, EPL, ROWN, HOMEP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid(gx, S), block(256);
    hipLaunchKernelGGL((l2q_topk_kernel<KC, EPL, ROWN, HOMEP>), grid, block, lds, st, Xpk, Ypk, tiles_per_split, tile_off,
                       lkeep, cand_idx, cand_key, cand_tau, pad_tile, dbg, wave_start);
#ifdef NABO_LISTS_PROF
    {
        unsigned long long h[8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(nabo_lists_prof), sizeof(h));
        fprintf(stderr, "[lists prof, cumulative] episodes %llu (x16 cyc %llu) drains %llu (x16 cyc %llu) rounds %llu (%llu) "
                        "records %llu appended %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
#endif
    return hipGetLastError();
}

template <int KC, int EPL, int ROWN>
static hipError_t qlaunch_one(const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S, int gx,
                              int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                              int64_t pad_tile, hipStream_t st, const int32_t *wave_start)
{
    return wave_start ? qlaunch_k<KC, EPL, ROWN, true>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key,
                                                       cand_tau, pad_tile, st, wave_start)
                      : qlaunch_k<KC, EPL, ROWN, false>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key,
                                                        cand_tau, pad_tile, st, nullptr);
}

// 512 rows per workgroup, one workgroup per CU (one wave per SIMD with the whole register file), lists of <= 32 kept
// entries.  (Half the rows per wave at two waves per SIMD, so that one wave's hit episodes overlap the other's MFMAs, was
// built and measured at 2.2x the time: every wave streams its own copy of the reference tiles, and at twice the traffic
// -- 64 bytes per clock and CU -- the vector L1 is the limit.)
void l2q_topk_geometry(int kc, int *rows_per_wg, int *wg_per_cu, int *lkeep_max)
{
    (void)kc;
    *rows_per_wg = 4 * 128;
    *wg_per_cu = 1;
    *lkeep_max = L2Q_ROW < 32 ? L2Q_ROW : 32;
}

// steps of 16 slots of the f16x3 operands for g components: 3 (g + 1) slots, instantiated values only (pack.hip)
int l2q_pick_kc(int g)
{
    const int need = (3 * (g + 1) + 15) / 16;
    const int inst[] = {2, 4, 6, 8, 10, 12};
    for (int v : inst)
        if (need <= v) return v;
    return -1;          // g >= 64: the fp32 kernel
}

// steps of 16 slots of the one-product operands (g components + two norm slots + the error slot), even (KS = KC / 2
// steps of 32 slots), instantiated values only
int l2q_pick_kc1(int g)
{
    const int need = 2 * ((g + 3 + 31) / 32);
    return need <= 12 ? need : -1;
}

hipError_t l2q_topk_launch(int kc, const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S,
                           int gx, int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                           int64_t pad_tile, hipStream_t st, const int32_t *wave_start)
{
    if ((int64_t)tiles_per_split * 32 >= NABO_LIST_SPLIT_REFS) return hipErrorInvalidValue;   // topk_lists.h: 25 bits of offset per entry
#define NABO_Q(KCV) case KCV: return qlaunch_one<KCV, 1, L2Q_ROW>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, pad_tile, st, wave_start);
    switch (kc) {
        NABO_Q(2) NABO_Q(4) NABO_Q(6) NABO_Q(8) NABO_Q(10) NABO_Q(12)
    default: return hipErrorInvalidValue;
    }
#undef NABO_Q
}

}  // namespace nabo
