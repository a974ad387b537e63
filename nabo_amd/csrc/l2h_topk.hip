// l2h_topk.hip -- Euclidean score + top-L filter with the contraction on the f16 matrix pipe ("f16x3" split),
// every wave streaming its own copy of the reference tiles (gfx950).  The default filter for g < 64.
//
// Every centred, scaled component v is split v = hi + lo (two f16 values, 22 significant bits together); the -2 x.y
// term is accumulated in fp32 from the three products  hi_y*hi_x + lo_y*hi_x + hi_y*lo_x  (lo*lo <= 2^-22 |x||y| is
// dropped) on v_mfma_f32_32x32x16_f16 (32 cycles per instruction, K = 16).  Operands are K-CONCATENATED (the layout of
// l2s_topk.hip's pack kernels, shared by both kernels): a reference cell is ONE vector [hi | lo | hi] of 3 (g+1) slots,
// a target [hi | hi | lo] (x -2), slot g of every segment carrying the norm term, so a 32x32 tile is
// KC = ceil(3 (g+1) / 16) MFMAs starting from C = 0 -- 10 at g = 50 (12 with the three products padded separately:
// the first version of this kernel) against 25 fp32 MFMAs of 64 cycles: 320 vs 1600 matrix-pipe cycles.
// The score only FILTERS candidates: refine.hip recomputes them in float64 and certifies the row with an error bound
// that accounts for the split (api.hip), so results are the same bits as the fp32 path.
//
// Structure: R = 4 row-blocks per wave (B operands resident: 4 x KC x 4 VGPRs), one wave per SIMD (the whole 512-
// register file), wave-private candidate lists in LDS (topk_lists.h), no barriers.  Reference tiles stream L2 -> VGPR
// into TWO register sets used alternately (tile t in one, tile t+1 in the other, loaded a whole tile ahead); a set is
// refilled in place with tile t+2, each register two MFMAs behind its last reader in the tile's last chain -- no
// register-to-register moves (the first version rotated a prefetch set into place: 8 v_mov per chain).
#include <cstdio>
#include <cstdlib>

#include <hip/hip_fp16.h>

#include "knn_common.h"
#include "topk_lists.h"

namespace nabo {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int L2H_NREC = 56;        // staging records per wave (topk_lists.h)
constexpr int L2H_ROW = 33;         // list entries per row (odd): 4 waves x (128 rows x 33 entries + staging) = 154 KB
constexpr int L2H_LAG = 2;          // a register is refilled this many MFMAs behind its last reader (l2_topk.hip)

// One chain: 32 refs x 32 targets x KC steps.  RELOAD: refill the tile's registers with tile `next` behind their last use.
template <int KC, bool RELOAD>
__device__ __forceinline__ f32x16 hchain(f16x8 (&a)[KC], const f16x8 (&b)[KC], const unsigned char *__restrict__ next, int lane)
{
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < KC + (RELOAD ? L2H_LAG : 0); ++s) {
        if (s < KC) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[s], acc, 0, 0, 0);
        const int r = s - L2H_LAG;
#ifdef NABO_L2H_NORELOAD        // timing experiments: the same tile again and again (garbage results)
        if (false) {
#else
        if (RELOAD && r >= 0) {
#endif
            __builtin_amdgcn_sched_barrier(0);
            a[r] = reinterpret_cast<const f16x8 *>(next)[r * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return acc;
}

// Grid: x = target super-blocks (4 waves x R tiles of 32 rows), y = reference splits.
template <int KC, int R, int EPL, int ROWN>
__global__ __launch_bounds__(256, 1) void l2h_topk_kernel(const unsigned char *__restrict__ Xpk,
                                                          const unsigned char *__restrict__ Ypk,
                                                          int tiles_per_split, int64_t tile_off, int lkeep,
                                                          uint32_t *__restrict__ cand_idx,
                                                          float *__restrict__ cand_key,
                                                          float *__restrict__ cand_tau, int64_t pad_tile, int dbg)
{
    constexpr int NREC = L2H_NREC;
    using C = ListCfg<EPL, ROWN, R, NREC>;
    constexpr int TB = KC * 1024;                      // bytes per packed tile (targets and references alike)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int lane = lane_id();
    const int hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int split = blockIdx.y;
    const int S = gridDim.y;
    const int64_t ltile0 = ((int64_t)blockIdx.x * 4 + wave) * R;
    const int64_t ttile0 = tile_off + ltile0;

    f16x8 xb[R][KC];
#pragma unroll
    for (int rb = 0; rb < R; ++rb) {
        const f16x8 *p = reinterpret_cast<const f16x8 *>(Xpk + (ttile0 + rb) * TB);
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            xb[rb][s] = p[s * 64 + lane];
            // The wave's B operands (160 of its ~370 registers, read-only from here on) are pinned in AGPRs, which the
            // MFMA reads directly.  Left to itself hipcc keeps them in arch VGPRs, runs out, "spills" them to AGPRs
            // and copies them back before every use: 0.86 v_accvgpr_read per MFMA in the loop.
            asm volatile("" : "+a"(xb[rb][s]));
        }
    }
    unsigned char *wl = smem_raw + (size_t)wave * C::BYTES;          // this wave's lists (topk_lists.h)
    float tauv[R];
    const float tau0 = (dbg & 1) ? -__builtin_inff() : __builtin_inff();
#pragma unroll
    for (int rb = 0; rb < R; ++rb) tauv[rb] = tau0;
    uint32_t scnt = 0;
    lists_init<C>(wl, lkeep, tau0, (uint32_t)split * (uint32_t)tiles_per_split * 32u);

    const int t_begin = split * tiles_per_split;
    const int t_end = t_begin + tiles_per_split;
    // Past the split's last tile the stream continues with an all-padding tile (+inf norms: nothing passes the filter),
    // so the loop below always runs its two steps -- see the note on the loop-head wait.
    // dbg & 2 (timing experiments only, results are garbage): the stream wraps inside a 128-tile window that stays
    // in the XCD's L2 -- the kernel's time without any L2 miss
    auto tile_ptr = [&](int t) {
        const int64_t tc = t < t_end ? (int64_t)t : pad_tile;
        return Ypk + ((dbg & 2) ? (int64_t)(t_begin + ((t - t_begin) & 127)) : tc) * TB;
    };

    f16x8 a0[KC], a1[KC];
    {
        const f16x8 *p0 = reinterpret_cast<const f16x8 *>(tile_ptr(t_begin)), *p1 = reinterpret_cast<const f16x8 *>(tile_ptr(t_begin + 1));
#pragma unroll
        for (int s = 0; s < KC; ++s) { a0[s] = p0[s * 64 + lane]; a1[s] = p1[s * 64 + lane]; }
    }
    f32x16 accP;
#pragma unroll
    for (int r = 0; r < 16; ++r) accP[r] = __builtin_inff();       // inf < tau is false: nothing pending

    // all R chains of tile t on register set `a`; the last chain refills `a` with tile t+2.  The previous chain's filter
    // is evaluated "before" the chain (no control flow: hipcc interleaves it with the chain's MFMAs; placed after a
    // refill chain, behind that chain's scheduling fences, it ran with the matrix pipe idle) and acted on after it.
    auto tile_step = [&](f16x8(&a)[KC], int t) {
        const unsigned char *next2 = tile_ptr(t + 2);
        f32x16 accA;
#pragma unroll
        for (int rb = 0; rb < R; ++rb) {
            const int prev = (rb + R - 1) % R;
            const uint32_t jbp = (uint32_t)((rb == 0 ? t - 1 : t) * 32 + 4 * hh);
            f32x16 &cur = (rb & 1) ? accP : accA;
            f32x16 &old = (rb & 1) ? accA : accP;
#ifndef NABO_L2H_NOFILTER
            const FilterVerdict v = filter_eval<R>(old, prev, tauv);
#endif
            if (rb == R - 1) cur = hchain<KC, true>(a, xb[rb], next2, lane);
            else cur = hchain<KC, false>(a, xb[rb], next2, lane);
#ifndef NABO_L2H_NOFILTER
            filter_stage<C, EPL, R, NREC>(old, v, prev, jbp, wl, scnt, lkeep, tauv);
#else
            asm volatile("" ::"v"(old));        // timing experiments: chains only, the accumulators kept alive
#endif
        }
        if (R & 1) accP = accA;        // odd R: the pending chain is the one just computed
    };
    // The wait hipcc places at the loop head is the strictest over every way into it, and it must be vmcnt(19..10): set
    // a0's refills done, a1's (issued a few hundred cycles ago) still in flight.  Two ways in used to make it
    // vmcnt(9..0) -- every second tile then waited for loads just issued, 24 % of the kernel's cycles (found with
    // GRBM_GUI_ACTIVE per compile-time ablation): the prologue's loads (hipcc interleaves the two sets
    // and moves them across fences, they are `const __restrict__`), hence the peeled first pair of steps; and an
    // `if (t + 1 < t_end)` around the second step, which gave the flow graph a path back to the head with one set's
    // refills outstanding, hence the padding tile.
    tile_step(a0, t_begin);
    tile_step(a1, t_begin + 1);
    for (int t = t_begin + 2; t < t_end; t += 2) {
        tile_step(a0, t);
        tile_step(a1, t + 1);                      // t + 1 == t_end: the padding tile
    }
    filter_and_stage<C, EPL, R, NREC>(accP, R - 1, (uint32_t)((t_end - 1) * 32 + 4 * hh), wl, scnt, lkeep, tauv);

    lists_flush<C, EPL, R>(wl, scnt, ltile0 * 32, split, S, lkeep, tauv, cand_idx, cand_key, cand_tau);
}

template <int KC, int R, int EPL, int ROWN>
static hipError_t hlaunch_one(const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S, int gx,
                              int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                              int64_t pad_tile, hipStream_t st)
{
    const int dbg = debug_ablate();
    constexpr size_t lds = (size_t)4 * ListCfg<EPL, ROWN, R, L2H_NREC>::BYTES;
    static_assert(lds <= 163840, "LDS budget");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&l2h_topk_kernel<KC, R, EPL, ROWN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid(gx, S), block(256);
    hipLaunchKernelGGL((l2h_topk_kernel<KC, R, EPL, ROWN>), grid, block, lds, st, Xpk, Ypk, tiles_per_split, tile_off,
                       lkeep, cand_idx, cand_key, cand_tau, pad_tile, dbg);
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

// Instantiated for lists of <= 32 kept entries (k + drop_first <= 28): 4 row-blocks per wave, rows of 33 list entries
// + staging = 154 KB of LDS, one workgroup per CU.
void l2h_topk_geometry(int kc, int *rows_per_wg, int *wg_per_cu, int *lkeep_max)
{
    (void)kc;
    *rows_per_wg = 4 * 4 * 32;
    *wg_per_cu = 1;
    *lkeep_max = L2H_ROW < 32 ? L2H_ROW : 32;              // emitted lists hold 32
}

// steps of 16 slots for g components: 3 (g+1) slots, instantiated values only (the packed layout of l2s_topk.hip)
int l2h_pick_kc(int g)
{
    const int need = (3 * (g + 1) + 15) / 16;
    const int inst[] = {2, 4, 6, 8, 10, 12};
    for (int v : inst)
        if (need <= v) return v;
    return -1;          // g >= 64: use the fp32 kernel
}

hipError_t l2h_topk_launch(int kc, const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S,
                           int gx, int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                           int64_t pad_tile, hipStream_t st)
{
    if ((int64_t)tiles_per_split * 32 >= NABO_LIST_SPLIT_REFS) return hipErrorInvalidValue;   // topk_lists.h: 25 bits of offset per entry
#define NABO_H(KCV) case KCV: return hlaunch_one<KCV, 4, 1, L2H_ROW>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, pad_tile, st);
    switch (kc) {
        NABO_H(2) NABO_H(4) NABO_H(6) NABO_H(8) NABO_H(10) NABO_H(12)
    default: return hipErrorInvalidValue;
    }
#undef NABO_H
}

}  // namespace nabo
