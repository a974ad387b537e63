// l2_topk.hip -- fused Euclidean score + per-target top-L candidate selection (gfx950).
//
// Replaces the arithmetic of nabo/_mapping.py:16-26 (_euclidean_dist) and the selection
// work of :139-145 (full argsort per row) for the ref<->ref path, WITHOUT materialising
// the m x n matrix.  This kernel produces, per target row, the L references with the
// smallest fp32 score  s = ||y||^2 - 2 x.y  (= d^2 - ||x||^2, same ranking as d) plus the
// threshold tau below which nothing was discarded; refine.hip re-evaluates the candidates
// in the reference's exact float64 arithmetic and certifies the result (guard) or sends the
// row to the exact fallback.  Indices are therefore bit-exact although this kernel is fp32.
//
// Mapping onto CDNA4
//   * the -2 x.y contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain):
//     A operand = 32 reference cells, B operand = 32 target cells (pre-scaled by -2),
//     C-in = ||y||^2 of the tile's references, so the accumulator IS the score;
//   * every wave owns R*32 target rows for the whole kernel: their B fragments
//     (R*KSTEPS VGPRs), thresholds and list counters live in registers; candidate lists live
//     in that wave's private LDS slice -> no barriers, no atomics, waves never synchronise;
//   * reference tiles stream straight from L2 into VGPRs as 16-byte-per-lane loads of the
//     pre-packed fragment layout (knn_common.h); the next tile is loaded in place while the
//     last chain of the current tile is still issuing (rolling prefetch);
//   * accumulator layout: lane (l&31) = target, register r / lane-half = reference, so a
//     row's threshold is ONE VGPR and the filter of a chain is 8 v_min3 + 1 v_cmp, scheduled
//     into the shadow of the next chain's MFMAs; only a hit takes the (out-of-line) append path;
//   * candidate lists per row: `lkeep` kept entries + a few pending slots per lane half; a lane
//     appends its own hits without talking to its partner half; a full pending list triggers a
//     wave-wide sort that keeps the lkeep smallest and tightens the threshold;
//   * two 4-wave workgroups fit a CU (2 x 80 KB LDS, <= 256 VGPRs), so each SIMD holds two waves
//     and one wave's append / compaction / L2 waits are covered by the other wave's MFMAs.
// Algorithmic work: 2*m*n*d flop on the MFMA pipe; HBM traffic is only the packed operands
// (4*KSTEPS*2 bytes per cell) and L indices per row -- the kernel is MFMA-bound.
#include <cstdlib>

#include "knn_common.h"
#include "topk_lists.h"

namespace nabo {

template <int KSTEPS>
struct RefTile {
    f32x4 f[q_groups(KSTEPS)];
    f32x16 n;
};

// Reference tiles are read with BUFFER loads: `buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen offset:imm` takes its
// lane address from ONE 32-bit VGPR (lane * 16) and everything uniform -- the tile's base in the descriptor, whole
// 4 KB blocks in the scalar offset -- from SGPRs.  With flat `global_load`s hipcc keeps 64-bit per-lane addresses in
// VGPR pairs for every load past the first 4 KB of a tile (and for all of them if asked nicely), and on gfx950, where
// the fp32 MFMA shares the vector ALU's operand paths, each such load costs ~20 ms per step at 1M x 1M
// (tools/ab: 7 of 11 loads per tile with VGPR-pair addresses 785 ms, all 11: 867 ms).
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float *tile_base)
{
    // raw buffer, no swizzle, no bounds (num_records = 2^31-1 bytes from the tile's own base); dword 3 as for
    // gfx90a / gfx94x / gfx950: DATA_FORMAT = 32-bit
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile_base), 0, 0x7FFFFFFF, 0x00020000);
}

// Load fragment group q of a packed tile: only the components that exist (the last group of a
// KSTEPS that is not a multiple of 4 is partial; loading its unused lanes would hand hipcc dead
// registers it reuses as temporaries -- and then guards with a full vmcnt(0)).
template <int KSTEPS>
__device__ __forceinline__ void load_group(f32x4 &dst, const float *__restrict__ tile_base, int q, int lane)
{
    constexpr int Q = q_groups(KSTEPS);
    constexpr int TAIL = KSTEPS - 4 * (Q - 1);
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(tile_base);
    const int voff = lane * 16 + (q & 3) * 1024, soff = (q >> 2) * 4096;
    if (q < Q - 1 || TAIL == 4) {
        dst = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    } else if (TAIL == 1) {
        dst[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
    } else if (TAIL == 2) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
        dst[0] = v[0]; dst[1] = v[1];
    } else {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
        dst[0] = v[0]; dst[1] = v[1];
        dst[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 8, soff, 0));
    }
}

template <int KSTEPS>
__device__ __forceinline__ f32x16 load_norm_block(const float *__restrict__ tile_base, int lane)
{
    constexpr int Q = q_groups(KSTEPS);
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(tile_base);
    const int voff = (lane >> 5) * 64;
    f32x16 n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * i, Q * 1024, 0));
        n[4 * i] = v[0]; n[4 * i + 1] = v[1]; n[4 * i + 2] = v[2]; n[4 * i + 3] = v[3];
    }
    return n;
}

template <int KSTEPS>
__device__ __forceinline__ void load_ref_tile(RefTile<KSTEPS> &y, const float *__restrict__ tile_base, int lane)
{
    constexpr int Q = q_groups(KSTEPS);
#pragma unroll
    for (int q = 0; q < Q; ++q) load_group<KSTEPS>(y.f[q], tile_base, q, lane);
    y.n = load_norm_block<KSTEPS>(tile_base, lane);
}

// One accumulation chain: 32 refs x 32 targets x (2*KSTEPS) components.
// RELOAD: overwrite the tile registers with `next` once their last reader is out of the matrix pipe.
//
// Where the refills sit matters more than anything else in this loop (hit-free kernel time at 1M x 1M x 50,
// tools/abl.sh history in DESIGN.md 4.1): a VMEM load whose destination is an operand of the MFMA that has just
// issued waits for that MFMA, and the next MFMA waits behind the load -- refilling each fragment group right
// behind its last reader cost 52 ms, no refills at all 678 ms against 776 ms.  Issued RELOAD_LAG MFMAs later the
// last reader has retired and the load goes out at once: 763 -> 711 ms hit-free, 818 -> 787 ms with hits
// (lag 1: 728 / 790, lag 2: 711 / 788, lag 3: 711 / 787, lag 4: 728).
constexpr int RELOAD_LAG = 2;

template <int KSTEPS, bool RELOAD>
__device__ __forceinline__ f32x16 mfma_chain(RefTile<KSTEPS> &y, const float (&xb)[KSTEPS],
                                             const float *__restrict__ next, int lane)
{
    constexpr int Q = q_groups(KSTEPS);
    f32x16 acc;
    if (RELOAD) {
        // C-in straight from the norm registers into a DIFFERENT destination.  Written as a builtin, hipcc ties
        // destination and C-in of this MFMA and copies the block first (8 v_mov_b64 per tile) -- and on gfx950 the
        // fp32 MFMA shares the vector ALU: every VALU cycle is a cycle the matrix chain does not get (tools/mfma_lab:
        // a 480-cycle VALU detour in one of a SIMD's two waves costs the SIMD 380 cycles).  The norm block is refilled
        // RELOAD_LAG MFMAs later, like the fragment groups.
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %3" : "=&v"(acc) : "v"(y.f[0][0]), "v"(xb[0]), "v"(y.n));
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.f[0][0], xb[0], y.n, 0, 0, 0);
    }
#pragma unroll
    for (int s = 1; s < KSTEPS + (RELOAD ? RELOAD_LAG : 0); ++s) {
        if (s < KSTEPS) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.f[s >> 2][s & 3], xb[s], acc, 0, 0, 0);
        const int r = s - RELOAD_LAG;           // MFMA r retired: if it was the last reader of its group, refill it
        if (RELOAD && r == 0) {
            __builtin_amdgcn_sched_barrier(0);
            y.n = load_norm_block<KSTEPS>(next, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (RELOAD && r >= 0 && ((r & 3) == 3 || r == KSTEPS - 1)) {
            __builtin_amdgcn_sched_barrier(0);
            load_group<KSTEPS>(y.f[r >> 2], next, r >> 2, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return acc;
}

// Grid: x = target super-blocks (4 waves x R tiles of 32 rows), y = reference splits.
// Xpk: packed target tiles (scaled by -2; padded tiles are zero); this launch covers tiles
//      [tile_off, tile_off + gridDim.x*4*R).
// Ypk: [S*tiles_per_split][rtile] packed reference tiles (padding: zero fragments, +inf norm).
// cand_idx/cand_key: [rows_launch][S][LMAX], cand_tau: [rows_launch][S] (rows local to the launch).
// Two waves per SIMD need <= 256 VGPRs; that holds while the resident target fragments
// (R*KSTEPS registers) stay <= 64 -- larger shapes run one wave per SIMD without spilling.
// staging records per wave (topk_lists.h): what the 80 KB (53 KB at three workgroups per CU) leave beside the rows
__host__ __device__ constexpr int l2_nrec(int r, int epl) { return epl == 2 ? 40 : r == 2 ? 36 : 32; }

template <int KSTEPS, int R, int EPL, int ROWN>
__global__ __launch_bounds__(256, (R * KSTEPS <= 25 && EPL == 1 ? 3 : R * KSTEPS <= 64 ? 2 : 1)) void l2_topk_kernel(const float *__restrict__ Xpk,
                                                                                  const float *__restrict__ Ypk,
                                                                                  int tiles_per_split,
                                                                                  int64_t tile_off, int lkeep,
                                                                                  uint32_t *__restrict__ cand_idx,
                                                                                  float *__restrict__ cand_key,
                                                                                  float *__restrict__ cand_tau,
                                                                                  int dbg /* ablation, 0 in production */)
{
    constexpr int L2_NREC = l2_nrec(R, EPL);
    using C = ListCfg<EPL, ROWN, R, L2_NREC>;
    constexpr int QTF = qtile_floats(KSTEPS);
    constexpr int RTF = rtile_floats(KSTEPS);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int lane = lane_id();
    const int hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int split = blockIdx.y;
    const int S = gridDim.y;
    const int64_t ltile0 = ((int64_t)blockIdx.x * 4 + wave) * R;     // local to this launch
    const int64_t ttile0 = tile_off + ltile0;

    // resident target fragments
    float xb[R][KSTEPS];
#pragma unroll
    for (int rb = 0; rb < R; ++rb) {
        const f32x4 *p = reinterpret_cast<const f32x4 *>(Xpk + (ttile0 + rb) * QTF);
#pragma unroll
        for (int q = 0; q < q_groups(KSTEPS); ++q) {
            f32x4 v = p[q * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < KSTEPS) xb[rb][4 * q + e] = v[e];
        }
    }
    unsigned char *wl = smem_raw + (size_t)wave * C::BYTES;          // this wave's lists (topk_lists.h)
    float tauv[R];
    const float tau0 = (dbg & 1) ? -__builtin_inff() : __builtin_inff();
#pragma unroll
    for (int rb = 0; rb < R; ++rb) tauv[rb] = tau0;
    uint32_t scnt = 0;
    lists_init<C>(wl, lkeep, tau0, (uint32_t)split * (uint32_t)tiles_per_split * 32u);

    // tile counters are 32-bit (n_ref < 2^32 - 16 => < 2^27 tiles): the loop test stays on the scalar unit
    const int t_begin = split * tiles_per_split;
    const int t_end = t_begin + tiles_per_split;
    const float *ybase = Ypk;

    RefTile<KSTEPS> y;
    load_ref_tile<KSTEPS>(y, ybase + (int64_t)t_begin * RTF, lane);

    f32x16 accP;                  // chain whose filter is still pending
#pragma unroll
    for (int r = 0; r < 16; ++r) accP[r] = __builtin_inff();   // inf < tau is false: nothing pending

    if (R == 2) {
        for (int t = t_begin; t < t_end; ++t) {
            const int tn = (t + 1 < t_end) ? t + 1 : t;
            f32x16 accA = mfma_chain<KSTEPS, false>(y, xb[0], nullptr, lane);
            filter_and_stage<C, EPL, R, L2_NREC>(accP, R - 1, ((uint32_t)(t - 1) * 32u + 4u * (uint32_t)hh), wl, scnt, lkeep, tauv);
            accP = mfma_chain<KSTEPS, true>(y, xb[R - 1], ybase + (int64_t)tn * RTF, lane);
            filter_and_stage<C, EPL, R, L2_NREC>(accA, 0, ((uint32_t)t * 32u + 4u * (uint32_t)hh), wl, scnt, lkeep, tauv);
        }
        filter_and_stage<C, EPL, R, L2_NREC>(accP, R - 1, ((uint32_t)(t_end - 1) * 32u + 4u * (uint32_t)hh), wl, scnt, lkeep, tauv);
    } else {
        for (int t = t_begin; t < t_end; ++t) {
            const int tn = (t + 1 < t_end) ? t + 1 : t;
            f32x16 accA = mfma_chain<KSTEPS, true>(y, xb[0], ybase + (int64_t)tn * RTF, lane);
            filter_and_stage<C, EPL, R, L2_NREC>(accP, 0, ((uint32_t)(t - 1) * 32u + 4u * (uint32_t)hh), wl, scnt, lkeep, tauv);
            accP = accA;
        }
        filter_and_stage<C, EPL, R, L2_NREC>(accP, 0, ((uint32_t)(t_end - 1) * 32u + 4u * (uint32_t)hh), wl, scnt, lkeep, tauv);
    }

    lists_flush<C, EPL, R>(wl, scnt, ltile0 * 32, split, S, lkeep, tauv, cand_idx, cand_key, cand_tau);
}

// ---- launch wrapper -------------------------------------------------------------------
template <int KSTEPS, int R, int EPL, int ROWN>
static hipError_t launch_one(const float *Xpk, const float *Ypk, int tiles_per_split, int S, int gx, int64_t tile_off,
                             int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau, hipStream_t st)
{
    const int dbg = debug_ablate();
    const size_t lds = (size_t)4 * ListCfg<EPL, ROWN, R, l2_nrec(R, EPL)>::BYTES;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&l2_topk_kernel<KSTEPS, R, EPL, ROWN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid(gx, S), block(256);
    hipLaunchKernelGGL((l2_topk_kernel<KSTEPS, R, EPL, ROWN>), grid, block, lds, st, Xpk, Ypk, tiles_per_split, tile_off,
                       lkeep, cand_idx, cand_key, cand_tau, dbg);
    return hipGetLastError();
}

// Rows per workgroup and co-resident workgroups per CU of the variant that will run.
void l2_topk_geometry(int ksteps, int epl, int *rows_per_wg, int *wg_per_cu, int *lkeep_max)
{
    int R = epl == 1 ? 2 : 1;
    *wg_per_cu = (R * ksteps <= 64) ? 2 : 1;
    *lkeep_max = epl == 2 ? 64 : 32;                   // row entries 33 / 65 (64-entry lists)
    if (epl == -1) { R = 1; *wg_per_cu = 3; }          // epl = -1: one row-block, three waves per SIMD
    *rows_per_wg = 4 * R * 32;
}

// ksteps must be one of the instantiated values; epl 1 -> lists of <= 32 (R=2), 2 -> <= 64 (R=1).
// Rows take 33 (65) entries + the per-wave staging area: two workgroups (2 x 80 KB) per CU.
hipError_t l2_topk_launch(int ksteps, int epl, const float *Xpk, const float *Ypk, int tiles_per_split, int S, int gx,
                          int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                          hipStream_t st)
{
    if ((int64_t)tiles_per_split * 32 >= NABO_LIST_SPLIT_REFS) return hipErrorInvalidValue;   // topk_lists.h: 25 bits of offset per entry
#define NABO_CASE(KS)                                                                                                 \
    case KS:                                                                                                          \
        if (epl == -1)                                                                                                \
            return launch_one<KS, 1, 1, 33>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key,      \
                                            cand_tau, st);                                                            \
        return epl == 1 ? launch_one<KS, 2, 1, 33>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, \
                                                  cand_tau, st)                                                       \
                        : launch_one<KS, 1, 2, 65>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, \
                                                  cand_tau, st);
    switch (ksteps) {
        NABO_CASE(8)
        NABO_CASE(16)
        NABO_CASE(25)
        NABO_CASE(32)
        NABO_CASE(50)
        NABO_CASE(64)
    default:
        return hipErrorInvalidValue;
    }
#undef NABO_CASE
}

}  // namespace nabo
