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
//   * candidate lists are L kept + PEND pending entries per row; two 4-wave workgroups fit a CU
//     (2 x 80 KB LDS, <= 256 VGPRs), so each SIMD holds two waves and one wave's append /
//     compaction / L2 waits are covered by the other wave's MFMAs.
// Algorithmic work: 2*m*n*d flop on the MFMA pipe; HBM traffic is only the packed operands
// (4*KSTEPS*2 bytes per cell) and L indices per row -- the kernel is MFMA-bound.
#include <cstdlib>

#include "knn_common.h"

namespace nabo {

template <int KSTEPS>
struct RefTile {
    f32x4 f[q_groups(KSTEPS)];
    f32x16 n;
};

// Load fragment group q of a packed tile: only the components that exist (the last group of a
// KSTEPS that is not a multiple of 4 is partial; loading its unused lanes would hand hipcc dead
// registers it reuses as temporaries -- and then guards with a full vmcnt(0)).
template <int KSTEPS>
__device__ __forceinline__ void load_group(f32x4 &dst, const float *__restrict__ tile_base, int q, int lane)
{
    constexpr int Q = q_groups(KSTEPS);
    constexpr int TAIL = KSTEPS - 4 * (Q - 1);
    const float *p = tile_base + (q * 64 + lane) * 4;
    if (q < Q - 1 || TAIL == 4) {
        dst = *reinterpret_cast<const f32x4 *>(p);
    } else if (TAIL == 1) {
        dst[0] = p[0];
    } else if (TAIL == 2) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 v = *reinterpret_cast<const f32x2 *>(p);
        dst[0] = v[0]; dst[1] = v[1];
    } else {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 v = *reinterpret_cast<const f32x2 *>(p);
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = p[2];
    }
}

template <int KSTEPS>
__device__ __forceinline__ void load_ref_tile(RefTile<KSTEPS> &y, const float *__restrict__ tile_base, int lane)
{
    constexpr int Q = q_groups(KSTEPS);
#pragma unroll
    for (int q = 0; q < Q; ++q) load_group<KSTEPS>(y.f[q], tile_base, q, lane);
    y.n = *reinterpret_cast<const f32x16 *>(tile_base + Q * 256 + (lane >> 5) * 16);
}

// One accumulation chain: 32 refs x 32 targets x (2*KSTEPS) components.
// RELOAD: overwrite the tile registers with `next` as soon as their last use has issued.
template <int KSTEPS, bool RELOAD>
__device__ __forceinline__ f32x16 mfma_chain(RefTile<KSTEPS> &y, const float (&xb)[KSTEPS],
                                             const float *__restrict__ next, int lane)
{
    constexpr int Q = q_groups(KSTEPS);
    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.f[0][0], xb[0], y.n, 0, 0, 0);
    // The sched_barriers pin each reload right behind the last MFMA that reads the old
    // registers: left alone, hipcc sinks some of them to the end of the chain and the next
    // chain then waits a full L2 round trip for its first fragment.
    if (RELOAD) {
        __builtin_amdgcn_sched_barrier(0);
        y.n = *reinterpret_cast<const f32x16 *>(next + Q * 256 + (lane >> 5) * 16);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 1; s < KSTEPS; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.f[s >> 2][s & 3], xb[s], acc, 0, 0, 0);
        if (RELOAD && ((s & 3) == 3 || s == KSTEPS - 1)) {
            __builtin_amdgcn_sched_barrier(0);
            load_group<KSTEPS>(y.f[s >> 2], next, s >> 2, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return acc;
}

// Two row-blocks against the same reference tile, MFMAs of the two accumulation chains
// interleaved: consecutive MFMAs are independent (no dependent-issue bubble) and the chain
// restart (C-in read, drain before the filter) is paid once per 2*KSTEPS MFMAs.  The tile
// registers are reloaded in place for the next tile behind their last use.
template <int KSTEPS>
__device__ __forceinline__ void mfma_chain_pair(RefTile<KSTEPS> &y, const float (&xb0)[KSTEPS],
                                                const float (&xb1)[KSTEPS], const float *__restrict__ next,
                                                int lane, f32x16 &accA, f32x16 &accB)
{
    constexpr int Q = q_groups(KSTEPS);
    accA = __builtin_amdgcn_mfma_f32_32x32x2f32(y.f[0][0], xb0[0], y.n, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x2f32(y.f[0][0], xb1[0], y.n, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    y.n = *reinterpret_cast<const f32x16 *>(next + Q * 256 + (lane >> 5) * 16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 1; s < KSTEPS; ++s) {
        accA = __builtin_amdgcn_mfma_f32_32x32x2f32(y.f[s >> 2][s & 3], xb0[s], accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_32x32x2f32(y.f[s >> 2][s & 3], xb1[s], accB, 0, 0, 0);
        if ((s & 3) == 3 || s == KSTEPS - 1) {
            __builtin_amdgcn_sched_barrier(0);
            load_group<KSTEPS>(y.f[s >> 2], next, s >> 2, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Candidate-list geometry: L = 32*EPL kept entries + PEND pending slots per row.
template <int EPL, int PEND>
struct ListCfg {
    static constexpr int L = 32 * EPL;
    static constexpr int CAP = L + PEND;
    static_assert(CAP <= 64 * EPL, "a row must fit one wave-wide sort");
};

// Sort one row's buffer, keep the L smallest, return the new threshold (key of rank L-1).
template <int EPL, int PEND>
__device__ __forceinline__ float compact_row(uint2 *rowbuf, uint32_t count, float (&key)[EPL], uint32_t (&val)[EPL])
{
    constexpr int L = ListCfg<EPL, PEND>::L;
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const int e = r * 64 + lane;
        key[r] = __builtin_inff();
        val[r] = 0xFFFFFFFFu;
        if ((uint32_t)e < count) {
            uint2 v = rowbuf[e];
            key[r] = __uint_as_float(v.x);
            val[r] = v.y;
        }
    }
    wave_bitonic_sort<EPL, float>(key, val);
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const int e = r * 64 + lane;
        if (e < L) rowbuf[e] = make_uint2(__float_as_uint(key[r]), val[r]);
    }
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, key[(L - 1) >> 6]), (L - 1) & 63));
}

// Append every accumulator element below its row's threshold to that row's list.
// lane l: target row (l & 31) of this row-block; register r / half (l >> 5): reference
// jbase + cd_row(r, l >> 5).  Lanes l and l+32 share a row and keep identical tau / cnt.
// A row whose list is full is compacted (sort, keep L, tighten tau) and its deferred hits are
// re-examined against the new threshold.
template <int EPL, int PEND>
__device__ __forceinline__ void append_hits(const f32x16 &acc, float &tau, uint32_t &cnt,
                                            uint2 *blockbuf /* this wave+rb: [32][CAP] */, uint32_t jbase)
{
    constexpr int CAP = ListCfg<EPL, PEND>::CAP, L = ListCfg<EPL, PEND>::L;
    const int lane = lane_id();
    const int tl = lane & 31;
    const int hh = lane >> 5;
    uint2 *rowbuf = blockbuf + tl * CAP;
    uint32_t pend = 0xFFFFu;
    for (;;) {
        uint32_t defer = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool hit = ((pend >> r) & 1u) && (acc[r] < tau);
            const uint64_t mask = __builtin_amdgcn_ballot_w64(hit);
            if (mask != 0) {
                const uint32_t h0 = ((uint32_t)mask >> tl) & 1u;
                const uint32_t h1 = ((uint32_t)(mask >> 32) >> tl) & 1u;
                const uint32_t pos = cnt + (hh ? h0 : 0u);
                const bool ok = hit && pos < (uint32_t)CAP;
                if (ok) rowbuf[pos] = make_uint2(__float_as_uint(acc[r]), jbase + (uint32_t)(cd_row(r, 0) + 4 * hh));
                if (hit && !ok) defer |= (1u << r);
                cnt = min(cnt + h0 + h1, (uint32_t)CAP);
            }
        }
        const uint64_t dm = __builtin_amdgcn_ballot_w64(defer != 0);
        if (dm == 0) break;
        uint32_t rows = (uint32_t)dm | (uint32_t)(dm >> 32);
        while (rows) {
            const int row = __builtin_ctz(rows);
            rows &= rows - 1;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cnt, row);
            float key[EPL];
            uint32_t val[EPL];
            const float nt = compact_row<EPL, PEND>(blockbuf + row * CAP, c, key, val);
            if (tl == row) { tau = nt; cnt = min(c, (uint32_t)L); }
        }
        pend = defer;
    }
}

template <int EPL, int PEND>
__device__ __forceinline__ void filter_and_append(const f32x16 &acc, float &tau, uint32_t &cnt,
                                                  uint2 *blockbuf, uint32_t jbase)
{
    bool any = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) any |= (acc[r] < tau);
    if (__builtin_amdgcn_ballot_w64(any) != 0) append_hits<EPL, PEND>(acc, tau, cnt, blockbuf, jbase);
}

// Final flush of one row-block: sort every row, emit L candidate indices (+ tau).
template <int EPL, int PEND>
__device__ __forceinline__ void flush_block(float &tau, uint32_t &cnt, uint2 *blockbuf,
                                            int64_t grow0, int split, int S,
                                            uint32_t *__restrict__ cand_idx, float *__restrict__ cand_key,
                                            float *__restrict__ cand_tau)
{
    constexpr int CAP = ListCfg<EPL, PEND>::CAP, L = ListCfg<EPL, PEND>::L;
    const int lane = lane_id();
    const int tl = lane & 31;
    for (int row = 0; row < 32; ++row) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cnt, row);
        float key[EPL];
        uint32_t val[EPL];
        float nt = compact_row<EPL, PEND>(blockbuf + row * CAP, c, key, val);
        float t_row = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tau), row));
        if (c > (uint32_t)L) t_row = nt;           // entries were dropped: threshold = rank L-1 key
        const int64_t o = ((grow0 + row) * S + split) * (int64_t)L;
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const int e = r * 64 + lane;
            if (e < L) {
                cand_idx[o + e] = val[r];
                if (cand_key) cand_key[o + e] = key[r];
            }
        }
        if (lane == 0) cand_tau[(grow0 + row) * S + split] = t_row;
        if (tl == row) { tau = t_row; cnt = min(c, (uint32_t)L); }
    }
}

// Grid: x = target super-blocks (4 waves x R tiles of 32 rows), y = reference splits.
// Xpk: [gridDim.x*4*R][qtile] packed target tiles (scaled by -2; padded tiles are zero).
// Ypk: [S*tiles_per_split][rtile] packed reference tiles (padding: zero fragments, +inf norm).
// cand_idx/cand_key: [rows_pad][S][L], cand_tau: [rows_pad][S], rows_pad = gridDim.x*4*R*32.
// Two waves per SIMD need <= 256 VGPRs; that holds while the resident target fragments
// (R*KSTEPS registers) stay <= 64 -- larger shapes run one wave per SIMD without spilling.
template <int KSTEPS, int R, int EPL, int PEND>
__global__ __launch_bounds__(256, (R * KSTEPS <= 64 ? 2 : 1)) void l2_topk_kernel(const float *__restrict__ Xpk,
                                                         const float *__restrict__ Ypk,
                                                         int tiles_per_split,
                                                         uint32_t *__restrict__ cand_idx,
                                                         float *__restrict__ cand_key,
                                                         float *__restrict__ cand_tau,
                                                         int dbg /* ablation switches, 0 in production */)
{
    constexpr int CAP = ListCfg<EPL, PEND>::CAP;
    constexpr int QTF = qtile_floats(KSTEPS);
    constexpr int RTF = rtile_floats(KSTEPS);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint2 *smem = reinterpret_cast<uint2 *>(smem_raw);

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int split = blockIdx.y;
    const int S = gridDim.y;
    const int64_t ttile0 = ((int64_t)blockIdx.x * 4 + wave) * R;

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
    float tau[R];
    uint32_t cnt[R];
#pragma unroll
    for (int rb = 0; rb < R; ++rb) { tau[rb] = (dbg & 1) ? -__builtin_inff() : __builtin_inff(); cnt[rb] = 0; }
    uint2 *wbuf = smem + (size_t)wave * R * 32 * CAP;

    const int64_t t_begin = (int64_t)split * tiles_per_split;
    const int64_t t_end = t_begin + tiles_per_split;
    const float *ybase = Ypk;

    RefTile<KSTEPS> y;
    load_ref_tile<KSTEPS>(y, ybase + t_begin * RTF, lane);
    if ((dbg & 4) && ((blockIdx.x >> 8) & 1)) {          // experiment: stagger the two workgroups of a CU
        for (int i = 0; i < (dbg >> 8); ++i) __builtin_amdgcn_s_sleep(16);
    }

    f32x16 accP;                  // chain whose filter is still pending
#pragma unroll
    for (int r = 0; r < 16; ++r) accP[r] = __builtin_inff();   // inf < tau is false: nothing pending

    if (R == 2) {
        for (int64_t t = t_begin; t < t_end; ++t) {
            const int64_t tn = (dbg & 2) ? t_begin + ((t + 1) & 63) : ((t + 1 < t_end) ? t + 1 : t);
            f32x16 accA, accB;
            mfma_chain_pair<KSTEPS>(y, xb[0], xb[R - 1], ybase + tn * RTF, lane, accA, accB);
            filter_and_append<EPL, PEND>(accA, tau[0], cnt[0], wbuf, (uint32_t)(t * 32));
            filter_and_append<EPL, PEND>(accB, tau[R - 1], cnt[R - 1], wbuf + (R - 1) * 32 * CAP, (uint32_t)(t * 32));
        }
    } else {
        for (int64_t t = t_begin; t < t_end; ++t) {
            const int64_t tn = (dbg & 2) ? t_begin + ((t + 1) & 63) : ((t + 1 < t_end) ? t + 1 : t);
            f32x16 accA = mfma_chain<KSTEPS, true>(y, xb[0], ybase + tn * RTF, lane);
            filter_and_append<EPL, PEND>(accP, tau[0], cnt[0], wbuf, (uint32_t)((t - 1) * 32));
            accP = accA;
        }
        filter_and_append<EPL, PEND>(accP, tau[0], cnt[0], wbuf, (uint32_t)((t_end - 1) * 32));
    }

#pragma unroll
    for (int rb = 0; rb < R; ++rb)
        flush_block<EPL, PEND>(tau[rb], cnt[rb], wbuf + rb * 32 * CAP, (ttile0 + rb) * 32, split, S,
                               cand_idx, cand_key, cand_tau);
}

// ---- launch wrapper -------------------------------------------------------------------
template <int KSTEPS, int R, int EPL, int PEND>
static hipError_t launch_one(const float *Xpk, const float *Ypk, int tiles_per_split, int S, int gx,
                             uint32_t *cand_idx, float *cand_key, float *cand_tau, hipStream_t st)
{
    static const int dbg = getenv("NABO_DEBUG_ABLATE") ? atoi(getenv("NABO_DEBUG_ABLATE")) : 0;
    const size_t lds = (size_t)4 * R * 32 * ListCfg<EPL, PEND>::CAP * sizeof(uint2);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&l2_topk_kernel<KSTEPS, R, EPL, PEND>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid(gx, S), block(256);
    hipLaunchKernelGGL((l2_topk_kernel<KSTEPS, R, EPL, PEND>), grid, block, lds, st, Xpk, Ypk, tiles_per_split,
                       cand_idx, cand_key, cand_tau, dbg);
    return hipGetLastError();
}

// ksteps must be one of the instantiated values; epl 1 -> L=32 (R=2), 2 -> L=64 (R=1).
// Lists are L + 8*EPL entries: two workgroups (2 x 80 KB) per CU.
hipError_t l2_topk_launch(int ksteps, int epl, const float *Xpk, const float *Ypk, int tiles_per_split,
                          int S, int gx, uint32_t *cand_idx, float *cand_key, float *cand_tau, hipStream_t st)
{
#define NABO_CASE(KS)                                                                                          \
    case KS:                                                                                                   \
        return epl == 1 ? launch_one<KS, 2, 1, 8>(Xpk, Ypk, tiles_per_split, S, gx, cand_idx, cand_key, cand_tau, st) \
                        : launch_one<KS, 1, 2, 16>(Xpk, Ypk, tiles_per_split, S, gx, cand_idx, cand_key, cand_tau, st);
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
