// topk_lists.h -- wave-private candidate lists in LDS shared by the Euclidean top-L kernels.
//
// Accumulator convention (v_mfma_f32_32x32x*): lane l holds, for target row (l & 31) of the wave's
// row-block, 16 scores a[r]; register r of lane half h = l >> 5 belongs to reference
// jb + cd_row(r, 0) where jb = first reference of the tile + 4*h.
#pragma once
#include "knn_common.h"

namespace nabo {

// Candidate-list geometry (per target row, ROW entries in the owning wave's LDS slice), with
// ph = (ROW - 2 - lkeep) / 2 pending slots per lane half (whatever the kept list does not need):
//   [0, lkeep)                      kept entries (the lkeep smallest seen so far, sorted)
//   [lkeep, lkeep+ph]               pending slots of lane half 0 (+1 scratch slot at index ph)
//   [lkeep+ph+1, lkeep+2ph+1]       pending slots of lane half 1 (+1 scratch slot)
// The second template parameter of everything below is ROW (entries per row in LDS).
template <int EPL, int ROWN>
struct ListCfg {
    static constexpr int LMAX = 32 * EPL;                // stride of the emitted candidate lists
    static constexpr int ROW = ROWN;
    static_assert(ROW - 2 <= 64 * EPL, "a row must fit one wave-wide sort");
    __device__ static int ph(int lkeep) { return (ROW - 2 - lkeep) >> 1; }     // needs lkeep <= ROW - 4
};

struct RowState {           // per lane; lanes l and l+32 hold the same tau / kc, their own pc
    float tau;              // nothing with score >= tau can still enter the kept list
    uint32_t pc;            // pending entries of this lane half
    uint32_t kc;            // kept entries
};

// Sort kept + both pending lists of one row, keep the lkeep smallest.  Returns the new kept
// count; `tau_out` is the key of rank lkeep-1 when at least lkeep entries exist.
template <int EPL, int ROWN>
__device__ __forceinline__ uint32_t compact_row(uint2 *rowbuf, uint32_t kc, uint32_t pa, uint32_t pb, int lkeep,
                                                float &tau_io, float (&key)[EPL], uint32_t (&val)[EPL])
{
    const int ph = ListCfg<EPL, ROWN>::ph(lkeep);
    const int lane = lane_id();
    const uint32_t total = kc + pa + pb;
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const uint32_t e = (uint32_t)(r * 64 + lane);
        key[r] = __builtin_inff();
        val[r] = 0xFFFFFFFFu;
        if (e < total) {
            uint32_t src = e;                                         // kept
            if (e >= kc) src = lkeep + (e - kc);                      // pending, half 0
            if (e >= kc + pa) src = lkeep + (ph + 1) + (e - kc - pa); // pending, half 1
            uint2 v = rowbuf[src];
            key[r] = __uint_as_float(v.x);
            val[r] = v.y;
        }
    }
    wave_bitonic_sort<EPL, float>(key, val);
    const uint32_t nk = total < (uint32_t)lkeep ? total : (uint32_t)lkeep;
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const uint32_t e = (uint32_t)(r * 64 + lane);
        if (e < nk) rowbuf[e] = make_uint2(__float_as_uint(key[r]), val[r]);
    }
    if (total >= (uint32_t)lkeep) {
        const int e = lkeep - 1;
        float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, key[0]), e & 63));
        if (EPL > 1 && e >= 64)
            t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, key[EPL - 1]), e & 63));
        tau_io = t;
    }
    return nk;
}

// Hit path.  lane l: target row (l & 31) of this row-block; register r of lane half h is
// reference jb + cd_row(r, 0) (jb already contains 4*h).  Per iteration every hitting lane appends
// its smallest outstanding score to its own pending list (one unconditional ds_write: lanes
// without a hit write their scratch slot), knocks that register out and looks again; rows whose
// pending list is full are compacted first.  No per-register branches: a VALU->SALU round trip
// costs more than the ~80 VALU instructions of an iteration.
template <int EPL, int ROWN>
__device__ __forceinline__ void slow_append(f32x16 a, float m, RowState &st, uint2 *blockbuf, uint32_t jb, int lkeep)
{
    constexpr int ROW = ListCfg<EPL, ROWN>::ROW;
    const int ph = ListCfg<EPL, ROWN>::ph(lkeep);
    const int lane = lane_id();
    const int tl = lane & 31, hh = lane >> 5;
    uint2 *sub = blockbuf + tl * ROW + lkeep + hh * (ph + 1);
    bool hit = m < st.tau;
    if (EPL == 1) {
        // (Short lists only: with 64-entry lists -- k' > 24 -- several hits per lane and full rows are common enough
        // that the test costs more than it saves: cosine d=100, k=50 at 1M x 1M 1680 vs 1660 ms.)
        // Fast path (nearly every episode): every hitting lane has exactly ONE score below its threshold and room for
        // it.  One compare per register finds the register (per lane) and, through the scalar unit, the number of
        // hits in the wave; if that equals the number of hitting lanes nothing else can be pending, and the
        // knock-out / second look of the general loop below (40 of its ~80 vector instructions) is not needed.
        // On gfx950 vector-ALU instructions are not hidden behind the fp32 MFMAs of the SIMD's other wave.
        uint32_t rs = 0;
        int total = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool below = a[r] < st.tau;
            total += __builtin_popcountll(__builtin_amdgcn_ballot_w64(below));
            rs = below ? (uint32_t)r : rs;
        }
        const uint64_t hm = __builtin_amdgcn_ballot_w64(hit);
        const uint64_t full = __builtin_amdgcn_ballot_w64(hit && st.pc >= (uint32_t)ph);
        if (total == __builtin_popcountll(hm) && full == 0) {
            const uint32_t slot = hit ? st.pc : (uint32_t)ph;
            sub[slot] = make_uint2(__float_as_uint(m), jb + (rs & 3u) + 8u * (rs >> 2));
            st.pc += hit ? 1u : 0u;
            return;
        }
    }
    for (;;) {
        const uint64_t fm = __builtin_amdgcn_ballot_w64(hit && st.pc >= (uint32_t)ph);
        if (fm != 0) {
            uint32_t rows = (uint32_t)fm | (uint32_t)(fm >> 32);
            while (rows) {
                const int row = __builtin_ctz(rows);
                rows &= rows - 1;
                const uint32_t kc = (uint32_t)__builtin_amdgcn_readlane((int)st.kc, row);
                const uint32_t pa = (uint32_t)__builtin_amdgcn_readlane((int)st.pc, row);
                const uint32_t pb = (uint32_t)__builtin_amdgcn_readlane((int)st.pc, row + 32);
                float nt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, st.tau), row));
                float key[EPL];
                uint32_t val[EPL];
                const uint32_t nk = compact_row<EPL, ROWN>(blockbuf + row * ROW, kc, pa, pb, lkeep, nt, key, val);
                if (tl == row) { st.tau = nt; st.pc = 0; st.kc = nk; }
            }
            hit = m < st.tau;
            if (__builtin_amdgcn_ballot_w64(hit) == 0) break;
            continue;
        }
        // register holding the lane minimum
        uint32_t rs = 0;
#pragma unroll
        for (int r = 15; r >= 1; --r) rs = (a[r] == m) ? (uint32_t)r : rs;
        const uint32_t slot = hit ? st.pc : (uint32_t)ph;
        sub[slot] = make_uint2(__float_as_uint(m), jb + (rs & 3u) + 8u * (rs >> 2));
        st.pc += hit ? 1u : 0u;
        // knock it out, look for another hit in the same lane
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = (hit && rs == (uint32_t)r) ? __builtin_inff() : a[r];
        m = a[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fminf(m, a[r]);
        hit = m < st.tau;
        if (__builtin_amdgcn_ballot_w64(hit) == 0) break;
    }
}

template <int EPL, int ROWN>
__device__ __forceinline__ void filter_and_append(const f32x16 &acc, RowState &st, uint2 *blockbuf, uint32_t jb,
                                                  int lkeep)
{
    float m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fminf(m, acc[r]);
    if (__builtin_amdgcn_ballot_w64(m < st.tau) != 0) slow_append<EPL, ROWN>(acc, m, st, blockbuf, jb, lkeep);
}

// Final flush of one row-block: sort every row, emit the kept candidate indices (+ tau).
template <int EPL, int ROWN>
__device__ __forceinline__ void flush_block(RowState &st, uint2 *blockbuf, int64_t lrow0, int split, int S, int lkeep,
                                            uint32_t *__restrict__ cand_idx, float *__restrict__ cand_key,
                                            float *__restrict__ cand_tau)
{
    constexpr int LMAX = ListCfg<EPL, ROWN>::LMAX, ROW = ListCfg<EPL, ROWN>::ROW;
    const int lane = lane_id();
    for (int row = 0; row < 32; ++row) {
        const uint32_t kc = (uint32_t)__builtin_amdgcn_readlane((int)st.kc, row);
        const uint32_t pa = (uint32_t)__builtin_amdgcn_readlane((int)st.pc, row);
        const uint32_t pb = (uint32_t)__builtin_amdgcn_readlane((int)st.pc, row + 32);
        float t_row = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, st.tau), row));
        float key[EPL];
        uint32_t val[EPL];
        const uint32_t nk = compact_row<EPL, ROWN>(blockbuf + row * ROW, kc, pa, pb, lkeep, t_row, key, val);
        const int64_t o = ((lrow0 + row) * S + split) * (int64_t)LMAX;
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const uint32_t e = (uint32_t)(r * 64 + lane);
            if (e < (uint32_t)LMAX) {
                cand_idx[o + e] = e < nk ? val[r] : 0xFFFFFFFFu;
                if (cand_key) cand_key[o + e] = e < nk ? key[r] : __builtin_inff();
            }
        }
        if (lane == 0) cand_tau[(lrow0 + row) * S + split] = t_row;
    }
}

}  // namespace nabo
