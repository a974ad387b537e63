// topk_lists.h -- wave-private candidate lists in LDS shared by the Euclidean top-L kernels (l2_topk.hip,
// l2h_topk.hip, l2s_topk.hip): STAGED hits, BATCHED appends.
//
// Accumulator convention (v_mfma_f32_32x32x*): lane l holds, for target row (l & 31) of a row-block, 16 scores a[i];
// register i of lane half h = l >> 5 belongs to reference jb + cd_row(i, 0) where jb = first reference of the tile + 4 h.
//
// A chain's filter is 8 v_min3 + 1 v_cmp against the row's threshold tau (one VGPR, lane = row).  What happens on a hit
// used to be the largest non-MFMA item of these kernels: the hitting lanes -- one or two of 64 -- searched their 16
// registers, appended, knocked out and looked again while the whole wave waited (~70 vector instructions and two
// scalar round trips per episode, 255 episodes per row over 1M references).  Now:
//   * STAGE (the episode, ~20 instructions, no loop): every hitting lane copies its 16 scores into a record of the
//     wave's staging area (4 ds_write_b128) with a header (row, first reference).  Nothing is searched.
//   * DRAIN (when the staging area is full, every ~15 episodes): ONE LANE PER RECORD.  Each lane compares its record's
//     16 scores with the row's current threshold (from LDS: the authoritative copy), reserves slots in the row's
//     pending list with one LDS atomic add and writes the qualifying (score, reference) pairs -- up to 64 records
//     in parallel instead of one hit at a time.  Scores >= the row's threshold are dropped (the threshold only ever
//     decreases, so they stay outside the final list).  Entries that find the pending list full stay flagged in their
//     record and are retried after the compaction below.
//   * MERGE (end of every drain): ONE LANE PER ROW.  The kept list of a row is UNSORTED, its threshold is the largest
//     kept key and `pmax` its position; a lane walks its row's few pending entries and, for each one below the
//     threshold, overwrites the largest kept entry and rescans its lkeep keys for the new maximum (~3 lkeep
//     instructions, all 64 rows of a pass at once).  The first version sorted kept + pending wave-wide, one row at a
//     time, whenever a pending list filled up: 4150 cycles per row compaction (tools: -DNABO_LISTS_PROF), 44 % of the
//     whole hit path.  Merging after every drain also keeps the thresholds exact, so fewer scores are staged at all.
// The VGPR copy of a threshold is refreshed after every drain; between drains it is stale (too large), which only
// stages a few scores that the drain then drops.
// Kept lists start as lkeep sentinel entries (+inf, 0xFFFFFFFF): no separate "kept count", and a row that never sees
// lkeep real candidates ends with threshold +inf (= nothing was dropped), as before.
#pragma once
#include "knn_common.h"

namespace nabo {

// EPL: emitted candidate lists hold 32 * EPL entries; ROWN: entries per row in LDS (kept + pending, odd);
// NB: row-blocks per wave; NREC: staging records per wave.
template <int EPL, int ROWN, int NB, int NREC>
struct ListCfg {
    static constexpr int LMAX = 32 * EPL;                // stride of the emitted candidate lists
    static constexpr int ROW = ROWN;
    static constexpr int NROWS = NB * 32;
    static_assert(ROW % 2 == 1, "odd row stride (lane-per-row accesses)");
    static_assert(NREC <= 64 && NREC % 2 == 0, "one lane per staged record");
    // per-wave LDS block (16-byte aligned; BYTES is a multiple of 16):
    //   srec [NREC][16] f32 | rows [NROWS][ROW] uint2 | shdr [NREC] uint2 | pcnt [NROWS] u32 | tauL [NROWS] f32 |
    //   pmax [NROWS] u32.  ROW is odd: a lane-per-row walk then spreads over the LDS banks
    static constexpr int OFF_ROWS = NREC * 64;
    static constexpr int OFF_SHDR = OFF_ROWS + NROWS * ROW * 8;
    static constexpr int OFF_PCNT = OFF_SHDR + NREC * 8;
    static constexpr int OFF_TAU = OFF_PCNT + NROWS * 4;
    static constexpr int OFF_PMAX = OFF_TAU + NROWS * 4;
#ifdef NABO_LISTS_PROF
    static constexpr int OFF_PROF = OFF_PMAX + NROWS * 4;    // 16 u32 event counters / cycle sums (profiling builds only)
    static constexpr int BYTES = OFF_PROF + 64;
    __device__ static uint32_t *prof(unsigned char *w) { return reinterpret_cast<uint32_t *>(w + OFF_PROF); }
#else
    static constexpr int BYTES = OFF_PMAX + NROWS * 4;
#endif
    __device__ static uint32_t *pmax(unsigned char *w) { return reinterpret_cast<uint32_t *>(w + OFF_PMAX); }
    static_assert(BYTES % 16 == 0, "per-wave list block must keep 16-byte alignment");
    __device__ static float *srec(unsigned char *w) { return reinterpret_cast<float *>(w); }
    __device__ static uint2 *rows(unsigned char *w) { return reinterpret_cast<uint2 *>(w + OFF_ROWS); }
    __device__ static uint2 *shdr(unsigned char *w) { return reinterpret_cast<uint2 *>(w + OFF_SHDR); }
    __device__ static uint32_t *pcnt(unsigned char *w) { return reinterpret_cast<uint32_t *>(w + OFF_PCNT); }
    __device__ static float *tauL(unsigned char *w) { return reinterpret_cast<float *>(w + OFF_TAU); }
};

// Profiling builds (-DNABO_LISTS_PROF, tools only): per-wave event counts and shader-clock sums in LDS, added to a
// global array by lists_flush and printed by the launch wrapper: [0] episodes [1] cycles staging [2] drains
// [3] cycles draining (compactions included) [4] compactions [5] cycles compacting [6] records [7] entries appended.
#ifdef NABO_LISTS_PROF
static __device__ unsigned long long nabo_lists_prof[8];
#define NABO_PROF_ADD(w, i, v)                                                    \
    do {                                                                          \
        if (lane_id() == 0) atomicAdd(&C::prof(w)[i], (uint32_t)(v));            \
    } while (0)
#define NABO_PROF_T0() const uint64_t prof_t0 = __builtin_readcyclecounter()
#define NABO_PROF_DT() (uint32_t)(__builtin_readcyclecounter() - prof_t0)
#else
#define NABO_PROF_ADD(w, i, v) do { } while (0)
#define NABO_PROF_T0() do { } while (0)
#define NABO_PROF_DT() 0u
#endif

// Sentinel kept lists, empty pending lists, thresholds +inf (tau0 = -inf: "no hits" timing experiments).
template <typename C>
__device__ __forceinline__ void lists_init(unsigned char *w, int lkeep, float tau0)
{
    const int lane = lane_id();
    uint2 *rows = C::rows(w);
    for (int e = lane; e < C::NROWS * lkeep; e += 64) {
        const int r = e / lkeep, s = e - r * lkeep;
        rows[r * C::ROW + s] = make_uint2(__float_as_uint(__builtin_inff()), 0xFFFFFFFFu);
    }
    for (int r = lane; r < C::NROWS; r += 64) {
        C::pcnt(w)[r] = 0u;
        C::tauL(w)[r] = tau0;
        C::pmax(w)[r] = 0u;
    }
#ifdef NABO_LISTS_PROF
    if (lane < 16) C::prof(w)[lane] = 0u;
#endif
}

// MERGE: one lane per row (see the header comment).  Pending entries [lkeep, lkeep + min(pcnt, P)) of every row are
// folded into its unsorted kept list [0, lkeep); threshold = largest kept key, pmax = where it sits.
template <typename C>
__device__ __forceinline__ void merge_rows(unsigned char *w, int lkeep)
{
    const int lane = lane_id();
    const int P = C::ROW - lkeep;
#pragma unroll
    for (int base = 0; base < C::NROWS; base += 64) {
        const int r = base + lane < C::NROWS ? base + lane : C::NROWS - 1;
        uint32_t np = base + lane < C::NROWS ? C::pcnt(w)[r] : 0u;
        np = np < (uint32_t)P ? np : (uint32_t)P;              // reservations past the end were never written
        if (__builtin_amdgcn_ballot_w64(np != 0) == 0) continue;
        uint2 *kept = C::rows(w) + r * C::ROW;
        const uint2 *pend = kept + lkeep;
        float tau = C::tauL(w)[r];
        uint32_t pm = C::pmax(w)[r];
        for (uint32_t e = 0; __builtin_amdgcn_ballot_w64(e < np) != 0; ++e) {
            bool repl = false;
            if (e < np) {
                const uint2 v = pend[e];
                repl = __uint_as_float(v.x) < tau;
                if (repl) kept[pm] = v;                         // evict the largest kept entry
            }
            if (__builtin_amdgcn_ballot_w64(repl) != 0) {       // new maximum of the rows that changed
                // eight keys per round trip: a one-key-at-a-time scan is a chain of lkeep dependent LDS latencies
                // (12,900 cycles per merge measured); reads past lkeep (pending slots, the next row) are masked
                float t = -__builtin_inff();
                uint32_t p = 0;
                for (int i0 = 0; i0 < lkeep; i0 += 8) {
                    float kq[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) kq[j] = __uint_as_float(kept[i0 + j].x);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool gt = (i0 + j < lkeep) && (kq[j] > t);
                        t = gt ? kq[j] : t;
                        p = gt ? (uint32_t)(i0 + j) : p;
                    }
                }
                tau = repl ? t : tau;
                pm = repl ? p : pm;
            }
        }
        if (np != 0) {
            C::tauL(w)[r] = tau;
            C::pmax(w)[r] = pm;
            C::pcnt(w)[r] = 0u;
        }
    }
}

// Batched appends: one lane per staged record (see the header comment).  Written to need few registers: a record's
// 16 scores are read from LDS four at a time, once to count the qualifying ones and once more to write them.
template <typename C, int EPL>
__device__ __forceinline__ void lists_drain_body(unsigned char *w, uint32_t scnt, int lkeep)
{
    const int lane = lane_id();
    const int P = C::ROW - lkeep;
    uint2 *rows = C::rows(w);
    uint32_t *pcnt = C::pcnt(w);
    float *tauL = C::tauL(w);
    const bool mine = (uint32_t)lane < scnt;
    uint32_t row = 0, qm = 0, jb = 0;
    const f32x4 *rp = reinterpret_cast<const f32x4 *>(C::srec(w) + (mine ? lane : 0) * 16);
    if (mine) {
        const uint2 h = C::shdr(w)[lane];
        row = h.x & 0xFFu;
        qm = h.x >> 8;
        jb = h.y;
    }
    for (;;) {
        // which of my record's scores still qualify, and how many pending slots they need
        uint32_t q = 0;
        if (qm != 0) {
            const float t = tauL[row];
#pragma unroll
            for (int q4 = 3; q4 >= 0; --q4) {
                const f32x4 v = rp[q4];
#pragma unroll
                for (int e = 3; e >= 0; --e) q = q + q + (v[e] < t ? 1u : 0u);       // bit 4 q4 + e
            }
            q &= qm;
        }
        const int c = __builtin_popcount(q);
        uint32_t slot = 0;
        if (c > 0) slot = __hip_atomic_fetch_add(&pcnt[row], (uint32_t)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t left = 0;
#ifdef NABO_LISTS_PROF
        if (c > 0) atomicAdd(&C::prof(w)[7], (uint32_t)c);
#endif
        {   // write my qualifying scores, lowest register first: as many rounds as the busiest lane has entries
            // (usually one or two) instead of sixteen predicated blocks
            uint2 *pend = rows + row * C::ROW + lkeep;
            const float *rf = reinterpret_cast<const float *>(rp);
            uint32_t todo = q;
            while (__builtin_amdgcn_ballot_w64(todo != 0) != 0) {
                if (todo != 0) {
                    const int i = __builtin_ctz(todo);
                    todo &= todo - 1;
                    if (slot < (uint32_t)P) pend[slot] = make_uint2(__float_as_uint(rf[i]), jb + (uint32_t)((i & 3) + 8 * (i >> 2)));
                    else left |= 1u << i;
                    ++slot;
                }
            }
        }
        qm = left;                                       // qualified but found the row full: again after its compaction
        {   // fold every row's pending entries into its kept list (and empty the pending lists)
            NABO_PROF_T0();
            merge_rows<C>(w, lkeep);
            NABO_PROF_ADD(w, 4, 1);
            NABO_PROF_ADD(w, 5, NABO_PROF_DT() >> 4);
        }
        if (__builtin_amdgcn_ballot_w64(qm != 0) == 0) break;
    }
}

// NABO_DRAIN_CALL (defined by the including kernel file): the drain as a REAL function call on the wave's LDS offset
// (the callee still uses ds_ instructions) instead of an inlined copy at every filter site -- for kernels whose MFMA
// loop leaves no registers for the drain's temporaries.
typedef __attribute__((address_space(3))) unsigned char lds_byte;

template <typename C, int EPL>
__device__ __noinline__ void lists_drain_fn(uint32_t w_off, uint32_t scnt, int lkeep)
{
    lists_drain_body<C, EPL>((unsigned char *)(lds_byte *)(uintptr_t)w_off, scnt, lkeep);
}

// drain + refresh of the register copies of the thresholds (lane = row of its row-block)
template <typename C, int EPL, int NB>
__device__ __forceinline__ void lists_drain(unsigned char *w, uint32_t scnt, int lkeep, float (&tauv)[NB])
{
    NABO_PROF_T0();
#ifdef NABO_DRAIN_CALL
    lists_drain_fn<C, EPL>((uint32_t)(uintptr_t)w, scnt, lkeep);
#else
    lists_drain_body<C, EPL>(w, scnt, lkeep);
#endif
    NABO_PROF_ADD(w, 2, 1);
    NABO_PROF_ADD(w, 3, NABO_PROF_DT() >> 4);
    NABO_PROF_ADD(w, 6, scnt);
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) tauv[rb] = C::tauL(w)[rb * 32 + (lane_id() & 31)];
}

// The episode: hitting lanes park their 16 scores; the staging area is drained when it is full.
template <typename C>
__device__ __forceinline__ void stage_write(const f32x16 &a, uint32_t p, int rb, uint32_t jb, unsigned char *w)
{
    f32x4 *rp = reinterpret_cast<f32x4 *>(C::srec(w) + p * 16);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        f32x4 v;
        v[0] = a[4 * q4]; v[1] = a[4 * q4 + 1]; v[2] = a[4 * q4 + 2]; v[3] = a[4 * q4 + 3];
        rp[q4] = v;
    }
    C::shdr(w)[p] = make_uint2((uint32_t)(rb * 32 + (lane_id() & 31)) | (0xFFFFu << 8), jb);
}

template <typename C, int EPL, int NB, int NREC>
__device__ __forceinline__ void stage_hits(const f32x16 &a, float m, int rb, uint32_t jb, unsigned char *w, uint32_t &scnt,
                                           int lkeep, float (&tauv)[NB])
{
    bool hit = m < tauv[rb];
    {   // the usual episode: everything fits -- one ballot, one rank, the writes
        const uint64_t bm = __builtin_amdgcn_ballot_w64(hit);
        const uint32_t n = (uint32_t)__builtin_popcountll(bm);
        if (scnt + n <= (uint32_t)NREC) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            if (hit) stage_write<C>(a, scnt + rank, rb, jb, w);
            scnt += n;
            return;
        }
    }
    for (;;) {      // more hitting lanes than free records: fill, drain, look again with the new thresholds
        const uint64_t bm = __builtin_amdgcn_ballot_w64(hit);
        if (bm == 0) return;
        const uint32_t room = (uint32_t)NREC - scnt;
        if (room == 0) {
            lists_drain<C, EPL, NB>(w, scnt, lkeep, tauv);
            scnt = 0;
            hit = hit && (m < tauv[rb]);                 // the threshold may have come down
            continue;
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
        const bool take = hit && rank < room;
        if (take) stage_write<C>(a, scnt + rank, rb, jb, w);
        const uint32_t n = (uint32_t)__builtin_popcountll(bm);
        scnt += n < room ? n : room;
        hit = hit && !take;
    }
}

// A chain's filter: lane minimum of the 16 scores against the row's threshold; a wave-uniform branch on "any hit".
template <typename C, int EPL, int NB, int NREC>
__device__ __forceinline__ void filter_and_stage(const f32x16 &acc, int rb, uint32_t jb, unsigned char *w, uint32_t &scnt,
                                                 int lkeep, float (&tauv)[NB])
{
    float m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fminf(m, acc[r]);
    if (__builtin_amdgcn_ballot_w64(m < tauv[rb]) != 0) {
        NABO_PROF_T0();
        stage_hits<C, EPL, NB, NREC>(acc, m, rb, jb, w, scnt, lkeep, tauv);
        NABO_PROF_ADD(w, 0, 1);
        NABO_PROF_ADD(w, 1, NABO_PROF_DT() >> 4);           // (drains inside the episode are counted here too)
    }
}

// Final flush of a wave: drain what is staged, sort every row, emit the kept candidate indices (+ threshold).
// lrow0: first row of the wave's first row-block, local to the launch (rows of a wave are consecutive).
template <typename C, int EPL, int NB>
__device__ __forceinline__ void lists_flush(unsigned char *w, uint32_t scnt, int64_t lrow0, int split, int S, int lkeep,
                                            float (&tauv)[NB], uint32_t *__restrict__ cand_idx,
                                            float *__restrict__ cand_key, float *__restrict__ cand_tau)
{
    constexpr int LMAX = C::LMAX;
    const int lane = lane_id();
    if (scnt > 0) lists_drain<C, EPL, NB>(w, scnt, lkeep, tauv);
#ifdef NABO_LISTS_PROF
    if (lane < 8) atomicAdd(&nabo_lists_prof[lane], (unsigned long long)C::prof(w)[lane]);
#endif
    const uint2 *rows = C::rows(w);
    for (int row = 0; row < C::NROWS; ++row) {
        // kept entries in list order (unsorted: refine.hip orders candidates by their exact distances anyway)
        const int64_t o = ((lrow0 + row) * S + split) * (int64_t)LMAX;
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const uint32_t e = (uint32_t)(r * 64 + lane);
            if (e < (uint32_t)LMAX) {
                uint2 v = make_uint2(__float_as_uint(__builtin_inff()), 0xFFFFFFFFu);
                if (e < (uint32_t)lkeep) v = rows[row * C::ROW + e];
                cand_idx[o + e] = v.y;
                if (cand_key) cand_key[o + e] = __uint_as_float(v.x);
            }
        }
        if (lane == 0) cand_tau[(lrow0 + row) * S + split] = C::tauL(w)[row];
    }
}

}  // namespace nabo
