// topk_lists.h -- wave-private candidate lists in LDS shared by the Euclidean top-L kernels (l2_topk.hip,
// l2h_topk.hip, l2s_topk.hip): STAGED hits, BATCHED list updates.
//
// Accumulator convention (v_mfma_f32_32x32x*): lane l holds, for target row (l & 31) of a row-block, 16 scores a[i];
// register i of lane half h = l >> 5 belongs to reference jb + cd_row(i, 0) where jb = first reference of the tile + 4 h.
// (v_mfma_f32_16x16x32, l2q_topk.hip: row-blocks of 16 rows, 8 scores per lane and record -- ListCfg's RPBv.)
//
// A chain's filter is 8 v_min3 + 1 v_cmp against the row's threshold tau (one VGPR, lane = row).  What happens on a hit
// used to be the largest non-MFMA item of these kernels: the hitting lanes -- one or two of 64 -- searched their 16
// registers, appended to the row's pending list, knocked the score out and looked again while the whole wave waited
// (~70 vector instructions and two scalar round trips per episode; ~300 list entries per row over 1M references),
// and every full pending list cost a wave-wide sort of ONE row (4150 cycles measured).  Now:
//   * STAGE (the episode, ~20 instructions, no search, no loop): every hitting lane copies its 16 scores into a record
//     of the wave's staging area (4 ds_write_b128) with a header (row, first reference).
//   * DRAIN (when the staging area is full): ONE LANE PER RECORD.  A row's kept list is UNSORTED, its threshold is the
//     largest kept key and `pmax` its position.  A lane compares its record's 16 scores with the row's current
//     threshold (LDS: the authoritative copy) and, for each one below it, overwrites the row's largest kept entry and
//     rescans the row's lkeep keys (eight per LDS round trip) for the new maximum.  Lanes whose records belong to the
//     same row (the two lane halves of a row, consecutive tiles) take turns: every contender writes its lane id to the
//     row's `owner` word, whoever reads its own id back goes first.  Up to 64 records are processed at once; a drain
//     is a few rounds of ~10 LDS round trips whatever the number of records.
// The VGPR copy of a threshold is refreshed after every drain; between drains it is stale (too large), which only
// stages a few scores that the drain then drops.  Thresholds are exact after every drain (no pending lists whose
// entries do not count yet), so fewer scores are staged than with deferred compaction.
// Kept lists start as lkeep sentinel entries (+inf, no index): no "kept count"; a row that never sees lkeep real
// candidates ends with threshold +inf (= nothing was dropped).  Lists are emitted unsorted: refine.hip orders the
// candidates by their exact float64 distances anyway.
// AN ENTRY IS A DOUBLE.  Its high word is the fp32 key, its low word (slot << 25) | (reference - first reference of the
// split): as IEEE doubles such words order exactly as their fp32 keys do (sign-magnitude, the low word only breaks ties;
// +-inf keys are finite doubles, NaN keys never get in), so the rescan after a replacement -- new maximum AND its
// slot -- is ONE v_max_f64 per entry and one ds_read2_b64 per two, where key / position pairs took a compare, a
// maximum and a select per key plus the wait states between them (~40 instead of ~105 instructions per rescan; the
// rescans are most of what a drain costs).  25 bits of index: a split streams fewer than 2^25 - 1 references
// (api.hip raises the split count for larger sets); the split's first reference is added back when lists are emitted.
#pragma once
#include "knn_common.h"

namespace nabo {

// EPL: emitted candidate lists hold 32 * EPL entries; ROWN: entries per row in LDS (odd, >= lkeep); NB: row-blocks per
// wave; NREC: staging records per wave (<= 64: one lane per record); RPBv: target rows per row-block -- 32 for the
// 32x32 MFMA shapes (a lane holds 16 scores of row l & 31, references jb + (i & 3) + 8 (i >> 2)), 16 for 16x16x32
// (l2q_topk.hip: 8 scores of row l & 15, references jb + (i & 3) + 16 (i >> 2)).
// GRPv (l2c_topk.hip on its 64-entry lists): per-row GROUP maxima -- eight groups of eight slots, gmax[row][g] = the largest
// entry of slots 8 g .. 8 g + 7 (a double like the entries).  A replacement rescans its own group and the eight group
// maxima, 16 entries instead of 64: on these lists the rescans were most of a kernel that spent half its time on hits.
template <int EPL, int ROWN, int NB, int NREC, int RPBv = 32, bool GRPv = false>
struct ListCfg {
    static constexpr int LMAX = 32 * EPL;                // stride of the emitted candidate lists
    static constexpr int ROW = ROWN;
    static constexpr int RPB = RPBv;                     // rows per row-block
    static constexpr int RS = RPBv / 2;                  // scores per lane and record (64 lanes x RS = RPB rows x 32 references)
    static constexpr int JSTRIDE = RPBv == 32 ? 8 : 16;  // reference stride between a lane's groups of four scores
    static_assert(RPBv == 32 || RPBv == 16, "row-blocks of 32 or 16 rows");
    static constexpr int NROWS = NB * RPBv;
    static_assert(ROW % 2 == 1, "odd row stride: lane-per-row walks spread over the LDS banks");
    static_assert(NREC <= 64 && NREC % 2 == 0, "one lane per staged record");
    // per-wave LDS block (16-byte aligned; BYTES is a multiple of 16):
    //   srec [NREC][RS] f32 | rows [NROWS][ROW] uint2 (x = slot | offset, y = key) | shdr [NREC] uint2 | tauL [NROWS] f32 |
    //   pmax [NROWS] u32 | owner [NROWS] u32 | base u32
    static constexpr int OFF_ROWS = NREC * RS * 4;
    static constexpr int OFF_SHDR = OFF_ROWS + NROWS * ROW * 8;
    static constexpr int OFF_TAU = OFF_SHDR + NREC * 8;
    static constexpr int OFF_PMAX = OFF_TAU + NROWS * 4;
    static constexpr int OFF_OWNER = OFF_PMAX + NROWS * 4;
    static constexpr int OFF_BASE = OFF_OWNER + NROWS * 4;   // first reference of the split (entries hold offsets from it)
    static constexpr bool GRP = GRPv;
    static constexpr int NGRP = 8;                           // groups of eight slots: rows of up to 64 kept entries
    static_assert(!GRPv || (ROWN >= 64 && ROWN <= 65), "group maxima: 64 kept entries in eight groups of eight");
    static constexpr int OFF_GMAX = OFF_BASE + 16;           // gmax [NROWS][NGRP] doubles (GRP only)
    static constexpr int GMAX_BYTES = GRPv ? NROWS * NGRP * 8 : 0;
    __device__ static double *gmax(unsigned char *w) { return reinterpret_cast<double *>(w + OFF_GMAX); }
#ifdef NABO_LISTS_PROF
    static constexpr int OFF_PROF = OFF_BASE + 16 + GMAX_BYTES;   // 16 u32 event counters / cycle sums (profiling builds only)
    static constexpr int BYTES = OFF_PROF + 64;
    __device__ static uint32_t *prof(unsigned char *w) { return reinterpret_cast<uint32_t *>(w + OFF_PROF); }
#else
    static constexpr int BYTES = OFF_BASE + 16 + GMAX_BYTES;
#endif
    static constexpr int IDX_BITS = 25;                      // entry low word: (slot << IDX_BITS) | offset of the reference
    static constexpr uint32_t IDX_MASK = (1u << IDX_BITS) - 1u;   // (all ones: no reference -- a sentinel entry)
    static_assert(ROWN <= 128, "seven bits of slot");
    __device__ static uint32_t *base(unsigned char *w) { return reinterpret_cast<uint32_t *>(w + OFF_BASE); }
    static_assert(BYTES % 16 == 0, "per-wave list block must keep 16-byte alignment");
    __device__ static float *srec(unsigned char *w) { return reinterpret_cast<float *>(w); }
    __device__ static uint2 *rows(unsigned char *w) { return reinterpret_cast<uint2 *>(w + OFF_ROWS); }
    __device__ static uint2 *shdr(unsigned char *w) { return reinterpret_cast<uint2 *>(w + OFF_SHDR); }
    __device__ static float *tauL(unsigned char *w) { return reinterpret_cast<float *>(w + OFF_TAU); }
    __device__ static uint32_t *pmax(unsigned char *w) { return reinterpret_cast<uint32_t *>(w + OFF_PMAX); }
    __device__ static uint32_t *owner(unsigned char *w) { return reinterpret_cast<uint32_t *>(w + OFF_OWNER); }
};

// Profiling builds (-DNABO_LISTS_PROF, tools only): per-wave event counts and shader-clock sums in LDS, added to a
// global array by lists_flush and printed by the launch wrapper: [0] episodes [1] cycles staging (drains inside an
// episode included) [2] drains [3] cycles draining [4] drain rounds [5] - [6] records [7] entries written to a list.
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

// Sentinel kept lists, thresholds +inf (tau0 = -inf: "no hits" timing experiments).
template <typename C>
__device__ __forceinline__ void lists_init(unsigned char *w, int lkeep, float tau0, uint32_t idx_base)
{
    const int lane = lane_id();
    uint2 *rows = C::rows(w);
    // slots past lkeep (ROW >= lkeep) hold -inf: never the maximum, never evicted, never emitted -- the rescan walks
    // whole compile-time batches without a bounds test per key
    for (int e = lane; e < C::NROWS * C::ROW; e += 64) {
        const int s = e % C::ROW;
        rows[e] = make_uint2(((uint32_t)s << C::IDX_BITS) | C::IDX_MASK, __float_as_uint(s < lkeep ? __builtin_inff() : -__builtin_inff()));
    }
    if (lane == 0) C::base(w)[0] = idx_base;
    if constexpr (C::GRP) {
        // a group's largest initial entry: its highest sentinel slot below lkeep (key +inf), or slot 8 g (key -inf) past it
        for (int e = lane; e < C::NROWS * C::NGRP; e += 64) {
            const int g = e % C::NGRP;
            const int top = lkeep - 1 < 8 * g + 7 ? lkeep - 1 : 8 * g + 7;
            const bool live = 8 * g < lkeep;
            const uint32_t slot = (uint32_t)(live ? top : 8 * g);
            const uint2 v = make_uint2((slot << C::IDX_BITS) | C::IDX_MASK, __float_as_uint(live ? __builtin_inff() : -__builtin_inff()));
            reinterpret_cast<uint2 *>(C::gmax(w))[e] = v;
        }
    }
    for (int r = lane; r < C::NROWS; r += 64) {
        C::tauL(w)[r] = tau0;
        C::pmax(w)[r] = 0u;
    }
#ifdef NABO_LISTS_PROF
    if (lane < 16) C::prof(w)[lane] = 0u;
#endif
}

// Batched list updates: one lane per staged record (see the header comment).
#ifndef NABO_RESCAN
#define NABO_RESCAN 12
#endif
// Largest entry of a row (as a double: the header comment) from slot I0 on, NABO_RESCAN entries per LDS round trip (a
// one-at-a-time scan is a chain of lkeep dependent LDS latencies).  Batches are compile-time pieces of the row's ROW
// slots -- the last one ends with the row -- and run while they begin below lkeep (wave-uniform): what a batch reads
// past lkeep has a -inf key (lists_init), so nothing needs a bounds test.
template <typename C, int I0>
__device__ __forceinline__ void lists_rescan(const uint2 *kept, int lkeep, double &best)
{
    if constexpr (I0 < C::ROW) {
        if (I0 < lkeep) {
            constexpr int LEN = C::ROW - I0 < NABO_RESCAN ? C::ROW - I0 : NABO_RESCAN;
            double e[LEN];
#pragma unroll
            for (int j = 0; j < LEN; ++j) e[j] = reinterpret_cast<const double *>(kept)[I0 + j];
            double other = e[LEN - 1];                        // two chains: a v_max_f64 need not wait for the one before it
#pragma unroll
            for (int j = 0; j + 1 < LEN; ++j) {
                if (j & 1) other = __builtin_fmax(other, e[j]);
                else best = __builtin_fmax(best, e[j]);
            }
            best = __builtin_fmax(best, other);
            lists_rescan<C, I0 + LEN>(kept, lkeep, best);
        }
    }
}

template <typename C>
__device__ __forceinline__ void lists_drain_body(unsigned char *w, uint32_t scnt, int lkeep)
{
    const int lane = lane_id();
    // volatile: lanes of one wave hand rows over to each other through these words.  The arbitration stores the lane
    // id and reads the word back to learn WHICH lane's store the LDS kept -- without volatile hipcc forwards the stored
    // value to the load and every contender believes it won; a lane that lost a round must re-read the threshold and
    // the position the winner left behind, not reuse what it loaded a round earlier.
    // (LDS address space spelled out: a volatile access through a generic pointer compiles to flat_load / flat_store,
    // which count on vmcnt and made every drain wait for the reference-tile loads in flight)
    typedef __attribute__((address_space(3))) volatile float lds_vf32;
    typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
    lds_vf32 *tauL = (lds_vf32 *)C::tauL(w);
    lds_vu32 *pmaxL = (lds_vu32 *)C::pmax(w);
    lds_vu32 *owner = (lds_vu32 *)C::owner(w);
    const bool mine = (uint32_t)lane < scnt;
    uint32_t row = 0, jb = 0, q = 0;
    const f32x4 *rp = reinterpret_cast<const f32x4 *>(C::srec(w) + (mine ? lane : 0) * C::RS);
    const float *rf = reinterpret_cast<const float *>(rp);
    if (mine) {
        const uint2 h = C::shdr(w)[lane];
        row = h.x;
        jb = h.y - C::base(w)[0];                            // entries hold offsets from the split's first reference
        const float t = tauL[row];                           // which scores are below the row's threshold right now
#pragma unroll
        for (int q4 = C::RS / 4 - 1; q4 >= 0; --q4) {
            const f32x4 v = rp[q4];
#pragma unroll
            for (int e = 3; e >= 0; --e) q = q + q + (v[e] < t ? 1u : 0u);       // bit 4 q4 + e
        }
    }
    uint2 *kept = C::rows(w) + row * C::ROW;
    while (__builtin_amdgcn_ballot_w64(q != 0) != 0) {
        NABO_PROF_ADD(w, 4, 1);
        // one contender per row goes now, the others in a later round
        if (q != 0) owner[row] = (uint32_t)lane;
        const bool go = q != 0 && owner[row] == (uint32_t)lane;
        float tau = go ? tauL[row] : 0.0f;
        uint32_t pm = go ? pmaxL[row] : 0u;
        uint32_t todo = go ? q : 0u;
        while (__builtin_amdgcn_ballot_w64(todo != 0) != 0) {
            bool repl = false;
            if (todo != 0) {
                const int i = __builtin_ctz(todo);
                todo &= todo - 1;
                const float key = rf[i];
                repl = key < tau;
                if (repl) {
                    // evict the largest kept entry
                    kept[pm] = make_uint2((pm << C::IDX_BITS) | (jb + (uint32_t)((i & 3) + C::JSTRIDE * (i >> 2))), __float_as_uint(key));
#ifdef NABO_LISTS_PROF
                    atomicAdd(&C::prof(w)[7], 1u);
#endif
                }
            }
            if (__builtin_amdgcn_ballot_w64(repl) != 0) {       // new maximum of the rows that changed
                double best = __builtin_bit_cast(double, (uint64_t)0xFF800000u << 32);       // key -inf
                if constexpr (C::GRP) {
                    // the replaced slot's group, then the eight group maxima (lanes without a replacement walk along on
                    // their stale position: harmless reads, their results are dropped below)
                    const uint32_t g = pm >> 3;
                    const double *ge = reinterpret_cast<const double *>(kept) + 8 * g;
                    double e8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) e8[j] = ge[j];
                    double gm = __builtin_fmax(__builtin_fmax(__builtin_fmax(e8[0], e8[1]), __builtin_fmax(e8[2], e8[3])),
                                               __builtin_fmax(__builtin_fmax(e8[4], e8[5]), __builtin_fmax(e8[6], e8[7])));
                    double *gmx = C::gmax(w) + row * C::NGRP;
                    if (repl) gmx[g] = gm;
                    double m8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) m8[j] = gmx[j];
                    best = __builtin_fmax(__builtin_fmax(__builtin_fmax(m8[0], m8[1]), __builtin_fmax(m8[2], m8[3])),
                                          __builtin_fmax(__builtin_fmax(m8[4], m8[5]), __builtin_fmax(m8[6], m8[7])));
                } else {
                    lists_rescan<C, 0>(kept, lkeep, best);
                }
                const uint64_t bb = __builtin_bit_cast(uint64_t, best);
                tau = repl ? __uint_as_float((uint32_t)(bb >> 32)) : tau;
                pm = repl ? (uint32_t)bb >> C::IDX_BITS : pm;
            }
        }
        if (go) {
            tauL[row] = tau;
            pmaxL[row] = pm;
            q = 0;
        }
    }
}

// NABO_DRAIN_CALL (defined by the including kernel file): the drain as a REAL function call on the wave's LDS offset
// (the callee still uses ds_ instructions) instead of an inlined copy at every filter site.
typedef __attribute__((address_space(3))) unsigned char lds_byte;

template <typename C>
__device__ __noinline__ void lists_drain_fn(uint32_t w_off, uint32_t scnt, int lkeep)
{
    lists_drain_body<C>((unsigned char *)(lds_byte *)(uintptr_t)w_off, scnt, lkeep);
}

// the drain alone (l2c_topk.hip refreshes its register copies of the thresholds once per tile, not per drain)
template <typename C>
__device__ __forceinline__ void lists_drain_only(unsigned char *w, uint32_t scnt, int lkeep)
{
    NABO_PROF_T0();
#ifdef NABO_DRAIN_CALL
    lists_drain_fn<C>((uint32_t)(uintptr_t)w, scnt, lkeep);
#else
    lists_drain_body<C>(w, scnt, lkeep);
#endif
    NABO_PROF_ADD(w, 2, 1);
    NABO_PROF_ADD(w, 3, NABO_PROF_DT() >> 4);
    NABO_PROF_ADD(w, 6, scnt);
}

// drain + refresh of the register copies of the thresholds (lane = row of its row-block)
template <typename C, int NB>
__device__ __forceinline__ void lists_drain(unsigned char *w, uint32_t scnt, int lkeep, float (&tauv)[NB])
{
    NABO_PROF_T0();
#ifdef NABO_DRAIN_CALL
    lists_drain_fn<C>((uint32_t)(uintptr_t)w, scnt, lkeep);
#else
    lists_drain_body<C>(w, scnt, lkeep);
#endif
    NABO_PROF_ADD(w, 2, 1);
    NABO_PROF_ADD(w, 3, NABO_PROF_DT() >> 4);
    NABO_PROF_ADD(w, 6, scnt);
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) tauv[rb] = C::tauL(w)[rb * C::RPB + (lane_id() & (C::RPB - 1))];
}

// The episode: hitting lanes park their scores (a[C::RS]); the staging area is drained when it is full.
template <typename C>
__device__ __forceinline__ void stage_write(const float (&a)[C::RS], uint32_t p, int rb, uint32_t jb, unsigned char *w)
{
    f32x4 *rp = reinterpret_cast<f32x4 *>(C::srec(w) + p * C::RS);
#pragma unroll
    for (int q4 = 0; q4 < C::RS / 4; ++q4) {
        f32x4 v;
        v[0] = a[4 * q4]; v[1] = a[4 * q4 + 1]; v[2] = a[4 * q4 + 2]; v[3] = a[4 * q4 + 3];
        rp[q4] = v;
    }
    C::shdr(w)[p] = make_uint2((uint32_t)(rb * C::RPB + (lane_id() & (C::RPB - 1))), jb);
}

template <typename C, int EPL, int NB, int NREC>
__device__ __forceinline__ void stage_hits(const float (&a)[C::RS], float m, int rb, uint32_t jb, unsigned char *w,
                                           uint32_t &scnt, int lkeep, float (&tauv)[NB])
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
            lists_drain<C, NB>(w, scnt, lkeep, tauv);
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

// Two row-blocks' hits in one episode (l2q_topk.hip: one filter branch covers a PAIR of 16-row blocks): one rank
// computation and one staging-area check for both; the rare overflow goes through the general path.
template <typename C, int EPL, int NB, int NREC>
__device__ __forceinline__ void stage_hits2(const float (&a0)[C::RS], float m0, const float (&a1)[C::RS], float m1, int rb0,
                                            uint32_t jb, unsigned char *w, uint32_t &scnt, int lkeep, float (&tauv)[NB])
{
    const bool h0 = m0 < tauv[rb0], h1 = m1 < tauv[rb0 + 1];
    const uint64_t b0 = __builtin_amdgcn_ballot_w64(h0), b1 = __builtin_amdgcn_ballot_w64(h1);
    const uint32_t n0 = (uint32_t)__builtin_popcountll(b0), n = n0 + (uint32_t)__builtin_popcountll(b1);
    if (scnt + n <= (uint32_t)NREC) {
        if (h0) stage_write<C>(a0, scnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)), rb0, jb, w);
        if (h1) stage_write<C>(a1, scnt + n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u)), rb0 + 1, jb, w);
        scnt += n;
        return;
    }
    stage_hits<C, EPL, NB, NREC>(a0, m0, rb0, jb, w, scnt, lkeep, tauv);
    stage_hits<C, EPL, NB, NREC>(a1, m1, rb0 + 1, jb, w, scnt, lkeep, tauv);
}

// A chain's filter (32x32 shapes): lane minimum of the 16 scores against the row's threshold; a wave-uniform branch on
// "any hit".
template <typename C, int EPL, int NB, int NREC>
__device__ __forceinline__ void filter_and_stage(const f32x16 &acc, int rb, uint32_t jb, unsigned char *w, uint32_t &scnt,
                                                 int lkeep, float (&tauv)[NB])
{
    static_assert(C::RS == 16, "32x32 accumulator layout");
    float m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fminf(m, acc[r]);
    if (__builtin_amdgcn_ballot_w64(m < tauv[rb]) != 0) {
        NABO_PROF_T0();
        float a[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = acc[r];
        stage_hits<C, EPL, NB, NREC>(a, m, rb, jb, w, scnt, lkeep, tauv);
        NABO_PROF_ADD(w, 0, 1);
        NABO_PROF_ADD(w, 1, NABO_PROF_DT() >> 4);           // (drains inside the episode are counted here too)
    }
}

// The same filter in two parts (l2h_topk.hip): the minimum / compare, free of control flow so that hipcc can schedule it
// between the MFMAs of the chain issued next, and the branch on its verdict.
struct FilterVerdict { float m; uint64_t any; };

template <int NB>
__device__ __forceinline__ FilterVerdict filter_eval(const f32x16 &acc, int rb, const float (&tauv)[NB])
{
    FilterVerdict v;
    v.m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) v.m = fminf(v.m, acc[r]);
    v.any = __builtin_amdgcn_ballot_w64(v.m < tauv[rb]);
    return v;
}

template <typename C, int EPL, int NB, int NREC>
__device__ __forceinline__ void filter_stage(const f32x16 &acc, const FilterVerdict &v, int rb, uint32_t jb, unsigned char *w,
                                             uint32_t &scnt, int lkeep, float (&tauv)[NB])
{
    static_assert(C::RS == 16, "32x32 accumulator layout");
    if (v.any != 0) {
        NABO_PROF_T0();
        float a[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = acc[r];
        stage_hits<C, EPL, NB, NREC>(a, v.m, rb, jb, w, scnt, lkeep, tauv);
        NABO_PROF_ADD(w, 0, 1);
        NABO_PROF_ADD(w, 1, NABO_PROF_DT() >> 4);
    }
}

// Final flush of a wave: drain what is staged, emit every row's kept candidate indices (+ threshold).
// lrow0: first row of the wave's first row-block, local to the launch (rows of a wave are consecutive).
template <typename C, int EPL, int NB>
__device__ __forceinline__ void lists_flush(unsigned char *w, uint32_t scnt, int64_t lrow0, int split, int S, int lkeep,
                                            float (&tauv)[NB], uint32_t *__restrict__ cand_idx,
                                            float *__restrict__ cand_key, float *__restrict__ cand_tau)
{
    const uint32_t idx_base = C::base(w)[0];
    constexpr int LMAX = C::LMAX;
    const int lane = lane_id();
    if (scnt > 0) lists_drain<C, NB>(w, scnt, lkeep, tauv);
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
                uint2 v = make_uint2(C::IDX_MASK, __float_as_uint(__builtin_inff()));
                if (e < (uint32_t)lkeep) v = rows[row * C::ROW + e];
                const uint32_t off = v.x & C::IDX_MASK;
                cand_idx[o + e] = off == C::IDX_MASK ? 0xFFFFFFFFu : idx_base + off;
                if (cand_key) cand_key[o + e] = __uint_as_float(v.y);
            }
        }
        if (lane == 0) cand_tau[(lrow0 + row) * S + split] = C::tauL(w)[row];
    }
}

}  // namespace nabo
