// l2c_topk.hip -- the ONE-PRODUCT Euclidean filter: the first pass of the default Euclidean / cosine query (gfx950).
//
// Operands: pack_ctiles_kernel<., true, 1> (l2s_topk.hip) -- a cell is ONE vector of g + 3 f16 slots,
//   references [hi (g) | nh | nl | ey],  targets [-2 hi (g) | 2^15 | 2^15 | -tx],
// so the contraction  ||rep_y||^2 - 2 hi_x.hi_y - tx ey  needs KS = ceil((g + 3) / 32) steps of v_mfma_f32_16x16x32_f16
// (2 at g = 50) where the f16x3 split of l2q_topk.hip needs 5.  tx ey >= the split's dropped terms (hi x lo, lo x hi,
// lo x lo), so the score is a rigorous LOWER bound of the f16x3 score: the same lists, the same certificate in
// refine.hip (with this kernel's accumulation coefficient), and a row the bound is too weak for goes on (api.hip:
// pass_level) -- first through THIS kernel again, seeded (tau_init: the row starts from the threshold its failed
// certificate implies and only collects what lies below it), then through the f16x3 / fp32 filter.  On 1M x 1M x 50 the
// bound costs 3 % more staged scores and 0.7 % of the rows.  KS = 1 .. 4: every g <= 125.
//
// What a tile costs was measured on the l2q kernel at KS = 2 (tools/r3_coarse_ab.sh, 1M x 1M): the bare MFMA loop
// 62 ms = the matrix pipe's rate, + 24 ms of filter instructions (14 per 8 MFMAs, bunched behind the chains they read),
// + 12-15 ms for the tile refills (fenced into the last chain of a tile), + 32 ms of hits -- nothing overlapped.  Hence:
//   * the filter of a pair of row-blocks is 10 instructions (2 x (3 v_min3 + v_min + v_cmp)): no NaN canonicalisation --
//     this file is compiled with -fno-honor-nans, and the operands make every score finite or +inf (padding TARGET rows
//     carry the norm slots too, a masked reference has no error slot).  (One minimum over both row-blocks against the
//     larger of the two thresholds is 9 -- and 1.5x the kernel time: a row's scores sit below its NEIGHBOUR's threshold
//     all the time, thresholds differ by the local density of the cells.)
//   * a ring of FOUR register sets for the reference tiles (a set is 4 KS registers): tile t + 3 is requested at the
//     top of step t, no scheduling fences -- three whole steps ahead instead of one;
//   * a tile's scores wait one step in registers: their filter reads nothing an MFMA in flight writes, and there is one
//     verdict branch per TILE (see the kernel body);
//   * MFMAs and filter instructions are hand-scheduled inline-assembly statements for KS = 2 and 4 (cpair: one or two
//     filter instructions behind each MFMA, an accumulator is ONE register quad from its first step to its last reader);
//     the other shapes leave the same loop to hipcc with scheduling groups;
//   * the staging / drain code sits out of line behind that unlikely branch, the drain itself behind a real call;
//   * every vector instruction next to the MFMAs costs its ~4 issue cycles whoever issues it (also a second wave of the
//     same SIMD): what counts is the instruction COUNT -- two waves per SIMD (geometry B) hide latency, not issue.
#include <cstdio>
#include <cstdlib>

#include <hip/hip_fp16.h>

#ifndef NABO_L2C_INLINE_DRAIN
#define NABO_DRAIN_CALL 1
#endif
#include "knn_common.h"
#include "topk_lists.h"

namespace nabo {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef NABO_L2C_BUILTIN
constexpr bool L2C_BUILTIN = true;
#else
constexpr bool L2C_BUILTIN = false;
#endif
#ifdef NABO_L2C_ABL
constexpr bool L2C_ABLATED = true;
#else
constexpr bool L2C_ABLATED = false;
#endif
// Three geometries (claunch_one), by the list length a pass wants:
//   A  one wave per SIMD: 4 waves x 128 rows, lists of <= 32 kept entries (33-entry rows), 64 staging records;
//   B  TWO waves per SIMD: 4 waves x 96 rows per workgroup, TWO workgroups per CU (round 3: one of 8 waves), lists of <= 23
//      kept entries (23-entry rows: k' + 8 at k' = 15), 32 staging records -- 2 x 4 x 20 096 bytes of LDS.  While one wave of a SIMD stages hits or drains its lists, the other one's
//      MFMAs keep the matrix pipe busy; the price is 8 x 4 KB of tile per 24 MFMAs instead of 4 x 4 KB per 32.
//   C  one wave per SIMD: 4 waves x 64 rows, lists of <= 64 kept entries (65-entry rows, emitted lists of 64): k' up to 56
//      (BASELINE configs[4]: k = 50), which the fp32-MFMA kernel served before.
constexpr int L2C_NREC = 64;        // staging records (8 scores each) per wave, geometry A
constexpr int L2C_ROW = 33;         // list entries per row (odd), geometry A
#ifndef NABO_L2C_NREC_B
#define NABO_L2C_NREC_B 32
#endif
constexpr int L2C_NREC_B = NABO_L2C_NREC_B;
constexpr int L2C_ROW_B = 23;
constexpr int L2C_ROW_C = 65;

struct cacc { f32x4 v[2][2]; };     // [row-block of the pair][reference half]
#define L2C_SG(n) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, n, 0);

// One pair of row-blocks against one reference tile: four accumulators one after another (l2q_topk.hip).
template <int KS>
__device__ __forceinline__ cacc cchain(const f16x8 (&a)[2][KS], const f16x8 (&b0)[KS], const f16x8 (&b1)[KS])
{
    cacc acc;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        f32x4 r = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
            r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c & 1][s], (c >> 1) ? b1[s] : b0[s], r, 0, 0, 0);
        acc.v[c >> 1][c & 1] = r;
    }
    return acc;
}

// lane minimum of the 8 scores of each row-block of the pair (3 v_min3 + v_min each)
struct cmins { float m0, m1; };
__device__ __forceinline__ cmins cmin8x2(const cacc &acc)
{
    cmins r;
    r.m0 = fminf(fminf(acc.v[0][0][0], acc.v[0][0][1]), acc.v[0][0][2]);
    r.m0 = fminf(fminf(r.m0, acc.v[0][0][3]), acc.v[0][1][0]);
    r.m0 = fminf(fminf(r.m0, acc.v[0][1][1]), acc.v[0][1][2]);
    r.m0 = fminf(r.m0, acc.v[0][1][3]);
    r.m1 = fminf(fminf(acc.v[1][0][0], acc.v[1][0][1]), acc.v[1][0][2]);
    r.m1 = fminf(fminf(r.m1, acc.v[1][0][3]), acc.v[1][1][0]);
    r.m1 = fminf(fminf(r.m1, acc.v[1][1][1]), acc.v[1][1][2]);
    r.m1 = fminf(r.m1, acc.v[1][1][3]);
    return r;
}

// One pair of row-blocks against one reference tile WITH the filter of another pair (`old`, complete long ago) beside it,
// scheduled by hand: 4 KS segments of one MFMA + its share of the filter's ten instructions (two 8-way minimum trees, two
// compares), a scheduling fence after each.  The MFMAs are inline assembly so that an accumulator is one register quad
// from its first step to its last reader -- hipcc gives the C = 0 step of every chain the SAME scratch quad, the trees'
// temporaries then land on it while the next MFMA still reads it as C, and the hazard pads (s_nop 2..6, several per
// tile) cost 11 ms of 87 at 1M x 1M.  What inline assembly MFMAs need (cdna_hip_programming.md, inline assembly, item 2):
// D -> next MFMA taking it whole as C: nothing; D -> any other reader: 12 wait states -- every reader here is a tile
// later, and the code after the loop pads explicitly; "=&v": D never overlaps an A operand whose last use this is.
#define L2C_PAIR_STATEMENTS(BC) \
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %8, 0\n\t" \
                 "v_min_f32 %2, %10, %11\n\t" \
                 "v_min3_f32 %2, %2, %12, %13\n\t" \
                 "v_mfma_f32_16x16x32_f16 %0, %5, %9, %0\n\t" \
                 "v_min3_f32 %2, %2, %14, %15\n\t" \
                 "v_min3_f32 %2, %2, %16, %17\n\t" \
                 "v_mfma_f32_16x16x32_f16 %1, %6, %8, 0\n\t" \
                 "v_min_f32 %3, %18, %19\n\t" \
                 "v_mfma_f32_16x16x32_f16 %1, %7, %9, %1\n\t" \
                 "v_min3_f32 %3, %3, %20, %21" \
                 : "=&v"(r00), "=&v"(r01), "=&v"(m0), "=&v"(m1) \
                 : "v"(a[0][0]), "v"(a[0][1]), "v"(a[1][0]), "v"(a[1][1]), BC(b0[0]), BC(b0[1]), \
                   "v"(old.v[0][0][0]), "v"(old.v[0][0][1]), "v"(old.v[0][0][2]), "v"(old.v[0][0][3]), \
                   "v"(old.v[0][1][0]), "v"(old.v[0][1][1]), "v"(old.v[0][1][2]), "v"(old.v[0][1][3]), \
                   "v"(old.v[1][0][0]), "v"(old.v[1][0][1]), "v"(old.v[1][0][2]), "v"(old.v[1][0][3])); \
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %5, %9, 0\n\t" \
                 "v_min3_f32 %4, %4, %11, %12\n\t" \
                 "v_mfma_f32_16x16x32_f16 %0, %6, %10, %0\n\t" \
                 "v_min3_f32 %4, %4, %13, %14\n\t" \
                 "v_mfma_f32_16x16x32_f16 %1, %7, %9, 0\n\t" \
                 "v_cmp_lt_f32_e64 %2, %15, %16\n\t" \
                 "v_mfma_f32_16x16x32_f16 %1, %8, %10, %1\n\t" \
                 "v_cmp_lt_f32_e64 %3, %4, %17" \
                 : "=&v"(r10), "=&v"(r11), "=&s"(h0), "=&s"(h1), "+v"(m1) \
                 : "v"(a[0][0]), "v"(a[0][1]), "v"(a[1][0]), "v"(a[1][1]), BC(b1[0]), BC(b1[1]), \
                   "v"(old.v[1][1][0]), "v"(old.v[1][1][1]), "v"(old.v[1][1][2]), "v"(old.v[1][1][3]), \
                   "v"(m0), "v"(tau0), "v"(tau1));
// The same for FOUR steps of 32 slots (94 <= g <= 125: BASELINE configs[4], d = 100): two statements of eight MFMAs, one
// filter instruction behind each of the first ten.
#define L2C_PAIR4_STATEMENTS(BC) \
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %12, 0\n\t" \
                 "v_min_f32 %2, %16, %17\n\t" \
                 "v_mfma_f32_16x16x32_f16 %0, %5, %13, %0\n\t" \
                 "v_min3_f32 %2, %2, %18, %19\n\t" \
                 "v_mfma_f32_16x16x32_f16 %0, %6, %14, %0\n\t" \
                 "v_min3_f32 %2, %2, %20, %21\n\t" \
                 "v_mfma_f32_16x16x32_f16 %0, %7, %15, %0\n\t" \
                 "v_min3_f32 %2, %2, %22, %23\n\t" \
                 "v_mfma_f32_16x16x32_f16 %1, %8, %12, 0\n\t" \
                 "v_min_f32 %3, %24, %25\n\t" \
                 "v_mfma_f32_16x16x32_f16 %1, %9, %13, %1\n\t" \
                 "v_min3_f32 %3, %3, %26, %27\n\t" \
                 "v_mfma_f32_16x16x32_f16 %1, %10, %14, %1\n\t" \
                 "v_mfma_f32_16x16x32_f16 %1, %11, %15, %1" \
                 : "=&v"(r00), "=&v"(r01), "=&v"(m0), "=&v"(m1) \
                 : "v"(a[0][0]), "v"(a[0][1]), "v"(a[0][2]), "v"(a[0][3]), "v"(a[1][0]), "v"(a[1][1]), "v"(a[1][2]), "v"(a[1][3]), \
                   BC(b0[0]), BC(b0[1]), BC(b0[2]), BC(b0[3]), \
                   "v"(old.v[0][0][0]), "v"(old.v[0][0][1]), "v"(old.v[0][0][2]), "v"(old.v[0][0][3]), \
                   "v"(old.v[0][1][0]), "v"(old.v[0][1][1]), "v"(old.v[0][1][2]), "v"(old.v[0][1][3]), \
                   "v"(old.v[1][0][0]), "v"(old.v[1][0][1]), "v"(old.v[1][0][2]), "v"(old.v[1][0][3])); \
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %5, %13, 0\n\t" \
                 "v_min3_f32 %4, %4, %17, %18\n\t" \
                 "v_mfma_f32_16x16x32_f16 %0, %6, %14, %0\n\t" \
                 "v_min3_f32 %4, %4, %19, %20\n\t" \
                 "v_mfma_f32_16x16x32_f16 %0, %7, %15, %0\n\t" \
                 "v_cmp_lt_f32_e64 %2, %21, %22\n\t" \
                 "v_mfma_f32_16x16x32_f16 %0, %8, %16, %0\n\t" \
                 "v_cmp_lt_f32_e64 %3, %4, %23\n\t" \
                 "v_mfma_f32_16x16x32_f16 %1, %9, %13, 0\n\t" \
                 "v_mfma_f32_16x16x32_f16 %1, %10, %14, %1\n\t" \
                 "v_mfma_f32_16x16x32_f16 %1, %11, %15, %1\n\t" \
                 "v_mfma_f32_16x16x32_f16 %1, %12, %16, %1" \
                 : "=&v"(r10), "=&v"(r11), "=&s"(h0), "=&s"(h1), "+v"(m1) \
                 : "v"(a[0][0]), "v"(a[0][1]), "v"(a[0][2]), "v"(a[0][3]), "v"(a[1][0]), "v"(a[1][1]), "v"(a[1][2]), "v"(a[1][3]), \
                   BC(b1[0]), BC(b1[1]), BC(b1[2]), BC(b1[3]), \
                   "v"(old.v[1][1][0]), "v"(old.v[1][1][1]), "v"(old.v[1][1][2]), "v"(old.v[1][1][3]), \
                   "v"(m0), "v"(tau0), "v"(tau1));
template <int KS, bool BAGPR>
__device__ __forceinline__ void cpair(const f16x8 (&a)[2][KS], const f16x8 (&b0)[KS], const f16x8 (&b1)[KS], cacc &cur,
                                      const cacc &old, float tau0, float tau1, cmins &mm, uint64_t &hit)
{
    static_assert(KS == 2 || KS == 4, "hand schedules for two (g <= 61) and four (94 <= g <= 125) steps of 32 slots; other shapes take the builtin path");
    float m0, m1;
    uint64_t h0, h1;
    f32x4 &r00 = cur.v[0][0], &r01 = cur.v[0][1], &r10 = cur.v[1][0], &r11 = cur.v[1][1];
    // Two statements of four MFMAs, each MFMA followed by its share of the filter (<= 2 instructions: what fits beside 16
    // matrix cycles).  Inside a statement nothing is padded and nothing needs to be: the filter's instructions depend on
    // each other only (vector-ALU interlocks), never on an MFMA of this tile.
    // (BAGPR: the B operands are pinned in AGPRs -- one wave per SIMD; at two waves per SIMD they stay in arch VGPRs, see
    // the kernel: with AGPRs in play hipcc parks accumulators there and copies them right behind these statements)
    if constexpr (KS == 4) {
        static_assert(BAGPR, "four steps: one wave per SIMD only");
        L2C_PAIR4_STATEMENTS("a")
    } else if constexpr (BAGPR) {
        L2C_PAIR_STATEMENTS("a")
    } else {
        L2C_PAIR_STATEMENTS("v")
    }
    mm.m0 = m0;
    mm.m1 = m1;
    hit = h0 | h1;
}

// Staging of a PAIR of row-blocks' hits (topk_lists.h: one record of 8 scores per hitting lane), with WAVE-UNIFORM control
// flow: the record count lives in a scalar register, the only per-lane code is the record write.  The usual episode has
// room for every hitting lane of both row-blocks: one room test, no loop.  Otherwise (cstage_full) lanes that do not fit
// wait for a drain and then look again at the threshold it left (NREC may be smaller than a wave: up to 64 lanes hit at
// once).  A drain does NOT touch the register copies of the thresholds here: the kernel reloads them once, after the
// tile's episodes (`drained`) -- refreshed inside, hipcc copied all of them out and back in around every episode's join
// with this rare path (twelve moves and a wait for the staging writes per episode).  Until then they are stale, i.e. too
// large, which only stages a few scores that the next drain drops.
template <typename C, int NREC>
__device__ __forceinline__ void cstage_full(bool h, const f32x4 &lo, const f32x4 &hi, float m, uint32_t row, uint32_t jb,
                                            unsigned char *w, uint32_t &scnt, int lkeep, bool &drained)
{
    uint64_t b = __builtin_amdgcn_ballot_w64(h);
    while (b != 0) {
        const uint32_t room = (uint32_t)NREC - scnt;
        if (room == 0) {
            lists_drain_only<C>(w, scnt, lkeep);
            scnt = 0;
            drained = true;
            h = h && (m < C::tauL(w)[row]);
            b = __builtin_amdgcn_ballot_w64(h);
            continue;
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
        const bool take = h && rank < room;
        if (take) {
            const uint32_t q = scnt + rank;
            f32x4 *rp = reinterpret_cast<f32x4 *>(C::srec(w) + q * C::RS);
            rp[0] = lo;
            rp[1] = hi;
            C::shdr(w)[q] = make_uint2(row, jb);
        }
        const uint32_t n = (uint32_t)__builtin_popcountll(b);
        scnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(scnt + (n < room ? n : room)));
        h = h && !take;
        b = __builtin_amdgcn_ballot_w64(h);
    }
}

template <typename C, int NB, int NREC>
__device__ __forceinline__ void cstage2(const cacc &acc, const cmins &m, int p, uint32_t jb, unsigned char *w, uint32_t &scnt,
                                        int lkeep, const float (&tauv)[NB], bool &drained)
{
    const bool h0 = m.m0 < tauv[2 * p], h1 = m.m1 < tauv[2 * p + 1];
    const uint64_t b0 = __builtin_amdgcn_ballot_w64(h0), b1 = __builtin_amdgcn_ballot_w64(h1);
    const uint32_t n0 = (uint32_t)__builtin_popcountll(b0), n = n0 + (uint32_t)__builtin_popcountll(b1);
    const uint32_t row = (uint32_t)(2 * p * C::RPB + (lane_id() & (C::RPB - 1)));
    if (__builtin_expect(scnt + n <= (uint32_t)NREC, 1)) {
        if (h0) {
            const uint32_t q = scnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u));
            f32x4 *rp = reinterpret_cast<f32x4 *>(C::srec(w) + q * C::RS);
            rp[0] = acc.v[0][0];
            rp[1] = acc.v[0][1];
            C::shdr(w)[q] = make_uint2(row, jb);
        }
        if (h1) {
            const uint32_t q = scnt + n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
            f32x4 *rp = reinterpret_cast<f32x4 *>(C::srec(w) + q * C::RS);
            rp[0] = acc.v[1][0];
            rp[1] = acc.v[1][1];
            C::shdr(w)[q] = make_uint2(row + (uint32_t)C::RPB, jb);
        }
        scnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(scnt + n));
        return;
    }
    // (the drain is a call: hipcc moves live registers around it, accumulators of the tile in flight included, and it does
    // not know the inline-assembly MFMAs' latency.  At least 16 instructions lie between the tile's last MFMA and this
    // point -- the 12 wait states an 8-pass MFMA's result needs; the pad makes it independent of that count)
    asm volatile("s_nop 15" ::: "memory");
    cstage_full<C, NREC>(h0, acc.v[0][0], acc.v[0][1], m.m0, row, jb, w, scnt, lkeep, drained);
    // (the first row-block's drain may have brought the second one's threshold down; stale is safe, fresh stages less)
    cstage_full<C, NREC>(h1, acc.v[1][0], acc.v[1][1], m.m1, row + (uint32_t)C::RPB, jb, w, scnt, lkeep, drained);
}

// Grid: x = target super-blocks (4 waves x 128 rows), y = reference splits.
// WPS: waves per SIMD the kernel is built for (1: the whole register file, B operands in AGPRs, a ring of four tile sets;
// 2: 256 registers, a ring of two).  Geometry B is WAVES = 4, WPS = 2: TWO 384-row workgroups per CU -- the same occupancy
// as one 768-row workgroup of eight waves, at half the granularity: 100k target rows are 261 workgroups on 512 slots (every
// CU busy, each wave alone on its SIMD) instead of 131 on 256 (half the chip idle), and the last round of a long query is cut finer.
template <int KS, int EPL, int ROWN, int NBv, int NRECv, int WAVES, int WPS, bool PCS = false>
__global__ __launch_bounds__(64 * WAVES, WPS) void l2c_topk_kernel(const unsigned char *__restrict__ Xpk,
                                                          const unsigned char *__restrict__ Ypk,
                                                          int tiles_per_split, int64_t tile_off, int lkeep,
                                                          uint32_t *__restrict__ cand_idx,
                                                          float *__restrict__ cand_key,
                                                          float *__restrict__ cand_tau, int64_t pad_tile, int dbg_arg,
                                                          int64_t rows_valid, const float *__restrict__ tau_init,
                                                          int tau_stride, int64_t tau_row0,
                                                          const int4 *__restrict__ pieces, int piece_S)
{
    // PIECES (api.hip: cut_pieces): with fewer column-workgroups than the chip has slots a launch of (columns x splits)
    // workgroups either leaves slots empty or spills a few workgroups into a second round that costs as much as the first.
    // Instead the linear space (column, reference tile) is cut into ~`slots` equal chunks, a chunk that crosses a column
    // boundary into two pieces, and the launch is ONE WORKGROUP PER PIECE = (column, list slot of the column, first tile, end
    // tile), longest first: the slots that finish a short piece pick up the next one.  Every piece has its own lists,
    // thresholds and emitted list (row, slot) of piece_S.
    // dbg: timing ablations of the experiment builds (knn_common.h: debug_ablate; the shipped library always passes 0).
    // The four-step kernel on the 64-entry lists keeps it a RUN-TIME value in the product build too: with the ablation
    // selects compiled in -- a few scalar instructions and never-taken branches per tile, same loads, same waits, same
    // MFMA statements -- that one instantiation runs 16 % faster than with them folded away (cosine 1M x 1M, d = 100,
    // k = 50, same box: 308 against 365 ms; no such effect, or 1-4 % the other way, in the other instantiations:
    // profiles/r3_coarse_experiments.txt item 13).  Not understood; measured.
#ifdef NABO_EXPERIMENTS
    constexpr bool DBG_RT = true;
#else
    constexpr bool DBG_RT = KS == 4 && EPL == 2;
#endif
    const int dbg = DBG_RT ? dbg_arg : 0;
    constexpr int NB = NBv;                            // row-blocks of 16 targets per wave
    constexpr int NP = NB / 2;                         // pairs
    constexpr int NREC = NRECv;
    using C = ListCfg<EPL, ROWN, NB, NREC, 16, (ROWN >= 64)>;
    constexpr int TB = 2 * KS * 1024;                  // bytes per packed 32-cell tile (targets and references alike)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int lane = lane_id();
    const int lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (PCS: its own instantiation; one workgroup per piece, longest pieces first -- a loop over a chunk's pieces inside the
    // kernel kept every kernel argument alive to the end and cost the 256-register geometry spills in its staging path)
    int4 piece = make_int4((int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.y * tiles_per_split, ((int)blockIdx.y + 1) * tiles_per_split);
    if (PCS) piece = pieces[blockIdx.x];
    const int colx = PCS ? __builtin_amdgcn_readfirstlane(piece.x) : (int)blockIdx.x;
    const int split = PCS ? __builtin_amdgcn_readfirstlane(piece.y) : (int)blockIdx.y;
    const int S = PCS ? piece_S : (int)gridDim.y;
    const int64_t ltile0 = ((int64_t)colx * WAVES + wave) * (NB / 2);        // in 32-row tiles
    const int64_t ttile0 = tile_off + ltile0;

    f16x8 xb[NB][KS];
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) {
        const f16x8 *p = reinterpret_cast<const f16x8 *>(Xpk + (ttile0 + (rb >> 1)) * TB);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xb[rb][s] = p[((rb & 1) * KS + s) * 64 + lane];
            // pinned in AGPRs (l2h_topk.hip) at one wave per SIMD.  NOT at two: with 256 registers per wave hipcc then
            // parks ACCUMULATORS in the spare AGPRs and copies them (v_accvgpr_write) right behind the inline-assembly
            // MFMA that is still writing them -- a hazard it cannot see (wrong neighbours in 3 of 9 test shapes).
            if (WPS == 1) asm volatile("" : "+a"(xb[rb][s]));
        }
    }
    unsigned char *wl = smem_raw + (size_t)wave * C::BYTES;          // this wave's lists (topk_lists.h)
    float tauv[NB];
    const float tau0 = (dbg & 1) ? -__builtin_inff() : __builtin_inff();
    // Padding rows (beyond the query's last row; a 1-row query still is a 512-row workgroup) start from threshold -inf:
    // nothing ever passes, so they cost no list work -- their operands are finite (pack_ctiles_kernel<.,.,1>) and would
    // otherwise fill and refine lists like any row: 1.5 ms for a one-row query over 125k references, ~0.1 ms without.
    const int64_t row0 = ttile0 * 32;                    // first row of the wave (position in the query)
    // SEEDED pass (tau_init, one value per row of the query; api.hip: the second pass of the rows whose first-pass
    // certificate failed): a row starts from the threshold refine.hip worked out for it -- below it lie the few
    // references that can still enter or tie with the first k', nothing is built up from +inf.  The kept list starts as
    // lkeep entries (tau_init, no index): its maximum IS the threshold until every one of them has been replaced, and from
    // then on the list behaves like any other (a row with more than lkeep references below its seed loses its
    // certificate again and goes on to the f16x3 pass).
    // The same start serves the TOURNAMENT seeds of l2c_pre_kernel (below): one value per row AND reference split
    // (tau_stride = the launch's split count, tau_row0 = its first row), an upper bound of the row's lkeep-th smallest
    // score among the split's first references -- the list warm-up from +inf, half of all list updates, is skipped.
    auto seed_of = [&](int64_t r) -> float {
        return tau_stride ? tau_init[(r - tau_row0) * tau_stride + split] : tau_init[r];
    };
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) {
        const int64_t r = row0 + rb * 16 + (lane & 15);
        tauv[rb] = r < rows_valid ? ((tau_init && !(dbg & 1)) ? seed_of(r) : tau0) : -__builtin_inff();
    }
    uint32_t scnt = 0;
    const int t_begin = PCS ? __builtin_amdgcn_readfirstlane(piece.z) : split * tiles_per_split;
    const int t_end = PCS ? __builtin_amdgcn_readfirstlane(piece.w) : t_begin + tiles_per_split;
    lists_init<C>(wl, lkeep, tau0, (uint32_t)t_begin * 32u);
    {
        const int64_t nv = rows_valid - row0;            // valid rows of this wave
        if (nv < C::NROWS)
            for (int r = lane; r < C::NROWS; r += 64)
                if (r >= nv) C::tauL(wl)[r] = -__builtin_inff();
        if (tau_init && !(dbg & 1)) {
            uint2 *rows = C::rows(wl);
            for (int e = lane; e < C::NROWS * lkeep; e += 64) {
                const int r = e / lkeep, sl = e - r * lkeep;
                if (r < nv) rows[r * C::ROW + sl].y = __float_as_uint(seed_of(row0 + r));
            }
            for (int r = lane; r < C::NROWS; r += 64)
                if (r < nv) C::tauL(wl)[r] = seed_of(row0 + r);
            if constexpr (C::GRP) {
                for (int e = lane; e < C::NROWS * C::NGRP; e += 64) {
                    const int r = e / C::NGRP, g = e % C::NGRP;
                    if (r < nv && 8 * g < lkeep) reinterpret_cast<uint2 *>(C::gmax(wl))[e].y = __float_as_uint(seed_of(row0 + r));
                }
            }
        }
    }

    // past the split's last tile: an all-padding tile (+inf norms, nothing passes) -- the loop runs in fours (twos)
    auto tile_ptr = [&](int ts) {
        const int64_t tc = ts < t_end ? (int64_t)ts : pad_tile;
        // dbg & 2 / dbg & 4 (timing experiments, garbage results): the stream wraps inside a window of 128 tiles / of 2
        return Ypk + ((dbg & 2) ? (int64_t)(t_begin + ((ts - t_begin) & 127)) : (dbg & 4) ? (int64_t)(t_begin + ((ts - t_begin) & 1)) : tc) * TB;
    };
    auto tile_load = [&](f16x8(&a)[2][KS], int ts) {
        const f16x8 *p = reinterpret_cast<const f16x8 *>(tile_ptr(ts));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < KS; ++s) a[h][s] = p[(h * KS + s) * 64 + lane];
    };

    // tile sets in flight: four at one wave per SIMD (tile t + 3 is requested at the top of step t); two at two waves per
    // SIMD, where the other wave covers the latency and 256 registers have to hold everything
    // (... and at four operand steps with eight row-blocks: 4 x 32 + 2 x 64 + ... arch VGPRs would not fit 256, hipcc would
    // park values in AGPRs -- see the B operands above for why that must not happen next to inline-assembly MFMAs)
    constexpr int RING = (WPS > 1 || (KS >= 4 && NBv >= 8)) ? 2 : 4;
    f16x8 a0[2][KS], a1[2][KS], a2[2][KS], a3[2][KS];
    tile_load(a0, t_begin);
    if (RING == 4) {
        tile_load(a1, t_begin + 1);
        tile_load(a2, t_begin + 2);
    }

    // Scores of a whole tile (NP pairs x 16 registers) stay in registers for one more step: their filter runs next to
    // the NEXT tile's chains, on operands that were complete long before (a filter instruction that reads what the
    // previous chain has just written costs three times one that does not: tools/coarse_lab.hip), and ONE branch per
    // tile acts on the four verdicts (a compare-and-branch per chain costs ~37 cycles even when it falls through).
    cacc accE[NP], accO[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) accO[p].v[r][h] = f32x4{__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};

#ifdef NABO_L2C_ABL
    float abl_run = 1e30f;
    int abl_cnt = 0;
#endif
    // chains of tile t into `cur` (set `a`; set `an` receives tile t + 3), filter of tile t - 1 (`old`) beside them
    auto tile_step = [&](const f16x8(&a)[2][KS], f16x8(&an)[2][KS], cacc(&cur)[NP], const cacc(&old)[NP], int t) {
        tile_load(an, t + RING - 1);
        cmins mm[NP];
        uint64_t hit[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#if defined(NABO_L2C_ABL) && NABO_L2C_ABL == 1          // timing ablations (tools/ab): no filter at all
            asm volatile("" ::"v"(old[p].v[0][0]), "v"(old[p].v[0][1]), "v"(old[p].v[1][0]), "v"(old[p].v[1][1]));
            hit[p] = 0;
#elif defined(NABO_L2C_ABL) && NABO_L2C_ABL == 2        // the minimum trees, no compare
            mm[p] = cmin8x2(old[p]);
            abl_run = fminf(abl_run, fminf(mm[p].m0, mm[p].m1));
            hit[p] = 0;
#elif defined(NABO_L2C_ABL) && NABO_L2C_ABL == 3        // trees + compares, verdicts folded into a register, no branch
            mm[p] = cmin8x2(old[p]);
            abl_cnt += ((mm[p].m0 < tauv[2 * p]) | (mm[p].m1 < tauv[2 * p + 1])) ? 1 : 0;
            hit[p] = 0;
#else
            if constexpr ((KS == 2 || KS == 4) && !L2C_BUILTIN) {
                cpair<KS, (WPS == 1)>(a, xb[2 * p], xb[2 * p + 1], cur[p], old[p], tauv[2 * p], tauv[2 * p + 1], mm[p], hit[p]);
            } else {                                    // hipcc's own schedule of builtin MFMAs (other shapes; A/B runs)
                mm[p] = cmin8x2(old[p]);
                hit[p] = __builtin_amdgcn_ballot_w64((mm[p].m0 < tauv[2 * p]) | (mm[p].m1 < tauv[2 * p + 1]));
            }
#endif
            if constexpr ((KS != 2 && KS != 4) || L2C_BUILTIN || L2C_ABLATED) {
                cur[p] = cchain<KS>(a, xb[2 * p], xb[2 * p + 1]);
                // one MFMA, then at most two of the filter's instructions (what fits beside a 16-cycle MFMA)
                L2C_SG(2) L2C_SG(2) L2C_SG(1) L2C_SG(1) L2C_SG(1) L2C_SG(1) L2C_SG(1) L2C_SG(1)
            }
        }
        uint64_t any = hit[0];
#pragma unroll
        for (int p = 1; p < NP; ++p) any |= hit[p];
        if (__builtin_expect(any != 0, 0)) {
            const uint32_t jb = (uint32_t)((t - 1) * 32 + 4 * lq);           // (a padding step stages nothing)
            // the staging wave goes AHEAD of the SIMD's other wave (s_setprio; two waves per SIMD): it is back in its MFMA
            // loop sooner -- 0.6 ms of 102 at 1M x 1M; the other way round (the MFMA loop ahead) costs 1.5 ms
            __builtin_amdgcn_s_setprio(3);
            bool drained = false;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (hit[p] != 0) {
                    NABO_PROF_T0();
                    cstage2<C, NB, NREC>(old[p], mm[p], p, jb, wl, scnt, lkeep, tauv, drained);
                    NABO_PROF_ADD(wl, 0, 1);
                    NABO_PROF_ADD(wl, 1, NABO_PROF_DT() >> 4);
                }
            }
            if (drained) {
#pragma unroll
                for (int rb = 0; rb < NB; ++rb) tauv[rb] = C::tauL(wl)[rb * C::RPB + (lane & (C::RPB - 1))];
            }
            __builtin_amdgcn_s_setprio(0);
        }
    };
    // Every load of the preamble lands HERE (s_waitcnt vmcnt(0), once per wave).  hipcc's scheduler is free to reorder the
    // preamble's tile loads, and its wait-count pass merges the preamble's pending loads with the loop's at the loop
    // header: with the first tile's loads issued last, the FIRST step of every round waited for all but the newest tile
    // (vmcnt(8) where 24 loads may be in flight) -- a ring of four that emptied once per round.  Seen in the product build of
    // every ring-of-four geometry, not in the experiment builds (whose extra branches happened to keep the order):
    // cosine 1M x 1M, d = 100, k = 50: kernel 425 -> 318 ms.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int t = t_begin;
    if (RING == 4) {
        for (; t < t_end; t += 4) {
            tile_step(a0, a3, accE, accO, t);
            tile_step(a1, a0, accO, accE, t + 1);
            tile_step(a2, a1, accE, accO, t + 2);
            tile_step(a3, a2, accO, accE, t + 3);
        }
    } else {
        for (; t < t_end; t += 2) {
            tile_step(a0, a1, accE, accO, t);
            tile_step(a1, a0, accO, accE, t + 1);
        }
    }
#ifdef NABO_L2C_ABL
    if (abl_run + (float)abl_cnt == 12345.0f) cand_tau[0] = abl_run;
#endif
    // the last step's scores (a padding step when the split's length is no multiple of four: all +inf); its MFMAs are
    // inline assembly, so the wait states between them and the first reader are ours to insert (12 for this shape)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    {
        const uint32_t jb = (uint32_t)((t - 1) * 32 + 4 * lq);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const cmins m = cmin8x2(accO[p]);
            bool drained = false;
            cstage2<C, NB, NREC>(accO[p], m, p, jb, wl, scnt, lkeep, tauv, drained);
        }
    }

    lists_flush<C, EPL, NB>(wl, scnt, ltile0 * 32, split, S, lkeep, tauv, cand_idx, cand_key, cand_tau);
}

// ---- tournament seeds: a START THRESHOLD per (row, reference split) without any list work ---------------------------------
// A streamed top-L makes L (1 + ln(n / L)) list updates per row and half of them fall into the first few thousand references
// of the stream, while the threshold is still far above where it ends (a shard of 125k references pays them as the whole
// set does: the reason one rank of eight costs far more than an eighth).  This kernel looks at the first `pre_tiles` tiles of
// every split BEFORE the filter does and hands it an upper bound tau0 of the row's lkeep-th smallest score among them:
//   * a lane (row-block rb, row l & 15, quarter lq = l >> 4) sees 8 scores of its row per tile; the minimum over `gt`
//     consecutive tiles is a GROUP minimum -- one distinct reference per group, groups of different lanes and steps are
//     disjoint;
//   * each lane keeps its q = ceil(lkeep / 4) smallest group minima (a bubble through q sorted registers per group);
//   * tau0 = the largest of the four lanes' q-th values: 4 q >= lkeep distinct references score <= tau0.
// The filter then starts the row's list as lkeep entries (tau0, no index) -- the seeded start of tau_init above -- and
// collects the ~1.4 lkeep references below tau0 instead of climbing down from +inf through lkeep (1 + ln(T / lkeep))
// updates.  Any tau0 gives a correct certificate (refine.hip bounds the non-candidates by the FINAL threshold <= tau0); a
// tau0 that is too low only leaves the list short and sends the row to the next pass, so nothing here needs to be exact --
// it is, though: the same operands through the same MFMA chain as the filter's.  Cost: pre_tiles of the stream's MFMA
// work and ~16 vector instructions per pair of row-blocks and tile, no LDS, no branches.
template <int KS, int NB, int QM>
__global__ __launch_bounds__(256) void l2c_pre_kernel(const unsigned char *__restrict__ Xpk, const unsigned char *__restrict__ Ypk,
                                                      int tiles_per_split, int64_t tile_off, int64_t rows, int pre_tiles, int gt,
                                                      int q, int64_t pad_tile, int64_t rows_valid, float *__restrict__ tau_out,
                                                      const int4 *__restrict__ ranges, int rows_per_col)
{
    constexpr int NP = NB / 2;
    constexpr int TB = 2 * KS * 1024;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int split = blockIdx.y, S = gridDim.y;
    const int64_t ltile0 = ((int64_t)blockIdx.x * 4 + wave) * NP;            // in 32-row tiles, local to the launch
    const int64_t ttile0 = tile_off + ltile0;
    // (wave-uniform: a wave beyond the launch's rows -- the filter's workgroups are whole multiples of a wave's rows -- or
    // with nothing but padding rows)
    if (ltile0 * 32 >= rows || ttile0 * 32 >= rows_valid) return;
    f16x8 xb[NB][KS];
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) {
        const f16x8 *p = reinterpret_cast<const f16x8 *>(Xpk + (ttile0 + (rb >> 1)) * TB);
#pragma unroll
        for (int s = 0; s < KS; ++s) xb[rb][s] = p[((rb & 1) * KS + s) * 64 + lane];
    }
    float srt[NB][QM];
#pragma unroll
    for (int rb = 0; rb < NB; ++rb)
#pragma unroll
        for (int j = 0; j < QM; ++j) srt[rb][j] = __builtin_inff();
    int t_begin = split * tiles_per_split, t_end = t_begin + tiles_per_split;
    if (ranges) {
        // a launch cut into pieces: the tournament of (column, slot) looks at the first tiles of THAT piece (a wave's rows lie
        // in one column: the columns are whole multiples of a wave's rows); no tournament for a short or unused piece --
        // its seeds stay +inf (the caller's fill)
        const int4 r = ranges[(ltile0 * 32 / rows_per_col) * S + split];
        t_begin = __builtin_amdgcn_readfirstlane(r.x);
        t_end = __builtin_amdgcn_readfirstlane(r.y);
        pre_tiles = __builtin_amdgcn_readfirstlane(r.z);
        gt = __builtin_amdgcn_readfirstlane(r.w);
        if (pre_tiles <= 0) return;
    }
    auto tile_load = [&](f16x8(&a)[2][KS], int ts) {
        const f16x8 *p = reinterpret_cast<const f16x8 *>(Ypk + (ts < t_end ? (int64_t)ts : pad_tile) * TB);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < KS; ++s) a[h][s] = p[(h * KS + s) * 64 + lane];
    };
    float gm[NB];
    auto tile_mins = [&](const f16x8(&a)[2][KS]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const cacc acc = cchain<KS>(a, xb[2 * p], xb[2 * p + 1]);
            const cmins mm = cmin8x2(acc);
            gm[2 * p] = fminf(gm[2 * p], mm.m0);
            gm[2 * p + 1] = fminf(gm[2 * p + 1], mm.m1);
        }
    };
    f16x8 a0[2][KS], a1[2][KS];
    tile_load(a0, t_begin);
    const int t_stop = t_begin + pre_tiles;                                  // (gt even, pre_tiles a multiple of gt: the caller's)
    for (int t = t_begin; t < t_stop;) {
#pragma unroll
        for (int rb = 0; rb < NB; ++rb) gm[rb] = __builtin_inff();
        for (int u = 0; u < gt; u += 2) {
            tile_load(a1, t + 1);
            tile_mins(a0);
            tile_load(a0, t + 2);
            tile_mins(a1);
            t += 2;
        }
#pragma unroll
        for (int rb = 0; rb < NB; ++rb) {                                    // the group minimum bubbles into the sorted registers
            float x = gm[rb];
#pragma unroll
            for (int j = 0; j < QM; ++j) {
                const float lo = fminf(srt[rb][j], x);
                x = fmaxf(srt[rb][j], x);
                srt[rb][j] = lo;
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) {
        float v = srt[rb][0];
#pragma unroll
        for (int j = 1; j < QM; ++j) v = (j == q - 1) ? srt[rb][j] : v;
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        v = fmaxf(v, __shfl_xor(v, 32, 64));
        const int64_t lrow = ltile0 * 32 + rb * 16 + (lane & 15);
        if (lane < 16 && tile_off * 32 + lrow < rows_valid) tau_out[lrow * S + split] = v;
    }
}

template <int KS, int NB, int QM>
static hipError_t cpre_launch(const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S, int64_t rows,
                              int64_t tile_off, int pre_tiles, int gt, int q, int64_t pad_tile, hipStream_t st,
                              int64_t rows_valid, float *tau_out, const int *ranges, int rows_per_col)
{
    const int64_t gx = (rows + 4 * NB * 16 - 1) / (4 * NB * 16);
    if (ranges && (rows_per_col <= 0 || rows_per_col % (NB * 16) != 0)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((l2c_pre_kernel<KS, NB, QM>), dim3((unsigned)gx, (unsigned)S), dim3(256), 0, st, Xpk, Ypk, tiles_per_split,
                       tile_off, rows, pre_tiles, gt, q, pad_tile, rows_valid, tau_out, reinterpret_cast<const int4 *>(ranges),
                       rows_per_col);
    return hipGetLastError();
}

// How much of a split's stream the tournament looks at, for lists of lkeep entries over splits of tiles_per_split tiles:
// *pre_tiles (0: not worth it), *gt tiles per group.  The optimum does not depend on the stream's length: a tile of the
// tournament costs a row ~2 SIMD cycles per step of 32 slots (kc / 2 steps), a list update ~120 (1M x 1M: 14 ms of hits for
// 269 updates per row), and a tournament over T tiles saves lkeep (ln(32 T / lkeep) - 0.4) updates: d/dT = 0 at
// T = 120 lkeep / kc tiles (690 at lkeep = 23, two steps) -- capped at a quarter of the stream; scale_pct: A/B runs.
void l2c_pre_plan(int kc, int lkeep, int tiles_per_split, int scale_pct, int *pre_tiles, int *gt)
{
    const int q = (lkeep + 3) / 4;
    // (round 4: 90 instead of the model's 120 -- 1M x 1M: 91.1 ms either way; one rank of eight, 15-entry lists: 13.95 instead of 14.18)
    int want = (int)((int64_t)90 * lkeep * scale_pct / 100 / (kc > 0 ? kc : 2));
    if (want > tiles_per_split / 4) want = tiles_per_split / 4;
    *pre_tiles = 0;
    *gt = 2;
    if (want < 6 * q) return;                         // fewer than 3 q groups of two tiles: the bound would be loose
    int g2 = want / (8 * q);                          // >= 8 q groups when the budget allows, in groups of 2, 4, .. tiles
    g2 = g2 < 2 ? 2 : (g2 > 8 ? 8 : g2 & ~1);
    *gt = g2;
    *pre_tiles = want / g2 * g2;
}

// tau_out [rows][S] (rows local to this launch: row tile_off * 32 of the query is row 0), rows_valid as in l2c_topk_launch
hipError_t l2c_pre_launch(int kc, int lkeep, const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S,
                          int64_t rows, int64_t tile_off, int pre_tiles, int gt, int64_t pad_tile, hipStream_t st,
                          int64_t rows_valid, float *tau_out, const int *ranges, int rows_per_col)
{
    const int q = (lkeep + 3) / 4;
    // (ranges: every (column, slot) brings its own tournament length -- l2c_pre_plan's, checked by the planner)
    if (!ranges && (pre_tiles <= 0 || gt < 2 || (gt & 1) || pre_tiles % gt || pre_tiles > tiles_per_split)) return hipErrorInvalidValue;
    if (q < 1 || q > 16 || rows <= 0) return hipErrorInvalidValue;
#define NABO_PRE(KSV)                                                                                                               \
    case 2 * KSV:                                                                                                                   \
        return q <= 8 ? cpre_launch<KSV, (KSV <= 2 ? 8 : 4), 8>(Xpk, Ypk, tiles_per_split, S, rows, tile_off, pre_tiles, gt, q, pad_tile, \
                                                              st, rows_valid, tau_out, ranges, rows_per_col)                       \
                      : cpre_launch<KSV, 4, 16>(Xpk, Ypk, tiles_per_split, S, rows, tile_off, pre_tiles, gt, q, pad_tile, st,     \
                                                rows_valid, tau_out, ranges, rows_per_col);
    switch (kc) {
        NABO_PRE(1) NABO_PRE(2) NABO_PRE(3) NABO_PRE(4)
    default: return hipErrorInvalidValue;
    }
#undef NABO_PRE
}

template <int KS, int EPL, int ROWN, int NBv, int NRECv, int WAVES, int WPS>
static hipError_t claunch_geo(const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S, int gx,
                              int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                              int64_t pad_tile, hipStream_t st, int64_t rows_valid, const float *tau_init, int tau_stride,
                              int64_t tau_row0, const L2cPieces &pcs)
{
    const int dbg = debug_ablate();
    constexpr size_t lds = (size_t)WAVES * ListCfg<EPL, ROWN, NBv, NRECv, 16, (ROWN >= 64)>::BYTES;
    static_assert(lds <= 163840, "LDS budget");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&l2c_topk_kernel<KS, EPL, ROWN, NBv, NRECv, WAVES, WPS, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 block(64 * WAVES);
#ifndef NABO_EXPERIMENTS
    if (pcs.pieces) return hipErrorInvalidValue;        // (the launch cut into pieces lost to uniform splits: experiments build only)
#else
    if (pcs.pieces) {
        // pieces: one workgroup per chunk of the (column, tile) space; gx / S then only describe the emitted lists
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&l2c_topk_kernel<KS, EPL, ROWN, NBv, NRECv, WAVES, WPS, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((l2c_topk_kernel<KS, EPL, ROWN, NBv, NRECv, WAVES, WPS, true>), dim3(pcs.n_pieces, 1), block, lds, st, Xpk, Ypk,
                           tiles_per_split, tile_off, lkeep, cand_idx, cand_key, cand_tau, pad_tile, dbg, rows_valid, tau_init, tau_stride,
                           tau_row0, reinterpret_cast<const int4 *>(pcs.pieces), S);
    } else
#endif
    {
        hipLaunchKernelGGL((l2c_topk_kernel<KS, EPL, ROWN, NBv, NRECv, WAVES, WPS, false>), dim3(gx, S), block, lds, st, Xpk, Ypk,
                           tiles_per_split, tile_off, lkeep, cand_idx, cand_key, cand_tau, pad_tile, dbg, rows_valid, tau_init, tau_stride,
                           tau_row0, nullptr, S);
    }
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

template <int KS>
static hipError_t claunch_one(int geo, const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S, int gx,
                              int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                              int64_t pad_tile, hipStream_t st, int64_t rows_valid, const float *tau_init, int tau_stride,
                              int64_t tau_row0, const L2cPieces &pcs)
{
    if constexpr (KS <= 2) {
        if (geo == 1)
            return claunch_geo<KS, 1, L2C_ROW_B, 6, L2C_NREC_B, 4, 2>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx,
                                                                  cand_key, cand_tau, pad_tile, st, rows_valid, tau_init, tau_stride, tau_row0, pcs);
    }
    if (geo == 2)
        return claunch_geo<KS, 2, L2C_ROW_C, 4, L2C_NREC, 4, 1>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key,
                                                            cand_tau, pad_tile, st, rows_valid, tau_init, tau_stride, tau_row0, pcs);
    return claunch_geo<KS, 1, L2C_ROW, 8, L2C_NREC, 4, 1>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau,
                                                      pad_tile, st, rows_valid, tau_init, tau_stride, tau_row0, pcs);
}

// Which geometry serves lists of `lkeep_want` kept entries: 1 = B (two waves per SIMD) up to 23 entries and KS <= 2 (its
// 256 registers per wave hold two steps of operands), 0 = A up to 32, 2 = C up to 64; -1: none.  pin (the index option
// "l2c_geo": 0 = A, 2 = C, anything else none) selects one where it can serve the lists at all.
int l2c_geometry(int kc, int lkeep_want, int pin)
{
    if (lkeep_want > 64) return -1;
    if (lkeep_want > 32 || pin == 2) return 2;
    if (kc > 4 || lkeep_want > L2C_ROW_B || pin == 0) return 0;
    return 1;
}

void l2c_topk_geometry(int kc, int lkeep_want, int pin, int *rows_per_wg, int *wg_per_cu, int *lkeep_max)
{
    switch (l2c_geometry(kc, lkeep_want, pin)) {
    case 1: *rows_per_wg = 4 * 96; *wg_per_cu = 2; *lkeep_max = L2C_ROW_B; return;
    case 2: *rows_per_wg = 4 * 64; *lkeep_max = 64; break;
    default: *rows_per_wg = 4 * 128; *lkeep_max = L2C_ROW < 32 ? L2C_ROW : 32; break;
    }
    *wg_per_cu = 1;
}

// steps of 16 slots of the one-product operands (g components + two norm slots + the error slot), even (KS = kc / 2
// steps of 32 slots), instantiated values only: g <= 125
int l2c_pick_kc(int g)
{
    const int need = 2 * ((g + 3 + 31) / 32);
    return need <= 8 ? need : -1;
}

// The split's padding must cover the ring: the kernel reads tiles up to t_end + 3 (as pad_tile) -- all of them are the
// caller's padding tile, never past the allocation.  geo: what l2c_geometry said when the caller sized its grid.
hipError_t l2c_topk_launch(int kc, int geo, const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S,
                           int gx, int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                           int64_t pad_tile, hipStream_t st, int64_t rows_valid, const float *tau_init, int tau_stride,
                           int64_t tau_row0, const L2cPieces *pieces)
{
    const L2cPieces pcs = pieces ? *pieces : L2cPieces{nullptr, 0, nullptr, 0};
    if (pcs.pieces && pcs.n_pieces < 1) return hipErrorInvalidValue;
    if ((int64_t)tiles_per_split * 32 >= NABO_LIST_SPLIT_REFS) return hipErrorInvalidValue;   // topk_lists.h: 25 bits of offset per entry
    if (geo < 0 || geo > 2 || (geo == 1 && (kc > 4 || lkeep > L2C_ROW_B)) || (geo == 0 && lkeep > 32) || lkeep > 64)
        return hipErrorInvalidValue;
    switch (kc) {
    case 2: return claunch_one<1>(geo, Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, pad_tile, st, rows_valid, tau_init, tau_stride, tau_row0, pcs);
    case 4: return claunch_one<2>(geo, Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, pad_tile, st, rows_valid, tau_init, tau_stride, tau_row0, pcs);
    case 6: return claunch_one<3>(geo, Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, pad_tile, st, rows_valid, tau_init, tau_stride, tau_row0, pcs);
    case 8: return claunch_one<4>(geo, Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, pad_tile, st, rows_valid, tau_init, tau_stride, tau_row0, pcs);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace nabo
