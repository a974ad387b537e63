// canberra_bits.hip -- the counting pass of the modified-Canberra filter as BIT-SLICED arithmetic on per-bucket bitmaps.
//
// Context (canberra_f32.hip): _mod_canberra_dist (nabo/_mapping.py:29-45) adds exactly 1 for every dimension whose
// |x - y| is outside the window f |x| and a quotient in [0, 1) otherwise, so `distance >= number of dimensions PROVEN out
// of window`; a pair whose count already reaches the row's threshold is dropped, the survivors get the fp32 lower bound,
// the kept candidates the exact float64 expression (refine.hip).  The SWAR form of the count (7-bit integers, four
// dimensions per word: v_sub, v_sub, v_bitop3, v_bcnt per four dimensions and ONE reference per lane) issues 1.22 vector
// instructions per pair and dimension and runs at 88 % of the vector issue peak: it cannot get faster, only smaller.
//
// This file turns the count by 90 degrees: a lane holds 32 REFERENCES as the bits of a word.
//   * Per dimension the references' values are cut into CBB_B = 32 QUANTILE buckets (edges from a sample of the
//     references, api.hip); b(v) = #{edges <= v} is a non-decreasing step function, so for any window (lo, hi)
//     y in (lo, hi)  =>  b(lo) <= b(y) <= b(hi): counting "may be in window" on bucket numbers can only err towards IN.
//   * Index side (once per set_ref): for every block of 2048 references, dimension d and bucket row r the word
//     tab[block][d][r][w] holds, for the 32 references of word w, the bits "b(y_d) <= r - 1" (row 0 is the empty set):
//     cumulative bitmaps, 50 x 33 x 64 words per block = 206 bytes per reference.
//   * Query side: a target's window in dimension d is two row numbers (lo_row = b(lo), hi_row = b(hi) + 1), and the 32
//     references of a word that MAY be in the window are  tab[hi_row] & ~tab[lo_row]  -- two reads and one vector
//     instruction per 32 pairs; the per-reference count over the dimensions lives in SIX BIT PLANES per target (a
//     bit-sliced counter: adding a 0/1 plane is a ripple of and / xor pairs), the comparison with the row's integer
//     threshold is a bit-sliced comparator.  ~14 vector instructions per dimension and 32 pairs: 0.44 per pair and
//     dimension against 1.22.
//   * Who shares what.  The table is 206 bytes per reference and every target needs all of it: a wave that streamed it
//     for its own 16-32 targets (the first version of this file) moved 14 TB per 1M x 1M step through L2 and was no
//     faster than the SWAR pass (1.88 s vs 1.75 s).  So a WORKGROUP of 8 waves (128 targets) walks the table together:
//     the rows of a PAIR of dimensions of the current block (2 x 65 x 256 B) are staged in LDS (double-buffered, one
//     barrier per pair), every wave reads the rows each of its 8 targets needs from there (row number wave-uniform, lane =
//     word: conflict-free) and adds the two masks to that target's counter, which stays in registers for the whole block
//     (8 targets x 6 planes = 48 VGPRs; sixteen waves per workgroup, four per SIMD): one 3:2 compressor (v_bitop3 0x96 /
//     0xE8) takes both masks into the lowest plane, one carry ripples upwards.  Table traffic: 412 MB per 128 targets.
//   * Survivors (7e-3 of the pairs with 32 quantile buckets, simulated on the bench's data; 2.2e-3 for the 128 uniform
//     buckets of the SWAR pass) leave through the same wave-private work ring, fp32 lower bound, candidate lists and
//     certificate as before.  The ring is filled one survivor per lane and round, i.e. NOT in ascending reference
//     order, so a list accepts a candidate by (key, index) < (tau, index of the last kept entry): the kept set is the
//     L smallest pairs in that order whatever the arrival order (refine.hip's plateau certificate relies on it).
//     Results are the reference's bits either way (tests/test_knn_gpu.py: canberra cases run both kernels).
#include <cstdlib>
#include <type_traits>
#include "knn_common.h"

namespace nabo {

#ifndef NABO_CBB_B
#define NABO_CBB_B 64
#endif
constexpr int CBB_B = NABO_CBB_B;         // quantile buckets per dimension (survivors of the count: 7e-3 of the pairs at 32, 2.6e-3 at 64)
constexpr int CBB_ROWS = CBB_B + 1;       // cumulative rows per dimension (row 0: empty set)
constexpr int CBB_BLK = 2048;             // references per block: 64 lanes x 32 bits
constexpr int CBB_T = 8;                  // target rows per wave (six count planes each, in registers: < 128 VGPRs per wave)
#ifndef NABO_CBB_TB
#define NABO_CBB_TB 8
#endif
constexpr int CBB_NW = 16;                // waves per workgroup (four per SIMD): they share the LDS copy of the table rows

int cbb_buckets() { return CBB_B; }
int cbb_rows_per_wg() { return CBB_T * CBB_NW; }
size_t cbb_table_bytes(int64_t n, int g)
{
    // (+ one dimension of slack: with an odd g the kernel stages dimension g of the last block -- never used)
    return ((size_t)((n + CBB_BLK - 1) / CBB_BLK) * g + 1) * CBB_ROWS * 64 * sizeof(uint32_t);
}
size_t cbb_valid_bytes(int64_t n) { return (size_t)((n + CBB_BLK - 1) / CBB_BLK) * 64 * sizeof(uint32_t); }

// b(v) = number of edges <= v, edges ascending [CBB_B - 1]
__device__ __forceinline__ int cbb_bucket(const double *__restrict__ edges, double v)
{
    int lo = 0, hi = CBB_B - 1;           // answer in [0, CBB_B - 1]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (edges[mid] <= v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// grid (blocks, g), 64 threads: lane w builds the 33 cumulative rows of its word for dimension blockIdx.y
__global__ __launch_bounds__(64) void cbb_pack_table_kernel(const double *__restrict__ Y, int64_t n, int g,
                                                            const double *__restrict__ edges, uint32_t *__restrict__ tab)
{
    __shared__ uint32_t eq[CBB_ROWS][64];
    const int w = threadIdx.x, d = blockIdx.y;
    const int64_t blk = blockIdx.x;
#pragma unroll
    for (int r = 0; r < CBB_ROWS; ++r) eq[r][w] = 0u;
    const double *ed = edges + (size_t)d * (CBB_B - 1);
    for (int r = 0; r < 32; ++r) {
        const int64_t j = blk * CBB_BLK + (int64_t)w * 32 + r;
        if (j < n) eq[cbb_bucket(ed, Y[j * g + d]) + 1][w] |= 1u << r;         // (column w is this lane's own: no races)
    }
    uint32_t acc = 0u;
    uint32_t *o = tab + ((size_t)(blk * g + d) * CBB_ROWS) * 64 + w;
    for (int r = 0; r < CBB_ROWS; ++r) {
        acc |= eq[r][w];
        o[(size_t)r * 64] = acc;
    }
}

// vbits[block][w]: bit r = reference block*2048 + 32 w + r exists and is not ignored
__global__ void cbb_valid_kernel(const uint8_t *__restrict__ mask, int64_t n, int64_t n_words, uint32_t *__restrict__ vbits)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t v = 0u;
    for (int r = 0; r < 32; ++r) {
        const int64_t j = w * 32 + r;
        if (j < n && !(mask && mask[j])) v |= 1u << r;
    }
    vbits[w] = v;
}

// rowoff[row][k] = b(lo) | (b(hi) + 1) << 8 (the two rows of dimension k's cumulative table, a byte each) for the window (lo, hi) of the reference's test widened by
// its own float64 roundings (T+ as in canberra_f32.hip: cbf_pack_targets8_kernel); padding dimensions: 0 (empty set)
__global__ void cbb_pack_targets_kernel(const double *__restrict__ X, int64_t m, int g, int gp, double f,
                                        const double *__restrict__ edges, uint16_t *__restrict__ rowoff)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * gp) return;
    const int64_t row = e / gp;
    const int k = (int)(e - row * gp);
    if (k >= g) { rowoff[e] = 0; return; }
    const double x = X[row * g + k];
    const double tp = (f * fabs(x)) * (1.0 + 2.3e-16) * (1.0 + 1e-12);             // T+ >= the reference's fl64(f |x|), padded
    int blo = 0, bhi = CBB_B - 1;                                                   // "cannot tell": every bucket
    if (tp == tp && tp < 1e300 && x == x) {
        const double *ed = edges + (size_t)k * (CBB_B - 1);
        // in-window  =>  |x - y| < T+  =>  y in (x - T+, x + T+); the two ends computed in float64 and moved one ulp outwards
        blo = cbb_bucket(ed, nextafter(x - tp, -__builtin_inf()));
        bhi = cbb_bucket(ed, nextafter(x + tp, __builtin_inf()));
    }
    rowoff[e] = (uint16_t)((uint32_t)blo | ((uint32_t)(bhi + 1) << 8));
}

template <int EPL>
__device__ __forceinline__ float cbb_compact(float *kb, uint32_t *ib, int count, float (&key)[EPL], uint32_t (&val)[EPL],
                                             uint32_t *last_idx = nullptr)
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
    if (last_idx) *last_idx = (uint32_t)__shfl((int)val[(L - 1) >> 6], (L - 1) & 63, 64);
    return __shfl(key[(L - 1) >> 6], (L - 1) & 63, 64);
}

// grid.x = ceil(m / (NW T)) workgroups of NW waves, grid.y = S splits of `blocks_per_split` reference blocks.
template <int GP, int EPL>
__global__ __launch_bounds__(64 * CBB_NW, 1)
void cbb_filter_kernel(const float2 *__restrict__ xq, const uint16_t *__restrict__ rowoff, int64_t m,
                       const float *__restrict__ yrow, const uint32_t *__restrict__ tab, const uint32_t *__restrict__ vbits,
                       int64_t n, int g, int64_t n_blocks, int64_t blocks_per_split, float slack, float plateau,
                       uint32_t *__restrict__ cand_idx, float *__restrict__ cand_tau)
{
    constexpr int T = CBB_T, NW = CBB_NW;
    constexpr int TB = T < NABO_CBB_TB ? T : NABO_CBB_TB;      // targets per batch of row reads (their LDS reads fly together)
    constexpr int L = 32 * EPL, CAP = L + 16 * EPL;      // kept + pending entries per list
    constexpr int WLN = 256;                             // work-list ring (entries; <= 63 pending + 64 new)
    constexpr int ROWW = CBB_ROWS * 64;                  // words of one dimension's rows
    constexpr int WAVE_BYTES = T * GP * 2 + T * CAP * 8 + T * 16 + WLN * 5;
    static_assert(WAVE_BYTES % 16 == 0, "wave block alignment");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // shared: rows [2][2][CBB_ROWS][64] u32 (a pair of dimensions of the current block, double-buffered)
    // per wave: ro [T][GP] u16 | keys [T][CAP] f32 | idx [T][CAP] u32 | tau [T] f32 | tidx [T] u32 | cnt [T] i32 | thr [T] u32 |
    //           wl [WLN] u32 | wl_t [WLN] u8
    uint32_t *rows = reinterpret_cast<uint32_t *>(smem_raw);
    unsigned char *wb = smem_raw + 4 * ROWW * 4 + (size_t)wave * WAVE_BYTES;
    uint16_t *ro = reinterpret_cast<uint16_t *>(wb);
    float *keys = reinterpret_cast<float *>(ro + T * GP);
    uint32_t *idxs = reinterpret_cast<uint32_t *>(keys + T * CAP);
    float *tau = reinterpret_cast<float *>(idxs + T * CAP);
    uint32_t *tidx = reinterpret_cast<uint32_t *>(tau + T);
    int *cnt = reinterpret_cast<int *>(tidx + T);
    uint32_t *thr_l = reinterpret_cast<uint32_t *>(cnt + T);
    uint32_t *wl = thr_l + T;
    unsigned char *wl_t = reinterpret_cast<unsigned char *>(wl + WLN);

    const int S = gridDim.y;
    const int split = blockIdx.y;
    const int64_t row0 = ((int64_t)blockIdx.x * NW + wave) * T;
    // (a wave without rows still loads and waits at the barriers with the others)
    int t_cnt = row0 >= m ? 0 : (row0 + T > m ? (int)(m - row0) : T);
    // Survivor test: n_out = g - inw dimensions are PROVEN out of window, distance >= n_out; a pair is dropped when
    // n_out >= t1 = tau + slack (+2e-5, rounded up), i.e. survivors have inw > g - t1 (canberra_f32.hip: count_threshold).
    auto count_threshold = [&](float tau_t) -> uint32_t {
        float t1 = tau_t + (2e-5f + slack);
        t1 = __uint_as_float(__float_as_uint(t1) + (t1 < __builtin_inff() ? 1u : 0u));
        const float need = (float)g - t1;
        return need < 0.0f ? 0u : (uint32_t)(int)floorf(need) + 1u;
    };
    for (int e = lane; e < T; e += 64) { tau[e] = __builtin_inff(); tidx[e] = 0xFFFFFFFFu; cnt[e] = 0; thr_l[e] = 0u; }
    for (int e = lane; e < T * GP; e += 64) {
        const int64_t row = row0 + e / GP;
        ro[e] = row < m ? rowoff[row * GP + e % GP] : (uint16_t)0;
    }
    const float below_plateau = __uint_as_float(__float_as_uint(plateau) - 1u);
    int wl_head = 0, wl_n = 0;                               // wave-uniform ring state

    // fp32 lower bound of `nb` (<= 64) work-list pairs, one per lane, then list insertion (canberra_f32.hip: drain);
    // the target's packed (x, thr) row comes from global memory here (rare: 7e-3 of the pairs).  A list accepts
    // (key, j) < (tau, tidx) lexicographically: arrival order does not matter (header).
    auto drain = [&](int nb) {
        const bool act = lane < nb;
        const int slot = (wl_head + lane) & (WLN - 1);
        const uint32_t j = act ? wl[slot] : 0u;
        const int t_p = act ? (int)wl_t[slot] : 0;
        wl_head = (wl_head + nb) & (WLN - 1);
        wl_n -= nb;
        const float4 *xp = reinterpret_cast<const float4 *>(xq + (row0 + t_p) * GP);     // (x, thr) of two dimensions
        const float4 *yp = reinterpret_cast<const float4 *>(yrow + (int64_t)j * GP);     // four dimensions
        float lb = 0.0f;
        int no_p = 0;
#pragma unroll 1
        for (int q8 = 0; q8 < GP / 8; ++q8) {
            const float4 y0 = yp[2 * q8], y1 = yp[2 * q8 + 1];
            const float4 x0 = xp[4 * q8], x1 = xp[4 * q8 + 1], x2 = xp[4 * q8 + 2], x3 = xp[4 * q8 + 3];
            const float xs_[8] = {x0.x, x0.z, x1.x, x1.z, x2.x, x2.z, x3.x, x3.z};
            const float th_[8] = {x0.y, x0.w, x1.y, x1.w, x2.y, x2.w, x3.y, x3.w};
            const float ys_[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float s = fabsf(xs_[k]) + fabsf(ys_[k]);
                const float ad = fabsf(xs_[k] - ys_[k]);
                const float nlb = fmaxf(__builtin_fmaf(s, -2.5e-07f, ad), 0.0f);
                const float den = __builtin_fmaf(s, 1.00000072f, 0.01000002f);
                const float q = nlb * __builtin_amdgcn_rcpf(den);
                const bool out = ad >= th_[k];
                no_p += out ? 1 : 0;
                lb += out ? 1.0f : q;
            }
        }
        const float key = (no_p == g) ? plateau : fminf(lb - slack, below_plateau);
        auto below = [&](float tk, uint32_t ti) { return key < tk || (key == tk && j < ti); };
        const bool hit = act && below(tau[t_p], tidx[t_p]);
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
                    uint32_t li = 0xFFFFFFFFu;
                    const float nt = cbb_compact<EPL>(keys + t2 * CAP, idxs + t2 * CAP, c, kr, vr, &li);
                    if (lane == 0) { tau[t2] = nt; tidx[t2] = li; cnt[t2] = L; thr_l[t2] = count_threshold(nt); }
                    pend = pend && below(nt, li);
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

    const int64_t b_begin = split * blocks_per_split;
    int64_t b_end = b_begin + blocks_per_split;
    if (b_end > n_blocks) b_end = n_blocks;
    // The rows of dimensions 2 dp, 2 dp + 1 of block blk (a PAIR of dimensions per step: one carry-save step serves
    // both), staged by the whole workgroup through registers into the LDS buffer the previous step has finished with:
    // the next pair is fetched before this one is counted and written behind the count; ONE barrier per step.
    // g odd: the last pair's second half is whatever follows in the table, with every target's rows (0, 0).
    constexpr int PIECES = 2 * ROWW / 4;                     // sixteen-byte pieces of a pair
    constexpr int NST = (PIECES + 64 * NW - 1) / (64 * NW);  // per thread
    static_assert(NST <= 6, "staging registers");
    uint4 st0, st1, st2, st3, st4, st5;                      // (scalars, not an array: an array captured by the lambdas went to scratch)
    st0 = st1 = st2 = st3 = st4 = st5 = make_uint4(0u, 0u, 0u, 0u);
    const int tid = (int)threadIdx.x;
    const int npair = (g + 1) / 2;
    auto fetch = [&](int64_t blk, int dp) {
        // (both dimensions of a pair are contiguous in the table; an odd g reads one dimension past the block's last --
        // the next block's first, or the table's slack dimension behind the very last block)
        const uint4 *src = reinterpret_cast<const uint4 *>(tab + ((size_t)blk * g + 2 * dp) * ROWW);
        auto ld = [&](int i) { const int pc = tid + i * 64 * NW; return src[pc < PIECES ? pc : PIECES - 1]; };
        st0 = ld(0);
        if constexpr (NST > 1) st1 = ld(1);
        if constexpr (NST > 2) st2 = ld(2);
        if constexpr (NST > 3) st3 = ld(3);
        if constexpr (NST > 4) st4 = ld(4);
        if constexpr (NST > 5) st5 = ld(5);
    };
    auto commit = [&](int buf) {
        uint4 *dst = reinterpret_cast<uint4 *>(rows + buf * 2 * ROWW);
        auto wr = [&](int i, const uint4 &v) { const int pc = tid + i * 64 * NW; if (pc < PIECES) dst[pc] = v; };
        wr(0, st0);
        if constexpr (NST > 1) wr(1, st1);
        if constexpr (NST > 2) wr(2, st2);
        if constexpr (NST > 3) wr(3, st3);
        if constexpr (NST > 4) wr(4, st4);
        if constexpr (NST > 5) wr(5, st5);
    };
    if (b_begin < b_end) {
        fetch(b_begin, 0);
        commit(0);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t blk = b_begin; blk < b_end; ++blk) {
        const uint32_t vmask = vbits[blk * 64 + lane];
        uint32_t pl[T][6];                                   // bit-sliced counters: pl[t][b] = bit b of the 32 counts
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int b = 0; b < 6; ++b) pl[t][b] = 0u;
        // Steps come in twos (round 3): the carries out of the lowest plane of two consecutive steps (weight 2 each) wait in a
        // register and enter plane 1 TOGETHER through a second 3:2 compressor, and only its carry (weight 4) ripples through
        // planes 2..5 -- 14 instructions per two steps and target instead of 24 (MODE 1: first of two, 2: second, 0: a
        // single step with the full ripple, the last one of an odd number).
        // One straight-line copy of the step per role, no lambdas around the plane arrays (captured arrays went to scratch):
        // STASH = first of two (the weight-2 carry waits in c1a), otherwise it combines (a lone last step combines with 0).
        uint32_t c1a[T];
#define CBB_STEP(DP, STASH)                                                                                              \
        {                                                                                                                \
            const int dp_ = (DP);                                                                                        \
            const bool more = dp_ + 1 < npair || blk + 1 < b_end;                                                        \
            if (more) fetch(dp_ + 1 < npair ? blk : blk + 1, dp_ + 1 < npair ? dp_ + 1 : 0);                             \
            const unsigned char *rb = reinterpret_cast<const unsigned char *>(rows + buf * 2 * ROWW + lane);            \
            const uint32_t myro = *reinterpret_cast<const uint32_t *>(ro + (lane & (T - 1)) * GP + 2 * dp_);             \
            _Pragma("unroll") for (int t0 = 0; t0 < T; t0 += TB) {                                                       \
                uint32_t m0[TB], m1[TB];                                                                                 \
                _Pragma("unroll") for (int i = 0; i < TB; ++i) {                                                         \
                    const uint32_t r4 = (uint32_t)__builtin_amdgcn_readlane((int)myro, t0 + i);                          \
                    m0[i] = *reinterpret_cast<const uint32_t *>(rb + ((r4 & 0xFF00u))) &                                 \
                            ~*reinterpret_cast<const uint32_t *>(rb + ((r4 & 0xFFu) << 8));                              \
                    m1[i] = *reinterpret_cast<const uint32_t *>(rb + ROWW * 4 + ((r4 >> 16) & 0xFF00u)) &                \
                            ~*reinterpret_cast<const uint32_t *>(rb + ROWW * 4 + ((r4 >> 8) & 0xFF00u));                 \
                }                                                                                                        \
                _Pragma("unroll") for (int i = 0; i < TB; ++i) {                                                         \
                    const int t = t0 + i;                                                                                \
                    uint32_t c = __builtin_amdgcn_bitop3_b32(pl[t][0], m0[i], m1[i], 0xE8);                              \
                    pl[t][0] = __builtin_amdgcn_bitop3_b32(pl[t][0], m0[i], m1[i], 0x96);                                \
                    if (STASH) {                                                                                         \
                        c1a[t] = c;                                                                                      \
                    } else {                                                                                             \
                        const uint32_t c4 = __builtin_amdgcn_bitop3_b32(pl[t][1], c1a[t], c, 0xE8);                      \
                        pl[t][1] = __builtin_amdgcn_bitop3_b32(pl[t][1], c1a[t], c, 0x96);                               \
                        c = c4;                                                                                          \
                        _Pragma("unroll") for (int b = 2; b < 6; ++b) {                                                  \
                            const uint32_t carry = pl[t][b] & c;                                                         \
                            pl[t][b] ^= c;                                                                               \
                            c = carry;                                                                                   \
                        }                                                                                                \
                    }                                                                                                    \
                }                                                                                                        \
            }                                                                                                            \
            if (more) commit(buf ^ 1);                                                                                   \
            __syncthreads();                                                                                             \
            buf ^= 1;                                                                                                    \
        }
        // ONE LDS read hands the pair's row offsets of all T targets to the wave (lane t holds target t's two words;
        // v_readlane then makes them scalars) -- a read + wait per target made the step a chain of LDS latencies; all the row
        // reads of a batch of TB targets fly together.  (An odd g: the second half of the last pair is a padding
        // dimension's (0, 0): the mask is x & ~x = 0.)
        for (int dp0 = 0; dp0 < npair; dp0 += 2) {
            const bool lone = dp0 + 1 >= npair;
            if (!lone) CBB_STEP(dp0, true)
            else {
#pragma unroll
                for (int t = 0; t < T; ++t) c1a[t] = 0u;
            }
            CBB_STEP(lone ? dp0 : dp0 + 1, false)
        }
#undef CBB_STEP
        // inw >= thr ?  bit-sliced comparator per target (all T unrolled: the planes are registers), then the survivors
        // into the ring, target by target
        uint32_t gev[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const uint32_t thr_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)thr_l[t]);
            uint32_t gt = 0u, eq = 0xFFFFFFFFu;
#pragma unroll
            for (int b = 5; b >= 0; --b) {                       // (branch-free: the threshold's bit selects per plane)
                const uint32_t tb = ((thr_in >> b) & 1u) ? 0xFFFFFFFFu : 0u;
                gt |= eq & pl[t][b] & ~tb;
                eq &= ~(pl[t][b] ^ tb);
            }
            gev[t] = thr_in < 64u ? ((gt | eq) & vmask) : 0u;
        }
#pragma unroll 1
        for (int t = 0; t < t_cnt; ++t) {
            uint32_t ge = gev[0];
#pragma unroll
            for (int tt = 1; tt < T; ++tt) ge = (t == tt) ? gev[tt] : ge;
            uint64_t anyb = __builtin_amdgcn_ballot_w64(ge != 0u);
            while (anyb != 0) {                                  // every lane with survivors hands over its lowest one
                const bool has = ge != 0u;
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(anyb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)anyb, 0u));
                if (has) {
                    const int r = __builtin_ctz(ge);
                    const int slot = (wl_head + wl_n + rank) & (WLN - 1);
                    wl[slot] = (uint32_t)(blk * CBB_BLK + lane * 32 + r);
                    wl_t[slot] = (unsigned char)t;
                    ge &= ge - 1u;
                }
                wl_n += __popcll(anyb);
                while (wl_n >= 64) drain(64);
                anyb = __builtin_amdgcn_ballot_w64(ge != 0u);
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
        const float nt = cbb_compact<EPL>(keys + t * CAP, idxs + t * CAP, c, kr, vr);
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

hipError_t cbb_pack_table_launch(const double *Y, int64_t n, int g, const double *edges, uint32_t *tab, hipStream_t st)
{
    const int64_t blocks = (n + CBB_BLK - 1) / CBB_BLK;
    hipLaunchKernelGGL(cbb_pack_table_kernel, dim3((unsigned)blocks, (unsigned)g), dim3(64), 0, st, Y, n, g, edges, tab);
    return hipGetLastError();
}

hipError_t cbb_valid_launch(const uint8_t *mask, int64_t n, uint32_t *vbits, hipStream_t st)
{
    const int64_t words = ((n + CBB_BLK - 1) / CBB_BLK) * 64;
    hipLaunchKernelGGL(cbb_valid_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, mask, n, words, vbits);
    return hipGetLastError();
}

hipError_t cbb_pack_targets_launch(const double *X, int64_t m, int g, int gp, double f, const double *edges, uint16_t *rowoff,
                                   hipStream_t st)
{
    const int64_t tot = m * gp;
    hipLaunchKernelGGL(cbb_pack_targets_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, X, m, g, gp, f, edges,
                       rowoff);
    return hipGetLastError();
}

void cbf_constants(int g, float *slack, float *plateau);

template <int GP, int EPL>
static hipError_t cbb_launch_one(const float *xq, const uint16_t *rowoff, int64_t m, const float *yrow, const uint32_t *tab,
                                 const uint32_t *vbits, int64_t n, int g, int S, uint32_t *cand_idx, float *cand_tau,
                                 hipStream_t st)
{
    constexpr int L = 32 * EPL, CAP = L + 16 * EPL;
    const int64_t n_blocks = (n + CBB_BLK - 1) / CBB_BLK;
    const int64_t bps = (n_blocks + S - 1) / S;
    float slack, plateau;
    cbf_constants(g, &slack, &plateau);
    constexpr size_t lds = (size_t)4 * CBB_ROWS * 64 * 4 + (size_t)CBB_NW * (CBB_T * GP * 2 + CBB_T * CAP * 8 + CBB_T * 16 + 256 * 5);
    static_assert(lds <= 163840, "LDS budget");
    auto kern = &cbb_filter_kernel<GP, EPL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int rpw = CBB_T * CBB_NW;
    dim3 grid((unsigned)((m + rpw - 1) / rpw), S), block(64 * CBB_NW);
    hipLaunchKernelGGL(kern, grid, block, lds, st, reinterpret_cast<const float2 *>(xq), rowoff, m, yrow, tab, vbits, n, g,
                       n_blocks, bps, slack, plateau, cand_idx, cand_tau);
    return hipGetLastError();
}

// instantiated for g <= 64 (six count planes hold 63) and 32-entry lists (k + drop_first <= 24)
bool cbb_available(int g, int gp, int epl) { return g <= 63 && gp <= 64 && epl == 1; }

hipError_t cbb_filter_launch(int gp, const float *xq, const uint16_t *rowoff, int64_t m, const float *yrow, const uint32_t *tab,
                             const uint32_t *vbits, int64_t n, int g, int S, uint32_t *cand_idx, float *cand_tau, hipStream_t st)
{
#define NABO_CBB(GPV) case GPV: return cbb_launch_one<GPV, 1>(xq, rowoff, m, yrow, tab, vbits, n, g, S, cand_idx, cand_tau, st);
    switch (gp) {
        NABO_CBB(8) NABO_CBB(16) NABO_CBB(24) NABO_CBB(32) NABO_CBB(40) NABO_CBB(48) NABO_CBB(56) NABO_CBB(64)
    default: return hipErrorInvalidValue;
    }
#undef NABO_CBB
}

}  // namespace nabo
