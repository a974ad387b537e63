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
constexpr int CBB_ROWS = CBB_B;           // STORED cumulative rows per dimension: rows 1 .. CBB_B (row 0, the empty set, is a zero row in LDS)
constexpr int CBB_BLK = 2048;             // references per block: 64 lanes x 32 bits
constexpr int CBB_T = 8;                  // target rows per wave (six count planes each, in registers: < 128 VGPRs per wave)
#ifndef NABO_CBB_TB
#define NABO_CBB_TB 8
#endif
constexpr int CBB_NW = 16;                // waves per workgroup (four per SIMD): they share the LDS copy of the table rows
constexpr int CBB_WLN = 128;              // work-list ring of a wave (entries)
constexpr int CBB_DIM_BYTES = CBB_ROWS * 256;                          // the rows of one dimension of one block: 16 KiB
constexpr int CBB_PAIR_BYTES = 2 * CBB_DIM_BYTES;                      // ... of a pair of dimensions: contiguous in the table
constexpr int CBB_PIECES = CBB_PAIR_BYTES / 1024;                      // 1-KiB LDS-DMA pieces per pair: two per wave (64 buckets)
constexpr int CBB_ZERO_OFF = CBB_PAIR_BYTES;                           // the zero row ("row 0") behind the pair, never overwritten
constexpr int CBB_BUF_BYTES = CBB_PAIR_BYTES + 256;                    // a row buffer in LDS
constexpr int CBB_BUF1 = 65536;                                        // LDS offset of the second row buffer (one address bit)
static_assert(CBB_PAIR_BYTES % 1024 == 0, "whole DMA pieces");
// Who requests the table pieces: waves 0 .. DMAW-1, CBB_PPW CONSECUTIVE 1-KiB pieces each (one M0 write, the instruction's
// immediate offset moves the global AND the LDS address of the pieces behind the first).
// (Tried in round 4 and dropped: an L2 prefetch of the pair three steps ahead -- a dword per lane and 128-byte line by LDS-DMA
// into a scratch row, vmcnt leaving it in flight for a step: 544 -> 571 ms.  The pass is not waiting for table misses; its
// SIMDs issue an instruction every 4.9 cycles, 82 % of what they can: what counts is the instruction count.)
#ifndef NABO_CBB_SEED
#define NABO_CBB_SEED 512
#endif
#ifndef NABO_CBB_DMAW
#define NABO_CBB_DMAW 4          // (16 waves x 2 pieces: 527 ms at 1M x 1M; 8 x 4: 507; 4 x 8: 503; 2 x 16: 512)
#endif
constexpr int CBB_DMAW = NABO_CBB_DMAW;                                // waves 0 .. DMAW-1 request the table pieces
constexpr int CBB_PPW = CBB_PIECES / CBB_DMAW;                         // consecutive pieces a DMA wave requests per pair
static_assert(CBB_PIECES % CBB_DMAW == 0 && CBB_DMAW <= CBB_NW && CBB_PPW >= 1 && (CBB_PPW <= 4 || CBB_PPW % 4 == 0), "pieces per DMA wave: up to four per statement (13-bit immediate offset)");
static_assert(CBB_BUF_BYTES <= CBB_BUF1 && CBB_BUF_BYTES <= 65536, "row addresses (16 bits) and the buffer bit must not overlap");
// per-wave LDS block: ro2 [npair <= gp/2][T] uint2 | keys, idx [T][CAP] | tau, tidx, cnt, thr [T] | wl [WLN] u32 | wl_t [WLN] u8
// (kept + pending list entries: 16 (12) pending per 32 kept; ro2 entry npair repeats entry 0 -- gp / 2 entries hold it for every
// g the kernel is instantiated for: g <= 63 < gp whenever gp / 2 == npair would be needed)
constexpr int cbb_cap(int gp, int epl) { return 32 * epl + (gp >= 64 ? 12 : 16) * epl; }   // (gp = 64: the LDS budget)
constexpr int cbb_ro2_pairs(int gp) { return gp / 2 + 1; }
constexpr int cbb_wave_bytes(int gp, int epl) { return cbb_ro2_pairs(gp) * CBB_T * 8 + CBB_T * cbb_cap(gp, epl) * 8 + CBB_T * 16 + CBB_WLN * 5; }
constexpr size_t cbb_lds_bytes(int gp, int epl)
{
    const int wbytes = cbb_wave_bytes(gp, epl);
    const int na = (CBB_BUF1 - CBB_BUF_BYTES) / wbytes;
    return (size_t)CBB_BUF1 + CBB_BUF_BYTES + (size_t)(CBB_NW > na ? CBB_NW - na : 0) * wbytes;
}

constexpr int cbb_gpad(int g) { return (g + 1) & ~1; }                 // dimensions per block in the table
int cbb_buckets() { return CBB_B; }
int cbb_rows_per_wg() { return CBB_T * CBB_NW; }
size_t cbb_table_bytes(int64_t n, int g)
{
    // a block holds an EVEN number of dimensions (an odd g: one dimension nobody writes or uses -- every target's rows there
    // are (0, 0), the empty mask), so that the kernel's fetch pointer advances by one pair per step and nothing else;
    // + one pair of slack: the step behind the last one requests a pair too
    return ((size_t)((n + CBB_BLK - 1) / CBB_BLK) * cbb_gpad(g) + 2) * CBB_ROWS * 64 * sizeof(uint32_t);
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
    __shared__ uint32_t eq[CBB_B][64];
    const int w = threadIdx.x, d = blockIdx.y;
    const int64_t blk = blockIdx.x;
#pragma unroll
    for (int r = 0; r < CBB_B; ++r) eq[r][w] = 0u;
    const double *ed = edges + (size_t)d * (CBB_B - 1);
    for (int r = 0; r < 32; ++r) {
        const int64_t j = blk * CBB_BLK + (int64_t)w * 32 + r;
        if (j < n) eq[cbb_bucket(ed, Y[j * g + d])][w] |= 1u << r;             // (column w is this lane's own: no races)
    }
    // stored row r - 1 = cumulative row r = buckets 0 .. r - 1 (cumulative row 0, the empty set, is not stored)
    uint32_t acc = 0u;
    uint32_t *o = tab + ((size_t)(blk * cbb_gpad(g) + d) * CBB_ROWS) * 64 + w;
    for (int r = 0; r < CBB_B; ++r) {
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

// LDS-DMA: NP consecutive pieces of 64 lanes x 16 bytes from gbase + lane_off (+ 1024 per piece) to lds_dst + lane * 16 (+ 1024
// per piece): M0 is written once, in the statement that reads it, and declared clobbered; the immediate offset of the
// instruction is added to the global and to the LDS address alike.  hipcc neither counts nor drains these loads -- the
// kernel waits for them itself (cbb_dma_wait).
template <int NP>
__device__ __forceinline__ void cbb_glds16(const void *gbase /* wave-uniform */, uint32_t lane_off, uint32_t lds_dst)
{
    static_assert(NP >= 1 && NP <= 4, "immediate offsets up to 3072");
#ifdef NABO_CBB_M0PER            // (A/B: the immediate offset moving the global address only)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory", "m0");
#pragma unroll
    for (int pc = 1; pc < NP; ++pc)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" : : "v"(lane_off), "s"(gbase), "s"(lds_dst + pc * 1024u), "n"(pc * 1024) : "memory", "m0");
#else
    if constexpr (NP == 1)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory", "m0");
    else if constexpr (NP == 2)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                     : : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072"
                     : : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory", "m0");
    static_assert(NP != 3, "1, 2 or 4 pieces");
#endif
}
__device__ __forceinline__ void cbb_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#ifdef NABO_CBB_NOBARRIER         // timing experiment (garbage results): the steps run without their workgroup barrier
#define CBB_STEP_BARRIER() do { } while (0)
#else
#define CBB_STEP_BARRIER() __syncthreads()
#endif
typedef uint32_t cbb_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t cbb_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const cbb_u32x4 cbb_lds_u4;
typedef __attribute__((address_space(3))) const cbb_u32x2 cbb_lds_u2;

// grid.x = ceil(m / (NW T)) workgroups of NW waves, grid.y = S splits of `blocks_per_split` reference blocks.
//
// Round 4: FOUR WORDS PER LANE.  The round-3 form (lane = one word of a row, eight targets one after another) issued 223
// instructions per step of eight targets -- 142 vector, 45 scalar, 36 LDS -- and, every instruction of a SIMD costing its ~4
// issue cycles whatever its kind, ran at exactly that price (4.7e11 instructions per 1M x 1M step = 0.77 s).  Now a lane holds
// FOUR consecutive words (128 references) and a group of 16 lanes covers a table row, so ONE ds_read_b128 fetches a row
// for FOUR targets at once (lane group q = lane >> 4 reads the row of target 4 s + q; rows are 256-byte aligned, so any
// combination of rows is conflict-free in the b128 lane grouping), row addresses are per-lane registers built from ONE LDS
// word per target and dimension pair (no v_readlane, no scalar unpacking), and the table rows arrive by LDS-DMA instead of
// through registers.  The counters take four dimension pairs per carry-save cycle (weight-2 carries of two steps enter
// plane 1 together, weight-4 carries of two such pairs enter plane 2 together, one weight-8 carry ripples on): 2.5 instead
// of 3.5 instructions per dimension and word.  ~95 instructions per step of eight targets.
template <int GP, int EPL>
__global__ __launch_bounds__(64 * CBB_NW, 1)
void cbb_filter_kernel(const float2 *__restrict__ xq, const uint16_t *__restrict__ rowoff, int64_t m,
                       const float *__restrict__ yrow, const uint32_t *__restrict__ tab, const uint32_t *__restrict__ vbits,
                       int64_t n, int g, int64_t n_blocks, int64_t blocks_per_split, float slack, float plateau,
                       uint32_t *__restrict__ cand_idx, float *__restrict__ cand_tau)
{
    constexpr int T = CBB_T, NW = CBB_NW, TS = T / 4;   // TS target slots: slot s, lane group q -> target 4 s + q of the wave
    constexpr int L = 32 * EPL, CAP = cbb_cap(GP, EPL);  // kept + pending entries per list
    constexpr int WLN = CBB_WLN;                         // work-list ring (entries; <= 63 pending + 64 new)
    constexpr int ROWW = CBB_ROWS * 64;                  // words of one dimension's rows in the table
    constexpr int WAVE_BYTES = cbb_wave_bytes(GP, EPL);
    constexpr int NA = (CBB_BUF1 - CBB_BUF_BYTES) / WAVE_BYTES;      // waves whose block lies between the two row buffers
    static_assert(WAVE_BYTES % 16 == 0, "wave block alignment");
    static_assert(T % 4 == 0 && TS >= 1, "targets come in fours (one per group of 16 lanes)");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q = lane >> 4, sub = lane & 15;
    // LDS map: [0, BUF_BYTES) row buffer 0 | blocks of waves 0 .. NA-1 | [BUF1, BUF1 + BUF_BYTES) row buffer 1 | the other
    // waves' blocks.  A row buffer holds a PAIR of dimensions of the current block (2 x 65 rows x 256 B, padded to whole
    // 1-KiB DMA pieces); buffer 1 starts at 2^16 so that "which buffer" is one address bit that ORs with row and lane bits.
    // per wave: ro2 [GP/2 + 1][T] uint2 | keys [T][CAP] f32 | idx [T][CAP] u32 | tau [T] f32 | tidx [T] u32 | cnt [T] i32 | thr [T] u32 |
    //           wl [WLN] u32 | wl_t [WLN] u8
    unsigned char *wb = smem_raw + (wave < NA ? CBB_BUF_BYTES + wave * WAVE_BYTES : CBB_BUF1 + CBB_BUF_BYTES + (wave - NA) * WAVE_BYTES);
    uint2 *ro2 = reinterpret_cast<uint2 *>(wb);
    float *keys = reinterpret_cast<float *>(ro2 + cbb_ro2_pairs(GP) * T);
    uint32_t *idxs = reinterpret_cast<uint32_t *>(keys + T * CAP);
    float *tau = reinterpret_cast<float *>(idxs + T * CAP);
    uint32_t *tidx = reinterpret_cast<uint32_t *>(tau + T);
    int *cnt = reinterpret_cast<int *>(tidx + T);
    uint32_t *thr_l = reinterpret_cast<uint32_t *>(cnt + T);
    uint32_t *wl = thr_l + T;
    unsigned char *wl_t = reinterpret_cast<unsigned char *>(wl + WLN);
    // LDS addresses below are byte offsets from the start of the dynamic segment, which IS LDS address 0: this kernel has no
    // static LDS (cbb_launch_one checks hipFuncGetAttributes once) -- going through the pointer costs a v_add per row read.
    const uint32_t ro2_off = (uint32_t)(wb - smem_raw);

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
    // ro2[dp][t] = the four row ADDRESSES inside a row buffer (16 bits each) of target t in dimensions 2 dp, 2 dp + 1:
    // x = lo | hi << 16 of the first, y of the second.  rowoff holds the cumulative row numbers b(lo) | (b(hi) + 1) << 8
    // per dimension (0 | 0 for padding): cumulative row r >= 1 is stored row r - 1 of its dimension's half of the buffer,
    // cumulative row 0 (the empty set) is the zero row behind the pair.
    auto row_addr = [](uint32_t r, uint32_t half) -> uint32_t {
        return r ? (r - 1u) * 256u + half * (uint32_t)CBB_DIM_BYTES : (uint32_t)CBB_ZERO_OFF;
    };
    const int npair = (g + 1) / 2;
    for (int e = lane; e < (npair + 1) * T; e += 64) {       // entry npair = entry 0: the step behind a block's last is the next block's first
        const int de = e / T, t = e - de * T;
        const int dp = de == npair ? 0 : de;
        const int64_t row = row0 + t;
        uint32_t xa = 0u, xc = 0u;
        if (row < m) { xa = rowoff[row * GP + 2 * dp]; xc = rowoff[row * GP + 2 * dp + 1]; }
        ro2[e] = make_uint2(row_addr(xa & 0xFFu, 0u) | (row_addr(xa >> 8, 0u) << 16), row_addr(xc & 0xFFu, 1u) | (row_addr(xc >> 8, 1u) << 16));
    }
    if (wave == 0) {             // the zero rows of both buffers (the DMA never touches them)
        *reinterpret_cast<uint32_t *>(smem_raw + CBB_ZERO_OFF + lane * 4) = 0u;
        *reinterpret_cast<uint32_t *>(smem_raw + CBB_BUF1 + CBB_ZERO_OFF + lane * 4) = 0u;
    }
    const float below_plateau = __uint_as_float(__float_as_uint(plateau) - 1u);
    int wl_head = 0, wl_n = 0;                               // wave-uniform ring state

    // fp32 lower bound of `nb` (<= 64) work-list pairs, one per lane, then list insertion (canberra_f32.hip: drain);
    // the target's packed (x, thr) row comes from global memory here (rare: 3e-3 of the pairs).  A list accepts
    // (key, j) < (tau, tidx) lexicographically: arrival order does not matter (header).
    // (Round 4, tried and dropped: the survivor's reference row requested up front -- one word per 128-byte line before the
    // loop -- and ONE run-time copy of the extraction loop instead of eight: 507 -> 532 / 556 ms.  A drain is not what the
    // step barrier waits for.)
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
                const float qq = nlb * __builtin_amdgcn_rcpf(den);
                const bool out = ad >= th_[k];
                no_p += out ? 1 : 0;
                lb += out ? 1.0f : qq;
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
    // The rows of dimensions 2 dp, 2 dp + 1 of block blk -- contiguous in the table, 32 KiB -- go to the row buffer the
    // previous step has finished with as 32 LDS-DMA pieces of 1 KiB (DMA wave w: pieces PPW w .. PPW w + PPW - 1): requested
    // at the top of a step, waited for (every wave for its own pieces) in front of the step's ONE barrier.  A block holds
    // an even number of dimensions in the table (cbb_gpad) and the pairs are visited in table order, so the fetch stream is
    // ONE scalar pointer that moves on by a pair per step; the step behind the split's last requests a pair nobody reads
    // (the next split's first, or the table's slack).
    const unsigned char *fsrc = reinterpret_cast<const unsigned char *>(tab) + ((size_t)b_begin * cbb_gpad(g) * ROWW) * 4 +
                                (size_t)wave * (CBB_PPW * 1024);
    const uint32_t lane_off = (uint32_t)lane * 16u;
    const bool dma_wave = CBB_DMAW == CBB_NW || wave < CBB_DMAW;
    const uint32_t fdst = (uint32_t)wave * (uint32_t)(CBB_PPW * 1024);
#ifdef NABO_CBB_NODMA            // timing experiment (garbage results): only the very first pair is ever fetched
    bool fetched = false;
#endif
    auto fetch = [&](int buf) {
#ifdef NABO_CBB_NODMA
        if (!fetched)
#endif
        if (dma_wave) {
            if constexpr (CBB_PPW <= 4) {
                cbb_glds16<CBB_PPW>(fsrc, lane_off, ((uint32_t)buf << 16) + fdst);
            } else {
#pragma unroll
                for (int q4 = 0; q4 < CBB_PPW / 4; ++q4)
                    cbb_glds16<4>(fsrc + q4 * 4096, lane_off, ((uint32_t)buf << 16) + fdst + (uint32_t)(q4 * 4096));
            }
        }
#ifdef NABO_CBB_DMA2X            // timing experiment (same results): every piece is requested twice
        if (dma_wave) cbb_glds16<CBB_PPW>(fsrc, lane_off, ((uint32_t)buf << 16) + fdst);
#endif
#ifdef NABO_CBB_NODMA
        fetched = true;
#endif
        fsrc += CBB_PAIR_BYTES;
    };
    if (b_begin < b_end) fetch(0);                           // (an empty split requests nothing: its range may lie past the table)
    cbb_u32x2 rnx[TS];                                      // row addresses of the step about to run (CBB_STEP)
#pragma unroll
    for (int s = 0; s < TS; ++s) rnx[s] = cbb_u32x2{0u, 0u};
    cbb_dma_wait();
    __syncthreads();                                         // (also: ro2 and the zero rows are written)
    const uint32_t ro_q = ro2_off + (uint32_t)(q * 8);       // this lane group's column of ro2
#pragma unroll
    for (int s = 0; s < TS; ++s) rnx[s] = *(cbb_lds_u2 *)(uintptr_t)(ro_q + (uint32_t)(4 * s * 8));
    // SEED (round 4).  A split's lists start with threshold +inf, i.e. count threshold 0: EVERY reference of its first block
    // survives the count and goes through the fp32 bound -- 2048 x 8 pairs, 256 drains per wave, as much as ~10 ordinary
    // blocks (2 % of a 1M-reference stream, a fifth of a 100k one).  So the first CBB_SEED references of the split go through
    // the bound BEFORE the count starts: the lists open with the 32 best of those (a threshold at their 6 % quantile), the
    // first block's count then lets a fraction of its references through instead of all, and its extraction skips the seeded words.
    // (Measured, same box: 1M x 1M kernel 504 -> 486 / 480 / 460-470 / 462 ms with 128 / 256 / 512 / 1024 seeds, 100k x 100k
    // 16.3 -> 13.0 / 11.5 / 10.6 / 11.4: the later blocks gain from the tighter start too.)
    constexpr int CBB_SEED = NABO_CBB_SEED;                  // references, a multiple of 128 (whole lanes of the block), <= 2048
    static_assert(CBB_SEED % 128 == 0 && CBB_SEED >= 0 && CBB_SEED <= CBB_BLK, "seeded references: whole lanes of the first block");
    if (CBB_SEED > 0 && b_begin < b_end) {
        const int64_t j0 = b_begin * CBB_BLK;
        for (int t = 0; t < t_cnt; ++t) {
            for (int ch = 0; ch < CBB_SEED / 64; ++ch) {
                const int64_t j = j0 + ch * 64 + lane;
                // (valid bits: word (j - block start) / 32 of the block, bit j % 32)
                const bool ok = j < n && ((vbits[b_begin * 64 + (ch * 64 + lane) / 32] >> (lane & 31)) & 1u) != 0u;
                const uint64_t mk = __builtin_amdgcn_ballot_w64(ok);
                if (ok) {
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                    const int slot = (wl_head + wl_n + rank) & (WLN - 1);
                    wl[slot] = (uint32_t)j;
                    wl_t[slot] = (unsigned char)t;
                }
                wl_n += __popcll(mk);
                while (wl_n >= 64) drain(64);
            }
        }
        while (wl_n > 0) drain(wl_n < 64 ? wl_n : 64);       // (the count thresholds of the first block come from these lists)
    }
    int buf = 0;
    for (int64_t blk = b_begin; blk < b_end; ++blk) {
        uint32_t rp = ro_q;                                  // ro2 entry of the current group of steps (DPI = 0)
        uint32_t pl[TS][4][6];                               // bit-sliced counters: pl[s][w][b] = bit b of the 32 counts of word w
        uint32_t c2a[TS][4], c4a[TS][4];                     // carries waiting for their partner (weight 2, weight 4)
#pragma unroll
        for (int s = 0; s < TS; ++s)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
#pragma unroll
                for (int b = 0; b < 6; ++b) pl[s][w][b] = 0u;
                c2a[s][w] = c4a[s][w] = 0u;
            }
        // One step = one pair of dimensions.  ROLE: 0 / 2 first of two (the weight-2 carry waits), 1 second of two and first
        // of four (the weight-4 carry waits), 3 last of four (weight-8 carry ripples through planes 3..5), 5 second of two at
        // the end of an odd number of pairs-of-steps (the weight-4 carry ripples through planes 2..5), 4 a lone last step.
#define CBB_STEP(DPI, ROLE)                                                                                                  \
        {                                                                                                                    \
            /* the row addresses come from rnx, read from ro2 BEFORE the previous step's barrier: all eight row reads of the  \
               step leave at once, the table rows of the next step are requested while they fly, slot 1's rows arrive under   \
               slot 0's arithmetic */                                                                                        \
            const uint32_t lb = (uint32_t)sub * 16u | ((uint32_t)buf << 16);      /* row | lane | buffer bits never overlap */  \
            cbb_u32x4 lA[TS], hA[TS], lB[TS], hB[TS];                                                                        \
            _Pragma("unroll") for (int s = 0; s < TS; ++s) {                                                                 \
                lA[s] = *(cbb_lds_u4 *)(uintptr_t)((rnx[s].x & 0xFFFFu) | lb);                                               \
                hA[s] = *(cbb_lds_u4 *)(uintptr_t)((rnx[s].x >> 16) | lb);                                                   \
                lB[s] = *(cbb_lds_u4 *)(uintptr_t)((rnx[s].y & 0xFFFFu) | lb);                                               \
                hB[s] = *(cbb_lds_u4 *)(uintptr_t)((rnx[s].y >> 16) | lb);                                                   \
            }                                                                                                                \
            fetch(buf ^ 1);                                                                                                  \
            /* ro2 of the NEXT step (wave-private: no barrier needed; constant offsets from the group's pointer) */          \
            _Pragma("unroll") for (int s = 0; s < TS; ++s)                                                                   \
                rnx[s] = *(cbb_lds_u2 *)(uintptr_t)(rp + (uint32_t)((((DPI) + 1) * T + 4 * s) * 8));                         \
            _Pragma("unroll") for (int s = 0; s < TS; ++s) {                                                                 \
                const uint32_t la_[4] = {lA[s].x, lA[s].y, lA[s].z, lA[s].w}, ha_[4] = {hA[s].x, hA[s].y, hA[s].z, hA[s].w}; \
                const uint32_t lb_[4] = {lB[s].x, lB[s].y, lB[s].z, lB[s].w}, hb_[4] = {hB[s].x, hB[s].y, hB[s].z, hB[s].w}; \
                _Pragma("unroll") for (int w = 0; w < 4; ++w) {                                                              \
                    const uint32_t m0 = ha_[w] & ~la_[w], m1 = hb_[w] & ~lb_[w];                                             \
                    uint32_t c = __builtin_amdgcn_bitop3_b32(pl[s][w][0], m0, m1, 0xE8);                                     \
                    pl[s][w][0] = __builtin_amdgcn_bitop3_b32(pl[s][w][0], m0, m1, 0x96);                                    \
                    if ((ROLE) == 0 || (ROLE) == 2) {                                                                        \
                        c2a[s][w] = c;                                                                                       \
                    } else if ((ROLE) == 4) {                                                                                \
                        _Pragma("unroll") for (int b = 1; b < 6; ++b) {                                                      \
                            const uint32_t carry = pl[s][w][b] & c;                                                          \
                            pl[s][w][b] ^= c;                                                                                \
                            c = carry;                                                                                       \
                        }                                                                                                    \
                    } else {                                                                                                 \
                        uint32_t c4 = __builtin_amdgcn_bitop3_b32(pl[s][w][1], c2a[s][w], c, 0xE8);                          \
                        pl[s][w][1] = __builtin_amdgcn_bitop3_b32(pl[s][w][1], c2a[s][w], c, 0x96);                          \
                        if ((ROLE) == 1) {                                                                                   \
                            c4a[s][w] = c4;                                                                                  \
                        } else {                                                                                             \
                            int b0 = 2;                                                                                      \
                            if ((ROLE) == 3) {                                                                               \
                                const uint32_t c8 = __builtin_amdgcn_bitop3_b32(pl[s][w][2], c4a[s][w], c4, 0xE8);           \
                                pl[s][w][2] = __builtin_amdgcn_bitop3_b32(pl[s][w][2], c4a[s][w], c4, 0x96);                 \
                                c4 = c8;                                                                                     \
                                b0 = 3;                                                                                      \
                            }                                                                                                \
                            _Pragma("unroll") for (int b = 2; b < 6; ++b) {                                                  \
                                if (b >= b0) {                                                                               \
                                    const uint32_t carry = pl[s][w][b] & c4;                                                 \
                                    pl[s][w][b] ^= c4;                                                                       \
                                    c4 = carry;                                                                              \
                                }                                                                                            \
                            }                                                                                                \
                        }                                                                                                    \
                    }                                                                                                        \
                }                                                                                                            \
                /* the step's results are materialised HERE: hipcc otherwise sinks the whole counter update of the first three  \
                   steps of a cycle into the fourth (their values are only read there), holding 3 x 32 mask registers: spills */  \
                _Pragma("unroll") for (int w = 0; w < 4; ++w) {                                                              \
                    asm volatile("" : "+v"(pl[s][w][0]));                                                                    \
                    if ((ROLE) == 0 || (ROLE) == 2) asm volatile("" : "+v"(c2a[s][w]));                                      \
                    else asm volatile("" : "+v"(pl[s][w][1]));                                                               \
                    if ((ROLE) == 1) asm volatile("" : "+v"(c4a[s][w]));                                                     \
                    if ((ROLE) >= 3) { _Pragma("unroll") for (int b = 2; b < 6; ++b) asm volatile("" : "+v"(pl[s][w][b])); }  \
                }                                                                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                                           \
            }                                                                                                                \
            cbb_dma_wait();                                                                                                  \
            CBB_STEP_BARRIER();                                                                                              \
            buf ^= 1;                                                                                                        \
        }
        int dp0 = 0;
        for (; dp0 + 4 <= npair; dp0 += 4) {
            CBB_STEP(0, 0)
            CBB_STEP(1, 1)
            CBB_STEP(2, 2)
            CBB_STEP(3, 3)
            rp += (uint32_t)(4 * T * 8);
        }
        if (dp0 + 2 <= npair) {
            CBB_STEP(0, 0)
            CBB_STEP(1, 5)
            rp += (uint32_t)(2 * T * 8);
            dp0 += 2;
        }
        if (dp0 < npair) CBB_STEP(0, 4)
#undef CBB_STEP
        // inw >= thr ?  Adding K = 64 - thr to the six-plane count carries out of plane 5 exactly when inw + K >= 64: ONE
        // majority per plane (the threshold's bits are per lane group: a register each, all ones or zero) instead of a
        // greater / equal pair per plane.  thr = 0: everything survives (K = 64 has no bits below 64); thr >= 64: nothing.
        uint32_t gev[TS][4];
        const uint4 vmask = reinterpret_cast<const uint4 *>(vbits)[blk * 16 + sub];
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            const uint32_t thr_in = thr_l[4 * s + q];
            const uint32_t kk = (64u - thr_in) & 63u;
            uint32_t kb[6];
#pragma unroll
            for (int b = 0; b < 6; ++b) kb[b] = (uint32_t)__builtin_amdgcn_sbfe((int)kk, b, 1);           // all ones where bit b of K is set
            const uint32_t all = thr_in == 0u ? 0xFFFFFFFFu : 0u;
            const uint32_t some = thr_in < 64u ? 0xFFFFFFFFu : 0u;        // (selects, not branches: the lane groups differ)
            const uint32_t vm_[4] = {vmask.x & some, vmask.y & some, vmask.z & some, vmask.w & some};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t c = pl[s][w][0] & kb[0];
#pragma unroll
                for (int b = 1; b < 6; ++b) c = __builtin_amdgcn_bitop3_b32(pl[s][w][b], kb[b], c, 0xE8);
                gev[s][w] = __builtin_amdgcn_bitop3_b32(c, all, vm_[w], 0xA8);      // (c | all) & vm
            }
        }
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            const int t_mine = 4 * s + q;                        // this lane group's target in slot s
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t ge = t_mine < t_cnt ? gev[s][w] : 0u;
                if (CBB_SEED > 0 && blk == b_begin && sub < CBB_SEED / 128) ge = 0u;       // (went through the bound as seeds)
                uint64_t anyb = __builtin_amdgcn_ballot_w64(ge != 0u);
                while (anyb != 0) {                              // every lane with survivors hands over its lowest one
                    const bool has = ge != 0u;
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(anyb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)anyb, 0u));
                    if (has) {
                        const int r = __builtin_ctz(ge);
                        const int slot = (wl_head + wl_n + rank) & (WLN - 1);
                        wl[slot] = (uint32_t)(blk * CBB_BLK + sub * 128 + w * 32 + r);
                        wl_t[slot] = (unsigned char)t_mine;
                        ge &= ge - 1u;
                    }
                    wl_n += __popcll(anyb);
                    while (wl_n >= 64) drain(64);
                    anyb = __builtin_amdgcn_ballot_w64(ge != 0u);
                }
            }
        }
    }
    cbb_dma_wait();                                          // (no LDS-DMA may outlive the workgroup's LDS)
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
    const int64_t n_blocks = (n + CBB_BLK - 1) / CBB_BLK;
    const int64_t bps = (n_blocks + S - 1) / S;
    float slack, plateau;
    cbf_constants(g, &slack, &plateau);
    constexpr size_t lds = cbb_lds_bytes(GP, EPL);
    static_assert(lds <= 163840, "LDS budget");
    auto kern = &cbb_filter_kernel<GP, EPL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    {   // the kernel addresses its dynamic LDS segment from 0: it must not have a static one
        static int static_lds = -1;
        if (static_lds < 0) {
            hipFuncAttributes fa;
            if ((e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern))) != hipSuccess) return e;
            static_lds = (int)fa.sharedSizeBytes;
        }
        if (static_lds != 0) return hipErrorInvalidConfiguration;
    }
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
