// l2s_topk.hip -- Euclidean score + top-L filter on the f16 matrix pipe, reference tiles SHARED per workgroup
// through an LDS ring (gfx950).  Same contract as l2_topk.hip / l2h_topk.hip: per target row the L references with
// the smallest score  s = ||y||^2 - 2 x.y  and the threshold below which nothing was dropped; refine.hip
// re-evaluates the candidates in the reference's float64 arithmetic (nabo/_mapping.py:16-26) and certifies the row
// with the error bound of the split, so the results are the same bits as every other path.
//
// What is different from l2h_topk.hip (per-wave streaming, one wave per SIMD):
//   * K-CONCATENATED f16x3 split.  v = hi + lo (two f16, 22 significant bits).  Instead of three separate products
//     per 16-component slab (12 MFMAs at g = 50), reference cells are packed as ONE vector [hi | lo | hi] of
//     3 (g+1) slots and targets as [hi | hi | lo] (x -2), so  <ref, tgt> = hi.hi + lo.hi + hi.lo  in
//     KC = ceil(3 (g+1) / 16) steps of v_mfma_f32_32x32x16_f16: 10 MFMAs at g = 50.  Slot g of every segment carries
//     the norm term (||y~||^2 2^-15 as hi + lo against the constant 2^15), so a chain starts from C = 0.
//   * ONE copy of each reference tile per CU.  A workgroup is 8 waves (two per SIMD); every wave owns R = 2 row-blocks
//     (64 target rows, B operands resident in VGPRs).  Reference tiles (KC KiB) arrive by LDS-DMA
//     (global_load_lds_dwordx4, no VGPR staging) into a 3-deep ring: per tile the CU fetches KC KiB instead of
//     4 waves x 8 KiB, and L2 / fabric traffic drops accordingly.  One barrier per tile: tile t+3's DMA is issued,
//     the wave runs its two chains on tile t (A operands in registers) while reading tile t+1's A operands from the
//     ring into a second register set, waits for ITS OWN pieces of tile t+2 with a counted s_waitcnt vmcnt, barrier.
//   * Two waves per SIMD: the f16 matrix pipe overlaps with another wave's vector / LDS work, so one wave's filter,
//     hit path (topk_lists.h) and waits are covered by its partner's MFMAs.
//   * Candidate lists: 28 entries per row (kept k'+8, the rest pending) + a staging area per wave (topk_lists.h),
//     wave-private, 127 KiB; ring 3 x KC KiB.
#include <cstdlib>

#include <hip/hip_fp16.h>

#include "knn_common.h"
#include "topk_lists.h"

namespace nabo {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int L2S_WAVES = 8;        // waves per workgroup (two per SIMD)
constexpr int L2S_R = 2;            // row-blocks per wave
constexpr int L2S_NBUF = 3;         // ring depth (tiles)

constexpr int L2S_NREC = 32;        // staging records per wave (topk_lists.h)
__host__ __device__ constexpr int l2s_row_entries(int kc) { (void)kc; return 25; }   // list entries per row (odd): LDS budget

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + lane * 16 (cdna_hip_programming.md,
// "What hipcc does not do": M0 written in the statement that reads it; hipcc does not count this load).
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int KC>
__device__ __forceinline__ f32x16 cchain(const f16x8 (&a)[KC], const f16x8 (&b)[KC])
{
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < KC; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[s], acc, 0, 0, 0);
    return acc;
}

// Grid: x = target super-blocks (8 waves x 2 tiles of 32 rows = 512 rows), y = reference splits.
//
// SYNC = 0: one workgroup barrier per tile.  SYNC = 1: no barriers -- two monotonic LDS counters per ring slot:
//   full[s]  += 1 per wave once ITS pieces of the tile bound for slot s have landed (8 per fill),
//   free[s]  += 1 per wave once it has copied the tile in slot s into its registers (8 per tile).
// A wave in step u (tile u in registers) waits only for events of step u-1 of the other waves (everyone has copied
// tile u out of its slot before tile u+3 is DMA'ed into it; everyone's pieces of tile u+1 have landed before it is
// read), and it signals its own events at the START of the step, ahead of its filters: a wave that falls into the hit
// path (topk_lists.h, hundreds of cycles) no longer stops the other seven at the next barrier -- they run up to a
// step ahead.  No wait ever depends on an event of the same step, so the slowest wave can always proceed.
template <int KC, int EPL, int ROWN, int SYNC>
__global__ __launch_bounds__(512, 1) void l2s_topk_kernel(const unsigned char *__restrict__ Xpk,
                                                          const unsigned char *__restrict__ Ypk, int tiles_per_split,
                                                          int64_t tile_off, int lkeep, uint32_t *__restrict__ cand_idx,
                                                          float *__restrict__ cand_key, float *__restrict__ cand_tau,
                                                          int dbg)
{
    constexpr int R = L2S_R, NW = L2S_WAVES, NBUF = L2S_NBUF;
    using C = ListCfg<EPL, ROWN, L2S_R, L2S_NREC>;
    constexpr int TB = KC * 1024;                      // bytes per packed tile (targets and references alike)
    constexpr int PPW = (KC + NW - 1) / NW;            // LDS-DMA pieces every wave issues per tile (uniform: counted waits)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char *lists = smem_raw + NBUF * TB;
    uint32_t *sync_full = reinterpret_cast<uint32_t *>(lists + (size_t)NW * C::BYTES);          // [NBUF], then free [NBUF]
    uint32_t *sync_free = sync_full + NBUF;

    const int lane = lane_id();
    const int hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int split = blockIdx.y;
    const int S = gridDim.y;
    const int64_t ltile0 = ((int64_t)blockIdx.x * NW + wave) * R;
    const int64_t ttile0 = tile_off + ltile0;

    if (SYNC == 1) {
        if (threadIdx.x < 2 * NBUF) sync_full[threadIdx.x] = 0;
        __syncthreads();
    }
    // resident target fragments
    f16x8 xb[R][KC];
#pragma unroll
    for (int rb = 0; rb < R; ++rb) {
        const f16x8 *p = reinterpret_cast<const f16x8 *>(Xpk + (ttile0 + rb) * TB);
#pragma unroll
        for (int s = 0; s < KC; ++s) xb[rb][s] = p[s * 64 + lane];
    }
    unsigned char *wl = lists + (size_t)wave * C::BYTES;             // this wave's lists (topk_lists.h)
    float tauv[R];
    const float tau0 = (dbg & 1) ? -__builtin_inff() : __builtin_inff();
#pragma unroll
    for (int rb = 0; rb < R; ++rb) tauv[rb] = tau0;
    uint32_t scnt = 0;
    lists_init<C>(wl, lkeep, tau0, (uint32_t)split * (uint32_t)tiles_per_split * 32u);

    const int t_begin = split * tiles_per_split;
    const int t_end = t_begin + tiles_per_split;
    const uint32_t ring = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)smem_raw);

    // this wave's PPW pieces of tile t -> ring buffer `buf`; a piece index past KC repeats one of the wave's own
    // (same bytes to the same place), so every wave issues the same number of loads per tile
    auto dma = [&](int t, int buf) {
        int tc = t < t_end ? t : t_end - 1;
        if (dbg & 2) tc = t_begin + ((tc - t_begin) & 127);         // timing experiments: L2-resident window (garbage results)
        const unsigned char *src = Ypk + (int64_t)tc * TB + lane * 16;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            int p = wave + NW * i;
            if (p >= KC) p = (KC >= NW) ? p - NW : wave % KC;
            glds16(src + p * 1024, ring + (uint32_t)(buf * TB + p * 1024));
        }
    };
    auto read_tile = [&](f16x8(&a)[KC], int buf) {
        const f16x8 *p = reinterpret_cast<const f16x8 *>(smem_raw + buf * TB);
#pragma unroll
        for (int s = 0; s < KC; ++s) a[s] = p[s * 64 + lane];
    };
    // counter protocol (SYNC = 1): one lane adds; everybody polls the same word (an LDS broadcast)
    auto signal = [&](uint32_t *ctr) {
        if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto await = [&](uint32_t *ctr, uint32_t need) {
        while ((uint32_t)__builtin_amdgcn_readfirstlane(
                   (int)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need)
            __builtin_amdgcn_s_sleep(1);
    };

    // The target fragments must have ARRIVED before the loop: hipcc waits for a load at its first use, which would be
    // inside the loop -- a counted vmcnt in front of every MFMA, every iteration, draining the LDS-DMA it cannot see.
    // Passing each register through an empty asm makes "first use" happen here.
#pragma unroll
    for (int rb = 0; rb < R; ++rb)
#pragma unroll
        for (int s = 0; s < KC; ++s) asm volatile("" : "+v"(xb[rb][s]));
    // prologue: tiles t_begin, +1, +2 in flight; A0 <- tile t_begin; tile t_begin+1 published
    f16x8 a0[KC], a1[KC];
    dma(t_begin, 0);
    dma(t_begin + 1, 1);
    dma(t_begin + 2, 2);
    wait_vmcnt<2 * PPW>();
    if (SYNC == 2) {
        wait_vmcnt<0>();                                          // (tile t_begin+2's DMA is re-issued by step 0: drain it here)
        __syncthreads();
    } else if (SYNC == 0) {
        __syncthreads();
        read_tile(a0, 0);
        wait_vmcnt<PPW>();
        __syncthreads();                                          // (lgkmcnt(0) inside: every wave has copied tile t_begin)
    } else {
        signal(&sync_full[0]);
        await(&sync_full[0], NW);
        read_tile(a0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < KC; ++s) asm volatile("" : "+v"(a0[s]));   // the copy is complete before slot 0 is declared free
        signal(&sync_free[0]);
        wait_vmcnt<PPW>();
        signal(&sync_full[1]);
    }

    f32x16 accP;
#pragma unroll
    for (int r = 0; r < 16; ++r) accP[r] = __builtin_inff();       // inf < tau is false: nothing pending

    // one step: tile t is in `cur`; ring slot b0 held it.  SYNC 0: tile t+1 is published in slot b1, tile t+2 is landing.
    int b0 = 0;                                                    // ring slot of tile t
    uint32_t need0 = NW;                                           // NW * (fills of slot b0 so far): tile t is its need0/NW-th tile
    uint32_t need1 = NW, need2 = NW;                               // the same for slots b1 (tile t+1) and b2 (tile t+2)
    auto step = [&](f16x8(&cur)[KC], f16x8(&nxt)[KC], int t) {
        const int b1 = b0 == NBUF - 1 ? 0 : b0 + 1;
        const int b2 = b1 == NBUF - 1 ? 0 : b1 + 1;
        if (SYNC == 2) {
            // ONE A-operand register set (KC = 10: two sets + the staging code do not fit 256 VGPRs without spills,
            // and a scratch reload in the MFMA loop makes hipcc wait vmcnt(0), i.e. for the LDS-DMA in flight):
            // tile t is read from the ring at the top of its own step; tile t+2's DMA goes into the slot tile t-1 left.
            dma(t + 2, b2);
            read_tile(cur, b0);
            f32x16 accA = cchain<KC>(cur, xb[0]);
            filter_and_stage<C, EPL, R, L2S_NREC>(accP, 1, (uint32_t)(t - 1) * 32u + 4u * (uint32_t)hh, wl, scnt, lkeep, tauv);
            accP = cchain<KC>(cur, xb[1]);
            filter_and_stage<C, EPL, R, L2S_NREC>(accA, 0, (uint32_t)t * 32u + 4u * (uint32_t)hh, wl, scnt, lkeep, tauv);
            wait_vmcnt<PPW>();                                      // my pieces of tile t+1 have landed (t+2's may be in flight)
            __syncthreads();                                        // tile t+1 published; every wave is done with tile t
        } else if (SYNC == 0) {
            dma(t + 3, b0);
            read_tile(nxt, b1);
            f32x16 accA = cchain<KC>(cur, xb[0]);
            filter_and_stage<C, EPL, R, L2S_NREC>(accP, 1, (uint32_t)(t - 1) * 32u + 4u * (uint32_t)hh, wl, scnt, lkeep, tauv);
            accP = cchain<KC>(cur, xb[1]);
            filter_and_stage<C, EPL, R, L2S_NREC>(accA, 0, (uint32_t)t * 32u + 4u * (uint32_t)hh, wl, scnt, lkeep, tauv);
            wait_vmcnt<PPW>();                                      // my pieces of tile t+2 have landed (t+3's may be in flight)
            __syncthreads();                                        // tile t+2 published; every wave has copied tile t+1
        } else {
            wait_vmcnt<0>();                                        // my pieces of tile t+2 (issued a whole step ago)
            signal(&sync_full[b2]);
            await(&sync_free[b0], need0);                           // every wave has copied tile t out of slot b0
            dma(t + 3, b0);
            await(&sync_full[b1], need1);                           // every wave's pieces of tile t+1 have landed
            read_tile(nxt, b1);
            f32x16 accA = cchain<KC>(cur, xb[0]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s = 0; s < KC; ++s) asm volatile("" : "+v"(nxt[s]));
            signal(&sync_free[b1]);                                 // tile t+1 is in my registers
            filter_and_stage<C, EPL, R, L2S_NREC>(accP, 1, (uint32_t)(t - 1) * 32u + 4u * (uint32_t)hh, wl, scnt, lkeep, tauv);
            accP = cchain<KC>(cur, xb[1]);
            filter_and_stage<C, EPL, R, L2S_NREC>(accA, 0, (uint32_t)t * 32u + 4u * (uint32_t)hh, wl, scnt, lkeep, tauv);
            const uint32_t n0 = need0 + NW;                         // slot b0 now awaits its next tile
            need0 = need1; need1 = need2; need2 = n0;
        }
        b0 = b1;
    };
    if (SYNC == 2) {
        for (int t = t_begin; t < t_end; ++t) step(a0, a0, t);
    } else {
        for (int t = t_begin; t < t_end; t += 2) {
            step(a0, a1, t);
            if (t + 1 < t_end) step(a1, a0, t + 1);
        }
    }
    filter_and_stage<C, EPL, R, L2S_NREC>(accP, 1, (uint32_t)(t_end - 1) * 32u + 4u * (uint32_t)hh, wl, scnt, lkeep, tauv);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // drain the look-ahead DMA before the epilogue's stores

    lists_flush<C, EPL, R>(wl, scnt, ltile0 * 32, split, S, lkeep, tauv, cand_idx, cand_key, cand_tau);
}

// ---- packing ------------------------------------------------------------------------------
// (pack_ctiles_kernel, the packer of these operands, lives in pack.hip)

template <int KC, int SYNC>
static hipError_t slaunch_sync(const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S, int gx,
                               int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                               hipStream_t st)
{
    const int dbg = debug_ablate();
    constexpr int ROWN = l2s_row_entries(KC);
    constexpr size_t lds = (size_t)L2S_NBUF * KC * 1024 + (size_t)L2S_WAVES * ListCfg<1, ROWN, L2S_R, L2S_NREC>::BYTES + 64;
    static_assert(lds <= 163840, "LDS budget");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&l2s_topk_kernel<KC, 1, ROWN, SYNC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid(gx, S), block(64 * L2S_WAVES);
    hipLaunchKernelGGL((l2s_topk_kernel<KC, 1, ROWN, SYNC>), grid, block, lds, st, Xpk, Ypk, tiles_per_split, tile_off, lkeep,
                       cand_idx, cand_key, cand_tau, dbg);
    return hipGetLastError();
}

template <int KC>
static hipError_t slaunch_one(const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S, int gx,
                              int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                              hipStream_t st)
{
    // NABO_L2S_SYNC: 0 = barrier per tile, two A-operand register sets; 1 = LDS counters instead of barriers;
    // 2 (default) = barrier per tile, ONE A-operand set read at the top of the step (no spills at KC = 10)
    static const int sync = getenv("NABO_L2S_SYNC") ? atoi(getenv("NABO_L2S_SYNC")) : 2;
    return sync == 0   ? slaunch_sync<KC, 0>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, st)
           : sync == 1 ? slaunch_sync<KC, 1>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, st)
                       : slaunch_sync<KC, 2>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, st);
}

// steps of 16 slots for g components: 3 (g+1) slots, instantiated values only
int l2s_pick_kc(int g)
{
    const int need = (3 * (g + 1) + 15) / 16;
    const int inst[] = {2, 4, 6, 8, 10};          // 12 (g <= 63) does not fit 256 VGPRs with two A-operand sets
    for (int v : inst)
        if (need <= v) return v;
    return -1;          // g >= 53: l2h_topk.hip (g < 64) or the fp32 kernel
}

void l2s_topk_geometry(int kc, int *rows_per_wg, int *wg_per_cu, int *lkeep_max)
{
    *rows_per_wg = L2S_WAVES * L2S_R * 32;
    *wg_per_cu = 1;
    *lkeep_max = l2s_row_entries(kc) - 1;
}

hipError_t l2s_topk_launch(int kc, const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S, int gx,
                           int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau, hipStream_t st)
{
    if ((int64_t)tiles_per_split * 32 >= NABO_LIST_SPLIT_REFS) return hipErrorInvalidValue;   // topk_lists.h: 25 bits of offset per entry
    switch (kc) {
    case 2: return slaunch_one<2>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, st);
    case 4: return slaunch_one<4>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, st);
    case 6: return slaunch_one<6>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, st);
    case 8: return slaunch_one<8>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, st);
    case 10: return slaunch_one<10>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace nabo
