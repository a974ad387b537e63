// l2h_topk.hip -- Euclidean score + top-L filter with the contraction on the f16 matrix pipe
// ("f16x3" split), gfx950.  EXPERIMENTAL: selected with NABO_L2_MODE=f16x3, same outputs.
//
// Every centred, scaled component v is split v = hi + lo (two f16 values, 22 significant bits
// together); the -2 x.y term is accumulated in fp32 from three f16 products
//        hi_y*hi_x + hi_y*lo_x + lo_y*hi_x          (lo*lo <= 2^-22 |x||y| is dropped)
// on v_mfma_f32_32x32x16_f16 (32 cycles per instruction, K = 16): 3*ceil(g/16) MFMAs per 32x32 tile
// instead of ceil(g/2) fp32 MFMAs of 64 cycles -- 384 vs 1600 matrix-pipe cycles at g = 50.  The
// score only FILTERS candidates: refine.hip recomputes them in float64 and certifies the row with an
// error bound that accounts for the split (api.hip), so results are the same bits as the fp32 path.
//
// Structure: as l2_topk.hip (wave-private rows, lists in LDS, filter of a chain in the shadow of the
// next chain) with R = 4 row-blocks per wave, one wave per SIMD (the chains are 4x shorter, so per-tile
// operand traffic has to be amortised over more rows: 8 KB per tile per wave feeds 48 MFMAs).  The next
// tile is fetched a whole tile ahead into a second register set and moved into place behind each
// register's last use.
#include <cstdlib>

#include <hip/hip_fp16.h>

#include "knn_common.h"
#include "topk_lists.h"

namespace nabo {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Packed f16 tile (32 cells), KS16 = ceil(g/16) K-steps:
//   part p (0 = hi, 1 = lo), step s: frag[p][s][lane l][j] = v_p[cell l & 31][16 s + 8 (l >> 5) + j], j < 8
// The ||y||^2 term rides in the contraction: component slot g (the first padding slot; KS16 is chosen so that
// one exists) holds  ||y~||^2 * 2^-15 (hi + lo, SCALED units; +inf for masked / padding cells) in reference
// tiles and the constant 2^15 in target tiles, so the MFMA chain starts from C = 0 (an inline constant: no
// norm registers, no C-in copies) and still ends with  s = ||y||^2 - 2 x.y.
__host__ __device__ constexpr int htile_bytes(int ks16, bool is_ref) { (void)is_ref; return 2 * ks16 * 1024; }

template <int KS16>
struct HTile {
    f16x8 hi[KS16], lo[KS16];
};

template <int KS16>
__device__ __forceinline__ void load_htile(HTile<KS16> &y, const unsigned char *__restrict__ base, int lane)
{
    const f16x8 *p = reinterpret_cast<const f16x8 *>(base);
#pragma unroll
    for (int s = 0; s < KS16; ++s) {
        y.hi[s] = p[s * 64 + lane];
        y.lo[s] = p[(KS16 + s) * 64 + lane];
    }
}

// 3*KS16 MFMAs: lo_y*hi_x, hi_y*lo_x, hi_y*hi_x (small terms first).
// ROLL: behind the last use of each reference register move the prefetched next tile into place
// and refill the prefetch register from `next2` (the tile after): a whole tile of prefetch distance.
// (A single register set refilled through a wave-private LDS staging slot by LDS-DMA was measured
// too: no register pressure, but the DMA has only 3 chains to land and the lists shrink to 31
// entries per row -- 477 ms against 391 ms at 1M x 1M x 50.)
template <int KS16, bool ROLL>
__device__ __forceinline__ f32x16 hchain(HTile<KS16> &y, HTile<KS16> &yn, const f16x8 (&xhi)[KS16],
                                         const f16x8 (&xlo)[KS16], const unsigned char *__restrict__ next2, int lane)
{
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < KS16; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(y.lo[s], xhi[s], acc, 0, 0, 0);
        if (ROLL) {
            __builtin_amdgcn_sched_barrier(0);
            y.lo[s] = yn.lo[s];
            yn.lo[s] = reinterpret_cast<const f16x8 *>(next2)[(KS16 + s) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int s = 0; s < KS16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(y.hi[s], xlo[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KS16; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(y.hi[s], xhi[s], acc, 0, 0, 0);
        if (ROLL) {
            __builtin_amdgcn_sched_barrier(0);
            y.hi[s] = yn.hi[s];
            yn.hi[s] = reinterpret_cast<const f16x8 *>(next2)[s * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return acc;
}

// Grid: x = target super-blocks (4 waves x R tiles of 32 rows), y = reference splits.
template <int KS16, int R, int EPL, int ROWN>
__global__ __launch_bounds__(256, 1) void l2h_topk_kernel(const unsigned char *__restrict__ Xpk,
                                                          const unsigned char *__restrict__ Ypk,
                                                          int tiles_per_split, int64_t tile_off, int lkeep,
                                                          uint32_t *__restrict__ cand_idx,
                                                          float *__restrict__ cand_key,
                                                          float *__restrict__ cand_tau, int dbg)
{
    constexpr int ROW = ListCfg<EPL, ROWN>::ROW;
    constexpr int XTB = htile_bytes(KS16, false);
    constexpr int YTB = htile_bytes(KS16, true);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint2 *smem = reinterpret_cast<uint2 *>(smem_raw);

    const int lane = lane_id();
    const int hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int split = blockIdx.y;
    const int S = gridDim.y;
    const int64_t ltile0 = ((int64_t)blockIdx.x * 4 + wave) * R;
    const int64_t ttile0 = tile_off + ltile0;

    f16x8 xhi[R][KS16], xlo[R][KS16];
#pragma unroll
    for (int rb = 0; rb < R; ++rb) {
        const f16x8 *p = reinterpret_cast<const f16x8 *>(Xpk + (ttile0 + rb) * XTB);
#pragma unroll
        for (int s = 0; s < KS16; ++s) {
            xhi[rb][s] = p[s * 64 + lane];
            xlo[rb][s] = p[(KS16 + s) * 64 + lane];
        }
    }
    RowState st[R];
#pragma unroll
    for (int rb = 0; rb < R; ++rb) {
        st[rb].tau = (dbg & 1) ? -__builtin_inff() : __builtin_inff();
        st[rb].pc = 0;
        st[rb].kc = 0;
    }
    uint2 *wbuf = smem + (size_t)wave * R * 32 * ROW;

    const int64_t t_begin = (int64_t)split * tiles_per_split;
    const int64_t t_end = t_begin + tiles_per_split;
    // dbg & 2 (timing experiments only, results are garbage): the stream wraps inside a 128-tile window that stays
    // in the XCD's L2 -- the kernel's time without any L2 miss
    auto tile_ptr = [&](int64_t t) {
        const int64_t tc = t < t_end ? t : t_end - 1;
        return Ypk + ((dbg & 2) ? t_begin + ((tc - t_begin) & 127) : tc) * (int64_t)YTB;
    };

    HTile<KS16> y, yn;
    load_htile<KS16>(y, tile_ptr(t_begin), lane);
    load_htile<KS16>(yn, tile_ptr(t_begin + 1), lane);

    f32x16 accP;
#pragma unroll
    for (int r = 0; r < 16; ++r) accP[r] = __builtin_inff();

    for (int64_t t = t_begin; t < t_end; ++t) {
        const unsigned char *next2 = tile_ptr(t + 2);
        f32x16 accA;
#pragma unroll
        for (int rb = 0; rb < R; ++rb) {
            const int prev = (rb + R - 1) % R;
            const int64_t tprev = rb == 0 ? t - 1 : t;
            if (rb & 1) {
                if (rb == R - 1) accP = hchain<KS16, true>(y, yn, xhi[rb], xlo[rb], next2, lane);
                else accP = hchain<KS16, false>(y, yn, xhi[rb], xlo[rb], next2, lane);
                filter_and_append<EPL, ROWN>(accA, st[prev], wbuf + prev * 32 * ROW, (uint32_t)(tprev * 32 + 4 * hh), lkeep);
            } else {
                if (rb == R - 1) accA = hchain<KS16, true>(y, yn, xhi[rb], xlo[rb], next2, lane);
                else accA = hchain<KS16, false>(y, yn, xhi[rb], xlo[rb], next2, lane);
                filter_and_append<EPL, ROWN>(accP, st[prev], wbuf + prev * 32 * ROW, (uint32_t)(tprev * 32 + 4 * hh), lkeep);
            }
        }
        if (R & 1) accP = accA;        // odd R: the pending chain is the one just computed
    }
    filter_and_append<EPL, ROWN>(accP, st[R - 1], wbuf + (R - 1) * 32 * ROW, (uint32_t)((t_end - 1) * 32 + 4 * hh), lkeep);

#pragma unroll
    for (int rb = 0; rb < R; ++rb)
        flush_block<EPL, ROWN>(st[rb], wbuf + rb * 32 * ROW, (ltile0 + rb) * 32, split, S, lkeep, cand_idx, cand_key,
                               cand_tau);
}

// ---- packing ------------------------------------------------------------------------------
// One wave per 32-cell tile.  v = (V - centre) * scale; hi = f16(v), lo = f16(v - hi).
// Targets carry the factor -2 (exact).  norm64 (targets): ||rep||^2 in UNSCALED units.
template <bool IS_REF>
__global__ __launch_bounds__(64) void pack_htiles_kernel(const double *__restrict__ V, int64_t ncell, int g,
                                                         const double *__restrict__ centre, double scale, int ks16,
                                                         int64_t ntiles_total, const uint8_t *__restrict__ mask,
                                                         unsigned char *__restrict__ out, double *__restrict__ norm64,
                                                         unsigned int *__restrict__ norm_max_bits)
{
    const int64_t tile = blockIdx.x;
    if (tile >= ntiles_total) return;
    const int lane = threadIdx.x;
    const int c = lane & 31, hh = lane >> 5;
    const int64_t cell = tile * 32 + c;
    const bool live = cell < ncell;
    unsigned char *o = out + tile * (int64_t)htile_bytes(ks16, IS_REF);
    double ss = 0.0;
    // cells whose scaled components leave the f16 range (targets carry a factor 2; references are scaled to
    // 2^12) or are not finite: zero fragments + NaN norm (targets: the row cannot be certified and goes to the
    // exact kernels) / +inf norm slot (references: out of the filter) -- as pack_tiles_kernel does for fp32
    bool bad = false;
    for (int k = hh; k < g && live; k += 2) {
        const float f = (float)((V[cell * g + k] - centre[k]) * scale);
        bad = bad || !(fabsf(f) <= 30000.0f);
    }
    bad = bad || (__shfl_xor((int)bad, 32, 64) != 0);
    f16x8 vh[4], vl[4];                                      // ks16 <= 4
    for (int s = 0; s < ks16; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * hh + j;
            float f = 0.0f;
            if (live && k < g && !bad) f = (float)((V[cell * g + k] - centre[k]) * scale);
            const _Float16 h = (_Float16)f;
            const _Float16 l = (_Float16)(f - (float)h);
            const double rep = (double)(float)h + (double)(float)l;
            ss += rep * rep;
            vh[s][j] = IS_REF ? h : (_Float16)(-2.0f * (float)h);
            vl[s][j] = IS_REF ? l : (_Float16)(-2.0f * (float)l);
        }
    }
    ss += __shfl_xor(ss, 32, 64);
    // the norm slot: component index g (s = g/16, half = (g%16)/8, j = g%8)
    float slot_hi, slot_lo = 0.0f;
    if (IS_REF) {
        float nf = __builtin_inff();
        if (live && !bad && !(mask && mask[cell])) {
            nf = (float)ss * 3.0517578125e-05f;              // ||y~||^2 (scaled units) * 2^-15, <= 2^15
            if (hh == 0) atomicMax(norm_max_bits, __float_as_uint((float)ss));        // SCALED units (host unscales)
        }
        const _Float16 h = (_Float16)nf;
        slot_hi = (float)h;
        if (nf < __builtin_inff()) slot_lo = (float)(_Float16)(nf - (float)h);
    } else {
        slot_hi = live ? 32768.0f : 0.0f;                     // 2^15, NOT scaled by -2: the product is +||y||^2
        if (hh == 0 && live) norm64[cell] = bad ? __builtin_nan("") : ss / (scale * scale);
    }
    for (int s = 0; s < ks16; ++s) {
        if (g / 16 == s && ((g % 16) >> 3) == hh) {
            vh[s][g & 7] = (_Float16)slot_hi;
            vl[s][g & 7] = (_Float16)slot_lo;
        }
        reinterpret_cast<f16x8 *>(o)[s * 64 + lane] = vh[s];
        reinterpret_cast<f16x8 *>(o)[(ks16 + s) * 64 + lane] = vl[s];
    }
}

// max |V - centre| over all components, in float64 (bits of a non-negative double order like unsigned 64-bit ints)
__global__ void maxabs_kernel(const double *__restrict__ V, int64_t n, int g, const double *__restrict__ centre,
                              unsigned long long *__restrict__ out_bits)
{
    double m = 0.0;
    const int64_t tot = n * g;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const double a = fabs(V[i] - centre[i % g]);
        if (a > m) m = a;                                   // NaN never wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(m, o, 64);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, (unsigned long long)__double_as_longlong(m));
}

hipError_t maxabs_launch(const double *V, int64_t n, int g, const double *centre, unsigned long long *out_bits, hipStream_t st)
{
    hipLaunchKernelGGL(maxabs_kernel, dim3(2048), dim3(256), 0, st, V, n, g, centre, out_bits);
    return hipGetLastError();
}

hipError_t pack_href_launch(const double *Y, int64_t n, int g, const double *centre, double scale, int ks16,
                            int64_t ntiles_total, const uint8_t *mask, unsigned char *out, unsigned int *norm_max_bits,
                            hipStream_t st)
{
    hipLaunchKernelGGL((pack_htiles_kernel<true>), dim3((unsigned)ntiles_total), dim3(64), 0, st, Y, n, g, centre, scale,
                       ks16, ntiles_total, mask, out, (double *)nullptr, norm_max_bits);
    return hipGetLastError();
}

hipError_t pack_hquery_launch(const double *X, int64_t m, int g, const double *centre, double scale, int ks16,
                              int64_t ntiles_total, unsigned char *out, double *xnorm, hipStream_t st)
{
    hipLaunchKernelGGL((pack_htiles_kernel<false>), dim3((unsigned)ntiles_total), dim3(64), 0, st, X, m, g, centre, scale,
                       ks16, ntiles_total, (const uint8_t *)nullptr, out, xnorm, (unsigned int *)nullptr);
    return hipGetLastError();
}

template <int KS16, int R, int EPL, int ROWN>
static hipError_t hlaunch_one(const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S, int gx,
                              int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                              hipStream_t st)
{
    static const int dbg = getenv("NABO_DEBUG_ABLATE") ? atoi(getenv("NABO_DEBUG_ABLATE")) : 0;
    const size_t lds = (size_t)4 * R * 32 * ROWN * sizeof(uint2);
    static_assert((size_t)4 * R * 32 * ROWN * sizeof(uint2) <= 163840, "LDS budget");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&l2h_topk_kernel<KS16, R, EPL, ROWN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid(gx, S), block(256);
    hipLaunchKernelGGL((l2h_topk_kernel<KS16, R, EPL, ROWN>), grid, block, lds, st, Xpk, Ypk, tiles_per_split, tile_off,
                       lkeep, cand_idx, cand_key, cand_tau, dbg);
    return hipGetLastError();
}

// The f16x3 variant is instantiated for lists of <= 32 kept entries (k + drop_first <= 24):
// 4 row-blocks per wave, rows of 40 list entries = 160 KB of LDS, one workgroup per CU.
void l2h_topk_geometry(int ks16, int *rows_per_wg, int *wg_per_cu, int *lkeep_max)
{
    (void)ks16;
    *rows_per_wg = 4 * 4 * 32;
    *wg_per_cu = 1;
    *lkeep_max = 32;
}

int l2h_pick_ks16(int g)
{
    const int need = (g + 1 + 15) / 16;       // one padding slot carries the norm term
    const int inst[] = {1, 2, 4};
    for (int v : inst)
        if (need <= v) return v;
    return -1;          // g >= 64: use the fp32 kernel
}

hipError_t l2h_topk_launch(int ks16, const unsigned char *Xpk, const unsigned char *Ypk, int tiles_per_split, int S,
                           int gx, int64_t tile_off, int lkeep, uint32_t *cand_idx, float *cand_key, float *cand_tau,
                           hipStream_t st)
{
    switch (ks16) {
    case 1: return hlaunch_one<1, 4, 1, 40>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, st);
    case 2: return hlaunch_one<2, 4, 1, 40>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, st);
    case 4: return hlaunch_one<4, 4, 1, 40>(Xpk, Ypk, tiles_per_split, S, gx, tile_off, lkeep, cand_idx, cand_key, cand_tau, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace nabo
