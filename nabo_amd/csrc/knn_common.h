// knn_common.h -- shared definitions for the gfx950 k-NN kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// topk_lists.h: references one reference split (grid.y) may stream at most -- a list entry holds a 25-bit offset into its
// split, all ones = "no reference"; api.hip raises the split count of larger sets
#define NABO_LIST_SPLIT_REFS (((int64_t)1 << 25) - 1)

// A filter launch cut into PIECES (l2c_topk.hip; api.hip: cut_pieces): device arrays.
//   pieces   [n_pieces] int4 (column-workgroup, list slot of that column, first reference tile, end tile), one workgroup each,
//            longest first
//   ranges   [columns x S] int4 (first tile, end tile, tournament tiles, tiles per tournament group) per (column, slot);
//            an unused slot is (0, 0, 0, 0)
struct L2cPieces { const int *pieces; int n_pieces; const int *ranges = nullptr; int rows_per_col = 0; };

namespace nabo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;

// NABO_DEBUG_ABLATE (kernel-timing experiments whose RESULTS ARE GARBAGE: no-hit runs, L2-resident streams, counters)
// exists only in builds made with -DNABO_EXPERIMENTS (tools/ab, tools/r*_call*.sh build such variants next to the
// product through `python -m nabo_amd._build --out ...`); the shipped library never reads the variable.
static inline int debug_ablate()
{
#ifdef NABO_EXPERIMENTS
    static const int v = getenv("NABO_DEBUG_ABLATE") ? atoi(getenv("NABO_DEBUG_ABLATE")) : 0;
    return v;
#else
    return 0;
#endif
}
constexpr int TILE = 32;                 // cells per MFMA tile (32x32x2 f32)

// ---------------------------------------------------------------------------------------
// Packed operand tiles for v_mfma_f32_32x32x2_f32.
// One tile = 32 cells.  K-step s consumes components k = 2s, 2s+1; lane l supplies
// cell (l & 31), component 2s + (l >> 5)  (A and B operand maps are the same shape).
// Four K-steps are packed per lane so a fragment group is ONE 16-byte load per lane:
//   frag[q][l][e] = v[cell = l & 31][k = 2*(4q + e) + (l >> 5)],   q < Q = ceil(KSTEPS/4)
// Reference tiles carry, behind the fragments, the accumulator-initialisation block
//   norm[h][r] = ||y||^2 of cell (r&3) + 8*(r>>2) + 4*h      (the C/D row map of the MFMA)
// so that lane (l>>5 == h) loads its 16 C-in values as one 64-byte read.
// ---------------------------------------------------------------------------------------
__host__ __device__ constexpr int q_groups(int ksteps) { return (ksteps + 3) / 4; }
__host__ __device__ constexpr int qtile_floats(int ksteps) { return q_groups(ksteps) * 256; }
__host__ __device__ constexpr int rtile_floats(int ksteps) { return q_groups(ksteps) * 256 + 32; }

// C/D row of accumulator register r in lane-half h (cdna_hip_programming.md section 3)
__host__ __device__ constexpr int cd_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Canonical order on (key, idx): key ascending, then idx ascending.
template <typename K>
__device__ __forceinline__ bool kv_less(K ka, uint32_t va, K kb, uint32_t vb)
{
    return (ka < kb) || (ka == kb && va < vb);
}

// Lane exchange with lane ^ m for the sorting networks.  __shfl_xor is ds_bpermute_b32: an LDS round trip (~120 cycles)
// per exchange, and a bitonic sort is a chain of 21 DEPENDENT exchanges -- a 64-entry sort measured 5400 cycles
// (tools: -DNABO_LISTS_PROF), almost all of it waiting.  For m < 32 the data-parallel-primitive forms are used instead
// (a VALU operand modifier: no LDS, no address register): quad_perm for m = 1, 2; two bank-masked row shifts for
// m = 4; row_ror:8 for m = 8; ds_swizzle (bit-mask mode, crossbar only) for m = 16; m = 32 keeps the permute.
// `m` must be a compile-time constant after unrolling (the dpp controls are immediates).
__device__ __forceinline__ int xor_lane_i32(int v, int m)
{
    switch (m) {
    case 1: return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);                    // quad_perm:[1,0,3,2]
    case 2: return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);                    // quad_perm:[2,3,0,1]
    case 4: {
        int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);               // row_shl:4 into banks 0, 2
        return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);                // row_shr:4 into banks 1, 3
    }
    case 8: return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);                   // row_ror:8
    case 16: return __builtin_amdgcn_ds_swizzle(v, 0x401F);                              // and 0x1F, or 0, xor 0x10
    default: return __shfl_xor(v, m, 64);
    }
}
__device__ __forceinline__ float shfl_xor_t(float v, int m) { return __builtin_bit_cast(float, xor_lane_i32(__builtin_bit_cast(int, v), m)); }
__device__ __forceinline__ uint32_t shfl_xor_t(uint32_t v, int m) { return (uint32_t)xor_lane_i32((int)v, m); }
__device__ __forceinline__ double shfl_xor_t(double v, int m)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)xor_lane_i32((int)(uint32_t)b, m), hi = (uint32_t)xor_lane_i32((int)(uint32_t)(b >> 32), m);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// Wave-wide bitonic sort of N = 64*EPL (key, val) pairs, ascending in the canonical order.
// Element index e = r*64 + lane (r = register slot).  After the call element e holds rank e.
template <int EPL, typename K>
__device__ __forceinline__ void wave_bitonic_sort(K (&key)[EPL], uint32_t (&val)[EPL])
{
    const int lane = lane_id();
    constexpr int N = 64 * EPL;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int dr = j >> 6;
#pragma unroll
                for (int r = 0; r < EPL; ++r) {
                    const int rp = r ^ dr;
                    if (rp > r) {
                        const bool asc = (((r * 64) & k) == 0) || (k == N);
                        const bool lt = kv_less<K>(key[rp], val[rp], key[r], val[r]);  // partner < self
                        if (lt == asc) {
                            K tk = key[r]; key[r] = key[rp]; key[rp] = tk;
                            uint32_t tv = val[r]; val[r] = val[rp]; val[rp] = tv;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < EPL; ++r) {
                    const int e = r * 64 + lane;
                    const bool asc = ((e & k) == 0) || (k == N);
                    const bool lower = (lane & j) == 0;
                    const K pk = shfl_xor_t(key[r], j);
                    const uint32_t pv = shfl_xor_t(val[r], j);
                    const bool plt = kv_less<K>(pk, pv, key[r], val[r]);   // partner < self
                    const bool keep_min = (lower == asc);
                    const bool take = keep_min ? plt : !plt && !(pk == key[r] && pv == val[r]);
                    if (take) { key[r] = pk; val[r] = pv; }
                }
            }
        }
    }
}

// ---- fast path for (float key, uint32 value) pairs: the candidate lists of the filter kernels ----------------------
// The generic network above spends ~27 instructions per stage on (key, value) comparisons and on working out, per
// lane, which side of the exchange it is on.  Here a pair is ONE 64-bit integer -- the key mapped to an unsigned int
// that orders like the float, in the high word, the value in the low word -- so "partner < self" is one
// v_cmp_lt_u64, and which lanes keep the minimum is a compile-time 64-bit constant per stage: XORed onto the compare
// mask on the scalar unit and turned back into a select mask (inverse ballot): 4 DPP moves + compare + 2 s_xor +
// 2 v_cndmask per stage.  A 64-entry compaction went from ~580 to ~200 instructions.  Same order as kv_less
// (key ascending, then value ascending); equal pairs are interchangeable.
__host__ __device__ constexpr uint64_t sort_keepmin_mask(int r, int k, int j, int n)
{
    uint64_t m = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int e = r * 64 + lane;
        const bool asc = ((e & k) == 0) || (k == n);
        const bool lower = (lane & j) == 0;
        if (lower == asc) m |= 1ull << lane;
    }
    return m;
}

__device__ __forceinline__ uint64_t xor_lane_u64(uint64_t v, int m)
{
    const uint32_t lo = (uint32_t)xor_lane_i32((int)(uint32_t)v, m), hi = (uint32_t)xor_lane_i32((int)(uint32_t)(v >> 32), m);
    return ((uint64_t)hi << 32) | lo;
}

template <int EPL, int K, int J, int R_>
__device__ __forceinline__ void sort64_stage_reg(uint64_t (&c)[EPL])
{
    constexpr int N = 64 * EPL;
    if constexpr (J >= 64) {
        constexpr int rp = R_ ^ (J >> 6);
        if constexpr (rp > R_) {
            constexpr bool asc = (((R_ * 64) & K) == 0) || (K == N);
            const uint64_t a = c[R_], b = c[rp];
            const bool sw = (b < a) == asc;
            c[R_] = sw ? b : a;
            c[rp] = sw ? a : b;
        }
    } else {
        constexpr uint64_t km = sort_keepmin_mask(R_, K, J, N);
        const uint64_t p = xor_lane_u64(c[R_], J);
        const uint64_t take = __builtin_amdgcn_ballot_w64(p < c[R_]) ^ ~km;
        c[R_] = __builtin_amdgcn_inverse_ballot_w64(take) ? p : c[R_];
    }
    if constexpr (R_ + 1 < EPL) sort64_stage_reg<EPL, K, J, R_ + 1>(c);
}

template <int EPL, int K, int J>
__device__ __forceinline__ void sort64_stages(uint64_t (&c)[EPL])
{
    sort64_stage_reg<EPL, K, J, 0>(c);
    if constexpr (J > 1) sort64_stages<EPL, K, J / 2>(c);
}

template <int EPL, int K>
__device__ __forceinline__ void sort64_phases(uint64_t (&c)[EPL])
{
    sort64_stages<EPL, K, K / 2>(c);
    if constexpr (K < 64 * EPL) sort64_phases<EPL, K * 2>(c);
}

// Wave-wide ascending sort of 64 * EPL (float key, uint32 value) pairs; element e = r * 64 + lane ends with rank e.
template <int EPL>
__device__ __forceinline__ void wave_sort_f32(float (&key)[EPL], uint32_t (&val)[EPL])
{
    uint64_t c[EPL];
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const uint32_t b = __float_as_uint(key[r]);
        const uint32_t u = b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);       // orders like the float
        c[r] = ((uint64_t)u << 32) | val[r];
    }
    sort64_phases<EPL, 2>(c);
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const uint32_t u = (uint32_t)(c[r] >> 32);
        key[r] = __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
        val[r] = (uint32_t)c[r];
    }
}

}  // namespace nabo
