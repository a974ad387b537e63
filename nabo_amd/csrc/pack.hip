// pack.hip -- operand preparation for the Euclidean MFMA kernel (gfx950).
//
// Host-side analogue in the reference: the per-cell HDF5 gathers into dense float64 chunks
// (nabo/_mapping.py:105,108,113,116).  Here the dense float64 arrays are already resident in
// HBM; these kernels centre them (distances are translation invariant; centring keeps the
// fp32 rounding error of the filter small), round to fp32 and lay them out as MFMA fragment
// tiles (knn_common.h).  O((m+n)*g) work, HBM-bound, negligible next to the 2*m*n*g kernel.
#include <hip/hip_fp16.h>

#include "knn_common.h"

namespace nabo {

// Deterministic centre: mean of up to `nsample` evenly strided reference rows.
// One block per column chunk; fixed summation order -> bit-reproducible.
__global__ void centre_kernel(const double *__restrict__ Y, int64_t n, int g, int64_t stride,
                              int64_t nsample, double *__restrict__ centre)
{
    __shared__ double part[256];
    const int k = blockIdx.x;               // one column per block
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < nsample; i += blockDim.x) {
        int64_t row = i * stride;
        if (row < n) s += Y[row * g + k];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) centre[k] = part[0] / (double)nsample;
}

// One wave per 32-cell tile.  IS_REF: also emits the norm block (+inf for masked / padded
// cells so they can never pass the filter) and the running max of ||y||^2 (bits of a
// non-negative float order like unsigned ints; SCALED units, the host unscales in double).  Every value is
// multiplied by `scale`, a power of two chosen from the references so that max |y~| lies in (1/2, 1]: exact, and
// it keeps squares and scores inside the fp32 range whatever the unit of the input.  Targets additionally carry
// the factor -2 (exact).
template <bool IS_REF>
__global__ __launch_bounds__(64) void pack_tiles_kernel(const double *__restrict__ V, int64_t ncell, int g,
                                                        const double *__restrict__ centre, double scale, int ksteps,
                                                        int64_t ntiles_total,
                                                        const uint8_t *__restrict__ mask,
                                                        float *__restrict__ out, double *__restrict__ norm64,
                                                        unsigned int *__restrict__ norm_max_bits)
{
    const int64_t tile = blockIdx.x;
    if (tile >= ntiles_total) return;
    const int lane = threadIdx.x;
    const int c = lane & 31, hh = lane >> 5;
    const int Q = q_groups(ksteps);
    const int64_t cell = tile * 32 + c;
    const bool live = cell < ncell;
    const int tile_floats = Q * 256 + (IS_REF ? 32 : 0);
    float *o = out + tile * tile_floats;
    // The score kernel is compiled with -fno-honor-nans: its scores must be finite or +inf BY CONSTRUCTION.
    // References are scaled to max |y~| <= 1; a target component beyond PACK_LIMIT (a target that dwarfs the
    // references, or non-finite input) could push a partial sum of the fma chain to +-inf and the next term to
    // inf - inf.  Such a cell gets ZERO fragments (its scores are then just ||y~||^2) and, for targets, a NaN
    // norm: refine.hip cannot certify the row (every comparison with its bound is false) and the exact float64
    // kernels answer it.  A reference cell with a non-finite component is taken out of the filter like a masked
    // one (+inf norm).  2 * 128 components * PACK_LIMIT * 1 stays far inside the fp32 range.
    constexpr float PACK_LIMIT = 1.0e30f;
    bool bad = false;
    for (int k = hh; k < g && live; k += 2) {
        const float f = (float)((V[cell * g + k] - centre[k]) * scale);
        bad = bad || !(fabsf(f) <= PACK_LIMIT);
    }
    bad = bad || (__shfl_xor((int)bad, 32, 64) != 0);
    double ss = 0.0;
    for (int q = 0; q < Q; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 2 * (4 * q + e) + hh;
            float f = 0.0f;
            if (live && k < g && !bad) {
                f = (float)((V[cell * g + k] - centre[k]) * scale);      // scale: a power of two (exact)
                ss += (double)f * (double)f;
            }
            v[e] = IS_REF ? f : -2.0f * f;
        }
        reinterpret_cast<f32x4 *>(o)[q * 64 + lane] = v;
    }
    ss += __shfl_xor(ss, 32, 64);           // both k-halves of the cell
    if (IS_REF) {
        float nf = __builtin_inff();
        if (live && !bad && !(mask && mask[cell])) {
            nf = (float)ss;
            if (hh == 0) atomicMax(norm_max_bits, __float_as_uint(nf));
        }
        if (hh == 0) o[Q * 256 + ((c >> 2) & 1) * 16 + (c & 3) + 4 * (c >> 3)] = nf;
    } else {
        if (hh == 0 && live) norm64[cell] = bad ? __builtin_nan("") : ss / (scale * scale);        // UNSCALED units
    }
}

hipError_t centre_launch(const double *Y, int64_t n, int g, double *centre, hipStream_t st)
{
    int64_t nsample = n < 16384 ? n : 16384;
    int64_t stride = n / nsample;
    if (stride < 1) stride = 1;
    hipLaunchKernelGGL(centre_kernel, dim3(g), dim3(256), 0, st, Y, n, g, stride, nsample, centre);
    return hipGetLastError();
}

hipError_t pack_ref_launch(const double *Y, int64_t n, int g, const double *centre, double scale, int ksteps,
                           int64_t ntiles_total, const uint8_t *mask, float *out,
                           unsigned int *norm_max_bits, hipStream_t st)
{
    hipLaunchKernelGGL((pack_tiles_kernel<true>), dim3((unsigned)ntiles_total), dim3(64), 0, st, Y, n, g, centre, scale,
                       ksteps, ntiles_total, mask, out, (double *)nullptr, norm_max_bits);
    return hipGetLastError();
}

hipError_t pack_query_launch(const double *X, int64_t m, int g, const double *centre, double scale, int ksteps,
                             int64_t ntiles_total, float *out, double *xnorm, hipStream_t st)
{
    hipLaunchKernelGGL((pack_tiles_kernel<false>), dim3((unsigned)ntiles_total), dim3(64), 0, st, X, m, g, centre, scale,
                       ksteps, ntiles_total, (const uint8_t *)nullptr, out, xnorm, (unsigned int *)nullptr);
    return hipGetLastError();
}

// ---- f16 operands of the matrix-pipe filters (l2c_topk.hip: one product; l2q_topk.hip: the f16x3 split) -----------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

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

// One wave per 32-cell tile.  v = (V - centre) * scale; hi = f16(v), lo = f16(v - hi); rep = hi + lo.
// norm64 (targets): ||rep||^2 in UNSCALED units.
// NSEG = 3 -- the f16x3 split.  Slot p of a cell's concatenated vector: segment p / (g+1), entry e = p % (g+1); e < g:
// component e, e == g: the norm slot.
//   references  [hi | lo | hi],  norm slots (nh, nl, 0) with nh + nl = ||rep_y||^2 2^-15 (+inf when masked / padding)
//   targets     [-2hi | -2hi | -2lo],  norm slots (2^15, 2^15, 0)
// NSEG = 1 -- the ONE-PRODUCT ("coarse") filter of l2q_topk.hip: g + 3 slots,
//   references  [hi (g) | nh | nl | ey],   targets  [-2hi (g) | 2^15 | 2^15 | -tx]
// Its score  ||rep_y||^2 - 2 hi_x.hi_y - tx ey  is a LOWER BOUND of the f16x3 score's exact value
// ||rep_y||^2 - 2 rep_x.rep_y:  |rep_x.rep_y - hi_x.hi_y| = |sum hi_x lo_y + lo_x hi_y + lo_x lo_y| <= (2u + u^2)(1 + u)
// sum |hi_x||hi_y| + (underflow terms, covered by the certificate's coefficient) with u = 2^-11, i.e. the score is off
// by at most 2^-9 1.002 ||hi_x|| ||hi_y||; tx = f16(2^-9 1.01 ||rep_x||) and ey = f16(1.002 ||rep_y||), both rounded
// to nearest AFTER a factor (1 + 2^-9) so that the f16 value is never below the real one, and never below 2^-13 (f16
// subnormals round with an absolute error).  The error term is a product of two slots: it costs no instruction, and it
// is per PAIR -- a per-row constant would have to assume max ||y||.
// Register layouts (16 bytes per lane and register, a tile is kc KiB either way):
//   L16 = false (v_mfma_f32_32x32x16_f16; l2h / l2s kernels): register s < kc, lane l: cell l & 31,
//                slots 16 s + 8 (l >> 5) + j;
//   L16 = true  (v_mfma_f32_16x16x32_f16; l2q kernel): register h (kc/2) + s, h < 2, s < kc/2, lane l: cell 16 h + (l & 15),
//                slots 32 s + 8 (l >> 4) + j.
template <bool IS_REF, bool L16, int NSEG>
__global__ __launch_bounds__(64) void pack_ctiles_kernel(const double *__restrict__ V, int64_t ncell, int g,
                                                         const double *__restrict__ centre, double scale, int kc,
                                                         int64_t ntiles_total, const uint8_t *__restrict__ mask,
                                                         unsigned char *__restrict__ out, double *__restrict__ norm64,
                                                         unsigned int *__restrict__ norm_max_bits,
                                                         const uint32_t *__restrict__ perm)
{
    const int64_t tile = blockIdx.x;
    if (tile >= ntiles_total) return;
    const int lane = threadIdx.x;
    const int g1 = g + 1;
    unsigned char *o = out + tile * (int64_t)kc * 1024;
    const int nh_cells = L16 ? 2 : 1;                        // cells this lane packs
    const int ks = L16 ? kc / 2 : kc;                        // registers per cell
    const int grp = L16 ? lane >> 4 : lane >> 5;             // which 8 slots of a step this lane supplies
    for (int hc = 0; hc < nh_cells; ++hc) {
        const int c = L16 ? 16 * hc + (lane & 15) : lane & 31;
        const int64_t cell = tile * 32 + c;
        const bool live = cell < ncell;
        // locality order (order.hip): packed position `cell` holds caller row perm[cell]; norm64 is indexed by POSITION
        const int64_t src = (live && perm) ? (int64_t)perm[cell] : cell;
        // whole-row pass: range check and ||rep||^2 (every lane of a cell computes the same)
        bool bad = false;
        double ss = 0.0;
        for (int e = 0; e < g && live; ++e) {
            const float f = (float)((V[src * g + e] - centre[e]) * scale);
            bad = bad || !(fabsf(f) <= 30000.0f);                 // f16 range (targets carry a factor 2); NaN / inf input
            const _Float16 h = (_Float16)f;
            const _Float16 l = (_Float16)(f - (float)h);
            const double rep = (double)(float)h + (double)(float)l;
            ss += rep * rep;
        }
        float nh = 0.0f, nl = 0.0f, er = 0.0f;
        if (IS_REF) {
            float nf = __builtin_inff();
            if (live && !bad && !(mask && mask[src])) {
                nf = (float)ss * 3.0517578125e-05f;                // ||y~||^2 (scaled units) * 2^-15
                if (grp == 0) atomicMax(norm_max_bits, __float_as_uint((float)ss));
            }
            const _Float16 h = (_Float16)nf;
            nh = (float)h;
            if (nf < __builtin_inff()) nl = (float)(_Float16)(nf - (float)h);
            if (NSEG == 1 && nf < __builtin_inff()) er = fmaxf((float)(sqrt(ss) * (1.002 * 1.001953125)), 1.220703125e-4f);
        } else {
            // segments 0 and 1; NOT scaled by -2: the product is +||y||^2.  (One-product operands: padding rows carry the
            // slots too -- 0 x inf would be the only NaN that kernel could see, and it is compiled with -fno-honor-nans.)
            nh = nl = (live || NSEG == 1) ? 32768.0f : 0.0f;
            if (grp == 0 && live) norm64[cell] = bad ? __builtin_nan("") : ss / (scale * scale);
            if (NSEG == 1 && live && !bad)
                er = -fmaxf((float)(sqrt(ss) * (0.001953125 * 1.01 * 1.001953125)), 1.220703125e-4f);
        }
        for (int s = 0; s < ks; ++s) {
            f16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = (L16 ? 32 : 16) * s + 8 * grp + j;
                float val = 0.0f;
                if (NSEG == 3) {
                    const int seg = p / g1, e = p - seg * g1;
                    if (seg < 3) {
                        if (e == g) {
                            val = seg == 0 ? nh : seg == 1 ? nl : 0.0f;
                        } else if (live && !bad) {
                            const float f = (float)((V[src * g + e] - centre[e]) * scale);
                            const _Float16 h = (_Float16)f;
                            const float lo = (float)(_Float16)(f - (float)h);
                            const bool want_lo = IS_REF ? seg == 1 : seg == 2;
                            val = want_lo ? lo : (float)h;
                            if (!IS_REF) val *= -2.0f;
                        }
                    }
                } else {
                    if (p == g) val = nh;
                    else if (p == g + 1) val = nl;
                    else if (p == g + 2) val = er;
                    else if (p < g && live && !bad) {
                        val = (float)(_Float16)(float)((V[src * g + p] - centre[p]) * scale);
                        if (!IS_REF) val *= -2.0f;
                    }
                }
                v[j] = (_Float16)val;
            }
            reinterpret_cast<f16x8 *>(o)[(hc * ks + s) * 64 + lane] = v;
        }
    }
}

// nseg: 3 = f16x3 operands, 1 = the one-product operands
hipError_t pack_cref_launch(const double *Y, int64_t n, int g, const double *centre, double scale, int kc,
                            int64_t ntiles_total, const uint8_t *mask, unsigned char *out, unsigned int *norm_max_bits,
                            bool layout16, hipStream_t st, const uint32_t *perm, int nseg)
{
    if (nseg == 1 && layout16)
        hipLaunchKernelGGL((pack_ctiles_kernel<true, true, 1>), dim3((unsigned)ntiles_total), dim3(64), 0, st, Y, n, g, centre,
                           scale, kc, ntiles_total, mask, out, (double *)nullptr, norm_max_bits, perm);
    else if (nseg == 1)
        hipLaunchKernelGGL((pack_ctiles_kernel<true, false, 1>), dim3((unsigned)ntiles_total), dim3(64), 0, st, Y, n, g, centre,
                           scale, kc, ntiles_total, mask, out, (double *)nullptr, norm_max_bits, perm);
    else if (layout16)
        hipLaunchKernelGGL((pack_ctiles_kernel<true, true, 3>), dim3((unsigned)ntiles_total), dim3(64), 0, st, Y, n, g, centre,
                           scale, kc, ntiles_total, mask, out, (double *)nullptr, norm_max_bits, perm);
    else
        hipLaunchKernelGGL((pack_ctiles_kernel<true, false, 3>), dim3((unsigned)ntiles_total), dim3(64), 0, st, Y, n, g, centre,
                           scale, kc, ntiles_total, mask, out, (double *)nullptr, norm_max_bits, perm);
    return hipGetLastError();
}

hipError_t pack_cquery_launch(const double *X, int64_t m, int g, const double *centre, double scale, int kc,
                              int64_t ntiles_total, unsigned char *out, double *xnorm, bool layout16, hipStream_t st,
                              const uint32_t *perm, int nseg)
{
    if (nseg == 1 && layout16)
        hipLaunchKernelGGL((pack_ctiles_kernel<false, true, 1>), dim3((unsigned)ntiles_total), dim3(64), 0, st, X, m, g, centre,
                           scale, kc, ntiles_total, (const uint8_t *)nullptr, out, xnorm, (unsigned int *)nullptr, perm);
    else if (nseg == 1)
        hipLaunchKernelGGL((pack_ctiles_kernel<false, false, 1>), dim3((unsigned)ntiles_total), dim3(64), 0, st, X, m, g, centre,
                           scale, kc, ntiles_total, (const uint8_t *)nullptr, out, xnorm, (unsigned int *)nullptr, perm);
    else if (layout16)
        hipLaunchKernelGGL((pack_ctiles_kernel<false, true, 3>), dim3((unsigned)ntiles_total), dim3(64), 0, st, X, m, g, centre,
                           scale, kc, ntiles_total, (const uint8_t *)nullptr, out, xnorm, (unsigned int *)nullptr, perm);
    else
        hipLaunchKernelGGL((pack_ctiles_kernel<false, false, 3>), dim3((unsigned)ntiles_total), dim3(64), 0, st, X, m, g, centre,
                           scale, kc, ntiles_total, (const uint8_t *)nullptr, out, xnorm, (unsigned int *)nullptr, perm);
    return hipGetLastError();
}

}  // namespace nabo
