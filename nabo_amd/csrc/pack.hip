// pack.hip -- operand preparation for the Euclidean MFMA kernel (gfx950).
//
// Host-side analogue in the reference: the per-cell HDF5 gathers into dense float64 chunks
// (nabo/_mapping.py:105,108,113,116).  Here the dense float64 arrays are already resident in
// HBM; these kernels centre them (distances are translation invariant; centring keeps the
// fp32 rounding error of the filter small), round to fp32 and lay them out as MFMA fragment
// tiles (knn_common.h).  O((m+n)*g) work, HBM-bound, negligible next to the 2*m*n*g kernel.
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

}  // namespace nabo
