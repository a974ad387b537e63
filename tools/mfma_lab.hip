// tools/mfma_lab.hip -- isolates what limits the l2_topk main loop (diagnostic only).
// Variants of: per "tile", two 25-MFMA chains (row-blocks 0/1) of v_mfma_f32_32x32x2_f32 with
// distinct A (25 regs "tile") and B (2x25 regs "targets") operands.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KS = 25;

// MODE bits: 1 = C-in from registers each chain (else continue accumulating: no restart)
//            2 = filter VALU (8 min3 + cmp -> flag) on the previous chain
//            4 = reload the A registers from memory every tile (rolling)
//            8 = interleave the two chains MFMA by MFMA
//           16 = (with 4) stream the reloads from a 228 MB array, one 7296-byte tile per iteration, as production does
//           64 = every second tile the wave takes a detour of 120 dependent VALU instructions (a stand-in for a hit
//                episode); 128 = (with 64) the detour runs at s_setprio 0, the loop at s_setprio 3
//           32 = (with 16) every wave touches one dword per 128-byte line of the tile PF tiles ahead (pulls it into L2)
template <int MODE, int PF = 2>
__global__ __launch_bounds__(256, 2) void lab(const float *__restrict__ src, float *out, int tiles)
{
    const int lane = threadIdx.x & 63;
    float a[KS], b0[KS], b1[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { a[s] = src[s * 64 + lane]; b0[s] = src[(s + 32) * 64 + lane]; b1[s] = src[(s + 64) * 64 + lane]; }
    f32x16 cin;
#pragma unroll
    for (int r = 0; r < 16; ++r) cin[r] = src[(100 + r) * 64 + lane];
    f32x16 accA = cin, accB = cin, accP = cin;
    float tau = -1e30f;
    int hits = 0;
    if (MODE & 128) __builtin_amdgcn_s_setprio(3);
    float pfv = 0.0f, junk = 0.0f;
    const float *p = src + 4096 + (blockIdx.x & 7) * 64;
    for (int t = 0; t < tiles; ++t) {
        if (MODE & 8) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b0[s], (s == 0 && (MODE & 1)) ? cin : accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b1[s], (s == 0 && (MODE & 1)) ? cin : accB, 0, 0, 0);
                if ((MODE & 4) && (s & 3) == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 v = *reinterpret_cast<const f32x4 *>(p + ((t & 63) * 1024 + (s >> 2) * 256 + lane * 4));
                    a[s - 3] = v[0]; a[s - 2] = v[1]; a[s - 1] = v[2]; a[s] = v[3];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE & 2) {
                float m = accA[0], m2 = accB[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) { m = fminf(m, accA[r]); m2 = fminf(m2, accB[r]); }
                if (__builtin_amdgcn_ballot_w64(fminf(m, m2) < tau)) ++hits;
            }
        } else {
            // chain rb0 -> accA, filter accP (previous rb1); chain rb1 -> accP, filter accA
#pragma unroll
            for (int s = 0; s < KS; ++s)
                accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b0[s], (s == 0 && (MODE & 1)) ? cin : accA, 0, 0, 0);
            if (MODE & 2) {
                float m = accP[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fminf(m, accP[r]);
                if (__builtin_amdgcn_ballot_w64(m < tau)) ++hits;
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                accP = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b1[s], (s == 0 && (MODE & 1)) ? cin : accP, 0, 0, 0);
                if ((MODE & 4) && (s & 3) == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 v = (MODE & 16) ? *reinterpret_cast<const f32x4 *>(src + ((size_t)(t % 31250) * 1824 + (s >> 2) * 256 + lane * 4))
                                          : *reinterpret_cast<const f32x4 *>(p + ((t & 63) * 1024 + (s >> 2) * 256 + lane * 4));
                    a[s - 3] = v[0]; a[s - 2] = v[1]; a[s - 1] = v[2]; a[s] = v[3];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if ((MODE & 64) && (t & 1)) {
                if (MODE & 128) __builtin_amdgcn_s_setprio(0);
                float z = junk;
#pragma unroll
                for (int i = 0; i < 120; ++i) z = __builtin_fmaf(z, 1.0000001f, 0.5f);
                junk = z;
                if (MODE & 128) __builtin_amdgcn_s_setprio(3);
            }
            if (MODE & 32) {
                junk += pfv;
                __builtin_amdgcn_sched_barrier(0);
                pfv = src[(size_t)((t + PF) % 31250) * 1824 + lane * 32];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE & 2) {
                float m = accA[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fminf(m, accA[r]);
                if (__builtin_amdgcn_ballot_w64(m < tau)) ++hits;
            }
        }
    }
    float s = hits + junk + pfv;
    for (int r = 0; r < 16; ++r) s += accA[r] + accB[r] + accP[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int PF = 2>
static void run(int wg_per_cu, int tiles, int rounds = 1)
{
    float *src, *out;
    const size_t sb = (MODE & 16) ? (size_t)31250 * 7296 + (1 << 22) : (size_t)(1 << 22);
    hipMalloc(&src, sb);
    hipMemset(src, 0, sb);
    int blocks = 256 * wg_per_cu * rounds;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((lab<MODE, PF>), dim3(blocks), dim3(256), 0, 0, src, out, tiles / 8);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((lab<MODE, PF>), dim3(blocks), dim3(256), 0, 0, src, out, tiles);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flop = (double)blocks * 4 * (double)tiles * 2 * KS * 4096.0;
    printf("mode=%2d pf=%d [%s%s%s%s%s] wg/cu=%d rounds=%d %8.3f ms  %6.1f TFLOP/s  %5.1f%%\n", MODE, (MODE & 32) ? PF : 0, (MODE & 1) ? "restart " : "",
           (MODE & 2) ? "filter " : "", (MODE & 4) ? "reload " : "", (MODE & 8) ? "interleave " : "", (MODE & 16) ? "stream228MB " : "", wg_per_cu, rounds, best,
           flop / best / 1e9, flop / best / 1e9 / 157.3 * 100);
    hipFree(src); hipFree(out);
}

int main(int argc, char **argv)
{
    const bool full = argc > 1;
    const int T = 4000;
    if (full) {
        run<0>(1, T); run<0>(2, T);
        run<1>(1, T); run<1>(2, T);
        run<3>(1, T); run<3>(2, T);
        run<7>(1, T); run<7>(2, T);
        run<8>(1, T); run<8>(2, T);
        run<9>(1, T); run<9>(2, T);
        run<11>(1, T); run<11>(2, T);
        run<15>(1, T); run<15>(2, T);
    }
    run<7>(2, 8000); run<71>(2, 8000); run<199>(2, 8000); run<7>(1, 8000); run<71>(1, 8000);
    if (!full) return 0;
    run<7>(2, 31250); run<23>(2, 31250); run<23>(2, 31250, 3); run<23>(1, 31250);
    run<55, 1>(2, 31250); run<55, 2>(2, 31250); run<55, 4>(2, 31250); run<55, 8>(2, 31250);
    run<55, 2>(2, 31250, 3); run<55, 4>(2, 31250, 3); run<55, 8>(2, 31250, 3);
    return 0;
}
