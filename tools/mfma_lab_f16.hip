// tools/mfma_lab_f16.hip -- ceiling of 12-MFMA f16 chains (v_mfma_f32_32x32x16_f16) with restart + filter,
// one wave per SIMD, 4 chains per "tile" (diagnostic only).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int KS = 4;

template <int MODE, int LEN>   // MODE bit0 restart from cin, bit1 filter; LEN = MFMAs per chain
__global__ __launch_bounds__(256, 1) void lab(const f16x8 *__restrict__ src, float *out, int tiles)
{
    const int lane = threadIdx.x & 63;
    f16x8 a[2 * KS], b[4][2 * KS];
#pragma unroll
    for (int s = 0; s < 2 * KS; ++s) {
        a[s] = src[s * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r][s] = src[(8 + r * 8 + s) * 64 + lane];
    }
    f32x16 cin;
#pragma unroll
    for (int r = 0; r < 16; ++r) cin[r] = (float)src[(100 + r) * 64 + lane][0];
    f32x16 accA = cin, accP = cin;
    float tau = -1e30f;
    int hits = 0;
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            f32x16 &acc = (rb & 1) ? accP : accA;
            f32x16 &old = (rb & 1) ? accA : accP;
#pragma unroll
            for (int s = 0; s < LEN; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s % (2 * KS)], b[rb][s % (2 * KS)], (s == 0 && (MODE & 1)) ? cin : acc, 0, 0, 0);
            if (MODE & 2) {
                float m = old[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fminf(m, old[r]);
                if (__builtin_amdgcn_ballot_w64(m < tau)) ++hits;
            }
        }
    }
    float s = hits;
    for (int r = 0; r < 16; ++r) s += accA[r] + accP[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int LEN>
static void run(int tiles)
{
    f16x8 *src; float *out;
    hipMalloc(&src, 1 << 22); hipMemset(src, 0, 1 << 22);
    int blocks = 256;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((lab<MODE, LEN>), dim3(blocks), dim3(256), 0, 0, src, out, tiles / 8);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((lab<MODE, LEN>), dim3(blocks), dim3(256), 0, 0, src, out, tiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double n_mfma = (double)blocks * 4 * (double)tiles * 4 * LEN;
    double flop = n_mfma * 32.0 * 32 * 16 * 2;
    printf("mode=%d len=%2d  %8.3f ms  %7.1f TFLOP/s  %5.1f%% of 2516   cycles/MFMA@2.4GHz=%.1f\n", MODE, LEN, best, flop / best / 1e9,
           flop / best / 1e9 / 2516.6 * 100, best * 1e-3 * 2.4e9 / (n_mfma / 1024));
    hipFree(src); hipFree(out);
}

int main()
{
    const int T = 20000;
    run<0, 12>(T); run<1, 12>(T); run<3, 12>(T); run<3, 24>(T / 2); run<3, 48>(T / 4); run<0, 48>(T / 4);
    return 0;
}
