// tools/mfma_peak.hip -- what can v_mfma_f32_32x32x2_f32 sustain on this chip, in the issue patterns
// l2_topk_kernel uses?  (diagnostic only, not part of the library)
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS, int LEN>
__global__ __launch_bounds__(256) void k_chain(float *out, int iters, float a0, float b0)
{
    f32x16 acc[CHAINS];
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    f32x16 sum = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = a;
#pragma unroll
        for (int s = 0; s < LEN; ++s)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) sum += acc[c];
        a += 1e-7f;
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += sum[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char *name, K kern, int wg_per_cu, int chains, int len, int iters)
{
    float *out;
    int blocks = 256 * wg_per_cu;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters / 10, 1.0f, 0.5f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flop = (double)blocks * 4 /*waves*/ * (double)iters * chains * len * 4096.0;
    printf("%-44s wg/cu=%d  %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)\n", name, wg_per_cu, best, flop / best / 1e9,
           flop / best / 1e9 / 157.3 * 100);
    hipFree(out);
}

int main()
{
    const int it = 4000;
    run("1 chain x25, dependent", k_chain<1, 25>, 1, 1, 25, it);
    run("1 chain x25, dependent", k_chain<1, 25>, 2, 1, 25, it);
    run("2 chains x25 interleaved", k_chain<2, 25>, 1, 2, 25, it);
    run("2 chains x25 interleaved", k_chain<2, 25>, 2, 2, 25, it);
    run("4 chains x25 interleaved", k_chain<4, 25>, 1, 4, 25, it / 2);
    run("1 chain x200, dependent", k_chain<1, 200>, 1, 1, 200, it / 8);
    run("1 chain x200, dependent", k_chain<1, 200>, 2, 1, 200, it / 8);
    return 0;
}
