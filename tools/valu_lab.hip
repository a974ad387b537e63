// valu_lab.hip -- issue-rate microbenchmark for the modified-Canberra counting pass (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_lab.hip -o /tmp/valu_lab && /tmp/valu_lab
// Variants of "count the dimensions with |x - y| >= thr" over 56 register-resident y values per lane and
// wave-uniform (x, thr) pairs; prints ns per (pair, dimension) per SIMD-lane and the implied clocks per
// wave-instruction assuming 3 VALU instructions per dimension.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int GP = 56;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR, int NCH>
__global__ __launch_bounds__(256) void k(const float2 *__restrict__ xq, const float *__restrict__ y, int iters, int T,
                                         int *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    float yv[NCH][GP];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int kk = 0; kk < GP; ++kk) yv[c][kk] = y[(c * GP + kk) * 64 + lane];
    int total = 0;
    float ftotal = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (int t = 0; t < T; ++t) {
            const f32x16 *xb = reinterpret_cast<const f32x16 *>(xq + (size_t)t * GP);
            int no[NCH];
            float fo[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) { no[c] = 0; fo[c] = 0.f; }
#pragma unroll
            for (int b = 0; b < GP / 8; ++b) {
                const f32x16 cur = xb[b];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const float d = cur[2 * i] - yv[c][b * 8 + i];
                        if (VAR == 0) no[c] += (fabsf(d) >= cur[2 * i + 1]) ? 1 : 0;
                        if (VAR == 1) fo[c] += (fabsf(d) >= cur[2 * i + 1]) ? 1.0f : 0.0f;
                        if (VAR == 2) fo[c] += __builtin_amdgcn_fmed3f(fabsf(d) * 1e30f - cur[2 * i + 1], 0.f, 1.f);
                        if (VAR == 3) {   // sign-bit accumulation: (thr - |d|) < 0 -> top bit set
                            const float e = cur[2 * i + 1] - fabsf(d);
                            no[c] += (int)(__float_as_uint(e) >> 31);
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) { total += no[c]; ftotal += fo[c]; }
        }
    }
    if (total == -12345 || ftotal == -1.5f) out[threadIdx.x] = total;
}

static size_t g_lds = 0;      // dynamic LDS per workgroup: limits occupancy like the real kernel's lists do

template <int VAR, int NCH>
static void run(const char *name, const float2 *xq, const float *y, int *out)
{
    const int T = 32, iters = 400, blocks = 256 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<VAR, NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<VAR, NCH>), dim3(blocks), dim3(256), g_lds, 0, xq, y, 10, T, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<VAR, NCH>), dim3(blocks), dim3(256), g_lds, 0, xq, y, iters, T, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double wave_dims = (double)blocks * 4 * iters * T * GP * NCH;       // (wave, dimension) units
    const double per_simd = wave_dims / 1024.0;                              // 256 CUs x 4 SIMDs
    const double clk = ms * 1e-3 * 2.4e9 / per_simd;
    printf("%-28s NCH=%d  %.2f ms  %.2f clk per (wave,dim) per SIMD  => %.2f clk/instr at 3 instr/dim\n", name, NCH, ms,
           clk, clk / 3.0);
}

int main(int argc, char **argv)
{
    if (argc > 1) g_lds = (size_t)atoi(argv[1]);
    printf("dynamic LDS per workgroup: %zu bytes\n", g_lds);
    float2 *xq;
    float *y;
    int *out;
    hipMalloc(&xq, 32 * GP * sizeof(float2));
    hipMalloc(&y, 2 * GP * 64 * sizeof(float));
    hipMalloc(&out, 1024);
    std::vector<float2> hx(32 * GP);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = make_float2((float)(i % 7) - 3.f, 0.3f + 0.01f * (i % 5));
    std::vector<float> hy(2 * GP * 64);
    for (size_t i = 0; i < hy.size(); ++i) hy[i] = (float)(i % 11) * 0.4f - 2.f;
    hipMemcpy(xq, hx.data(), hx.size() * sizeof(float2), hipMemcpyHostToDevice);
    hipMemcpy(y, hy.data(), hy.size() * sizeof(float), hipMemcpyHostToDevice);
    run<0, 1>("int count (cmp+addc)", xq, y, out);
    run<0, 2>("int count (cmp+addc)", xq, y, out);
    run<1, 1>("float count (cmp+cndmask+add)", xq, y, out);
    run<1, 2>("float count (cmp+cndmask+add)", xq, y, out);
    run<2, 2>("mul+sub+med3+add", xq, y, out);
    run<3, 1>("sign-bit (sub,sub,lshr+add)", xq, y, out);
    run<3, 2>("sign-bit (sub,sub,lshr+add)", xq, y, out);
    return 0;
}
