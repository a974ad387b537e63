// mfma_clock_lab.hip -- what clock does an MI355X hold under a dense f16 MFMA stream, and does the MFMA shape matter?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_clock_lab.hip -o tools/mfma_clock_lab.bin && tools/mfma_clock_lab.bin
// The f16x3 Euclidean filter (nabo_amd/csrc/l2h_topk.hip) is a stream of 10-MFMA chains of v_mfma_f32_32x32x16_f16,
// one wave per SIMD.  Its floor is 32 cycles per MFMA at WHATEVER clock the chip holds under that load
// (MI355X_MICROARCH.md, "DVFS give-back"): this lab measures that clock on random operands (zeros hold 2.4 GHz and
// prove nothing), for the 32x32x16 and the 16x16x32 shape at the same output tile per wave (4 x 32x32), operands in
// registers, chains of 10 accumulating MFMAs like the kernel's.  In-kernel clock = d(s_memtime) / d(s_memrealtime)
// x 100 MHz, stamped around the loop of the last of a series of back-to-back launches (>= 2 s of load).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ inline uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__device__ inline h8 rnd8(uint32_t seed, int zero)
{
    h8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t u = mix(seed * 8u + (uint32_t)i);
        r[i] = zero ? (_Float16)0.f : (_Float16)(((int)(u & 0xffff) - 32768) * (1.0f / 32768.0f));
    }
    return r;
}

// SHAPE 0: 4 accumulators of 32x32 (16 regs), chains of KC MFMAs 32x32x16.  SHAPE 1: 16 accumulators of 16x16
// (4 regs), chains of KC/2 MFMAs 16x16x32 -- same flops, same output tile, same operand registers per chain.
template <int SHAPE, int KC, bool SEQ>
__global__ __launch_bounds__(256) void lab(int iters, int zero, float *sink, unsigned long long *stamps)
{
    h8 a[KC], b[4][KC > 10 ? 1 : KC];
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        a[i] = rnd8(t * 64u + (uint32_t)i, zero);
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r][i] = rnd8(t * 64u + 16u + (uint32_t)(r * KC + i), zero);
    }
    float keep = 0.f;
    f16v acc32[4];
    f4v acc16[16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc32[r][i] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc16[r] = f4v{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        // a chain's result is read one whole iteration later (the kernel's filter reads it one chain later): the loop
        // measures the matrix pipe, not the latency of the last MFMA of a chain
        if (SHAPE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                keep = fminf(keep, acc32[r][0]);
                f16v acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < KC; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[r][i], acc, 0, 0, 0);
                acc32[r] = acc;
                if (SEQ) __builtin_amdgcn_sched_barrier(0);     // chain after chain (dependent MFMAs back to back)
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                keep = fminf(keep, acc16[r][0]);
                f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < KC / 2; ++i)       // 16x16x32: K = 32 per MFMA, 8 f16 per lane as well
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) % KC], b[r & 3][(i + (r >> 2)) % KC], acc, 0, 0, 0);
                acc16[r] = acc;
                if (SEQ) __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("" : "+v"(a[0]));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
    #pragma unroll
    for (int r = 0; r < 4; ++r) { asm volatile("" ::"v"(acc32[r])); keep += acc32[r][1]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) { asm volatile("" ::"v"(acc16[r])); keep += acc16[r][1]; }
    if (keep == -1.f) sink[t] = keep;
}

template <int SHAPE, int KC, bool SEQ>
static void run(const char *name, int zero, float *sink, unsigned long long *stamps)
{
    const int blocks = 256, iters = 60000;          // 4 x KC (or 16 x KC/2) MFMAs per iteration and wave
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 30; ++rep) {            // ~2+ s of load before the launch that is read
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((lab<SHAPE, KC, SEQ>), dim3(blocks), dim3(256), 0, 0, iters, zero, sink, stamps);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz(blocks);
    for (int i = 0; i < blocks; ++i) ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    std::sort(ghz.begin(), ghz.end());
    const double flop = (double)blocks * 4 * iters * 4.0 * KC * 32768.0;
    const double cyc_per_mfma32 = (ms * 1e-3) * ghz[blocks / 2] * 1e9 / ((double)iters * 4.0 * KC);
    printf("%-44s %s  %8.2f ms  %7.1f TFLOP/s  clock %.3f GHz (median WG; min %.3f max %.3f)  %.1f cycles per 32x32x16-equivalent\n",
           name, zero ? "zeros " : "random", ms, flop / (ms * 1e-3) / 1e12, ghz[blocks / 2], ghz[0], ghz[blocks - 1],
           cyc_per_mfma32);
}

int main()
{
    float *sink;
    unsigned long long *stamps;
    (void)hipMalloc(&sink, 256 * 256 * 4);
    (void)hipMalloc(&stamps, 256 * 16);
    run<0, 10, true>("32x32x16_f16, 4 chains of 10 one after another", 1, sink, stamps);
    run<0, 10, true>("32x32x16_f16, 4 chains of 10 one after another", 0, sink, stamps);
    run<0, 10, false>("32x32x16_f16, 4 chains of 10 interleaved", 0, sink, stamps);
    run<1, 10, true>("16x16x32_f16, 16 chains of 5 one after another", 0, sink, stamps);
    run<1, 10, false>("16x16x32_f16, 16 chains of 5 interleaved", 0, sink, stamps);
    run<0, 10, true>("32x32x16_f16, 4 chains of 10 one after another", 0, sink, stamps);
    return 0;
}
