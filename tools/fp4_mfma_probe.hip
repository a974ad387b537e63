// fp4_mfma_probe.hip -- does v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (e2m1) operands count 0/1 matches exactly,
// with which operand layout, and how fast?     hipcc -O3 --offload-arch=gfx950 tools/fp4_mfma_probe.hip -o tools/fp4_mfma_probe.bin
// Idea probed (mod-Canberra counting pass on the matrix pipe): a reference's bucket per dimension as a one-hot nibble
// vector, a target's window as a 0/1 nibble mask; their dot product is the number of dimensions that may be in window.
// Layout hypothesis: lane l = (r = l & 31, h = l >> 5) supplies row / column r and 32 of the 64 K values (half h) as
// the 32 nibbles of its first four operand VGPRs; A and B pair up position by position.  D: col = lane & 31 (B's r),
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (A's r) -- the dtype-independent 32x32 map.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const uint32_t *A, const uint32_t *B, float *D)
{
    const int lane = threadIdx.x;
    i32x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = i < 4 ? (int)A[lane * 4 + i] : 0; b[i] = i < 4 ? (int)B[lane * 4 + i] : 0; }
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    // cbsz = blgp = 4: fp4 e2m1; scales: E8M0 127 = 2^0 in every byte
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
#pragma unroll
    for (int i = 0; i < 16; ++i) D[lane * 16 + i] = c[i];
}

__global__ __launch_bounds__(256) void rate(int iters, const uint32_t *A, float *sink, unsigned long long *stamps)
{
    const int lane = threadIdx.x & 63;
    i32x8 a[13], b[4][13];
#pragma unroll
    for (int s = 0; s < 13; ++s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a[s][i] = i < 4 ? (int)A[(lane * 13 + s) * 4 + i] : 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r][s][i] = i < 4 ? (int)A[((lane + 7 * r + 3) % 64 * 13 + s) * 4 + i] : 0;
        }
    }
    f32x16 acc[4];
    float keep = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            keep = fmaxf(keep, acc[r][0]);
            f32x16 c;
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 13; ++s)
                c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s], b[r][s], c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            acc[r] = c;
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" : "+v"(a[0]));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { asm volatile("" ::"v"(acc[r])); keep += acc[r][1]; }
    if (keep == -1.f) sink[threadIdx.x] = keep;
}

int main()
{
    std::vector<uint32_t> hA(64 * 4), hB(64 * 4);
    srand(5);
    for (auto &v : hA) { v = 0; for (int n = 0; n < 8; ++n) v |= (uint32_t)((rand() & 3) == 0 ? 0x2 : 0x0) << (4 * n); }     // 1.0 = 0x2
    for (auto &v : hB) { v = 0; for (int n = 0; n < 8; ++n) v |= (uint32_t)((rand() & 1) ? 0x2 : 0x0) << (4 * n); }
    uint32_t *dA, *dB;
    float *dD;
    (void)hipMalloc(&dA, 64 * 13 * 16);
    (void)hipMalloc(&dB, 1024);
    (void)hipMalloc(&dD, 64 * 16 * 4);
    (void)hipMemcpy(dA, hA.data(), 1024, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    std::vector<float> hD(64 * 16);
    (void)hipMemcpy(hD.data(), dD, hD.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int reg = 0; reg < 16; ++reg) {
            const int col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            int want = 0;
            for (int h = 0; h < 2; ++h)
                for (int i = 0; i < 4; ++i)
                    for (int n = 0; n < 8; ++n) {
                        const int an = (hA[(row + 32 * h) * 4 + i] >> (4 * n)) & 0xF, bn = (hB[(col + 32 * h) * 4 + i] >> (4 * n)) & 0xF;
                        want += (an == 2 && bn == 2) ? 1 : 0;
                    }
            if ((float)want != hD[lane * 16 + reg]) { if (bad < 5) printf("lane %d reg %d: got %g want %d\n", lane, reg, hD[lane * 16 + reg], want); ++bad; }
        }
    printf("fp4 one-hot counts: %d of 1024 outputs differ from the layout hypothesis (D[5] = %g)\n", bad, hD[5]);
    // rate: 13 MFMAs per chain (K = 832), four chains per iteration, one wave per SIMD
    std::vector<uint32_t> big(64 * 13 * 4);
    for (auto &v : big) { v = 0; for (int n = 0; n < 8; ++n) v |= (uint32_t)((rand() & 7) == 0 ? 0x2 : 0x0) << (4 * n); }
    (void)hipMemcpy(dA, big.data(), big.size() * 4, hipMemcpyHostToDevice);
    float *sink;
    unsigned long long *stamps;
    (void)hipMalloc(&sink, 4096);
    (void)hipMalloc(&stamps, 256 * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 40000;
    float ms = 0;
    for (int rep = 0; rep < 20; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, iters, dA, sink, stamps);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> st(512);
    (void)hipMemcpy(st.data(), stamps, 4096, hipMemcpyDeviceToHost);
    const double ghz = (double)st[256] / (double)st[257] * 0.1;
    const double mfmas = (double)iters * 4 * 13;
    printf("v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4), chains of 13: %.2f ms, clock %.3f GHz, %.1f cycles per MFMA, %.0f G pair-dimension-buckets/s\n",
           ms, ghz, ms * 1e-3 * ghz * 1e9 / mfmas, 1024.0 * mfmas * 32 * 32 * 64 / (ms * 1e-3) / 1e9);
    return 0;
}
