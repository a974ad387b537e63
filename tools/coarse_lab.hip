// tools/coarse_lab.hip -- issue lab for the one-product filter's inner loop (l2c_topk.hip): chains of KS = 2 dependent
// v_mfma_f32_16x16x32_f16 per accumulator, four accumulators per pair of row-blocks, the filter of the PREVIOUS pair
// (two 8-way minimum trees + compares + an unlikely branch) issued next to them.  What does a filter instruction cost
// next to the MFMAs, and in which order?  One wave per SIMD, operands in registers (random, non-zero).  Diagnostic only.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans tools/coarse_lab.hip -o tools/coarse_lab.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int KS = 2, NB = 8, NP = 4;

struct cacc { f32x4 v[2][2]; };
#define SG(n) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, n, 0);

// MODE bit0: filter; bit1: round-robin accumulators (no two dependent MFMAs adjacent); bit2: sched_group_barrier pattern
// (1 MFMA, <= 2 VALU); bit3: four 16-byte loads per tile (ring of three sets); bit4: filter = ONE 16-way tree (9 instr.)
template <int MODE>
__device__ __forceinline__ cacc chain(const f16x8 (&a)[2][KS], const f16x8 (&b0)[KS], const f16x8 (&b1)[KS])
{
    cacc acc;
    if (MODE & 2) {
        f32x4 r[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                r[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c & 1][s], (c >> 1) ? b1[s] : b0[s], r[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc.v[c >> 1][c & 1] = r[c];
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s)
                r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c & 1][s], (c >> 1) ? b1[s] : b0[s], r, 0, 0, 0);
            acc.v[c >> 1][c & 1] = r;
        }
    }
    return acc;
}

struct cmins { float m0, m1; };
__device__ __forceinline__ cmins mins(const cacc &acc)
{
    cmins r;
    r.m0 = fminf(fminf(acc.v[0][0][0], acc.v[0][0][1]), acc.v[0][0][2]);
    r.m0 = fminf(fminf(r.m0, acc.v[0][0][3]), acc.v[0][1][0]);
    r.m0 = fminf(fminf(r.m0, acc.v[0][1][1]), acc.v[0][1][2]);
    r.m0 = fminf(r.m0, acc.v[0][1][3]);
    r.m1 = fminf(fminf(acc.v[1][0][0], acc.v[1][0][1]), acc.v[1][0][2]);
    r.m1 = fminf(fminf(r.m1, acc.v[1][0][3]), acc.v[1][1][0]);
    r.m1 = fminf(fminf(r.m1, acc.v[1][1][1]), acc.v[1][1][2]);
    r.m1 = fminf(r.m1, acc.v[1][1][3]);
    return r;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void lab(const f16x8 *__restrict__ src, float *out, int tiles, float tau_in)
{
    const int lane = threadIdx.x & 63;
    f16x8 xb[NB][KS];
#pragma unroll
    for (int rb = 0; rb < NB; ++rb)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xb[rb][s] = src[(rb * KS + s) * 64 + lane];
            asm volatile("" : "+a"(xb[rb][s]));
        }
    f16x8 a0[2][KS], a1[2][KS], a2[2][KS];
    const f16x8 *stream = src + 4096;
    auto load = [&](f16x8(&a)[2][KS], int t) {
        const f16x8 *p = stream + (size_t)((MODE & 8) ? (t & 127) : 0) * 256;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < KS; ++s) a[h][s] = p[(h * KS + s) * 64 + lane];
    };
    load(a0, 0); load(a1, 1); load(a2, 2);
    float tauv[NB];
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) tauv[rb] = tau_in + rb;
    cacc accP;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) accP.v[r][h] = f32x4{1e30f, 1e30f, 1e30f, 1e30f};
    int hits = 0;
    float run0 = 1e30f, run1 = 1e30f;
    float u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) u[i] = tau_in * (float)(i + lane);
    // FILT (MODE >> 4): 0 = the real thing (two trees on the MFMA results, compares, unlikely branch); 1 = the same ten
    // instructions on registers no MFMA writes, no branch; 2 = the trees on the MFMA results folded into running minima, no
    // compare, no branch; 3 = trees + ONE compare + branch on a single bit-OR of the verdicts (v_cmp, v_cmp, s_or)
    constexpr int FILT = MODE >> 4;
    auto act = [&](const cacc &acc, const cmins &m, int pr) {
        if (FILT == 0 || FILT == 3) {
            if (__builtin_expect(__builtin_amdgcn_ballot_w64((m.m0 < tauv[2 * pr]) | (m.m1 < tauv[2 * pr + 1])) != 0, 0)) {
                ++hits;
                out[threadIdx.x] = acc.v[0][0][0] + acc.v[1][1][3];
            }
        } else if (FILT == 2) {
            run0 = fminf(run0, m.m0);
            run1 = fminf(run1, m.m1);
        }
    };
    auto unrelated = [&]() {
        float r0 = fminf(fminf(u[0], u[1]), u[2]);
        r0 = fminf(fminf(r0, u[3]), u[4]); r0 = fminf(fminf(r0, u[5]), u[6]); r0 = fminf(r0, u[7]);
        float r1 = fminf(fminf(u[8], u[9]), u[10]);
        r1 = fminf(fminf(r1, u[11]), u[12]); r1 = fminf(fminf(r1, u[13]), u[14]); r1 = fminf(r1, u[15]);
        u[0] = r0 + 1.0f; u[8] = r1 + 1.0f;
        asm volatile("" : "+v"(u[0]), "+v"(u[8]));
    };
    auto step = [&](const f16x8(&a)[2][KS], f16x8(&an)[2][KS], int t) {
        if (MODE & 8) load(an, t + 2);
        cacc accA;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int prev = (p + NP - 1) % NP;
            if (p & 1) {
                cmins m;
                if ((MODE & 1) && FILT != 1) m = mins(accA);
                if ((MODE & 1) && FILT == 1) unrelated();
                accP = chain<MODE>(a, xb[2 * p], xb[2 * p + 1]);
                if (MODE & 4) {
                    SG(2) SG(2) SG(1) SG(1) SG(1) SG(1) SG(1) SG(1)
                }
                if ((MODE & 1) && FILT != 1) act(accA, m, prev);
                else asm volatile("" ::"v"(accA.v[0][0]), "v"(accA.v[0][1]), "v"(accA.v[1][0]), "v"(accA.v[1][1]));
            } else {
                cmins m;
                if ((MODE & 1) && FILT != 1) m = mins(accP);
                if ((MODE & 1) && FILT == 1) unrelated();
                accA = chain<MODE>(a, xb[2 * p], xb[2 * p + 1]);
                if (MODE & 4) {
                    SG(2) SG(2) SG(1) SG(1) SG(1) SG(1) SG(1) SG(1)
                }
                if ((MODE & 1) && FILT != 1) act(accP, m, prev);
                else asm volatile("" ::"v"(accP.v[0][0]), "v"(accP.v[0][1]), "v"(accP.v[1][0]), "v"(accP.v[1][1]));
            }
        }
    };
    const long long c0 = __builtin_readcyclecounter();
    for (int t = 0; t < tiles; t += 3) {
        step(a0, a2, t);
        step(a1, a0, t + 1);
        step(a2, a1, t + 2);
    }
    const long long c1 = __builtin_readcyclecounter();
    float s = (float)hits + run0 + run1 + u[0] + u[8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) s += accP.v[r][h][0];
    out[256 + blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long *>(out)[64] = c1 - c0;
}

template <int MODE>
static void run(int tiles, f16x8 *src, float *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256;
    hipLaunchKernelGGL((lab<MODE>), dim3(blocks), dim3(256), 0, 0, src, out, tiles / 8, -1e30f);
    hipDeviceSynchronize();
    float best = 1e30f;
    long long cyc = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((lab<MODE>), dim3(blocks), dim3(256), 0, 0, src, out, tiles, -1e30f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(&cyc, reinterpret_cast<long long *>(out) + 64, 8, hipMemcpyDeviceToHost); }
    }
    const double chains = (double)((tiles + 2) / 3 * 3) * NP;        // pair-chains per wave
    printf("mode=%2d filt=%d (%s%s%s%s)  %8.3f ms  ns per pair-chain %.1f  (counter ticks per pair-chain %.1f)  MFMA-only floor: 128 cycles\n", MODE,
           (MODE & 1) ? "filter " : "", (MODE & 2) ? "round-robin " : "", (MODE & 4) ? "sched-groups " : "", (MODE & 8) ? "loads " : "",
           best, best * 1e6 / chains, (double)cyc / chains);
}

int main()
{
    f16x8 *src; float *out;
    const size_t bytes = 64u << 20;
    hipMalloc(&src, bytes);
    {
        _Float16 *h = (_Float16 *)malloc(bytes);
        srand(1);
        for (size_t i = 0; i < bytes / 2; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
        hipMemcpy(src, h, bytes, hipMemcpyHostToDevice);
        free(h);
    }
    hipMalloc(&out, (256 + 256 * 256) * sizeof(float) + 1024);
    const int T = 30000;
    run<8>(T, src, out); run<9>(T, src, out); run<9 + 16>(T, src, out); run<9 + 32>(T, src, out); run<9 + 48>(T, src, out);
    run<13 + 16>(T, src, out); run<13 + 32>(T, src, out); run<11 + 32>(T, src, out); run<15 + 32>(T, src, out);
    return 0;
}
