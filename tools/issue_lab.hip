// issue_lab.hip -- vector-ALU ISSUE rate of one MI355X SIMD as a function of waves per SIMD (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/issue_lab.hip -o tools/issue_lab.bin && tools/issue_lab.bin
// The roofline of the modified-Canberra counting pass (packed-f16 add / fma / dot2 per dimension pair) is the rate
// at which a SIMD issues such instructions, so it is measured, not assumed: W waves per SIMD each run a long
// stream of INDEPENDENT instructions (16 accumulators round-robin) of one kind or of the counting loop's 1:1:1 mix.
// Output: wave-instructions per clock and SIMD at the nominal 2.4 GHz, and G wave-instructions/s for the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(OP)                                                                                                  \
    OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, float *out)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i) * 1e-3f;
    float x = 1.0001f, y = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0) {
#define OP(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                REP16(OP)
#undef OP
            } else if (KIND == 1) {
#define OP(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                REP16(OP)
#undef OP
            } else if (KIND == 2) {
#define OP(i) asm volatile("v_pk_add_f16 %0, %1, %0" : "+v"(a[i]) : "v"(x));
                REP16(OP)
#undef OP
            } else if (KIND == 3) {
#define OP(i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                REP16(OP)
#undef OP
            } else if (KIND == 4) {      // the counting loop's mix: add, fma(clamp), dot2c on independent chains
#define OP(i) asm volatile("v_pk_add_f16 %0, %1, %0\n\tv_pk_fma_f16 %0, %0, %0, %2 clamp\n\tv_dot2c_f32_f16 %3, %0, %1" \
                           : "+v"(a[i]), "+v"(x) : "v"(y), "v"(a[(i + 8) & 15]));
                REP16(OP)
#undef OP
            } else if (KIND == 5) {
#define OP(i) asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                REP16(OP)
#undef OP
            } else if (KIND == 7) {      // the SWAR counting loop of round 2: sub, sub, bitop3, bcnt (accumulating)
                uint32_t *u32 = reinterpret_cast<uint32_t *>(a);
                const uint32_t xi = __float_as_uint(x), yi = __float_as_uint(y);
#define OP(i) asm volatile("v_sub_u32 %0, %2, %1\n\tv_sub_u32 %1, %1, %3\n\tv_bitop3_b32 %0, %0, %1, %4 bitop3:0x80\n\tv_bcnt_u32_b32 %1, %0, %1" \
                           : "+v"(u32[i]), "+v"(u32[(i + 8) & 15]) : "v"(xi), "v"(yi), "s"(0x80808080u));
                REP16(OP)
#undef OP
            } else if (KIND == 8) {
                uint32_t *u32 = reinterpret_cast<uint32_t *>(a);
#define OP(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(u32[i]) : "v"(__float_as_uint(x)));
                REP16(OP)
#undef OP
            } else if (KIND == 9) {
                uint32_t *u32 = reinterpret_cast<uint32_t *>(a);
#define OP(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(u32[i]) : "v"(__float_as_uint(x)));
                REP16(OP)
#undef OP
            } else if (KIND == 10) {
                uint32_t *u32 = reinterpret_cast<uint32_t *>(a);
#define OP(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(u32[i]) : "v"(__float_as_uint(x)), "v"(__float_as_uint(y)));
                REP16(OP)
#undef OP
            } else if (KIND >= 11 && KIND <= 16) {
                uint32_t *u32 = reinterpret_cast<uint32_t *>(a);
                const uint32_t xi = __float_as_uint(x), yi = __float_as_uint(y);
#define OP(i)                                                                                                         \
    if (KIND == 11) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(u32[i]) : "v"(xi), "v"(yi));                         \
    if (KIND == 12) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(u32[i]) : "v"(xi), "v"(yi));                    \
    if (KIND == 13) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(u32[i]) : "v"(xi), "v"(yi));                    \
    if (KIND == 14) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(u32[i]) : "v"(xi), "v"(yi));                       \
    if (KIND == 15) asm volatile("v_lshl_add_u32 %0, %1, 1, %0" : "+v"(u32[i]) : "v"(xi));                             \
    if (KIND == 16) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(u32[i]) : "v"(xi), "v"(yi));
                REP16(OP)
#undef OP
            } else {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*reinterpret_cast<double *>(&a[(i & 7) * 2])) : "v"(*reinterpret_cast<double *>(&a[0])), "v"(*reinterpret_cast<double *>(&a[2])));
                REP16(OP)
#undef OP
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == -12345.f) out[threadIdx.x] = s;
}

template <int KIND>
static void run(const char *name, int per_op, float *out)
{
    for (int W : {1, 2, 4, 8}) {
        const int iters = 20000, blocks = 256 * W;
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, 100, out);
        hipEventRecord(a);
        hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, iters, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double insts = (double)blocks * 4 * iters * 64.0 * per_op;      // wave-instructions
        const double per_simd_clk = insts / 1024.0 / (ms * 1e-3 * 2.4e9);
        printf("%-34s W=%d waves/SIMD  %8.3f ms  %.3f wave-instr/clk/SIMD (at 2.4 GHz)  %.1f G wave-instr/s\n", name, W, ms,
               per_simd_clk, insts / (ms * 1e-3) / 1e9);
    }
}

int main()
{
    float *out;
    hipMalloc(&out, 4096);
    run<0>("v_fma_f32", 1, out);
    run<1>("v_pk_fma_f16", 1, out);
    run<2>("v_pk_add_f16", 1, out);
    run<3>("v_dot2c_f32_f16", 1, out);
    run<4>("pk_add + pk_fma(clamp) + dot2c", 3, out);
    run<5>("v_min3_f32", 1, out);
    run<6>("v_pk_fma_f32", 1, out);
    run<8>("v_sub_u32", 1, out);
    run<9>("v_bcnt_u32_b32", 1, out);
    run<10>("v_bitop3_b32", 1, out);
    run<7>("sub + sub + bitop3 + bcnt (SWAR)", 4, out);
    run<11>("v_sad_u8", 1, out);
    run<12>("v_dot4_u32_u8", 1, out);
    run<13>("v_mad_u32_u24", 1, out);
    run<14>("v_add3_u32", 1, out);
    run<15>("v_lshl_add_u32", 1, out);
    run<16>("v_sad_u32", 1, out);
    return 0;
}
