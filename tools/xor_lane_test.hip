// xor_lane_test.hip -- checks knn_common.h's DPP / swizzle lane exchanges against the definition (lane ^ m) and the
// bitonic sort built on them against std::sort, on the GPU.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/xor_lane_test.hip -o tools/xor_lane_test.bin && tools/xor_lane_test.bin
#include "../nabo_amd/csrc/knn_common.h"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace nabo;

__global__ void kx(const int *in, int *out)
{
    const int v = in[threadIdx.x];
    out[0 * 64 + threadIdx.x] = xor_lane_i32(v, 1);
    out[1 * 64 + threadIdx.x] = xor_lane_i32(v, 2);
    out[2 * 64 + threadIdx.x] = xor_lane_i32(v, 4);
    out[3 * 64 + threadIdx.x] = xor_lane_i32(v, 8);
    out[4 * 64 + threadIdx.x] = xor_lane_i32(v, 16);
    out[5 * 64 + threadIdx.x] = xor_lane_i32(v, 32);
}

template <int EPL, typename K, bool FAST>
__global__ void ks(const K *kin, const uint32_t *vin, K *kout, uint32_t *vout)
{
    K key[EPL];
    uint32_t val[EPL];
    for (int r = 0; r < EPL; ++r) { key[r] = kin[blockIdx.x * 64 * EPL + r * 64 + threadIdx.x]; val[r] = vin[blockIdx.x * 64 * EPL + r * 64 + threadIdx.x]; }
    if constexpr (FAST) wave_sort_f32<EPL>(key, val);
    else wave_bitonic_sort<EPL, K>(key, val);
    for (int r = 0; r < EPL; ++r) { kout[blockIdx.x * 64 * EPL + r * 64 + threadIdx.x] = key[r]; vout[blockIdx.x * 64 * EPL + r * 64 + threadIdx.x] = val[r]; }
}

template <int EPL, typename K, bool FAST = false>
static int check_sort(const char *name)
{
    const int B = 200, N = 64 * EPL;
    std::vector<K> k(B * N), ko(B * N);
    std::vector<uint32_t> v(B * N), vo(B * N);
    unsigned s = 12345;
    for (int i = 0; i < B * N; ++i) { s = s * 1664525u + 1013904223u; k[i] = (K)((s >> 20) % 97) - 40; if ((s >> 9) % 50 == 0) k[i] = (K)__builtin_inff(); if ((s >> 11) % 60 == 0) k[i] = -k[i] * (K)1e30; v[i] = (uint32_t)(i % N) * 7919u % 1000u; }   // many ties
    K *dk, *dko; uint32_t *dv, *dvo;
    hipMalloc(&dk, sizeof(K) * B * N); hipMalloc(&dko, sizeof(K) * B * N); hipMalloc(&dv, 4 * B * N); hipMalloc(&dvo, 4 * B * N);
    hipMemcpy(dk, k.data(), sizeof(K) * B * N, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), 4 * B * N, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((ks<EPL, K, FAST>), dim3(B), dim3(64), 0, 0, dk, dv, dko, dvo);
    hipMemcpy(ko.data(), dko, sizeof(K) * B * N, hipMemcpyDeviceToHost); hipMemcpy(vo.data(), dvo, 4 * B * N, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < B; ++b) {
        std::vector<std::pair<K, uint32_t>> ref(N);
        for (int i = 0; i < N; ++i) ref[i] = {k[b * N + i], v[b * N + i]};
        std::sort(ref.begin(), ref.end());
        for (int i = 0; i < N; ++i) bad += !(ref[i].first == ko[b * N + i] && ref[i].second == vo[b * N + i]);
    }
    printf("%s: %d mismatches\n", name, bad);
    return bad;
}

int main()
{
    int h[64], o[6 * 64], *di, *dout, bad = 0;
    for (int i = 0; i < 64; ++i) h[i] = 1000 + i;
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kx, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    const int ms[6] = {1, 2, 4, 8, 16, 32};
    for (int c = 0; c < 6; ++c) {
        int b = 0;
        for (int i = 0; i < 64; ++i) b += o[c * 64 + i] != 1000 + (i ^ ms[c]);
        printf("xor %2d: %d wrong lanes\n", ms[c], b);
        bad += b;
    }
    bad += check_sort<1, float>("sort 64 float");
    bad += check_sort<2, float>("sort 128 float");
    bad += check_sort<1, float, true>("fast sort 64 float");
    bad += check_sort<2, float, true>("fast sort 128 float");
    bad += check_sort<1, double>("sort 64 double");
    bad += check_sort<4, double>("sort 256 double");
    printf(bad ? "FAILED\n" : "ok\n");
    return bad ? 1 : 0;
}
