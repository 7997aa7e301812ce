// wave_sum_check.hip -- the DPP / permlane-swap butterfly of csrc/elem.hip's wave_sum against the __shfl_xor loop it replaces:
// same pairings in the same order (32, 16, 8, 4, 2, 1), so every lane must hold the same BITS.  Prints the number of differing lanes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wave_sum_check wave_sum_check.hip && ./wave_sum_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
#include "../../3d-re-gen_amd/csrc/wave_sum.h"

__global__ void k(const float* in, float* a, float* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float v = i < n ? in[i] : 0.f;
    a[i] = wave_sum_shfl(v);
    b[i] = r3g::wave_sum(v);
}

int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    std::mt19937 g(7);
    std::normal_distribution<float> d(0.f, 1.f);
    for (int i = 0; i < n; ++i) h[i] = d(g) * (i % 97 == 0 ? 1e4f : 1.f) + (i % 3 == 0 ? 3.f : 0.f);
    float *in, *a, *b;
    (void)hipMalloc(&in, n * 4); (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4);
    (void)hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, in, a, b, n);
    std::vector<float> ha(n), hb(n);
    (void)hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) bad += *(uint32_t*)&ha[i] != *(uint32_t*)&hb[i];
    printf("wave_sum: %ld of %d lanes differ from the __shfl_xor butterfly (first values %.9g %.9g)\n", bad, n, ha[0], hb[0]);
    return bad != 0;
}
