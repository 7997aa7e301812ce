// MFMA issue-rate microbenchmark (gfx950): cycles per instruction for one wave per SIMD and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void k(unsigned long long* out, float* sink, int iters) {
    f32x4 a4[8];
    f32x16 a16[4];
    for (int i = 0; i < 8; ++i) a4[i] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) a16[i] = (f32x16){0};
    i32x8 x = {(int)threadIdx.x, 2, 3, 4, 5, 6, 7, 8}, y = {9, 8, 7, 6, 5, 4, 3, (int)threadIdx.x};
    bf16x8 p = __builtin_bit_cast(bf16x8, (int __attribute__((ext_vector_type(4)))){1, 2, 3, (int)threadIdx.x});
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, p, a4[i], 0, 0, 0);
            if (KIND == 1) a4[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x, y, a4[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if (KIND == 2) a16[i & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, y, a16[i & 3], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if (KIND == 3) a16[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, p, a16[i & 3], 0, 0, 0);
            if (KIND == 4) a4[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x, y, a4[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a4[i][0];
    for (int i = 0; i < 4; ++i) s += a16[i][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 12345.f) sink[0] = s;
}

template <int KIND>
void run(const char* name, double flop_per_instr) {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 4);
    for (int threads : {256, 512}) {
        const int iters = 2000;
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, sink, iters);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, sink, iters);
        unsigned long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
        const double per = (double)c / (iters * 8.0);
        printf("%-34s %d waves/SIMD: %.1f cycles per instruction per wave -> %.0f FLOP/cycle/SIMD\n", name, threads / 256, per,
               flop_per_instr * (threads / 256) / per);
    }
}

int main() {
    run<0>("mfma_f32_16x16x32_bf16", 16.0 * 16 * 32 * 2);
    run<3>("mfma_f32_32x32x16_bf16", 32.0 * 32 * 16 * 2);
    run<1>("mfma_scale_16x16x128 fp8", 16.0 * 16 * 128 * 2);
    run<2>("mfma_scale_32x32x64 fp8", 32.0 * 32 * 64 * 2);
    run<4>("mfma_scale_16x16x128 fp4", 16.0 * 16 * 128 * 2);
    return 0;
}
