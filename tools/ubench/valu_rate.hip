// valu_rate.hip -- what the softmax instructions of the attention kernels cost on gfx950: cycles (s_memtime ticks) per
// wave-instruction for 1, 2 and 3 waves per SIMD, independent instructions back to back.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(unsigned long long* out, float* sink, int iters, float seed) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15]));
            if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (KIND == 3 && (i & 1) == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(f32x2*)&a[i]) : "v"(*(f32x2*)&a[(i + 2) & 14]));
            if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15]));
            if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 7) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15]));
            if (KIND == 8) asm volatile("v_fma_mix_f32 %0, %0, %1, 0 op_sel_hi:[0,1,0]" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (KIND == 10) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
            if (KIND == 11) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (KIND == 12) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (KIND == 13) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (KIND == 14) asm volatile("v_exp_f16_e64 %0, %0" : "+v"(a[i]));
            if (KIND == 9 && i < 8) asm volatile("s_nop 0\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 8]));
            if (KIND == 15) asm volatile("v_mov_b32 %0, 0" : "=v"(a[i]));
            if (KIND == 16 && (i & 1) == 0) asm volatile("v_mov_b64 %0, 0" : "=v"(*(f32x2*)&a[i]));
            if (KIND == 17 && (i & 1) == 0) asm volatile("v_pk_mov_b32 %0, 0, 0" : "=v"(*(f32x2*)&a[i]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 12345.f) sink[0] = s;
}

template <int KIND>
void run(const char* name, int per_iter) {
    unsigned long long* d; float* sink;
    (void)hipMalloc(&d, 8); (void)hipMalloc(&sink, 4);
    printf("%-22s", name);
    for (int threads : {256, 512, 768}) {
        const int iters = 4000;
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, sink, iters, 0.5f);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, sink, iters, 0.5f);
        unsigned long long c; (void)hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
        const double per = (double)c / ((double)iters * per_iter);
        printf("  %d wave/SIMD: %6.2f ticks per instruction per wave = %6.2f per SIMD", threads / 256, per, per / (threads / 256));
    }
    printf("\n");
}

int main() {
    run<4>("v_add_f32", 16);
    run<5>("v_fma_f32", 16);
    run<1>("v_max3_f32", 16);
    run<2>("v_cvt_pk_bf16_f32", 16);
    run<3>("v_pk_add_f32", 8);
    run<7>("v_pk_fma_f16", 16);
    run<8>("v_fma_mix_f32", 16);
    run<0>("v_exp_f32", 16);
    run<6>("v_rcp_f32", 16);
    run<10>("v_exp_f16", 16);
    run<14>("v_exp_f16 hi->hi", 16);
    run<11>("v_pk_max_f16", 16);
    run<12>("v_pk_add_f16", 16);
    run<13>("v_cvt_pk_f16_f32", 16);
    run<9>("v_permlane32_swap", 8);
    run<15>("v_mov_b32 0", 16);
    run<16>("v_mov_b64 0", 8);
    run<17>("v_pk_mov_b32 0, 0", 8);
    return 0;
}
