// mfma_shape.hip -- VERDICT r4 item 1(b): is the MFMA SHAPE an energy lever on a power-limited MI355X?
//
// The phased GEMM's k-loop (csrc/gemm.hip) issues v_mfma_f32_16x16x32_bf16: 8192 multiply-adds per 8 operand registers read.
// v_mfma_f32_32x32x16_bf16 does 16384 per 8 registers: half the operand-register bytes per FLOP.  This program runs the
// MFMA-only skeleton of one 256 x 256 x 1024 tile stream -- 8 waves per compute unit (2 per SIMD), each wave the 128 x 64
// sub-tile of the phased kernel = 128 accumulator registers, operand fragments held in registers and filled with RANDOM bf16
// values (the clock a kernel sustains depends on its operand bits: cdna_hip_programming.md rule 25) -- once per shape, for about
// a second of wall time each, alternating A/B/A/B, and prints per run: wall time, TFLOP/s, and the shader clock the run sustained
// (s_memtime ticks of one wave over the run / wall time).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_shape mfma_shape.hip && ./mfma_shape [seconds per run]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// SHAPE 0: 16x16x32 -- per k-tile of 64: 8 (m) x 4 (n) accumulator tiles x 2 k-steps = 64 MFMAs of 8192 MACs
// SHAPE 1: 32x32x16 -- per k-tile of 64: 4 (m) x 2 (n) accumulator tiles x 4 k-steps = 32 MFMAs of 16384 MACs
template <int SHAPE>
__global__ __launch_bounds__(512) void stream(const uint4* __restrict__ frag, float* sink, unsigned long long* ticks, int ktiles) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    bf16x8 a[8], b[8];      // 16 operand fragments of 4 registers: what the phased kernel keeps live (af[4][2], wf0[2][2], wf1[2][2])
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 va = frag[((blockIdx.x * 8 + wid) * 16 + i) * 64 + lane], vb = frag[((blockIdx.x * 8 + wid) * 16 + 8 + i) * 64 + lane];
        a[i] = __builtin_bit_cast(bf16x8, va);
        b[i] = __builtin_bit_cast(bf16x8, vb);
    }
    f32x4 c4[32];
    f32x16 c16[8];
#pragma unroll
    for (int i = 0; i < 32; ++i) c4[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c16[i][r] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ktiles; ++t) {
        if constexpr (SHAPE == 0) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        c4[j * 8 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[(j & 3) * 2 + kk], a[(i & 3) * 2 + kk], c4[j * 8 + i], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        c16[j * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j * 4 + kk], a[(i & 1) * 4 + kk], c16[j * 4 + i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += c4[i][0] + c4[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c16[i][0] + c16[i][15];
    if (s == 1234.5f) sink[0] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    const double want_s = argc > 1 ? atof(argv[1]) : 1.0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t nfrag = (size_t)cus * 8 * 16 * 64;
    std::vector<uint4> h(nfrag);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    auto bf = [&]() {   // bf16 bits of a uniform value in [-1, 1): random sign, exponent 2^-1 .. 2^-8, random mantissa
        const unsigned long long r = rnd();
        const unsigned e = 126 - (unsigned)(__builtin_ctzll((r >> 16) | 0x80) );
        return (unsigned)(((r & 1) << 15) | (e << 7) | ((r >> 1) & 0x7F));
    };
    for (auto& v : h) {
        v.x = bf() | (bf() << 16); v.y = bf() | (bf() << 16); v.z = bf() | (bf() << 16); v.w = bf() | (bf() << 16);
    }
    uint4* d; float* sink; unsigned long long* ticks;
    hipMalloc(&d, nfrag * 16); hipMalloc(&sink, 4); hipMalloc(&ticks, cus * 8);
    hipMemcpy(d, h.data(), nfrag * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // MACs per k-tile and wave are equal for both shapes: 64 x 8192 = 32 x 16384 = 524288
    const double flop_per_ktile = 2.0 * 524288 * 8 * cus;
    int ktiles = 200000;
    for (int pass = 0; pass < 2; ++pass) {     // pass 0: calibrate the k-tile count to the requested wall time
        for (int rep = 0; rep < (pass == 0 ? 1 : 2); ++rep)
            for (int shape = 0; shape < 2; ++shape) {
                hipEventRecord(e0);
                if (shape == 0) hipLaunchKernelGGL(stream<0>, dim3(cus), dim3(512), 0, 0, d, sink, ticks, ktiles);
                else hipLaunchKernelGGL(stream<1>, dim3(cus), dim3(512), 0, 0, d, sink, ticks, ktiles);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> tk(cus);
                hipMemcpy(tk.data(), ticks, cus * 8, hipMemcpyDeviceToHost);
                double mean = 0;
                for (auto v : tk) mean += (double)v;
                mean /= cus;
                if (pass == 1)
                    printf("{\"shape\": \"%s\", \"ktiles\": %d, \"wall_ms\": %.1f, \"tflops\": %.1f, \"memtime_ticks_per_ktile\": %.2f, "
                           "\"ticks_per_second_GHz\": %.4f, \"us_per_256x256x1024_tile\": %.2f}\n",
                           shape == 0 ? "16x16x32" : "32x32x16", ktiles, ms, flop_per_ktile * ktiles / (ms * 1e-3) / 1e12,
                           mean / ktiles, mean / (ms * 1e-3) / 1e9, ms * 1e3 / ktiles * 16);
                else if (shape == 0) ktiles = (int)(ktiles * want_s * 1e3 / ms);
            }
    }
    return 0;
}
