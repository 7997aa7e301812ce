// gemm_common.h -- arithmetic shared by every GEMM kernel (gemm.hip; round 4's stream kernel, now tools/ubench/gemm4_stream.hip): the epilogue functions must be the same
// code in all of them, because the kernels are interchangeable per launch and their results are compared bit for bit.
#ifndef R3G_GEMM_COMMON_H
#define R3G_GEMM_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace r3g {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ float gelu_tanh(float x) {
    // torch GELU(approximate="tanh"): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3), written as x sigmoid(2u) =
    // x / (1 + 2^(x (c1 + c3 x^2))): 4 plain VALU + v_exp_f32 + v_rcp_f32 (round 3; the textbook form took 10 + 2 and lost
    // relative accuracy in the negative tail to the cancellation in 1 + tanh).  Saturates cleanly: 2^(+inf) -> rcp -> 0.
    // Round 4 measured a transcendental-free form (S(x) = 0.5 + t P(z), minimax polynomials of degree 10 / 11, |error| 4e-6,
    // tools/fit_gelu.py) with v_pk_fma_f32: SLOWER (MLP-in launch 256 -> 283 us).  On this chip a packed f32 instruction
    // takes two issue cycles per wave-instruction pair -- no gain over two plain ones -- and v_exp_f32 + v_rcp_f32 (8 cycles
    // each) are cheaper than the 12 fused multiply-adds (2 cycles each) that replace them (profiles/r04_gemm_stream.md).
    constexpr float c1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    constexpr float c3 = c1 * 0.044715f;
    const float z = x * fmaf(x * x, c3, c1);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
}

// exact-form GELU, x Phi(x), with erfc from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 output step):
// h = erfc(|x| / sqrt 2) / 2 = t (b1 + t (b2 + ...)) 2^(-x^2 log2(e) / 2), Phi = x >= 0 ? 1 - h : h (no cancellation in the tail).
// Round 4 also measured Phi(x) = sigmoid(p(x)) with an odd quintic p fitted to the normal distribution function (2.6e-5
// absolute, 7 plain VALU + 2 transcendentals instead of 14 + 2): the geo decoder's c_fc launch went 1149 -> 1140 us (0.8 %) --
// its epilogue arithmetic is not what that launch waits for -- and the form was not kept: 1.5e-7 stays.
__device__ __forceinline__ float gelu_erf(float x) {
    const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.3275911f * 0.7071067811865476f, 1.0f));
    const float e = __builtin_amdgcn_exp2f(x * x * -0.7213475204444817f);
    const float h = t * (0.127414796f + t * (-0.142248368f + t * (0.7107068705f + t * (-0.7265760135f + t * 0.5307027145f)))) * e;
    return x * (x >= 0.f ? 1.0f - h : h);
}
// four values (one accumulator register group)
__device__ __forceinline__ f32x4 gelu_tanh4(f32x4 v) {
    return (f32x4){gelu_tanh(v[0]), gelu_tanh(v[1]), gelu_tanh(v[2]), gelu_tanh(v[3])};
}
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
    return (f32x4){gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// v_cvt_pk_bf16_f32 (round to nearest even)
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<const uint32_t*>(&h);
}

// a 16-bit residual stream (EPI_RESID_BF16 / EPI_RESID_F16): two stream values of one 32-bit word as floats, and back (round
// to nearest even both ways; the sum old + update is formed in fp32 by the caller and rounded ONCE here)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <bool F16>
__device__ __forceinline__ f32x2 unpack16(uint32_t w) {
    if constexpr (F16) {
        return __builtin_convertvector(*reinterpret_cast<const f16x2*>(&w), f32x2);
    } else {
        return (f32x2){__uint_as_float(w << 16), __uint_as_float(w & 0xFFFF0000u)};
    }
}
template <bool F16>
__device__ __forceinline__ uint32_t pack16(float a, float b) {
    if constexpr (F16) {
        const f16x2 h = __builtin_convertvector((f32x2){a, b}, f16x2);
        return *reinterpret_cast<const uint32_t*>(&h);
    } else {
        return pack_bf16(a, b);
    }
}

}  // namespace r3g
#endif
