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
// ---- round 5: GELU in PACKED fp16 (option "gelu_pk", default on) -------------------------------------------------------------
// gelu(x) = x S(x) with S = 0.5 + (x / 4) P(z), z = 2 clamp((x / 4)^2, 0, 1) - 1 in [-1, 1], P a degree-6 polynomial fitted to
// (S(4 t) - 1/2) / t under the constraint P(1) = 1/2 (tools/fit_gelu_pk.py: the fit alone is within 1.2e-4 of S for both flavours, the fp16 evaluation of it within 6.5e-4 (tanh) / 7.2e-4 (erf) -- the bound include/r3g.h states and tests/test_gelu_pk_cpu.py asserts is 7.5e-4; on [-1, 1]
// the monomial coefficients stay below 0.71, so Horner's rule is stable in fp16 -- the same polynomial in (x/4)^2 on [0, 1] has
// coefficients up to 22 and loses three digits).  Everything between the accumulator and the product runs on v_pk_*_f16, two
// values per lane and instruction at the plain VALU rate: one v_cvt_pk_f16_f32, two multiplies (the second with the clamp
// modifier), seven fused multiply-adds and the clamped last one (11 instructions) for a PAIR of values, then one v_fma_mix_f32 per value for
// x * S in fp32 -- 6.5 issue slots per value against 6 plain + 2 quarter-rate transcendentals (= 14 slots of the vector pipe) for
// x * rcp(1 + exp2(.)).  The clamp modifiers make the ends exact: for |x| >= 4 the constraint gives S = 0.5 +- 0.5 exactly, so
// gelu(x) = x resp. 0 there (the true values differ from that by < 7e-5), +infinity included (-inf * 0 is NaN, as x Phi(x) is in
// the reference's own arithmetic); NaN stays NaN.
// What it costs in accuracy: S carries ~1.6e-4 rms of fp16 rounding; behind the bf16 rounding of the output (1.66e-3 rms relative)
// the relative L2 error of the stored values grows from 1.655e-3 to 1.675e-3 (+1.2 %), with no bias (2.6e-6 relative).
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h16x2 h16_splat(float v) { return (h16x2){(_Float16)v, (_Float16)v}; }

__device__ __forceinline__ h16x2 h16_clamp01(h16x2 x) {     // folds into the producing instruction's clamp modifier
    return __builtin_elementwise_min(__builtin_elementwise_max(x, h16_splat(0.f)), h16_splat(1.f));
}

// S for a pair of values, as packed fp16
template <bool ERF>
__device__ __forceinline__ h16x2 gelu_pk_s(float x0, float x1) {
    // coefficients of P / 4 in powers of z (the 1/4 belongs to t = x / 4 of the last fused multiply-add); fp16 values, the
    // constant term adjusted so that Horner's rule in fp16 gives P(1) / 4 = 0.125 EXACTLY (tools/fit_gelu_pk.py)
    constexpr float c0 = (ERF ? 0.7041015625f : 0.7041015625f) * 0.25f;
    constexpr float c1 = (ERF ? -0.33837890625f : -0.3388671875f) * 0.25f;
    constexpr float c2 = (ERF ? 0.2225341796875f : 0.22216796875f) * 0.25f;
    constexpr float c3 = (ERF ? -0.1378173828125f : -0.13525390625f) * 0.25f;
    constexpr float c4 = (ERF ? 0.0916748046875f : 0.08990478515625f) * 0.25f;
    constexpr float c5 = (ERF ? -0.07086181640625f : -0.07281494140625f) * 0.25f;
    constexpr float c6 = (ERF ? 0.0289154052734375f : 0.0306396484375f) * 0.25f;
    const h16x2 t = {(_Float16)x0, (_Float16)x1};                   // v_cvt_pk_f16_f32 (round to nearest even)
    const h16x2 u = h16_clamp01(t * (t * h16_splat(0.0625f)));      // min((x / 4)^2, 1)
    const h16x2 z = u * h16_splat(2.f) + h16_splat(-1.f);
    h16x2 acc = h16_splat(c6);
    acc = acc * z + h16_splat(c5);
    acc = acc * z + h16_splat(c4);
    acc = acc * z + h16_splat(c3);
    acc = acc * z + h16_splat(c2);
    acc = acc * z + h16_splat(c1);
    acc = acc * z + h16_splat(c0);
    return h16_clamp01(t * acc + h16_splat(0.5f));                  // S in [0, 1]
}

// The same for NP pairs IN LOCKSTEP (round 6): Horner's rule is a chain of dependent packed instructions, and on gfx950 a packed
// fp16 instruction that reads the result of the one before it needs a wait state -- hipcc scheduled one pair's chain after the
// other with an s_nop behind every link (301 s_nop among the 846 instructions of a 64-row GELU pass).  Step-outer, pair-inner: the
// neighbours' links fill each other's wait states.  The arithmetic per pair is gelu_pk_s's, instruction for instruction: same bits.
template <bool ERF, int NP>
__device__ __forceinline__ void gelu_pk_sn(const float (&x)[2 * NP], h16x2 (&s)[NP]) {
    constexpr float c0 = (ERF ? 0.7041015625f : 0.7041015625f) * 0.25f;
    constexpr float c1 = (ERF ? -0.33837890625f : -0.3388671875f) * 0.25f;
    constexpr float c2 = (ERF ? 0.2225341796875f : 0.22216796875f) * 0.25f;
    constexpr float c3 = (ERF ? -0.1378173828125f : -0.13525390625f) * 0.25f;
    constexpr float c4 = (ERF ? 0.0916748046875f : 0.08990478515625f) * 0.25f;
    constexpr float c5 = (ERF ? -0.07086181640625f : -0.07281494140625f) * 0.25f;
    constexpr float c6 = (ERF ? 0.0289154052734375f : 0.0306396484375f) * 0.25f;
    constexpr float cs[6] = {c5, c4, c3, c2, c1, c0};
    h16x2 t[NP], z[NP], a[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) t[p] = (h16x2){(_Float16)x[2 * p], (_Float16)x[2 * p + 1]};
#pragma unroll
    for (int p = 0; p < NP; ++p) z[p] = t[p] * h16_splat(0.0625f);
#pragma unroll
    for (int p = 0; p < NP; ++p) z[p] = h16_clamp01(t[p] * z[p]);
#pragma unroll
    for (int p = 0; p < NP; ++p) z[p] = z[p] * h16_splat(2.f) + h16_splat(-1.f);
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p] = h16_splat(c6) * z[p] + h16_splat(c5);
#pragma unroll
    for (int k = 1; k < 6; ++k)
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = a[p] * z[p] + h16_splat(cs[k]);
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = h16_clamp01(t[p] * a[p] + h16_splat(0.5f));
}

// x * S for four values in fp32: v_fma_mix_f32 reads S as the low / high fp16 half of its pair (no conversion instruction; the
// compiler does not select the instruction from C++, hence the asm -- ONE statement, so that it pads at most once around it)
__device__ __forceinline__ f32x4 gelu_pk_mix(f32x4 v, h16x2 s01, h16x2 s23) {
    float g0, g1, g2, g3;
    asm("v_fma_mix_f32 %0, %4, %8, 0 op_sel_hi:[0,1,0]\n\t"
        "v_fma_mix_f32 %1, %5, %8, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n\t"
        "v_fma_mix_f32 %2, %6, %9, 0 op_sel_hi:[0,1,0]\n\t"
        "v_fma_mix_f32 %3, %7, %9, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]"
        : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(s01), "v"(s23));
    return (f32x4){g0, g1, g2, g3};
}
// NV register groups at once: 2 NV pair chains in lockstep
template <bool ERF, int NV>
__device__ __forceinline__ void gelu_pk4n(f32x4 (&v)[NV]) {
    float x[4 * NV];
    h16x2 s[2 * NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) x[4 * i + c] = v[i][c];
    gelu_pk_sn<ERF, 2 * NV>(x, s);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = gelu_pk_mix(v[i], s[2 * i], s[2 * i + 1]);
}

template <bool ERF>
__device__ __forceinline__ f32x4 gelu_pk4(f32x4 v) {
    const h16x2 s01 = gelu_pk_s<ERF>(v[0], v[1]), s23 = gelu_pk_s<ERF>(v[2], v[3]);
    float g0, g1, g2, g3;
    asm("v_fma_mix_f32 %0, %4, %8, 0 op_sel_hi:[0,1,0]\n\t"
        "v_fma_mix_f32 %1, %5, %8, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n\t"
        "v_fma_mix_f32 %2, %6, %9, 0 op_sel_hi:[0,1,0]\n\t"
        "v_fma_mix_f32 %3, %7, %9, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]"
        : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(s01), "v"(s23));
    return (f32x4){g0, g1, g2, g3};
}

// four values (one accumulator register group); PK: the packed-fp16 form
template <bool PK>
__device__ __forceinline__ f32x4 gelu_tanh4(f32x4 v) {
    if constexpr (PK) {
        return gelu_pk4<false>(v);
    } else {
        return (f32x4){gelu_tanh(v[0]), gelu_tanh(v[1]), gelu_tanh(v[2]), gelu_tanh(v[3])};
    }
}
template <bool PK>
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
    if constexpr (PK) {
        return gelu_pk4<true>(v);
    } else {
        return (f32x4){gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
    }
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
