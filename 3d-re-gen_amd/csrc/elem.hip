// elem.hip -- HBM-bound elementwise / normalisation / layout kernels of the hot path (gfx950).
//
// These carry what the reference executes as separate PyTorch elementwise passes around its GEMMs
// (upstream hunyuan3ddit.py: LayerNorm + adaLN modulate, QKNorm(RMSNorm), rearrange "B L (K H D) ->
// K B H L D", timestep_embedding, Modulation; attention_blocks.py: LayerNorm, q/k LayerNorm, per-head
// interleaved qkv split, FourierEmbedder, ln_post + output_proj; pipelines.py: CFG combine + Euler step;
// Dinov2: patch embedding unfold, SwiGLU).  One wave64 per row for the row reductions, 16-byte
// accesses wherever the layout allows.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <math.h>
#include <stdint.h>

#include "kernels.h"
#include "prof.h"
#include "wave_sum.h"

namespace r3g {
namespace {

__device__ __forceinline__ uint16_t f2bf(float f) {  // v_cvt_pk_bf16_f32, round to nearest even
    const __bf16 h = (__bf16)f;
    return *reinterpret_cast<const uint16_t*>(&h);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
// wave_sum: csrc/wave_sum.h (round 6: the xor butterfly on DPP / permlane swaps instead of six ds_bpermute round trips; same bits)
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// ---------------------------------------------------------------- LayerNorm (+affine) (+adaLN modulate) -> bf16
constexpr int LN_MAX_V4 = 8;  // up to C = 64*4*8 = 2048
// four consecutive elements of a row held as f32 (XBF16 = false) or bf16 (true), as they sit in memory (the conversion of a prefetched row must not sit next to its load: it
// would wait for the data at once)
// (XF: 0 f32 | 1 bf16 | 2 fp16 -- the DiT's 16-bit residual stream of round 4 is fp16, the reference's own activation type)
struct Half4Raw { uint2 u; };
template <int XF> struct Row4Raw { typedef float4 type; };
template <> struct Row4Raw<1> { typedef uint2 type; };
template <> struct Row4Raw<2> { typedef Half4Raw type; };
template <int XF>
__device__ __forceinline__ typename Row4Raw<XF>::type load_row4_raw(const float* base, int64_t elem) {
    if constexpr (XF == 1) return *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem);
    else if constexpr (XF == 2) return Half4Raw{*reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem)};
    else return *reinterpret_cast<const float4*>(base + elem);
}
__device__ __forceinline__ float4 row4_cvt(const float4& r) { return r; }
__device__ __forceinline__ float4 row4_cvt(const uint2& u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ float4 row4_cvt(const Half4Raw& h) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = *reinterpret_cast<const h2*>(&h.u.x), b = *reinterpret_cast<const h2*>(&h.u.y);
    return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
}

// One wave per row, the row held in registers.
// NV: float4 per lane as a COMPILE-TIME constant (C = 256 NV; 0 = any C % 256 == 0 up to 2048, decided at run time).
//   With a run-time count every `if (i < nv)` and every optional operand (`if (p.w)` ...) is a branch around a load, and
//   the compiler closes each with s_waitcnt vmcnt(0): the loads of a row went out ONE PER MEMORY LATENCY (round 2's ISA:
//   global_load, vmcnt(0), branch, global_load, ...).  With NV fixed the row's loads are issued back to back, and the
//   optional operands are fetched in four straight-line groups up front.
// RPW: rows per wave, one after the other, the next row's loads in flight while the current one is reduced and written: a
//   workgroup of 4 rows lives ~6 us however little it does (dispatch, kernarg fetch, exit), which at 131 072 rows per
//   launch is as long as its memory time; 4 x RPW rows per workgroup amortise it.
// The arithmetic per row (element -> lane map, order of every sum) is the same in all instantiations.
// MODE (round 6): which optional terms a launch has, known at compile time for the two combinations the path is made of -- 1: affine
// (w and b; VAE / DINOv2 / geo decoder), 2: adaLN modulation (scale and shift, no affine; every DiT LayerNorm); -1: decided at run time
// (anything else, and the fp8 output).  The run-time form tests four pointers per register group inside its unrolled loops: 66
// v_cndmask and 19 branches in 686 instructions per row.  Same operations in the same order: bit-identical (tests/test_model_gpu.py).
template <int XBF16, int NV, int RPW, int MODE = -1>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs p) {
    constexpr int NVC = NV ? NV : LN_MAX_V4;
    const int lane = threadIdx.x & 63;
    const int nv = NV ? NV : (p.C >> 8);
    const bool has_w = MODE < 0 ? p.w != nullptr : (MODE & 1) != 0, has_b = MODE < 0 ? p.b != nullptr : (MODE & 1) != 0;
    typedef typename Row4Raw<XBF16>::type Raw;
    float4 v[NVC];
    Raw vr[NVC];
    auto load = [&](int r) {
        const int bt = r / p.rows_per_batch, lr = r - bt * p.rows_per_batch;
        const int64_t xr = (int64_t)bt * p.x_batch_stride + (int64_t)lr * p.ldx;
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) vr[i] = load_row4_raw<XBF16>(p.x, xr + (i * 64 + lane) * 4);
    };
    int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row >= p.rows) return;
    load(row);
    // the affine weights do not depend on the row: fetched once per wave
    float4 wv[NVC], bv[NVC];
    if (has_w) {
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) wv[i] = *reinterpret_cast<const float4*>(p.w + (i * 64 + lane) * 4);
    }
    if (has_b) {
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) bv[i] = *reinterpret_cast<const float4*>(p.b + (i * 64 + lane) * 4);
    }
    float4 av[NVC], hv[NVC];
    const float *sc_held = nullptr, *sh_held = nullptr;
    // one row of the wave (round 6: a function of its own, so that the one-row-per-wave instantiations are straight-line code --
    // as the single iteration of a rolled loop the compiler kept the loop-carried copies: 69 v_mov per row)
    auto one_row = [&](const int rr) __attribute__((always_inline)) -> bool {
        const bool more = RPW > 1 && rr + 1 < RPW && row + 1 < p.rows;   // wave-uniform
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) v[i] = row4_cvt(vr[i]);
        const int batch = row / p.rows_per_batch, lrow = row - batch * p.rows_per_batch;
        const bool seg2 = row >= p.seg2_row0 && row < p.seg2_row1;   // wave-uniform (one row per wave at a time)
        const float* sc = MODE >= 0 && !(MODE & 2) ? nullptr : seg2 ? p.scale2 : (p.scale ? p.scale + (int64_t)batch * p.mod_stride : nullptr);
        const float* sh = MODE >= 0 && !(MODE & 2) ? nullptr : seg2 ? p.shift2 : (p.shift ? p.shift + (int64_t)batch * p.mod_stride : nullptr);
        const bool has_sc = MODE < 0 ? sc != nullptr : (MODE & 2) != 0, has_sh = MODE < 0 ? sh != nullptr : (MODE & 2) != 0;
        // the modulation vectors are 8 bytes per element against the row's 2-4: a wave's consecutive rows nearly always
        // share them (same object, same segment), so they stay in registers until the pointer changes (wave-uniform)
        if (has_sc && sc != sc_held) {
#pragma unroll
            for (int i = 0; i < NVC; ++i)
                if (NV || i < nv) av[i] = *reinterpret_cast<const float4*>(sc + (i * 64 + lane) * 4);
        }
        if (has_sh && sh != sh_held) {
#pragma unroll
            for (int i = 0; i < NVC; ++i)
                if (NV || i < nv) hv[i] = *reinterpret_cast<const float4*>(sh + (i * 64 + lane) * 4);
        }
        sc_held = sc;
        sh_held = sh;
        if (more) load(row + 1);   // youngest loads of the iteration: the counted waits for the operands above leave them in flight
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) s += v[i].x + v[i].y + v[i].z + v[i].w;
        // (C = 1024 at compile time: 1 / 1024 is a power of two, the product IS the quotient; other lengths keep the division)
        const float mean = NV == 4 ? wave_sum(s) * (1.0f / 1024.0f) : wave_sum(s) / (float)p.C;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                ss = fmaf(d, d, fmaf(c, c, fmaf(b, b, fmaf(a, a, ss))));   // spelled out: see the note on contraction below
            }
        const float rstd = rsqrtf((NV == 4 ? wave_sum(ss) * (1.0f / 1024.0f) : wave_sum(ss) / (float)p.C) + p.eps);
        uint16_t* y = p.y + (int64_t)batch * p.y_batch_stride + (int64_t)lrow * p.ldy;
        const bool to_fp8 = MODE < 0 && p.y8 != nullptr;      // wave-uniform
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) {
                const int c = (i * 64 + lane) * 4;
                float o[4] = {(v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd};
                // every product and sum rounded on its own: whether the compiler fuses `o * w + b` depends on the control flow
                // around it, which differs between instantiations.  __fmul_rn / __fadd_rn do NOT prevent that (they are inline
                // header functions whose product and sum carry the translation unit's `contract` flag: in the straight-line
                // compile-time-MODE kernels of round 6 they came out as v_pk_fma_f32 and every downstream value moved by a bf16
                // rounding flip) -- plain operators under `fp contract(off)` do.
                {
#pragma clang fp contract(off)
                    if (has_w) { o[0] = o[0] * wv[i].x; o[1] = o[1] * wv[i].y; o[2] = o[2] * wv[i].z; o[3] = o[3] * wv[i].w; }
                    if (has_b) { o[0] = o[0] + bv[i].x; o[1] = o[1] + bv[i].y; o[2] = o[2] + bv[i].z; o[3] = o[3] + bv[i].w; }
                    if (has_sc) {
                        const float a0 = 1.f + av[i].x, a1 = 1.f + av[i].y, a2 = 1.f + av[i].z, a3 = 1.f + av[i].w;
                        o[0] = o[0] * a0; o[1] = o[1] * a1; o[2] = o[2] * a2; o[3] = o[3] * a3;
                    }
                    if (has_sh) { o[0] = o[0] + hv[i].x; o[1] = o[1] + hv[i].y; o[2] = o[2] + hv[i].z; o[3] = o[3] + hv[i].w; }
                }
                if (to_fp8) {   // keep the finished values (the row's maximum decides their scale), write below
                    v[i] = make_float4(o[0], o[1], o[2], o[3]);
                    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                    continue;
                }
                uint2 pk;
                pk.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
                pk.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
                *reinterpret_cast<uint2*>(y + c) = pk;
            }
        if (to_fp8) {
            // e4m3 operand of an fp8 GEMM: q = round(o / scale), scale = amax / 448 (one per row: the row is all here)
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d, 64));
            const float qs = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
            const float inv = 1.0f / qs;
            if (lane == 0) p.y_scale[row] = qs;
            uint8_t* y8 = p.y8 + (int64_t)row * p.ldy8;
#pragma unroll
            for (int i = 0; i < NVC; ++i)
                if (NV || i < nv) {
                    int w = 0;
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].x * inv, v[i].y * inv, w, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].z * inv, v[i].w * inv, w, true);
                    *reinterpret_cast<uint32_t*>(y8 + (i * 64 + lane) * 4) = (uint32_t)w;
                }
        }
        return more;
    };
    if constexpr (RPW == 1) {
        (void)one_row(0);
    } else {
#pragma unroll 1
        for (int rr = 0; rr < RPW; ++rr, ++row)
            if (!one_row(rr)) break;
    }   // rows of this wave
}

// generic-C variant (C % 64 == 0, C <= 2048): scalar per-lane elements (tiny configs, C = 64 / 128)
__global__ __launch_bounds__(256) void layernorm_small_kernel(LnArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int batch = row / p.rows_per_batch, lrow = row - batch * p.rows_per_batch;
    const float* x = p.x + (int64_t)batch * p.x_batch_stride + (int64_t)lrow * p.ldx;
    const int n = p.C >> 6;
    float v[32];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i)
        if (i < n) { v[i] = x[i * 64 + lane]; s += v[i]; }
    const float mean = wave_sum(s) / (float)p.C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i)
        if (i < n) { const float a = v[i] - mean; ss += a * a; }
    const float rstd = rsqrtf(wave_sum(ss) / (float)p.C + p.eps);
    const bool seg2 = row >= p.seg2_row0 && row < p.seg2_row1;   // wave-uniform (one row per wave)
    const float* sc = seg2 ? p.scale2 : (p.scale ? p.scale + (int64_t)batch * p.mod_stride : nullptr);
    const float* sh = seg2 ? p.shift2 : (p.shift ? p.shift + (int64_t)batch * p.mod_stride : nullptr);
    uint16_t* y = p.y + (int64_t)batch * p.y_batch_stride + (int64_t)lrow * p.ldy;
#pragma unroll
    for (int i = 0; i < 32; ++i)
        if (i < n) {
            const int c = i * 64 + lane;
            float o = (v[i] - mean) * rstd;
            if (p.w) o *= p.w[c];
            if (p.b) o += p.b[c];
            if (sc) o *= 1.f + sc[c];
            if (sh) o += sh[c];
            y[c] = f2bf(o);
        }
}

// ---------------------------------------------------------------- qkv split + per-head q/k norm + V transpose
// grid (ceil(L/64), H, B); 4 waves x 16 tokens; lane = head dim d.
__global__ __launch_bounds__(256) void qkv_split_kernel(QkvSplitArgs p) {
    __shared__ uint16_t vt[64][66];  // [d][token], padded
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int b = blockIdx.z, h = blockIdx.y, tok0 = blockIdx.x * 64;
    const uint16_t* src = p.src + (int64_t)b * p.src_batch_stride;
    float qw = 1.f, qb = 0.f, kw = 1.f, kb = 0.f;
    if (p.norm != QKN_NONE) {
        if (p.qw) qw = p.qw[lane];
        if (p.kw) kw = p.kw[lane];
        if (p.norm == QKN_LAYERNORM) {
            if (p.qb) qb = p.qb[lane];
            if (p.kb) kb = p.kb[lane];
        }
    }
    for (int i = 0; i < 16; ++i) {
        const int tl = wid * 16 + i, tok = tok0 + tl;
        const bool valid = tok < p.L;
        const uint16_t* row = src + (int64_t)(valid ? tok : 0) * p.ld + h * p.head_stride + lane;
        const int64_t drow = (int64_t)p.dst_row0 + tok;
        if (p.q_off >= 0 && p.Q) {
            float q = bf2f(row[p.q_off]);
            if (p.norm == QKN_RMS) {
                // upstream RMSNorm: x.float() * rsqrt(mean(x^2)+eps) * scale (kept in fp32 here)
                const float r = rsqrtf(wave_sum(q * q) * (1.f / 64.f) + p.eps);
                q = q * r * qw;
            } else if (p.norm == QKN_LAYERNORM) {
                const float mean = wave_sum(q) * (1.f / 64.f);
                const float dlt = q - mean;
                const float r = rsqrtf(wave_sum(dlt * dlt) * (1.f / 64.f) + p.eps);
                q = dlt * r * qw + qb;
            }
            if (p.q_scale != 0.f) q *= p.q_scale;
            if (valid) p.Q[(((int64_t)b * p.H + h) * p.Lq_pad + drow) * 64 + lane] = f2bf(q);
        }
        if (p.k_off >= 0 && p.K) {
            float k = bf2f(row[p.k_off]);
            if (p.norm == QKN_RMS) {
                const float r = rsqrtf(wave_sum(k * k) * (1.f / 64.f) + p.eps);
                k = k * r * kw;
            } else if (p.norm == QKN_LAYERNORM) {
                const float mean = wave_sum(k) * (1.f / 64.f);
                const float dlt = k - mean;
                const float r = rsqrtf(wave_sum(dlt * dlt) * (1.f / 64.f) + p.eps);
                k = dlt * r * kw + kb;
            }
            if (valid) p.K[(((int64_t)b * p.H + h) * p.Lk_pad + drow) * 64 + lane] = f2bf(k);
        }
        if (p.v_off >= 0 && p.Vt) vt[lane][tl] = valid ? row[p.v_off] : (uint16_t)0;
    }
    if (p.v_off >= 0 && p.Vt) {
        __syncthreads();
        // each wave writes 16 d-rows of 64 tokens (128 B per row); pad tokens are written as zeros
        for (int i = 0; i < 16; ++i) {
            const int d = wid * 16 + i;
            const int64_t col = (int64_t)p.dst_row0 + tok0 + lane;   // 64-aligned run of keys: vt_key_pos stays inside it
            if (col < p.Lk_pad) p.Vt[(((int64_t)b * p.H + h) * 64 + d) * (int64_t)p.Lk_pad + vt_key_pos(col)] = vt[d][lane];
        }
    }
}

// ---------------------------------------------------------------- small-batch GEMV (modulation / time embedding MLP)
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ x, int B, int K,
                                                   const uint16_t* __restrict__ W, int64_t ldw,
                                                   const float* __restrict__ bias, float* __restrict__ y, int N,
                                                   int silu_in, int silu_out) {
    extern __shared__ float xs[];  // [B][K]
    for (int i = threadIdx.x; i < B * K; i += 256) {
        const float v = x[i];
        xs[i] = silu_in ? silu(v) : v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    for (int n = wave; n < N; n += nwaves) {
        float acc[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[b] = 0.f;
        const uint16_t* w = W + (int64_t)n * ldw;
        for (int k = lane * 8; k < K; k += 512) {
            const uint4 pk = *reinterpret_cast<const uint4*>(w + k);
            const uint32_t u[4] = {pk.x, pk.y, pk.z, pk.w};
            float wf[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wf[2 * e] = __uint_as_float(u[e] << 16);
                wf[2 * e + 1] = __uint_as_float(u[e] & 0xFFFF0000u);
            }
#pragma unroll
            for (int b = 0; b < 8; ++b)
                if (b < B) {
                    const float* xb = xs + b * K + k;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[b] += wf[e] * xb[e];
                }
        }
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (b < B) {
                float v = wave_sum(acc[b]);
                if (lane == 0) {
                    v += bias ? bias[n] : 0.f;
                    y[(int64_t)b * N + n] = silu_out ? silu(v) : v;
                }
            }
    }
}

// grid (64, njobs): block (bx, job) computes rows bx*4+wave, +256, ... of job's matrix for all B inputs
__global__ __launch_bounds__(256) void gemv_multi_kernel(const float* __restrict__ x, int B, int K,
                                                         const GemvJob* __restrict__ jobs, float* __restrict__ y,
                                                         int silu_in) {
    extern __shared__ float xs[];
    for (int i = threadIdx.x; i < B * K; i += 256) {
        const float v = x[i];
        xs[i] = silu_in ? silu(v) : v;
    }
    __syncthreads();
    const GemvJob job = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63;
    float* out = y + job.out_off;
    for (int n = blockIdx.x * 4 + (threadIdx.x >> 6); n < job.N; n += gridDim.x * 4) {
        float acc[2] = {0.f, 0.f};
        const uint16_t* w = job.W + (int64_t)n * job.ldw;
        for (int k = lane * 8; k < K; k += 512) {
            const uint4 pk = *reinterpret_cast<const uint4*>(w + k);
            const uint32_t u[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float w0 = __uint_as_float(u[e] << 16), w1 = __uint_as_float(u[e] & 0xFFFF0000u);
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    if (b < B) acc[b] += w0 * xs[b * K + k + 2 * e] + w1 * xs[b * K + k + 2 * e + 1];
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if (b < B) {
                const float v = wave_sum(acc[b]);
                if (lane == 0) out[(int64_t)b * job.N + n] = v + (job.bias ? job.bias[n] : 0.f);
            }
    }
}

__global__ void timestep_embedding_kernel(const float* t, float t_scalar, int B, float time_factor, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 256) return;
    const int b = i >> 8, c = i & 255, j = c & 127;
    const float freq = expf(-logf(10000.0f) * (float)j / 128.0f);
    const float arg = (t ? t[b] : t_scalar) * time_factor * freq;
    out[i] = c < 128 ? cosf(arg) : sinf(arg);
}

__global__ void cast_pad_kernel(const float* in, int64_t ldi, uint16_t* out, int64_t ldo, int rows, int C, int Cpad,
                                float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * Cpad) return;
    const int r = (int)(i / Cpad), c = (int)(i % Cpad);
    out[(int64_t)r * ldo + c] = c < C ? f2bf(in[(int64_t)r * ldi + c] * scale) : (uint16_t)0;
}

__global__ void fill_rows_kernel(float* dst, int64_t ld, int rows, int C, const float* vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * C) return;
    const int r = (int)(i / C), c = (int)(i % C);
    dst[(int64_t)r * ld + c] = vals[c];
}

__global__ void cfg_euler_kernel(float* lat, const float* v2, int64_t n, float g, float ds) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float vc = v2[i], vu = v2[n + i];
    lat[i] = lat[i] + ds * (vu + g * (vc - vu));
}

// *flag |= 1 when x holds a NaN or an infinity (one atomic per wave that saw one)
__global__ void nonfinite_flag_kernel(const float* __restrict__ x, int64_t n, int* flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad |= (__float_as_uint(x[i]) & 0x7F800000u) == 0x7F800000u;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ void swiglu_kernel(const uint16_t* in, int64_t ldi, uint16_t* out, int64_t ldo, int rows, int F) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= (int64_t)rows * F) return;
    const int r = (int)(i / F), c = (int)(i % F);  // F % 8 == 0
    const uint4 a = *reinterpret_cast<const uint4*>(in + (int64_t)r * ldi + c);
    const uint4 g = *reinterpret_cast<const uint4*>(in + (int64_t)r * ldi + F + c);
    const uint32_t au[4] = {a.x, a.y, a.z, a.w}, gu[4] = {g.x, g.y, g.z, g.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a0 = __uint_as_float(au[e] << 16), a1 = __uint_as_float(au[e] & 0xFFFF0000u);
        const float g0 = __uint_as_float(gu[e] << 16), g1 = __uint_as_float(gu[e] & 0xFFFF0000u);
        o[e] = (uint32_t)f2bf(silu(a0) * g0) | ((uint32_t)f2bf(silu(a1) * g1) << 16);
    }
    *reinterpret_cast<uint4*>(out + (int64_t)r * ldo + c) = make_uint4(o[0], o[1], o[2], o[3]);
}

// dense grid point `idx` = (i*(R+1) + j)*(R+1) + k, coords = np.linspace(-bound, bound, R+1, dtype=float32)
__device__ __forceinline__ float lin_coord(int i, int R, double bound) {
    if (i == R) return (float)bound;
    const double step = (2.0 * bound) / (double)R;
    return (float)((double)i * step + (-bound));
}

__global__ void fourier_grid_kernel(uint16_t* out, int64_t start, int count, int R, double bound, int nf, int pi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t idx = start + i;
    const int n = R + 1;
    const int kk = (int)(idx % n), jj = (int)((idx / n) % n), ii = (int)(idx / ((int64_t)n * n));
    float xyz[3] = {0.f, 0.f, 0.f};
    if (ii < n) { xyz[0] = lin_coord(ii, R, bound); xyz[1] = lin_coord(jj, R, bound); xyz[2] = lin_coord(kk, R, bound); }
    uint16_t* o = out + (int64_t)i * 64;
    const int dim = 3 + 6 * nf;
    for (int c = 0; c < 3; ++c) o[c] = f2bf(xyz[c]);
    for (int c = 0; c < 3; ++c)
        for (int f = 0; f < nf; ++f) {
            float fr = (float)(1 << f);
            if (pi) fr *= 3.14159265358979323846f;
            const float e = xyz[c] * fr;
            o[3 + c * nf + f] = f2bf(sinf(e));
            o[3 + 3 * nf + c * nf + f] = f2bf(cosf(e));
        }
    for (int c = dim; c < 64; ++c) o[c] = 0;
}

template <bool XBF16, int NV, int RPW>
__global__ __launch_bounds__(256) void ln_dot_kernel(const float* x, int64_t ldx, int rows, int C, int do_ln,
                                                     const float* lnw, const float* lnb, float eps, const float* w,
                                                     float b, float* out) {
    // a wave takes RPW rows one after the other, the row held in registers (C % 256 == 0, C <= 2048): one HBM pass, the
    // next row's loads in flight behind the current row's reductions; NV as in layernorm_kernel
    constexpr int NVC = NV ? NV : LN_MAX_V4;
    const int lane = threadIdx.x & 63;
    const int nv = NV ? NV : (C >> 8);
    typedef typename Row4Raw<XBF16>::type Raw;
    float4 v[NVC];
    Raw vr[NVC];
    auto load = [&](int r) {
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) vr[i] = load_row4_raw<XBF16>(x, (int64_t)r * ldx + (i * 64 + lane) * 4);
    };
    int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row >= rows) return;
    load(row);
    const bool affine = do_ln && lnw;
    float4 gv[NVC], hv[NVC], ww[NVC];
    if (affine) {
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) {
                gv[i] = *reinterpret_cast<const float4*>(lnw + (i * 64 + lane) * 4);
                hv[i] = *reinterpret_cast<const float4*>(lnb + (i * 64 + lane) * 4);
            }
    }
#pragma unroll
    for (int i = 0; i < NVC; ++i)
        if (NV || i < nv) ww[i] = *reinterpret_cast<const float4*>(w + (i * 64 + lane) * 4);
#pragma unroll 1
    for (int rr = 0; rr < RPW; ++rr, ++row) {
        const bool more = RPW > 1 && rr + 1 < RPW && row + 1 < rows;   // wave-uniform
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) v[i] = row4_cvt(vr[i]);
        if (more) load(row + 1);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) s += v[i].x + v[i].y + v[i].z + v[i].w;
        float mean = wave_sum(s) / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) {
                const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
                ss = fmaf(a3, a3, fmaf(a2, a2, fmaf(a1, a1, fmaf(a0, a0, ss))));
            }
        float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
        if (!do_ln) { mean = 0.f; rstd = 1.f; }
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (NV || i < nv) {
                float o[4] = {(v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd};
                if (affine) {
                    o[0] = fmaf(o[0], gv[i].x, hv[i].x); o[1] = fmaf(o[1], gv[i].y, hv[i].y);
                    o[2] = fmaf(o[2], gv[i].z, hv[i].z); o[3] = fmaf(o[3], gv[i].w, hv[i].w);
                }
                acc += fmaf(o[3], ww[i].w, fmaf(o[2], ww[i].z, fmaf(o[1], ww[i].y, __fmul_rn(o[0], ww[i].x))));
            }
        acc = wave_sum(acc);
        if (lane == 0) out[row] = acc + b;
        if (!more) break;
    }   // rows of this wave
}

// generic-C fallback (C % 64 == 0)
__global__ __launch_bounds__(256) void ln_dot_small_kernel(const float* x, int64_t ldx, int rows, int C, int do_ln,
                                                           const float* lnw, const float* lnb, float eps,
                                                           const float* w, float b, float* out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * ldx;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    float mean = wave_sum(s) / (float)C;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; ss += d * d; }
    float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
    if (!do_ln) { mean = 0.f; rstd = 1.f; }
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) {
        float v = (xr[c] - mean) * rstd;
        if (do_ln && lnw) v = v * lnw[c] + lnb[c];
        acc += v * w[c];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row] = acc + b;
}

__global__ void im2col_kernel(const float* img, int S, int ps, uint16_t* out, int Kpad) {
    const int P = S / ps;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)P * P * Kpad) return;
    const int pidx = (int)(i / Kpad), k = (int)(i % Kpad);
    const int K = 3 * ps * ps;
    uint16_t v = 0;
    if (k < K) {
        const int c = k / (ps * ps), r = k % (ps * ps), dy = r / ps, dx = r % ps;
        const int py = pidx / P, px = pidx % P;
        v = f2bf(img[((int64_t)c * S + py * ps + dy) * S + px * ps + dx]);
    }
    out[i] = v;
}

__global__ void add_rows_kernel(float* x, int64_t ldx, const float* pos, int64_t ldp, int rows, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * C) return;
    const int r = (int)(i / C), c = (int)(i % C);
    x[(int64_t)r * ldx + c] += pos[(int64_t)r * ldp + c];
}

__global__ void f32_to_bf16_kernel(const float* in, uint16_t* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f2bf(in[i]);
}

inline int blocks_for(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

}  // namespace

// instantiations: C = 1024 (DiT, VAE, geo decoder) and 1536 (DINOv2-g) with the count fixed, anything else at run time
#define R3G_LN_LAUNCH(XB)                                                                                                  \
    {                                                                                                                      \
        const bool r4 = ln_rows_per_wave(p.rows) == 4;                                                                     \
        const dim3 g1((p.rows + 3) / 4), g4((p.rows + 15) / 16), blk(256);                                                 \
        /* the two combinations the path is made of, with their optional terms known at compile time (MODE) */            \
        const bool m_aff = g_ln_modes && p.w && p.b && !p.scale && !p.shift && !p.scale2 && !p.shift2 && !p.y8;            \
        const bool m_mod = g_ln_modes && !p.w && !p.b && p.scale && p.shift && !p.y8 && (p.seg2_row1 <= p.seg2_row0 || (p.scale2 && p.shift2)); \
        if (p.C == 1024 && g_ln_fixed && m_mod) {                                                                          \
            if (r4) hipLaunchKernelGGL((layernorm_kernel<XB, 4, 4, 2>), g4, blk, 0, s, p);                                 \
            else hipLaunchKernelGGL((layernorm_kernel<XB, 4, 1, 2>), g1, blk, 0, s, p);                                    \
        } else if (p.C == 1024 && g_ln_fixed && m_aff) {                                                                   \
            if (r4) hipLaunchKernelGGL((layernorm_kernel<XB, 4, 4, 1>), g4, blk, 0, s, p);                                 \
            else hipLaunchKernelGGL((layernorm_kernel<XB, 4, 1, 1>), g1, blk, 0, s, p);                                    \
        } else if (p.C == 1024 && g_ln_fixed) {                                                                            \
            if (r4) hipLaunchKernelGGL((layernorm_kernel<XB, 4, 4>), g4, blk, 0, s, p);                                    \
            else hipLaunchKernelGGL((layernorm_kernel<XB, 4, 1>), g1, blk, 0, s, p);                                       \
        } else if (p.C == 1536 && g_ln_fixed) {                                                                            \
            if (r4) hipLaunchKernelGGL((layernorm_kernel<XB, 6, 4>), g4, blk, 0, s, p);                                    \
            else hipLaunchKernelGGL((layernorm_kernel<XB, 6, 1>), g1, blk, 0, s, p);                                       \
        } else {                                                                                                           \
            if (r4) hipLaunchKernelGGL((layernorm_kernel<XB, 0, 4>), g4, blk, 0, s, p);                                    \
            else hipLaunchKernelGGL((layernorm_kernel<XB, 0, 1>), g1, blk, 0, s, p);                                       \
        }                                                                                                                  \
    }
// rows per wave of the row-in-registers kernels: 4 when that still leaves >= 4096 workgroups (two per workgroup slot of
// the device), otherwise 1 (the DiT's 7 552-row launches need every wave they can get).  g_ln_rows: 0 automatic | 1 | 4.
static int g_ln_rows = 0;
static bool g_ln_modes = true;   // option "ln_modes" (round 6): compile-time MODE instantiations for the affine-only / modulation-only launches
void ln_set_modes(bool on) { g_ln_modes = on; }
static bool g_ln_fixed = true;   // compile-time element counts for C = 1024 / 1536 (0: round 2's run-time count everywhere)
void ln_set_fixed_count(bool on) { g_ln_fixed = on; }
void ln_set_rows_per_wave(int rows) { g_ln_rows = rows == 1 || rows == 4 ? rows : 0; }
// 4 rows per wave from 65 536 rows per launch (the geo decoder's 131 072): below that, one row per wave.  Round 4 tried 4 rows
// from 16 384 (a launch group's 30 060 modulated DiT rows, with the modulation vectors held across a wave's rows): LayerNorm
// family 45.5 -> 46.8 ms per object, A/B twice on one box (profiles/r04_layernorm_rows.md) -- the option stays for the next try
static int g_ln_rows4_min = 65536;
void ln_set_rows4_min(int rows) { g_ln_rows4_min = rows > 0 ? rows : 65536; }
static int ln_rows_per_wave(int rows) { return g_ln_rows ? g_ln_rows : (rows >= g_ln_rows4_min ? 4 : 1); }

hipError_t layernorm_launch(const LnArgs& p, hipStream_t s) {
    if (p.rows <= 0) return hipSuccess;
    if (p.C % 64 || p.C > 2048) return hipErrorInvalidValue;
    ProfScope ps(PC_LAYERNORM, 6.0 * (double)p.rows * p.C, s);
    if (p.y8 && (p.C % 256 || (p.ldy8 & 3) || !p.y_scale)) return hipErrorInvalidValue;
    if (p.x_bf16 == 2) {
        if (p.C % 256 || (p.ldx & 3) || (p.ldy & 3) || (p.x_batch_stride & 3)) return hipErrorInvalidValue;
        R3G_LN_LAUNCH(2)
    } else if (p.x_bf16) {
        if (p.C % 256 || (p.ldx & 3) || (p.ldy & 3) || (p.x_batch_stride & 3)) return hipErrorInvalidValue;
        R3G_LN_LAUNCH(1)
    } else if (p.C % 256 == 0 && (p.ldx & 3) == 0 && (p.ldy & 3) == 0) {
        R3G_LN_LAUNCH(0)
    } else {
        hipLaunchKernelGGL(layernorm_small_kernel, dim3((p.rows + 3) / 4), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

hipError_t qkv_split_launch(const QkvSplitArgs& p, hipStream_t s) {
    if (p.L <= 0) return hipSuccess;
    if (p.dst_row0 % 64) return hipErrorInvalidValue;
    ProfScope ps(PC_QKV_SPLIT, 4.0 * 64.0 * p.L * p.H * p.B * ((p.q_off >= 0) + (p.k_off >= 0) + (p.v_off >= 0)), s);
    hipLaunchKernelGGL(qkv_split_kernel, dim3((p.L + 63) / 64, p.H, p.B), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t gemv_launch(const float* x, int B, int K, const uint16_t* W, int64_t ldw, const float* bias, float* y,
                       int N, int act_silu_in, int act_silu_out, hipStream_t s) {
    if (B > 8 || K % 8 || (size_t)B * K * 4 > 64 * 1024) return hipErrorInvalidValue;
    const int blocks = N >= 1024 ? 256 : (N + 3) / 4;
    ProfScope ps(PC_GEMV, 2.0 * (double)N * K, s);
    hipLaunchKernelGGL(gemv_kernel, dim3(blocks), dim3(256), (size_t)B * K * 4, s, x, B, K, W, ldw, bias, y, N,
                       act_silu_in, act_silu_out);
    return hipGetLastError();
}

hipError_t gemv_multi_launch(const float* x, int B, int K, const GemvJob* d_jobs, int njobs, float* y, int silu_in,
                             hipStream_t s) {
    if (B > 2 || K % 8 || njobs <= 0) return hipErrorInvalidValue;
    ProfScope ps(PC_GEMV, 0.0, s);
    hipLaunchKernelGGL(gemv_multi_kernel, dim3(64, njobs), dim3(256), (size_t)B * K * 4, s, x, B, K, d_jobs, y, silu_in);
    return hipGetLastError();
}

hipError_t timestep_embedding_launch(const float* t, float t_scalar, int B, float time_factor, float* out,
                                     hipStream_t s) {
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(B), dim3(256), 0, s, t, t_scalar, B, time_factor, out);
    return hipGetLastError();
}

hipError_t cast_pad_launch(const float* in, int64_t ldi, uint16_t* out, int64_t ldo, int rows, int C, int Cpad,
                           float scale, hipStream_t s) {
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(cast_pad_kernel, dim3(blocks_for((int64_t)rows * Cpad, 256)), dim3(256), 0, s, in, ldi, out, ldo,
                       rows, C, Cpad, scale);
    return hipGetLastError();
}

hipError_t fill_rows_launch(float* dst, int64_t ld, int rows, int C, const float* vals, hipStream_t s) {
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(fill_rows_kernel, dim3(blocks_for((int64_t)rows * C, 256)), dim3(256), 0, s, dst, ld, rows, C, vals);
    return hipGetLastError();
}

hipError_t cfg_euler_launch(float* latents, const float* v2, int64_t n, float guidance, float dsigma, hipStream_t s) {
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(cfg_euler_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, latents, v2, n, guidance, dsigma);
    return hipGetLastError();
}

hipError_t nonfinite_flag_launch(const float* x, int64_t n, int* flag, hipStream_t s) {
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    const int blocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, x, n, flag);
    return hipGetLastError();
}

hipError_t swiglu_launch(const uint16_t* in, int64_t ldi, uint16_t* out, int64_t ldo, int rows, int F, hipStream_t s) {
    if (F % 8) return hipErrorInvalidValue;
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(swiglu_kernel, dim3(blocks_for((int64_t)rows * F / 8, 256)), dim3(256), 0, s, in, ldi, out, ldo,
                       rows, F);
    return hipGetLastError();
}

hipError_t fourier_grid_launch(uint16_t* out, int64_t start, int count, int R, double bound, int num_freqs,
                               int include_pi, hipStream_t s) {
    if (3 + 6 * num_freqs > 64) return hipErrorInvalidValue;
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(fourier_grid_kernel, dim3(blocks_for(count, 256)), dim3(256), 0, s, out, start, count, R, bound,
                       num_freqs, include_pi);
    return hipGetLastError();
}

#define R3G_LNDOT_LAUNCH(XB)                                                                                               \
    {                                                                                                                      \
        const bool r4 = ln_rows_per_wave(rows) == 4;                                                                       \
        const dim3 g1((rows + 3) / 4), g4((rows + 15) / 16), blk(256);                                                     \
        if (C == 1024 && g_ln_fixed) {                                                                                     \
            if (r4) hipLaunchKernelGGL((ln_dot_kernel<XB, 4, 4>), g4, blk, 0, s, x, ldx, rows, C, do_ln, lnw, lnb, eps, w, b, out); \
            else hipLaunchKernelGGL((ln_dot_kernel<XB, 4, 1>), g1, blk, 0, s, x, ldx, rows, C, do_ln, lnw, lnb, eps, w, b, out);    \
        } else {                                                                                                           \
            if (r4) hipLaunchKernelGGL((ln_dot_kernel<XB, 0, 4>), g4, blk, 0, s, x, ldx, rows, C, do_ln, lnw, lnb, eps, w, b, out); \
            else hipLaunchKernelGGL((ln_dot_kernel<XB, 0, 1>), g1, blk, 0, s, x, ldx, rows, C, do_ln, lnw, lnb, eps, w, b, out);    \
        }                                                                                                                  \
    }
// ---- round 6: ln_post + output_proj folded into the last residual GEMM of the geo decoder (EPI_RESID_BF16_LND, kernels.h) -------------
// gw[n] = lnw[n] * w[n];  consts[0] = sum gw,  consts[1] = sum lnb[n] * w[n] + b   (one workgroup, fixed summation order)
__global__ __launch_bounds__(256) void lnd_prepare_kernel(const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                          const float* __restrict__ w, float b, int N, float* __restrict__ gw,
                                                          float* __restrict__ consts) {
    __shared__ double sg[256], sb[256];
    double ag = 0.0, ab = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float g = (lnw ? lnw[n] : 1.0f) * w[n];
        gw[n] = g;
        ag += (double)g;
        ab += (double)(lnb ? lnb[n] : 0.0f) * (double)w[n];
    }
    sg[threadIdx.x] = ag;
    sb[threadIdx.x] = ab;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { sg[threadIdx.x] += sg[threadIdx.x + st]; sb[threadIdx.x] += sb[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { consts[0] = (float)sg[0]; consts[1] = (float)(sb[0] + (double)b); }
}

// part [rows][parts][4] = {sum, sum of squared deviations from the chunk mean, sum x gw, -} of 64-column chunks -> out[row]:
// mean = sum S / N;  M2 = sum m2_i + 64 sum (S_i / 64 - mean)^2  (Chan et al.: exact merge of per-chunk moments);
// logit = rsqrt(M2 / N + eps) * (dot - mean * consts[0]) + consts[1].  One thread per row, chunks in ascending order.
// Round 6: the chunk statistics of a workgroup's 256 rows ([rows][parts][4] floats: one contiguous block) come in through LDS --
// coalesced 16-byte loads by consecutive threads, then every thread reads ITS row's `parts` records in order (the sums below keep
// their order: same bits).  One thread per row straight from memory touched 64 different lines per load instruction and read every
// record twice: 21 us per launch for 33.5 MB (1.6 TB/s).  Row stride parts + 1 records: conflict-free 128-bit LDS reads.
constexpr int kLnPartsMax = 16;
__device__ __forceinline__ const float4* ln_parts_tile(const float4* __restrict__ part, int rows, int parts, float4* tile) {
    const int row0 = blockIdx.x * 256;
    const int nrows = min(256, rows - row0);
    const float4* src = part + (int64_t)row0 * parts;
    const int total = nrows * parts;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int r = e / parts, i = e - r * parts;
        tile[r * (parts + 1) + i] = src[e];
    }
    __syncthreads();
    return tile + threadIdx.x * (parts + 1);
}

__global__ __launch_bounds__(256) void lnd_finalize_kernel(const float4* __restrict__ part, int rows, int parts, float eps,
                                                           const float* __restrict__ consts, float* __restrict__ out) {
    __shared__ float4 tile[256 * (kLnPartsMax + 1)];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const float4* p = parts <= kLnPartsMax ? ln_parts_tile(part, rows, parts, tile) : part + (int64_t)r * parts;   // (uniform)
    if (r >= rows) return;
    float S = 0.f, D = 0.f;
    for (int i = 0; i < parts; ++i) { S += p[i].x; D += p[i].z; }
    const float N = 64.0f * (float)parts;
    const float mean = S / N;
    float M2 = 0.f;
    for (int i = 0; i < parts; ++i) {
        const float dm = p[i].x * (1.0f / 64.0f) - mean;
        M2 += p[i].y + 64.0f * dm * dm;
    }
    out[r] = rsqrtf(M2 / N + eps) * (D - mean * consts[0]) + consts[1];
}

// ---- ln_3 fold (EPI_RESID_BF16_ST / EPI_BF16_GELU_ERF_LNF) ---------------------------------------------------------------------
// One wave per output row n: W'[n][k] = bf16(W[n][k] * gamma[k]); c1[n] = sum_k W'[n][k] (of the ROUNDED values: the algebra
// LN(x) W^T = rstd (x W'^T - mean c1) + c2 is then exact in W'); c2[n] = sum_k beta[k] W[n][k] + b[n].  K % 64 == 0.
__global__ __launch_bounds__(256) void lnf_prepare_kernel(const uint16_t* __restrict__ w, int64_t ldw, const float* __restrict__ b,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, int N, int K,
                                                          uint16_t* __restrict__ w2, float* __restrict__ c1, float* __restrict__ c2) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float a1 = 0.f, a2 = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float wv = __uint_as_float((uint32_t)w[(int64_t)n * ldw + k] << 16);
        const uint16_t r = f2bf(wv * gamma[k]);
        w2[(int64_t)n * K + k] = r;
        a1 += __uint_as_float((uint32_t)r << 16);
        a2 = fmaf(beta[k], wv, a2);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a1 += __shfl_xor(a1, d, 64);
        a2 += __shfl_xor(a2, d, 64);
    }
    if (lane == 0) {
        c1[n] = a1;
        c2[n] = a2 + (b ? b[n] : 0.f);
    }
}

// chunk statistics [rows][parts][4] -> (mean, rstd) per row (the same merge as lnd_finalize_kernel)
__global__ __launch_bounds__(256) void lnf_stats_kernel(const float4* __restrict__ part, int rows, int parts, float eps,
                                                        float2* __restrict__ stats) {
    __shared__ float4 tile[256 * (kLnPartsMax + 1)];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const float4* p = parts <= kLnPartsMax ? ln_parts_tile(part, rows, parts, tile) : part + (int64_t)r * parts;   // (uniform)
    if (r >= rows) return;
    float S = 0.f;
    for (int i = 0; i < parts; ++i) S += p[i].x;
    const float N = 64.0f * (float)parts;
    const float mean = S / N;
    float M2 = 0.f;
    for (int i = 0; i < parts; ++i) {
        const float dm = p[i].x * (1.0f / 64.0f) - mean;
        M2 += p[i].y + 64.0f * dm * dm;
    }
    stats[r] = make_float2(mean, rsqrtf(M2 / N + eps));
}

hipError_t lnf_prepare_launch(const uint16_t* w, int64_t ldw, const float* b, const float* gamma, const float* beta, int N, int K,
                              uint16_t* w2, float* c1, float* c2, hipStream_t s) {
    if (K % 64 || !gamma || !beta) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lnf_prepare_kernel, dim3((N + 3) / 4), dim3(256), 0, s, w, ldw, b, gamma, beta, N, K, w2, c1, c2);
    return hipGetLastError();
}

hipError_t lnf_stats_launch(const float* part, int rows, int parts, float eps, float* stats, hipStream_t s) {
    ProfScope prof_scope_(PC_LAYERNORM, 0.0, s);
    hipLaunchKernelGGL(lnf_stats_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, (const float4*)part, rows, parts, eps, (float2*)stats);
    return hipGetLastError();
}

hipError_t lnd_prepare_launch(const float* lnw, const float* lnb, const float* w, float b, int N, float* gw, float* consts, hipStream_t s) {
    hipLaunchKernelGGL(lnd_prepare_kernel, dim3(1), dim3(256), 0, s, lnw, lnb, w, b, N, gw, consts);
    return hipGetLastError();
}

hipError_t lnd_finalize_launch(const float* part, int rows, int parts, float eps, const float* consts, float* out, hipStream_t s) {
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(lnd_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, (const float4*)part, rows, parts, eps, consts, out);
    return hipGetLastError();
}

hipError_t ln_dot_launch(const float* x, int64_t ldx, int rows, int C, int do_ln, const float* lnw, const float* lnb,
                         float eps, const float* w, float b, float* out, hipStream_t s, int x_bf16) {
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    if (x_bf16) {
        if (C % 256 || C > 2048 || (ldx & 3)) return hipErrorInvalidValue;
        R3G_LNDOT_LAUNCH(true)
    } else if (C % 256 == 0 && C <= 2048 && (ldx & 3) == 0) {
        R3G_LNDOT_LAUNCH(false)
    } else {
        hipLaunchKernelGGL(ln_dot_small_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, rows, C, do_ln, lnw, lnb,
                           eps, w, b, out);
    }
    return hipGetLastError();
}

// bf16 [rows][K] -> fp8 e4m3 (OCP) [rows][K] with one fp32 scale per row: q = round(x / scale), scale = amax / 448.
// One wave per row, 8 elements per lane and step (K % 8 == 0): one pass for the maximum, one for the conversion (the row
// stays in L2 between them).  v_cvt_pk_fp8_f32 saturates and rounds to nearest even.
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const uint16_t* __restrict__ x, int64_t ldx, int rows, int K,
                                                             uint8_t* __restrict__ q, int64_t ldq, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t* xr = x + (int64_t)row * ldx;
    float amax = 0.f;
    for (int c = lane * 8; c < K; c += 512) {
        const uint4 u = *reinterpret_cast<const uint4*>(xr + c);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            amax = fmaxf(amax, fabsf(__uint_as_float(w[k] << 16)));
            amax = fmaxf(amax, fabsf(__uint_as_float(w[k] & 0xFFFF0000u)));
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d, 64));
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
    uint8_t* qr = q + (int64_t)row * ldq;
    for (int c = lane * 8; c < K; c += 512) {
        const uint4 u = *reinterpret_cast<const uint4*>(xr + c);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w[0] << 16) * inv, __uint_as_float(w[0] & 0xFFFF0000u) * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w[1] << 16) * inv, __uint_as_float(w[1] & 0xFFFF0000u) * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w[2] << 16) * inv, __uint_as_float(w[2] & 0xFFFF0000u) * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w[3] << 16) * inv, __uint_as_float(w[3] & 0xFFFF0000u) * inv, hi, true);
        *reinterpret_cast<uint2*>(qr + c) = make_uint2((unsigned)lo, (unsigned)hi);
    }
}

hipError_t quant_fp8_rows_launch(const uint16_t* x, int64_t ldx, int rows, int K, uint8_t* q, int64_t ldq, float* scale,
                                 hipStream_t s) {
    if (K % 8 || (ldx & 7) || (ldq & 7) || rows < 0) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, rows, K, q, ldq, scale);
    return hipGetLastError();
}

hipError_t im2col_launch(const float* img, int S, int ps, uint16_t* out, int Kpad, hipStream_t s) {
    const int P = S / ps;
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(im2col_kernel, dim3(blocks_for((int64_t)P * P * Kpad, 256)), dim3(256), 0, s, img, S, ps, out, Kpad);
    return hipGetLastError();
}

hipError_t add_rows_launch(float* x, int64_t ldx, const float* pos, int64_t ldp, int rows, int C, hipStream_t s) {
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(add_rows_kernel, dim3(blocks_for((int64_t)rows * C, 256)), dim3(256), 0, s, x, ldx, pos, ldp, rows, C);
    return hipGetLastError();
}

hipError_t f32_to_bf16_launch(const float* in, uint16_t* out, int64_t n, hipStream_t s) {
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

}  // namespace r3g
