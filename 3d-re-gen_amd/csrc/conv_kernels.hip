// conv_kernels.hip -- HBM-bound kernels of the SD-2.1-class UNet blocks of the texture stage (gfx950).
//
// What diffusers executes as cuDNN conv / GroupNorm / SiLU / GEGLU passes (ResnetBlock2D, Transformer2DModel, Downsample2D),
// here on rows: an activation is f32 or bf16 [H*W][C] (NHWC -- a pixel's channels are contiguous), so that every linear
// layer AND every convolution is the MFMA GEMM of gemm.hip over those rows:
//   * im2col3x3   : bf16 [H][W][C] -> bf16 [Ho*Wo][9 C], column (ky*3 + kx)*C + c, zero padding 1 (or 0 before / 1 after), stride 1 | 2; the 3x3
//                   convolution is then C_out = A . W^T with W re-laid to [C_out][ky][kx][C_in] (r3g/unet.py).  16 bytes per
//                   thread, a wave covers 1 KiB of contiguous output.  (An implicit-GEMM staging path that gathers the nine
//                   shifted rows straight into LDS would save this matrix's round trip through HBM: next step.)
//   * group_norm  : GroupNorm(32 groups) [+ SiLU] f32 [HW][C] -> bf16, in two launches: per-block partial sums per group
//                   (fp64 from the group level on, combined in a FIXED order: no float atomics, the result does not depend
//                   on scheduling), then normalise + affine (+ SiLU).
//   * geglu       : bf16 [rows][2F] -> bf16 [rows][F] = x[:, :F] * gelu_erf(x[:, F:])   (diffusers GEGLU)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "kernels.h"
#include "prof.h"

namespace r3g {
namespace {

__device__ __forceinline__ uint16_t f2bf(float f) {
    const __bf16 h = (__bf16)f;
    return *reinterpret_cast<const uint16_t*>(&h);
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
// torch.nn.functional.gelu (exact), x Phi(x) with erfc by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): the GEMM epilogue's form
__device__ __forceinline__ float gelu_erf(float x) {
    const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.3275911f * 0.7071067811865476f, 1.0f));
    const float e = __builtin_amdgcn_exp2f(x * x * -0.7213475204444817f);
    const float h = t * (0.127414796f + t * (-0.142248368f + t * (0.7107068705f + t * (-0.7265760135f + t * 0.5307027145f)))) * e;
    return x * (x >= 0.f ? 1.0f - h : h);
}
inline int blocks_for(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

__global__ __launch_bounds__(256) void im2col3x3_kernel(const uint16_t* __restrict__ x, int H, int W, int C, int stride, int pad,
                                                        int Ho, int Wo, uint16_t* __restrict__ out) {
    const int c8n = C >> 3;
    const int64_t total = (int64_t)Ho * Wo * 9 * c8n;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    x += (int64_t)blockIdx.y * H * W * C;                    // blockIdx.y = sample (round 4: all samples of a call in one launch)
    out += (int64_t)blockIdx.y * Ho * Wo * 9 * C;
    const int c8 = (int)(i % c8n);
    const int64_t r = i / c8n;
    const int tap = (int)(r % 9);
    const int64_t m = r / 9;
    const int oy = (int)(m / Wo), ox = (int)(m % Wo);
    const int iy = oy * stride + tap / 3 - pad, ix = ox * stride + tap % 3 - pad;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const uint4*>(x + ((int64_t)iy * W + ix) * C + c8 * 8);
    *reinterpret_cast<uint4*>(out + m * (9 * (int64_t)C) + (int64_t)tap * C + c8 * 8) = v;
}

// ---- GroupNorm, pass 1: block b sums rows [b*rpb, (b+1)*rpb) per channel, folds the channels of a group in LDS and writes
// (sum, sum of squares) per group as fp64 to partial[b][g][2]
constexpr int GN_MAX_CPT = 12;   // channels per thread: C <= 3072 (the up blocks normalise cat(hidden, skip): 2560 channels)
// Round 4: all samples of a call in ONE launch (blockIdx.y = sample; x and partial advance by a sample): the six views of the
// multiview UNet used to be 6 x 2 launches of a few microseconds of work each -- GroupNorm was 39 % of the texture step's kernel
// time (profiles/r04_texture_stage.md).
// Round 5: float4 columns instead of scalar channels (thread = (row lane p, column q) with 256 / (C / 4) row lanes when C <= 1024,
// up to three columns per thread above), four rows' loads in flight, and the groups folded by 256 / groups lanes each -- the
// round-4 kernel took 28 us for 64 rows and 43 us for 6 x 4096 rows x 320 channels (31 MB), 10.7 % of the texture step's kernel
// time (profiles/r05_texture_stage.md).  Every sum keeps a fixed order: rows ascending per (p, channel); per group lane j adds
// its items j, j + L, ... (item = channel-major, then row lane), then a fixed xor tree over the L lanes.
// addv (may be null): a vector per sample [sample][add_stride] added to every row of the sample BEFORE the statistics -- the
// resnet's time_emb_proj(silu(temb)) between conv1 and norm2, which used to be a pass of its own over the rows per sample.
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, int rows, int C, int groups, int rpb,
                                                         double* __restrict__ partial, const float* __restrict__ addv,
                                                         int64_t add_stride) {
    __shared__ float s_sum[256 * GN_MAX_CPT], s_sq[256 * GN_MAX_CPT];     // [row lane][C], P * C <= max(1024, C)
    constexpr int NQ = GN_MAX_CPT / 4;
    const int t = threadIdx.x;
    x += (int64_t)blockIdx.y * rows * C;
    partial += (int64_t)blockIdx.y * gridDim.x * groups * 2;
    const int r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
    const int cq = C >> 2;
    const int P = cq >= 256 ? 1 : 256 / cq;          // row lanes
    const int p = cq >= 256 ? 0 : t / cq;
    const int q0 = cq >= 256 ? t : t - p * cq;
    const bool active = p < P;
    float4 a[NQ], sq[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) { a[k] = make_float4(0.f, 0.f, 0.f, 0.f); sq[k] = a[k]; }
    if (addv) addv += (int64_t)blockIdx.y * add_stride;
    float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&av](float4& acc, float4& acc2, float4 v) {
        v.x += av.x; v.y += av.y; v.z += av.z; v.w += av.w;
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        acc2.x += v.x * v.x; acc2.y += v.y * v.y; acc2.z += v.z * v.z; acc2.w += v.w * v.w;
    };
    if (active) {
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int q = q0 + k * 256;
            if (q < cq) {
                const float* col = x + (int64_t)q * 4;
                if (addv) av = *reinterpret_cast<const float4*>(addv + q * 4);
                int r = r0 + p;
                for (; r + 3 * P < r1; r += 4 * P) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(col + (int64_t)(r + u * P) * C);
#pragma unroll
                    for (int u = 0; u < 4; ++u) add(a[k], sq[k], v[u]);
                }
                for (; r < r1; r += P) add(a[k], sq[k], *reinterpret_cast<const float4*>(col + (int64_t)r * C));
            }
        }
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int q = q0 + k * 256;
            if (q < cq) {
                *reinterpret_cast<float4*>(s_sum + (int64_t)p * C + q * 4) = a[k];
                *reinterpret_cast<float4*>(s_sq + (int64_t)p * C + q * 4) = sq[k];
            }
        }
    }
    __syncthreads();
    const int cpg = C / groups;
    int L = 1;                                        // lanes per group: a power of two, L * groups <= 256, L <= 8
    while (L < 8 && 2 * L * groups <= 256) L *= 2;
    const int g = t / L, j = t - g * L;
    double sd = 0.0, ssd = 0.0;
    if (g < groups) {
        const int items = cpg * P;
        for (int it = j; it < items; it += L) {
            const int c = g * cpg + it / P, pp = it - (it / P) * P;
            sd += (double)s_sum[pp * C + c];
            ssd += (double)s_sq[pp * C + c];
        }
    }
    for (int d = L >> 1; d >= 1; d >>= 1) {
        sd += __shfl_xor(sd, d, 64);
        ssd += __shfl_xor(ssd, d, 64);
    }
    if (g < groups && j == 0) {
        partial[((int64_t)blockIdx.x * groups + g) * 2] = sd;
        partial[((int64_t)blockIdx.x * groups + g) * 2 + 1] = ssd;
    }
}

// ---- pass 2: one WAVE per (sample, group) combines the blocks' partials -> (mean, rstd): lane l adds the partials of blocks
// l, l + 64, l + 128, ... in that order, then a fixed shuffle tree (xor 32, 16, ..., 1) adds the lanes -- fp64 and a fixed order, so
// the result is a pure function of the partials whatever the launch geometry.  Round 4 had one THREAD per (sample, group) walk
// all blocks: 256 dependent double additions behind strided loads, 26 us per call for a few hundred additions (6.8 % of the
// texture step's kernel time, profiles/r04_texture_stage.md); rounds 2-3 had every block of pass 3 redo the sum.
__global__ __launch_bounds__(256) void gn_stats_kernel(const double* __restrict__ partial, int nblk, int groups, int rows, int cpg,
                                                       float eps, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), smp = blockIdx.y;
    if (g >= groups) return;
    const double2* p = reinterpret_cast<const double2*>(partial) + (int64_t)smp * nblk * groups + g;
    double s = 0.0, ss = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        const double2 v = p[(int64_t)b * groups];
        s += v.x;
        ss += v.y;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s += __shfl_xor(s, d, 64);
        ss += __shfl_xor(ss, d, 64);
    }
    if (lane) return;
    const double n = (double)rows * cpg;
    const double mean = s / n;
    double var = ss / n - mean * mean;      // biased, as torch.nn.GroupNorm
    if (var < 0.0) var = 0.0;
    stats[((int64_t)smp * groups + g) * 2] = (float)mean;
    stats[((int64_t)smp * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// ---- pass 3: y = ((x - mean) * rstd * gamma + beta) [-> SiLU] -> bf16, 4 channels per thread
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, int rows, int C, int groups,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int do_silu, int rpb,
                                                       uint16_t* __restrict__ y, const float* __restrict__ addv, int64_t add_stride) {
    __shared__ float s_mean[256], s_rstd[256];
    const int t = threadIdx.x;
    const int cpg = C / groups;
    x += (int64_t)blockIdx.y * rows * C;
    y += (int64_t)blockIdx.y * rows * C;
    if (t < groups) {
        s_mean[t] = stats[((int64_t)blockIdx.y * groups + t) * 2];
        s_rstd[t] = stats[((int64_t)blockIdx.y * groups + t) * 2 + 1];
    }
    __syncthreads();
    const int r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
    // thread = (row lane p, float4 column q) as in pass 1: the column's gamma / beta / mean / rstd live in registers, four rows'
    // loads are in flight, no division in the loop (round 5; the arithmetic per element is round 4's, bit for bit)
    const int cq = C >> 2;
    const int P = cq >= 256 ? 1 : 256 / cq;
    const int p = cq >= 256 ? 0 : t / cq;
    const int q0 = cq >= 256 ? t : t - p * cq;
    if (p >= P) return;
    for (int q = q0; q < cq; q += 256) {
        const int c = q * 4;
        const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
        const float4 bt = *reinterpret_cast<const float4*>(beta + c);
        const float gw[4] = {gm.x, gm.y, gm.z, gm.w}, bw[4] = {bt.x, bt.y, bt.z, bt.w};
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
        if (addv) av = *reinterpret_cast<const float4*>(addv + (int64_t)blockIdx.y * add_stride + c);
        float mu[4], rs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = (c + e) / cpg;
            mu[e] = s_mean[g];
            rs[e] = s_rstd[g];
        }
        auto emit = [&](int64_t r, const float4& v) {
            const float in[4] = {v.x + av.x, v.y + av.y, v.z + av.z, v.w + av.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z = (in[e] - mu[e]) * rs[e] * gw[e] + bw[e];
                o[e] = do_silu ? silu(z) : z;
            }
            uint2 pk;
            pk.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
            pk.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
            *reinterpret_cast<uint2*>(y + r * C + c) = pk;
        };
        int r = r0 + p;
        for (; r + 3 * P < r1; r += 4 * P) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + (int64_t)(r + u * P) * C + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) emit(r + u * P, v[u]);
        }
        for (; r < r1; r += P) emit(r, *reinterpret_cast<const float4*>(x + (int64_t)r * C + c));
    }
}

__global__ __launch_bounds__(256) void geglu_kernel(const uint16_t* __restrict__ in, int64_t ldi, uint16_t* __restrict__ out,
                                                    int64_t ldo, int rows, int F) {
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
        o[e] = (uint32_t)f2bf(a0 * gelu_erf(g0)) | ((uint32_t)f2bf(a1 * gelu_erf(g1)) << 16);
    }
    *reinterpret_cast<uint4*>(out + (int64_t)r * ldo + c) = make_uint4(o[0], o[1], o[2], o[3]);
}

// nearest-neighbour 2x upsampling of f32 rows [H][W][C] -> bf16 rows [2H][2W][C] (the operand of Upsample2D's convolution)
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ x, int H, int W, int C, uint16_t* __restrict__ y) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)4 * H * W * c4n;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % c4n) * 4;
    const int64_t m = i / c4n;
    const int oy = (int)(m / (2 * W)), ox = (int)(m % (2 * W));
    const float4 v = *reinterpret_cast<const float4*>(x + ((int64_t)(oy >> 1) * W + (ox >> 1)) * C + c);
    uint2 pk;
    pk.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
    pk.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
    *reinterpret_cast<uint2*>(y + m * C + c) = pk;
}

// softmax over the rows of an fp32 score matrix S [rows][n] (times `scale`) -> bf16 probabilities P [rows][n]: the single-head,
// head-dim-512 attention of the VAE's mid block is two GEMMs around this (attn.hip's flash kernel is built for head dim 64).
// One block per row, three sweeps over a row that stays in L2 (n <= 16 384 floats); exp2 with the scale folded into log2(e).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int64_t lds, uint16_t* __restrict__ P,
                                                           int64_t ldp, int n, float scale_log2) {
    __shared__ float red[4];
    const float* row = S + (int64_t)blockIdx.x * lds;
    uint16_t* out = P + (int64_t)blockIdx.x * ldp;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    float mx = -INFINITY;
    for (int i = t * 4; i < n; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(row + i);
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale_log2;
    __syncthreads();
    float sum = 0.f;
    for (int i = t * 4; i < n; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(row + i);
        sum += __builtin_amdgcn_exp2f(fmaf(v.x, scale_log2, -mx)) + __builtin_amdgcn_exp2f(fmaf(v.y, scale_log2, -mx)) +
               __builtin_amdgcn_exp2f(fmaf(v.z, scale_log2, -mx)) + __builtin_amdgcn_exp2f(fmaf(v.w, scale_log2, -mx));
    }
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int i = t * 4; i < n; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(row + i);
        uint2 pk;
        pk.x = (uint32_t)f2bf(__builtin_amdgcn_exp2f(fmaf(v.x, scale_log2, -mx)) * inv) |
               ((uint32_t)f2bf(__builtin_amdgcn_exp2f(fmaf(v.y, scale_log2, -mx)) * inv) << 16);
        pk.y = (uint32_t)f2bf(__builtin_amdgcn_exp2f(fmaf(v.z, scale_log2, -mx)) * inv) |
               ((uint32_t)f2bf(__builtin_amdgcn_exp2f(fmaf(v.w, scale_log2, -mx)) * inv) << 16);
        *reinterpret_cast<uint2*>(out + i) = pk;
    }
}

// ---- the sampling loop of an InstructPix2Pix-class pipeline (upstream's delighting model), around r3g_unet_forward --------
// UNet input rows: out[p] = (latent[p] / sqrt(sigma^2 + 1) | cond[p])   (scheduler.scale_model_input + torch.cat(dim=1)); cond =
// the image latents (InstructPix2Pix: ic = zc) or the normal- and position-map latents of a view (multiview UNet: ic = 2 zc)
__global__ __launch_bounds__(256) void model_input_kernel(const float* __restrict__ lat, const float* __restrict__ cond, int zc, int ic,
                                                          int64_t n, float inv, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // over pixels x (zc + ic)
    const int oc = zc + ic;
    if (i >= n * oc) return;
    const int64_t p = i / oc;
    const int c = (int)(i - p * oc);
    out[i] = c < zc ? lat[p * zc + c] * inv : cond[p * ic + (c - zc)];
}

// classifier-free guidance: out = uncond + scale (cond - uncond)
__global__ __launch_bounds__(256) void cfg_combine_kernel(const float* __restrict__ u, const float* __restrict__ c, int64_t n, float g,
                                                          float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = u[i] + g * (c[i] - u[i]);
}

// diffusers EulerAncestralDiscreteScheduler.step, in place on the sample: x0 = x - sigma eps (epsilon) or
// v (-sigma / sqrt(sigma^2 + 1)) + x / (sigma^2 + 1) (v_prediction); derivative = (x - x0) / sigma;
// x <- x + derivative (sigma_down - sigma) + noise sigma_up
__global__ __launch_bounds__(256) void euler_ancestral_step_kernel(float* __restrict__ x, const float* __restrict__ m,
                                                                   const float* __restrict__ noise, int64_t n, float sigma, float dt,
                                                                   float sigma_up, int vpred) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xi = x[i], mi = m[i];
    const float s2 = sigma * sigma + 1.0f;
    const float x0 = vpred ? mi * (-sigma / sqrtf(s2)) + xi / s2 : xi - sigma * mi;
    const float d = (xi - x0) / sigma;
    x[i] = xi + d * dt + noise[i] * sigma_up;
}

// diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): out[dim] = [cos(t f_k) | sin(t f_k)],
// f_k = exp(-ln(10000) k / (dim/2))
__global__ void unet_timestep_kernel(float t, int dim, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim >> 1;
    if (k >= half) return;
    const float f = expf(-9.210340371976184f * (float)k / (float)half);
    out[k] = cosf(t * f);
    out[half + k] = sinf(t * f);
}

__global__ void vec_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (a ? a[i] : 0.f) + (b ? b[i] : 0.f);
}

}  // namespace

hipError_t im2col3x3_launch(const uint16_t* x, int H, int W, int C, int stride, int pad, uint16_t* out, hipStream_t s, int nb) {
    if (C % 8 || (stride != 1 && stride != 2) || (pad != 0 && pad != 1) || H < 1 || W < 1 || H + pad < 2 || W + pad < 2)
        return hipErrorInvalidValue;
    // zero padding: `pad` rows / columns before, one after (pad 1: the symmetric padding of the UNet's convolutions; pad 0: the
    // F.pad(x, (0, 1, 0, 1)) in front of the stride-2 convolution of the VAE encoder's Downsample2D)
    const int Ho = (H + pad + 1 - 3) / stride + 1, Wo = (W + pad + 1 - 3) / stride + 1;
    const int64_t total = (int64_t)Ho * Wo * 9 * (C / 8);
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(blocks_for(total, 256), nb < 1 ? 1 : nb), dim3(256), 0, s, x, H, W, C, stride, pad, Ho, Wo, out);
    return hipGetLastError();
}

int group_norm_blocks(int rows) {
    int nblk = (rows + 15) / 16;          // at least 16 rows per block, at most 256 blocks
    if (nblk > 256) nblk = 256;
    if (nblk < 1) nblk = 1;
    return nblk;
}

// nb samples of `rows` rows each (contiguous); partial: workspace of nb * (group_norm_blocks(rows) * groups * 2 doubles + groups floats)
hipError_t group_norm_launch(const float* x, int rows, int C, int groups, const float* gamma, const float* beta, float eps,
                             int do_silu, uint16_t* y, double* partial, hipStream_t s, int nb, const float* addv, int64_t add_stride) {
    if (addv && (add_stride & 3)) return hipErrorInvalidValue;
    if (C % 4 || groups < 1 || groups > 256 || C % groups || C > 256 * GN_MAX_CPT || rows < 1 || nb < 1) return hipErrorInvalidValue;
    const int nblk = group_norm_blocks(rows);
    const int rpb = (rows + nblk - 1) / nblk;
    float* stats = reinterpret_cast<float*>(partial + (int64_t)nb * nblk * groups * 2);
    ProfScope prof_scope_(PC_LAYERNORM, 0.0, s);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk, nb), dim3(256), 0, s, x, rows, C, groups, rpb, partial, addv, add_stride);
    hipLaunchKernelGGL(gn_stats_kernel, dim3((groups + 3) / 4, nb), dim3(256), 0, s, (const double*)partial, nblk, groups, rows, C / groups, eps, stats);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(nblk, nb), dim3(256), 0, s, x, rows, C, groups, (const float*)stats, gamma, beta,
                       do_silu, rpb, y, addv, add_stride);
    return hipGetLastError();
}

hipError_t geglu_launch(const uint16_t* in, int64_t ldi, uint16_t* out, int64_t ldo, int rows, int F, hipStream_t s) {
    if (F % 8 || (ldi & 7) || (ldo & 7)) return hipErrorInvalidValue;
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(geglu_kernel, dim3(blocks_for((int64_t)rows * F / 8, 256)), dim3(256), 0, s, in, ldi, out, ldo, rows, F);
    return hipGetLastError();
}

hipError_t upsample2x_launch(const float* x, int H, int W, int C, uint16_t* y, hipStream_t s) {
    if (C % 4 || H < 1 || W < 1) return hipErrorInvalidValue;
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(blocks_for((int64_t)4 * H * W * (C / 4), 256)), dim3(256), 0, s, x, H, W, C, y);
    return hipGetLastError();
}

hipError_t unet_timestep_launch(float t, int dim, float* out, hipStream_t s) {
    if (dim < 2 || dim % 2) return hipErrorInvalidValue;
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(unet_timestep_kernel, dim3(blocks_for(dim / 2, 256)), dim3(256), 0, s, t, dim, out);
    return hipGetLastError();
}

hipError_t vec_add_launch(const float* a, const float* b, float* out, int n, hipStream_t s) {
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(vec_add_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, a, b, out, n);
    return hipGetLastError();
}

hipError_t softmax_rows_launch(const float* S, int64_t lds, uint16_t* P, int64_t ldp, int rows, int n, float scale, hipStream_t s) {
    if (rows < 1 || n < 4 || n % 4 || (lds & 3) || (ldp & 3)) return hipErrorInvalidValue;
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, s, S, lds, P, ldp, n, scale * 1.4426950408889634f);
    return hipGetLastError();
}

hipError_t model_input_launch(const float* lat, int zc, const float* cond, int ic, int64_t pixels, float sigma, float* out, hipStream_t s) {
    if (zc < 1 || ic < 1 || pixels < 1) return hipErrorInvalidValue;
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    const float inv = 1.0f / sqrtf(sigma * sigma + 1.0f);
    hipLaunchKernelGGL(model_input_kernel, dim3(blocks_for(pixels * (zc + ic), 256)), dim3(256), 0, s, lat, cond, zc, ic, pixels, inv, out);
    return hipGetLastError();
}

hipError_t cfg_combine_launch(const float* uncond, const float* cond, int64_t n, float scale, float* out, hipStream_t s) {
    if (n < 1) return hipErrorInvalidValue;
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, uncond, cond, n, scale, out);
    return hipGetLastError();
}

hipError_t euler_ancestral_step_launch(float* x, const float* model_out, const float* noise, int64_t n, float sigma_from,
                                       float sigma_to, int vpred, hipStream_t s) {
    if (n < 1 || !(sigma_from > 0.0f) || sigma_to < 0.0f || sigma_to > sigma_from) return hipErrorInvalidValue;
    ProfScope prof_scope_(PC_ELEMWISE, 0.0, s);
    // fp32, in diffusers' order of operations
    const float f2 = sigma_from * sigma_from, t2 = sigma_to * sigma_to;
    const float sigma_up = sqrtf(t2 * (f2 - t2) / f2);
    const float sigma_down = sqrtf(t2 - sigma_up * sigma_up);
    hipLaunchKernelGGL(euler_ancestral_step_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, x, model_out, noise, n, sigma_from,
                       sigma_down - sigma_from, sigma_up, vpred);
    return hipGetLastError();
}

}  // namespace r3g
