// unet.cpp -- host-side orchestration of SD-2.1-class UNet blocks on one MI355X (compiled by hipcc).
//
// SURVEY.md 8(f) rank 3, first slice: the building blocks of the two diffusion UNets behind upstream's
// `Hunyuan3DPaintPipeline.__call__` (reference call site src/2d_to_3d_models/run.py:97; built at :126-128) -- diffusers
// ResnetBlock2D, Transformer2DModel (use_linear_projection) with one BasicTransformerBlock (self-attention, cross-attention
// over the text / image context, GEGLU feed-forward), Downsample2D, and their compositions CrossAttnDownBlock2D and
// UNetMidBlock2DCrossAttn.  Weights are registered under diffusers' state-dict names ("down_blocks.0.resnets.0.conv1.weight").
//
// Activations are rows: f32 [H*W][C] (the hidden state; a pixel's channels contiguous), GEMM operands bf16.  Every
// convolution is im2col3x3 (conv_kernels.hip) + the MFMA GEMM of gemm.hip with its fused epilogues (bias, fp32 residual
// add); a 1x1 convolution is the GEMM itself.  Attention is attn.hip's flash kernel (head dim 64: SD 2.x's heads).
// r3g_unet_forward strings them into the whole UNet2DConditionModel.forward of the SD-2.1 layout (conv_in, time embedding,
// down path with skip connections, mid block, up path on cat(hidden, skip) with nearest upsampling, conv_norm_out, conv_out).
// The SD-family VAE (AutoencoderKL encode / decode) runs on the same blocks (r3g_aekl_encode / r3g_aekl_decode, below).
// What is NOT here yet: the text / image encoders, and upstream's multiview / reference attention
// extensions of this UNet -- the texture stage keeps reporting its `texture_source` (stage/run.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/r3g.h"
#include "kernels.h"
#include "prof.h"
#include "r3g_ctx.h"

namespace r3g {

struct UTensor {
    const void* p;
    int dtype;  // 0 f32, 1 bf16
    int64_t rows, cols;
};

struct Unet {
    r3g_unet_config c{};
    std::unordered_map<std::string, UTensor> w;
    char* arena = nullptr;
    // activation buffers (sized for max_hw rows x max_channels)
    float *h = nullptr, *t1 = nullptr;     // transformer-level hidden state, resnet intermediate
    uint16_t *xn = nullptr;                // normalised / cast operand [hw][C]
    uint16_t *col = nullptr;               // im2col matrix [hw][9 C]
    uint16_t *Q = nullptr, *K = nullptr, *Vt = nullptr, *att = nullptr;
    uint16_t *ff = nullptr, *ff2 = nullptr;   // [hw][8 C], [hw][4 C]
    uint16_t *ctxK = nullptr, *ctxVt = nullptr;
    float *vec = nullptr;                  // [4][max_channels] small vectors
    double* gn_partial = nullptr;
    const uint16_t* zeros = nullptr;       // 256 bytes of zeros (the implicit convolution's taps outside the image)
    float* splitws = nullptr;              // split-K workspace of the 3 x 3 convolutions (kSplitWsElems floats, GemmArgs::split_ws)
    // whole-model forward (r3g_unet_forward): the concatenated input of an up-block resnet, two hidden-state buffers, the
    // time embedding, and the stack of skip connections (allocated on first use for the resolution at hand)
    float *catbuf = nullptr, *hb[2] = {nullptr, nullptr}, *emb = nullptr;
    float* skips = nullptr;
    size_t skips_bytes = 0;
    // several samples per call (the views of the multiview UNet): rows are [nb][H*W][C]; per-sample small vectors
    float *vecn = nullptr;                 // [kMaxViews][max_channels]: time_emb_proj(silu(emb_n)) of the resnet at hand
    float *embn = nullptr;                 // [kMaxViews][temb_dim]: time embedding + class embedding of every sample
    float *gate = nullptr;                 // [2][max_channels]: constant vectors mva_scale / ref_scale (GEMM epilogue gates)
    // 2.5D transformer blocks ([UPSTREAM-RECALLED] hunyuanpaint/unet/modules.py): what the reference pass writes per
    // transformer ("condition_embed_dict"), keyed by the transformer's prefix
    struct Cond { uint16_t* p = nullptr; int64_t rows = 0, cols = 0; size_t cap = 0; };
    std::unordered_map<std::string, Cond> cond;
    int nb = 1;                            // samples in the running forward
    int mv_flags = 0;                      // 1: write the condition store | 2: read it (reference attention)
    int groups = 1;                        // 2 (flag 4): the samples are a classifier-free-guidance PAIR of nb / 2 views each
    float mva_scale = 1.0f, ref_scale = 1.0f;
};
constexpr int kMaxViews = 16;
constexpr int64_t kSplitWsElems = 512LL * 128 * 128;   // slices x tiles <= 512 workgroups of 128 x 128 (gemm.hip: splitk128_factor)

#define U_TRY(expr)                                            \
    do {                                                       \
        hipError_t e__ = (expr);                               \
        if (e__ != hipSuccess) return hip_fail(e__, #expr);    \
    } while (0)
#define U_RC(expr)                  \
    do {                            \
        int rc__ = (expr);          \
        if (rc__) return rc__;      \
    } while (0)

static inline int64_t rup(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

struct ULin {
    const uint16_t* w = nullptr;
    const float* b = nullptr;
    int N = 0, K = 0;
};

static int u_lin(const Unet& u, const std::string& base, bool need_bias, int N, int K, ULin* out) {
    auto it = u.w.find(base + ".weight");
    if (it == u.w.end() || it->second.dtype != 1) return fail(R3G_ERR_STATE, "unet: missing bf16 weight '%s.weight'", base.c_str());
    if (it->second.rows != N || it->second.cols != K)
        return fail(R3G_ERR_INVALID, "unet: '%s.weight' is [%lld][%lld], expected [%d][%d]", base.c_str(),
                    (long long)it->second.rows, (long long)it->second.cols, N, K);
    out->w = (const uint16_t*)it->second.p;
    out->N = N; out->K = K;
    auto ib = u.w.find(base + ".bias");
    out->b = nullptr;
    if (ib != u.w.end()) {
        if (ib->second.dtype != 0 || ib->second.rows * ib->second.cols != N)
            return fail(R3G_ERR_INVALID, "unet: '%s.bias' must be f32 [%d]", base.c_str(), N);
        out->b = (const float*)ib->second.p;
    } else if (need_bias) {
        return fail(R3G_ERR_STATE, "unet: missing '%s.bias'", base.c_str());
    }
    return R3G_OK;
}

static int u_vec(const Unet& u, const std::string& name, int n, const float** out) {
    auto it = u.w.find(name);
    if (it == u.w.end() || it->second.dtype != 0) return fail(R3G_ERR_STATE, "unet: missing f32 tensor '%s'", name.c_str());
    if (it->second.rows * it->second.cols != n) return fail(R3G_ERR_INVALID, "unet: '%s' has the wrong size", name.c_str());
    *out = (const float*)it->second.p;
    return R3G_OK;
}

static int u_gemm(const uint16_t* A, int64_t lda, const ULin& l, const float* bias, void* C, int64_t ldc, int M, int epi,
                  hipStream_t s, float* split_ws = nullptr) {
    GemmArgs p{};
    p.A = A; p.lda = lda; p.W = l.w; p.ldw = l.K; p.bias = bias; p.C = C; p.ldc = ldc;
    p.M = M; p.N = l.N; p.K = l.K; p.epi = epi;
    p.split_ws = split_ws; p.split_ws_elems = split_ws ? kSplitWsElems : 0;
    hipError_t e = gemm_launch(p, 1, s);
    if (e != hipSuccess) return hip_fail(e, "gemm_launch(unet)");
    return R3G_OK;
}

static int u_check_shape(const Unet& u, int H, int W, int C, const char* what, int nb = 1) {
    if (H < 1 || W < 1 || nb < 1 || nb > kMaxViews || (int64_t)nb * H * W > u.c.max_hw || C > u.c.max_channels || C % 64 || C % u.c.groups)
        return fail(R3G_ERR_INVALID, "%s: %d x %d x %d does not fit the unet arena (max_hw %d, max_channels %d; C %% 64 == 0)", what,
                    H, W, C, u.c.max_hw, u.c.max_channels);
    return R3G_OK;
}

// conv 3x3 (pad 1) of bf16 rows src [H*W][Cin] -> epilogue(dst [Ho*Wo][Cout])
// pad 1: zero padding 1 all round; pad 0: F.pad(x, (0, 1, 0, 1)) (the VAE encoder's stride-2 Downsample2D)
// nb samples stacked as rows: the im2col runs per sample (a 3x3 window must not reach into the neighbour), the GEMM over all
static int u_conv3x3(Unet& u, const uint16_t* src, int H, int W, int Cin, int stride, const ULin& l, const float* bias, void* dst,
                     int epi, hipStream_t s, int pad = 1, int nb = 1) {
    const int Ho = (H + pad - 2) / stride + 1, Wo = (W + pad - 2) / stride + 1;
    const int M = nb * Ho * Wo;
    // round 5: implicit GEMM -- the 128 x 128 kernel gathers the nine shifted rows itself (conv_gemm_kernel); the im2col matrix is
    // written only for the problems the tile rule gives to the 256 x 256 kernels (N % 256 == 0 on a grid that fills the chip)
    if (gemm_conv_implicit() && (epi == EPI_F32 || epi == EPI_RESID_F32) && Cin % 64 == 0 && l.K == 9 * Cin && !gemm_auto_takes_256(M, l.N, l.K)) {
        GemmArgs p{};
        p.W = l.w; p.ldw = l.K; p.bias = bias; p.C = dst; p.ldc = l.N;
        p.M = M; p.N = l.N; p.K = l.K; p.epi = epi;
        p.split_ws = u.splitws; p.split_ws_elems = kSplitWsElems;
        p.conv.x = src; p.conv.zero = u.zeros; p.conv.H = H; p.conv.W = W; p.conv.Cin = Cin; p.conv.stride = stride; p.conv.pad = pad;
        p.conv.Ho = Ho; p.conv.Wo = Wo;
        hipError_t e = gemm_launch(p, 1, s);
        if (e != hipSuccess) return hip_fail(e, "gemm_launch(unet implicit convolution)");
        return R3G_OK;
    }
    U_TRY(im2col3x3_launch(src, H, W, Cin, stride, pad, u.col, s, nb));       // every sample of the call in one launch
    // few rows over a deep K (the coarse levels): slices of K side by side, summed in a fixed order (gemm.hip: split-K of the 128x128 kernel)
    return u_gemm(u.col, 9 * (int64_t)Cin, l, bias, dst, l.N, M, epi, s, u.splitws);
}

// GroupNorm (+ SiLU) per sample of f32 rows [nb][hw][C] -> bf16
static int u_group_norm(Unet& u, const float* x, int hw, int C, const float* g, const float* b, float eps, int silu, uint16_t* y,
                        hipStream_t s, int nb = 1, const float* addv = nullptr, int64_t add_stride = 0) {
    U_TRY(group_norm_launch(x, hw, C, u.c.groups, g, b, eps, silu, y, u.gn_partial, s, nb, addv, add_stride));     // all samples in one set of launches
    return R3G_OK;
}

// diffusers ResnetBlock2D.forward: out = shortcut(x) + conv2(silu(norm2(conv1(silu(norm1(x))) + time_emb_proj(silu(temb)))))
// (temb == nullptr: the VAE's resnets, which have no time_emb_proj).  nb samples as rows [nb][H*W][C]; temb_stride != 0: every
// sample has its own embedding (temb + n temb_stride), added after conv1 as a row vector; otherwise it rides in conv1's bias.
static int unet_resnet(Unet& u, const std::string& pre, const float* x, int H, int W, int Cin, int Cout, const float* temb,
                       float* out, hipStream_t s, int nb = 1, int64_t temb_stride = 0) {
    U_RC(u_check_shape(u, H, W, Cin, "r3g_unet_resnet", nb));
    U_RC(u_check_shape(u, H, W, Cout, "r3g_unet_resnet", nb));
    const int hw = H * W, rows = nb * hw;
    const float *g1, *b1, *g2, *b2;
    U_RC(u_vec(u, pre + ".norm1.weight", Cin, &g1));
    U_RC(u_vec(u, pre + ".norm1.bias", Cin, &b1));
    U_RC(u_vec(u, pre + ".norm2.weight", Cout, &g2));
    U_RC(u_vec(u, pre + ".norm2.bias", Cout, &b2));
    ULin c1, c2, tp;
    U_RC(u_lin(u, pre + ".conv1", true, Cout, 9 * Cin, &c1));
    U_RC(u_lin(u, pre + ".conv2", true, Cout, 9 * Cout, &c2));
    // per-channel constant of conv1's epilogue: conv1.bias + time_emb_proj(silu(temb))
    const float* cb = c1.b;
    // (per-sample embeddings -- class labels -- take this path for ONE sample too: a guidance pair of one view per half (nb = 2) must
    // equal its two calls (nb = 1) bit for bit, and "bias + projection folded into conv1's epilogue" associates the fp32 sum the other way)
    const bool per_sample = temb && temb_stride != 0;
    if (temb) {
        U_RC(u_lin(u, pre + ".time_emb_proj", true, Cout, u.c.temb_dim, &tp));
        if (per_sample) {
            // every sample's projection: rows of one GEMV launch (8 at a time) where the embeddings lie back to back
            if (temb_stride == u.c.temb_dim) {
                for (int n0 = 0; n0 < nb; n0 += 8)
                    U_TRY(gemv_launch(temb + n0 * temb_stride, std::min(8, nb - n0), u.c.temb_dim, tp.w, tp.K, tp.b,
                                      u.vecn + (int64_t)n0 * Cout, Cout, 1, 0, s));
            } else {
                for (int n = 0; n < nb; ++n)
                    U_TRY(gemv_launch(temb + n * temb_stride, 1, u.c.temb_dim, tp.w, tp.K, tp.b, u.vecn + (int64_t)n * Cout, Cout, 1, 0, s));
            }
        } else {
            float* tb = u.vec;
            float* sum = u.vec + u.c.max_channels;
            U_TRY(gemv_launch(temb, 1, u.c.temb_dim, tp.w, tp.K, tp.b, tb, Cout, 1, 0, s));
            U_TRY(vec_add_launch(c1.b, tb, sum, Cout, s));
            cb = sum;
        }
    }
    U_RC(u_group_norm(u, x, hw, Cin, g1, b1, u.c.resnet_eps, 1, u.xn, s, nb));
    U_RC(u_conv3x3(u, u.xn, H, W, Cin, 1, c1, cb, u.t1, EPI_F32, s, 1, nb));
    // per sample: norm2 reads conv1's rows + the sample's projection (u.vecn [nb][Cout]); the sum is not stored
    U_RC(u_group_norm(u, u.t1, hw, Cout, g2, b2, u.c.resnet_eps, 1, u.xn, s, nb, per_sample ? u.vecn : nullptr, Cout));
    // the residual: x itself, or conv_shortcut (1x1) of x, lands in `out` first; conv2's epilogue adds onto it
    if (Cin != Cout || u.w.count(pre + ".conv_shortcut.weight")) {
        ULin sc;
        U_RC(u_lin(u, pre + ".conv_shortcut", true, Cout, Cin, &sc));
        uint16_t* xb = u.ff;       // bf16 copy of x (free here)
        U_TRY(f32_to_bf16_launch(x, xb, (int64_t)rows * Cin, s));
        U_RC(u_gemm(xb, Cin, sc, sc.b, out, Cout, rows, EPI_F32, s));
    } else if (out != x) {
        U_TRY(hipMemcpyAsync(out, x, (size_t)rows * Cout * 4, hipMemcpyDeviceToDevice, s));
    }
    return u_conv3x3(u, u.xn, H, W, Cout, 1, c2, c2.b, out, EPI_RESID_F32, s, 1, nb);
}

static int u_layernorm(const float* x, uint16_t* y, int rows, int C, const float* w, const float* b, float eps, hipStream_t s) {
    LnArgs p{};
    p.x = x; p.ldx = C; p.y = y; p.ldy = C; p.w = w; p.b = b;
    p.rows = rows; p.C = C; p.rows_per_batch = rows; p.eps = eps;
    hipError_t e = layernorm_launch(p, s);
    if (e != hipSuccess) return hip_fail(e, "layernorm_launch(unet)");
    return R3G_OK;
}

// B independent sequences of Lq queries (Q [B][H][Lq_pad][64]) over Lk keys each; O rows [B][Lq][C]
static int u_attention(Unet& u, int heads, int Lq, int Lq_pad, int Lk, int Lk_pad, const uint16_t* K, const uint16_t* Vt, int C,
                       hipStream_t s, int B = 1) {
    AttnArgs p{};
    p.Q = u.Q; p.K = K; p.Vt = Vt; p.O = u.att; p.ldo = C; p.strideO = (int64_t)Lq * C;
    p.B = B; p.H = heads; p.Lq = Lq; p.Lq_pad = Lq_pad; p.Lk = Lk; p.Lk_pad = Lk_pad;
    p.scale = 0.125f;
    p.q_prescaled = attn_q_scale(0.125f) != 1.0f;
    hipError_t e = attention_launch(p, s);
    if (e != hipSuccess) return hip_fail(e, "attention_launch(unet)");
    return R3G_OK;
}

static GemmArgs u_qkv_args(const uint16_t* A, int64_t lda, const ULin& l, int M, int heads, int layout, uint16_t* Q, uint16_t* K,
                           uint16_t* Vt, int Lq_pad, int Lk_pad) {
    GemmArgs p{};
    p.A = A; p.lda = lda; p.W = l.w; p.ldw = l.K; p.bias = l.b;
    p.M = M; p.N = l.N; p.K = l.K; p.epi = EPI_QKV;
    p.qkv.Q = Q; p.qkv.K = K; p.qkv.Vt = Vt; p.qkv.Lq_pad = Lq_pad; p.qkv.Lk_pad = Lk_pad;
    p.qkv.heads = heads; p.qkv.layout = layout; p.qkv.norm = QKN_NONE; p.qkv.q_scale = attn_q_scale(0.125f);
    return p;
}

// h += gate * (att . W^T + b)   (gate: a constant per-column vector, or null for 1)
static int u_gemm_resid(const uint16_t* A, int64_t lda, const ULin& l, float* C, int64_t ldc, int M, const float* gate, hipStream_t s,
                        float* split_ws = nullptr) {
    GemmArgs p{};
    p.A = A; p.lda = lda; p.W = l.w; p.ldw = l.K; p.bias = l.b; p.C = C; p.ldc = ldc;
    p.M = M; p.N = l.N; p.K = l.K; p.epi = EPI_RESID_F32;
    p.gate = gate; p.strideGate = 0;
    p.split_ws = split_ws; p.split_ws_elems = split_ws ? kSplitWsElems : 0;   // the feed-forward's way back (K = 4 C) at the coarse levels
    hipError_t e = gemm_launch(p, 1, s);
    if (e != hipSuccess) return hip_fail(e, "gemm_launch(unet residual)");
    return R3G_OK;
}

// diffusers Transformer2DModel (use_linear_projection, one BasicTransformerBlock), in place on x f32 [nb][H*W][C].
// Fused projection weights are registered by r3g/unet.py (pure re-layouts): attn1.to_qkv = cat(to_q, to_k, to_v) rows,
// attn2.to_kv = per head (k rows, v rows) of to_k / to_v.
// With nb > 1 samples (the views of one object) and the weights of upstream's Basic2p5DTransformerBlock registered
// ([UPSTREAM-RECALLED] hy3dgen/texgen/hunyuanpaint/unet/modules.py) the block also runs, all on norm1's output:
//   * mv_flags & 1 ("w"): the normalised hidden states of all samples are kept per transformer (condition_embed_dict);
//   * mv_flags & 2 ("r") and "<blk>.attn_refview.*": h += ref_scale * attn_refview(norm_h, K / V from the kept states of the
//     reference pass, registered as "cond:<prefix>");
//   * nb > 1 and "<blk>.attn_multiview.*": h += mva_scale * self-attention over the tokens of ALL views as one sequence.
static int unet_transformer(Unet& u, const std::string& pre, float* x, int H, int W, int C, const uint16_t* ctx, int tokens,
                            hipStream_t s, int nb = 1) {
    U_RC(u_check_shape(u, H, W, C, "r3g_unet_transformer", nb));
    if (tokens < 1 || tokens > u.c.ctx_tokens) return fail(R3G_ERR_INVALID, "r3g_unet_transformer: %d context tokens (max %d)", tokens, u.c.ctx_tokens);
    const int hw = H * W, rows = nb * hw, heads = C / 64, Lp = (int)rup(hw, 128), Lkp = (int)rup(tokens, 64);
    // sample groups (round 5): G = 2 is the guidance pair -- group 0 the conditional evaluation (context rows [0, tokens), reference
    // attention under flag 2), group 1 the unconditional one (context rows [tokens, 2 tokens), no reference attention); the
    // multiview attention's sequence is a GROUP's views.  One launch set instead of two; G = 1 is exactly rounds 3-4.
    const int G = (nb > 1 && u.groups == 2 && nb % 2 == 0) ? 2 : 1;
    const int rows_g = rows / G, Lall = (int)rup(rows_g, 128);
    const std::string blk = pre + ".transformer_blocks.0";
    const float *gw, *gb, *w1, *b1, *w2, *b2, *w3, *b3;
    U_RC(u_vec(u, pre + ".norm.weight", C, &gw));
    U_RC(u_vec(u, pre + ".norm.bias", C, &gb));
    U_RC(u_vec(u, blk + ".norm1.weight", C, &w1));
    U_RC(u_vec(u, blk + ".norm1.bias", C, &b1));
    U_RC(u_vec(u, blk + ".norm2.weight", C, &w2));
    U_RC(u_vec(u, blk + ".norm2.bias", C, &b2));
    U_RC(u_vec(u, blk + ".norm3.weight", C, &w3));
    U_RC(u_vec(u, blk + ".norm3.bias", C, &b3));
    ULin pin, pout, qkv, o1, q2, kv2, o2, f0, f2;
    U_RC(u_lin(u, pre + ".proj_in", true, C, C, &pin));
    U_RC(u_lin(u, pre + ".proj_out", true, C, C, &pout));
    U_RC(u_lin(u, blk + ".attn1.to_qkv", false, 3 * C, C, &qkv));
    U_RC(u_lin(u, blk + ".attn1.to_out.0", true, C, C, &o1));
    U_RC(u_lin(u, blk + ".attn2.to_q", false, C, C, &q2));
    U_RC(u_lin(u, blk + ".attn2.to_kv", false, 2 * C, u.c.ctx_dim, &kv2));
    U_RC(u_lin(u, blk + ".attn2.to_out.0", true, C, C, &o2));
    U_RC(u_lin(u, blk + ".ff.net.0.proj", true, 8 * C, C, &f0));
    U_RC(u_lin(u, blk + ".ff.net.2", true, C, 4 * C, &f2));
    // (per GROUP: a guidance pair of ONE view is two groups of one sample each -- the two-call form runs no multiview attention
    // on a single view (upstream's block: `n > 1`), and the pair must equal it)
    const bool has_mv = nb / G > 1 && u.w.count(blk + ".attn_multiview.to_qkv.weight");
    const auto cond_it = u.w.find("cond:" + pre);
    const bool has_ref = (u.mv_flags & 2) && cond_it != u.w.end() && u.w.count(blk + ".attn_refview.to_q.weight");
    if (nb * (int64_t)Lp > (int64_t)G * Lall + 128 * kMaxViews) return fail(R3G_ERR_INVALID, "r3g_unet_transformer: padded query rows exceed the arena");
    // norm (GroupNorm, no activation) -> proj_in -> hidden state h
    U_RC(u_group_norm(u, x, hw, C, gw, gb, 1e-6f, 0, u.xn, s, nb));
    U_RC(u_gemm(u.xn, C, pin, pin.b, u.h, C, rows, EPI_F32, s));
    // self-attention, every sample on its own: the GEMM's batch dimension puts sample n into attention batch n
    U_RC(u_layernorm(u.h, u.xn, rows, C, w1, b1, 1e-5f, s));
    {
        GemmArgs p = u_qkv_args(u.xn, C, qkv, hw, heads, QKV_KHD, u.Q, u.K, u.Vt, Lp, Lp);
        p.strideA = (int64_t)hw * C;
        hipError_t e = gemm_launch(p, nb, s);
        if (e != hipSuccess) return hip_fail(e, "gemm_launch(unet qkv)");
    }
    U_RC(u_attention(u, heads, hw, Lp, hw, Lp, u.K, u.Vt, C, s, nb));
    U_RC(u_gemm_resid(u.att, C, o1, u.h, C, rows, nullptr, s));
    if (u.mv_flags & 1) {          // "w": keep norm1's output of all samples, tokens (n l)
        Unet::Cond& c = u.cond[pre];
        const size_t need = (size_t)rows * C * 2;
        if (need > c.cap) {
            if (c.p) { U_TRY(hipStreamSynchronize(s)); (void)hipFree(c.p); c.p = nullptr; c.cap = 0; }
            U_TRY(hipMalloc((void**)&c.p, need));
            c.cap = need;
        }
        c.rows = rows; c.cols = C;
        U_TRY(hipMemcpyAsync(c.p, u.xn, need, hipMemcpyDeviceToDevice, s));
    }
    if (has_ref) {                 // reference attention: every view's tokens query the reference pass's tokens
        const UTensor& ct = cond_it->second;
        if (ct.dtype != 1 || ct.cols != C || ct.rows < 1 || ct.rows > u.c.ctx_tokens)
            return fail(R3G_ERR_INVALID, "r3g_unet_transformer: 'cond:%s' must be bf16 [tokens <= ctx_tokens %d][%d]", pre.c_str(), u.c.ctx_tokens, C);
        const int rt = (int)ct.rows, Lrp = (int)rup(rt, 64);
        ULin rq, rkv, ro;
        U_RC(u_lin(u, blk + ".attn_refview.to_q", false, C, C, &rq));
        U_RC(u_lin(u, blk + ".attn_refview.to_kv", false, 2 * C, C, &rkv));
        U_RC(u_lin(u, blk + ".attn_refview.to_out.0", true, C, C, &ro));
        // (the guidance pair: only group 0's rows -- the first rows_g -- take the reference attention)
        const GemmArgs pq = u_qkv_args(u.xn, C, rq, rows_g, heads, QKV_Q_ONLY, u.Q, nullptr, nullptr, Lall, 0);
        hipError_t e = gemm_launch(pq, 1, s);
        if (e != hipSuccess) return hip_fail(e, "gemm_launch(unet refview q)");
        const GemmArgs pk = u_qkv_args((const uint16_t*)ct.p, C, rkv, rt, heads, QKV_HEAD_KV, nullptr, u.ctxK, u.ctxVt, 0, Lrp);
        e = gemm_launch(pk, 1, s);
        if (e != hipSuccess) return hip_fail(e, "gemm_launch(unet refview kv)");
        U_RC(u_attention(u, heads, rows_g, Lall, rt, Lrp, u.ctxK, u.ctxVt, C, s));
        U_RC(u_gemm_resid(u.att, C, ro, u.h, C, rows_g, u.ref_scale != 1.0f ? u.gate + u.c.max_channels : nullptr, s));
    }
    if (has_mv) {                  // multiview attention: the tokens of all views form one sequence
        ULin mq, mo;
        U_RC(u_lin(u, blk + ".attn_multiview.to_qkv", false, 3 * C, C, &mq));
        U_RC(u_lin(u, blk + ".attn_multiview.to_out.0", true, C, C, &mo));
        GemmArgs p = u_qkv_args(u.xn, C, mq, rows_g, heads, QKV_KHD, u.Q, u.K, u.Vt, Lall, Lall);
        p.strideA = (int64_t)rows_g * C;           // a group per GEMM batch = per attention batch
        hipError_t e = gemm_launch(p, G, s);
        if (e != hipSuccess) return hip_fail(e, "gemm_launch(unet multiview qkv)");
        U_RC(u_attention(u, heads, rows_g, Lall, rows_g, Lall, u.K, u.Vt, C, s, G));
        U_RC(u_gemm_resid(u.att, C, mo, u.h, C, rows, u.mva_scale != 1.0f ? u.gate : nullptr, s));
    }
    // cross-attention over the context tokens (one context per group: a group's rows are one batch of queries over its context)
    U_RC(u_layernorm(u.h, u.xn, rows, C, w2, b2, 1e-5f, s));
    {
        GemmArgs pq = u_qkv_args(u.xn, C, q2, rows_g, heads, QKV_Q_ONLY, u.Q, nullptr, nullptr, Lall, 0);
        pq.strideA = (int64_t)rows_g * C;
        hipError_t e = gemm_launch(pq, G, s);
        if (e != hipSuccess) return hip_fail(e, "gemm_launch(unet q)");
        GemmArgs pk = u_qkv_args(ctx, u.c.ctx_dim, kv2, tokens, heads, QKV_HEAD_KV, nullptr, u.ctxK, u.ctxVt, 0, Lkp);
        pk.strideA = (int64_t)tokens * u.c.ctx_dim;
        e = gemm_launch(pk, G, s);
        if (e != hipSuccess) return hip_fail(e, "gemm_launch(unet kv)");
    }
    U_RC(u_attention(u, heads, rows_g, Lall, tokens, Lkp, u.ctxK, u.ctxVt, C, s, G));
    U_RC(u_gemm_resid(u.att, C, o2, u.h, C, rows, nullptr, s));
    // GEGLU feed-forward
    U_RC(u_layernorm(u.h, u.xn, rows, C, w3, b3, 1e-5f, s));
    U_RC(u_gemm(u.xn, C, f0, f0.b, u.ff, 8 * (int64_t)C, rows, EPI_BF16, s));
    U_TRY(geglu_launch(u.ff, 8 * (int64_t)C, u.ff2, 4 * (int64_t)C, rows, 4 * C, s));
    U_RC(u_gemm_resid(u.ff2, 4 * (int64_t)C, f2, u.h, C, rows, nullptr, s, u.splitws));
    // proj_out + the block's input
    U_TRY(f32_to_bf16_launch(u.h, u.xn, (int64_t)rows * C, s));
    return u_gemm_resid(u.xn, C, pout, x, C, rows, nullptr, s);
}

// diffusers Downsample2D (conv 3x3, stride 2, padding 1)
static int unet_downsample(Unet& u, const std::string& pre, const float* x, int H, int W, int C, float* out, hipStream_t s, int nb = 1) {
    U_RC(u_check_shape(u, H, W, C, "r3g_unet_downsample", nb));
    ULin cv;
    U_RC(u_lin(u, pre + ".conv", true, C, 9 * C, &cv));
    U_TRY(f32_to_bf16_launch(x, u.xn, (int64_t)nb * H * W * C, s));
    return u_conv3x3(u, u.xn, H, W, C, 2, cv, cv.b, out, EPI_F32, s, 1, nb);
}

// diffusers Upsample2D: nearest 2x, then conv 3x3 (padding 1)
static int unet_upsample(Unet& u, const std::string& pre, const float* x, int H, int W, int C, float* out, hipStream_t s, int nb = 1) {
    U_RC(u_check_shape(u, 2 * H, 2 * W, C, "r3g_unet_upsample", nb));
    ULin cv;
    U_RC(u_lin(u, pre + ".conv", true, C, 9 * C, &cv));
    for (int n = 0; n < nb; ++n)
        U_TRY(upsample2x_launch(x + (int64_t)n * H * W * C, H, W, C, u.xn + (int64_t)n * 4 * H * W * C, s));
    return u_conv3x3(u, u.xn, 2 * H, 2 * W, C, 1, cv, cv.b, out, EPI_F32, s, 1, nb);
}

// diffusers UNet2DConditionModel.forward on the SD-2.1 layout: conv_in, time embedding, CrossAttnDownBlock2D x (n-1) +
// DownBlock2D, UNetMidBlock2DCrossAttn, UpBlock2D + CrossAttnUpBlock2D x (n-1) (every resnet of the up path takes
// cat(hidden, skip)), conv_norm_out + SiLU + conv_out.  sample f32 [nb][H*W][in_channels] -> out f32 [nb][H*W][out_channels].
// labels (host, [nb], or null): class_labels -- emb_n = time_embedding(t) + class_embedding.weight[labels[n]] (diffusers
// class_embed_type None with an nn.Embedding put in its place: the camera embedding of upstream's multiview UNet).
static int unet_forward(Unet& u, const float* sample, int H, int W, float timestep, const uint16_t* ctx, int tokens, float* out,
                        hipStream_t s, int nb = 1, const int32_t* labels = nullptr) {
    const r3g_unet_config& c = u.c;
    const int n = c.n_levels, L = c.layers_per_block;
    if (n < 1 || n > 4 || L < 1) return fail(R3G_ERR_STATE, "r3g_unet_forward: the configuration has no block structure");
    if ((H % (1 << (n - 1))) || (W % (1 << (n - 1)))) return fail(R3G_ERR_INVALID, "r3g_unet_forward: %d x %d is not divisible by %d", H, W, 1 << (n - 1));
    const int* ch = c.block_out_channels;
    const int c0 = ch[0];
    U_RC(u_check_shape(u, H, W, c0, "r3g_unet_forward", nb));
    u.nb = nb;
    // skip stack: conv_in output + every state of the down path
    struct Skip { float* p; int c, h, w; };
    std::vector<Skip> stack;
    {
        size_t need = 0;
        int h = H, w = W;
        need += (size_t)h * w * c0;
        for (int i = 0; i < n; ++i) {
            need += (size_t)L * h * w * ch[i];
            if (i < n - 1) { h /= 2; w /= 2; need += (size_t)h * w * ch[i]; }
        }
        need *= 4 * (size_t)nb;
        if (need > u.skips_bytes) {
            if (u.skips) { U_TRY(hipStreamSynchronize(s)); (void)hipFree(u.skips); u.skips = nullptr; u.skips_bytes = 0; }
            U_TRY(hipMalloc((void**)&u.skips, need));
            u.skips_bytes = need;
        }
    }
    float* next_slot = u.skips;
    auto push = [&](int cc, int h, int w) { Skip k{next_slot, cc, h, w}; next_slot += (size_t)nb * h * w * cc; stack.push_back(k); return k.p; };
    // time embedding: Timesteps(c0) -> linear_1 -> SiLU -> linear_2
    ULin l1, l2;
    U_RC(u_lin(u, "time_embedding.linear_1", true, c.temb_dim, c0, &l1));
    U_RC(u_lin(u, "time_embedding.linear_2", true, c.temb_dim, c.temb_dim, &l2));
    float* tsin = u.emb + c.temb_dim;   // [c0] sinusoidal features; linear_1's output goes through u.t1 (free here)
    U_TRY(unet_timestep_launch(timestep, c0, tsin, s));
    U_TRY(gemv_launch(tsin, 1, c0, l1.w, l1.K, l1.b, u.t1 /* scratch */, c.temb_dim, 0, 1, s));
    U_TRY(gemv_launch(u.t1, 1, c.temb_dim, l2.w, l2.K, l2.b, u.emb, c.temb_dim, 0, 0, s));
    const float* emb = u.emb;
    int64_t emb_stride = 0;
    if (labels) {
        auto it = u.w.find("class_embedding.weight");
        if (it == u.w.end() || it->second.dtype != 0 || it->second.cols != c.temb_dim)
            return fail(R3G_ERR_STATE, "r3g_unet_forward: class labels need 'class_embedding.weight' as f32 [classes][%d]", c.temb_dim);
        for (int k = 0; k < nb; ++k) {
            if (labels[k] < 0 || labels[k] >= it->second.rows) return fail(R3G_ERR_INVALID, "r3g_unet_forward: class label %d out of range", labels[k]);
            U_TRY(vec_add_launch(u.emb, (const float*)it->second.p + (int64_t)labels[k] * c.temb_dim, u.embn + (int64_t)k * c.temb_dim,
                                 c.temb_dim, s));
        }
        emb = u.embn;
        emb_stride = c.temb_dim;
    }
    // conv_in (input channels zero-padded to 64 in the operand and in the re-laid weight)
    {
        ULin ci;
        U_RC(u_lin(u, "conv_in", true, c0, 9 * 64, &ci));
        if (c.in_channels < 1 || c.in_channels > 64) return fail(R3G_ERR_INVALID, "r3g_unet_forward: in_channels must be in [1, 64]");
        U_TRY(cast_pad_launch(sample, c.in_channels, u.xn, 64, nb * H * W, c.in_channels, 64, 1.0f, s));
        float* x0 = push(c0, H, W);
        U_RC(u_conv3x3(u, u.xn, H, W, 64, 1, ci, ci.b, x0, EPI_F32, s, 1, nb));
    }
    const float* cur = stack.back().p;
    int h = H, w = W, cc = c0;
    for (int i = 0; i < n; ++i) {
        const bool last = i == n - 1;
        const std::string pre = "down_blocks." + std::to_string(i);
        for (int j = 0; j < L; ++j) {
            float* dst = push(ch[i], h, w);
            U_RC(unet_resnet(u, pre + ".resnets." + std::to_string(j), cur, h, w, cc, ch[i], emb, dst, s, nb, emb_stride));
            if (!last) U_RC(unet_transformer(u, pre + ".attentions." + std::to_string(j), dst, h, w, ch[i], ctx, tokens, s, nb));
            cur = dst;
            cc = ch[i];
        }
        if (!last) {
            float* dst = push(cc, h / 2, w / 2);
            U_RC(unet_downsample(u, pre + ".downsamplers.0", cur, h, w, cc, dst, s, nb));
            cur = dst;
            h /= 2; w /= 2;
        }
    }
    // mid block
    int side = 0;
    U_RC(unet_resnet(u, "mid_block.resnets.0", cur, h, w, cc, cc, emb, u.hb[side], s, nb, emb_stride));
    U_RC(unet_transformer(u, "mid_block.attentions.0", u.hb[side], h, w, cc, ctx, tokens, s, nb));
    U_RC(unet_resnet(u, "mid_block.resnets.1", u.hb[side], h, w, cc, cc, emb, u.hb[side], s, nb, emb_stride));
    cur = u.hb[side];
    // up path
    for (int i = 0; i < n; ++i) {
        const int cout = ch[n - 1 - i];
        const std::string pre = "up_blocks." + std::to_string(i);
        for (int j = 0; j <= L; ++j) {
            if (stack.empty()) return fail(R3G_ERR_STATE, "r3g_unet_forward: skip stack underflow");
            const Skip sk = stack.back();
            stack.pop_back();
            if (sk.h != h || sk.w != w) return fail(R3G_ERR_STATE, "r3g_unet_forward: skip resolution mismatch");
            const int cin = cc + sk.c;
            if (cin > c.max_channels) return fail(R3G_ERR_INVALID, "r3g_unet_forward: %d concatenated channels exceed max_channels", cin);
            const size_t hwn = (size_t)nb * h * w;
            U_TRY(hipMemcpy2DAsync(u.catbuf, (size_t)cin * 4, cur, (size_t)cc * 4, (size_t)cc * 4, hwn, hipMemcpyDeviceToDevice, s));
            U_TRY(hipMemcpy2DAsync(u.catbuf + cc, (size_t)cin * 4, sk.p, (size_t)sk.c * 4, (size_t)sk.c * 4, hwn, hipMemcpyDeviceToDevice, s));
            side ^= 1;
            U_RC(unet_resnet(u, pre + ".resnets." + std::to_string(j), u.catbuf, h, w, cin, cout, emb, u.hb[side], s, nb, emb_stride));
            if (i > 0) U_RC(unet_transformer(u, pre + ".attentions." + std::to_string(j), u.hb[side], h, w, cout, ctx, tokens, s, nb));
            cur = u.hb[side];
            cc = cout;
        }
        if (i < n - 1) {
            side ^= 1;
            U_RC(unet_upsample(u, pre + ".upsamplers.0", cur, h, w, cc, u.hb[side], s, nb));
            cur = u.hb[side];
            h *= 2; w *= 2;
        }
    }
    // conv_norm_out + SiLU + conv_out
    const float *gw, *gb;
    U_RC(u_vec(u, "conv_norm_out.weight", c0, &gw));
    U_RC(u_vec(u, "conv_norm_out.bias", c0, &gb));
    ULin co;
    U_RC(u_lin(u, "conv_out", true, c.out_channels, 9 * c0, &co));
    U_RC(u_group_norm(u, cur, H * W, c0, gw, gb, 1e-5f, 1, u.xn, s, nb));
    return u_conv3x3(u, u.xn, H, W, c0, 1, co, co.b, out, EPI_F32, s, 1, nb);
}

// ---------------------------------------------------------------------------------------------------------------------
// AutoencoderKL of the SD family (diffusers Encoder / Decoder: the same ResnetBlock2D without a time embedding, one single-head
// attention of head dim = channels in the mid block, Downsample2D with the one-sided padding, Upsample2D), on the same arena and
// the same weight table.  Weights under diffusers' names: "encoder.*", "quant_conv.*", "post_quant_conv.*", "decoder.*".

// diffusers Attention(heads 1, residual_connection, group_norm) of UNetMidBlock2D, in place on x f32 [hw][C]:
// x += to_out(softmax(q k^T / sqrt(C)) v), q / k / v = linear(GroupNorm(x)).  Head dim C (512) is outside attn.hip's flash
// kernel (head dim 64), and there is one such layer per VAE pass: two GEMMs around a row softmax.  V is produced transposed
// (V^T = W_v . xn^T: the weight is the GEMM's A operand) so that P . V is the "A . W^T" the GEMM computes; v's bias moves
// behind the softmax (rows of P sum to one: P (V + 1 b^T) = P V + b^T) and becomes the bias of that GEMM.
static int vae_attention(Unet& u, const std::string& pre, float* x, int H, int W, int C, hipStream_t s) {
    U_RC(u_check_shape(u, H, W, C, "vae attention"));
    const int hw = H * W;
    if (hw % 64) return fail(R3G_ERR_INVALID, "vae attention: %d x %d pixels is not a multiple of 64", H, W);
    const float *gw, *gb;
    U_RC(u_vec(u, pre + ".group_norm.weight", C, &gw));
    U_RC(u_vec(u, pre + ".group_norm.bias", C, &gb));
    ULin q, k, v, o;
    U_RC(u_lin(u, pre + ".to_q", true, C, C, &q));
    U_RC(u_lin(u, pre + ".to_k", true, C, C, &k));
    U_RC(u_lin(u, pre + ".to_v", true, C, C, &v));
    U_RC(u_lin(u, pre + ".to_out.0", true, C, C, &o));
    // scores fp32 [hw][hw] + probabilities bf16 [hw][hw] (64 + 32 MiB at 64 x 64 latents) live in the lazily sized side buffer
    const size_t need = (size_t)hw * hw * 6;
    if (need > u.skips_bytes) {
        if (u.skips) { U_TRY(hipStreamSynchronize(s)); (void)hipFree(u.skips); u.skips = nullptr; u.skips_bytes = 0; }
        U_TRY(hipMalloc((void**)&u.skips, need));
        u.skips_bytes = need;
    }
    float* S = u.skips;
    uint16_t* P = reinterpret_cast<uint16_t*>(u.skips + (size_t)hw * hw);
    U_TRY(group_norm_launch(x, hw, C, u.c.groups, gw, gb, u.c.resnet_eps, 0, u.xn, u.gn_partial, s));
    U_RC(u_gemm(u.xn, C, q, q.b, u.Q, C, hw, EPI_BF16, s));
    U_RC(u_gemm(u.xn, C, k, k.b, u.K, C, hw, EPI_BF16, s));
    {
        ULin xt;                       // "weight" = the normalised activations [hw][C]: V^T [C][hw] = W_v . xn^T
        xt.w = u.xn; xt.N = hw; xt.K = C;
        U_RC(u_gemm(v.w, C, xt, nullptr, u.Vt, hw, C, EPI_BF16, s));
    }
    {
        ULin kt;                       // S = Q . K^T
        kt.w = u.K; kt.N = hw; kt.K = C;
        U_RC(u_gemm(u.Q, C, kt, nullptr, S, hw, hw, EPI_F32, s));
    }
    U_TRY(softmax_rows_launch(S, hw, P, hw, hw, hw, 1.0f / sqrtf((float)C), s));
    {
        ULin vt;                       // O = P . V (+ b_v)
        vt.w = u.Vt; vt.N = C; vt.K = hw;
        U_RC(u_gemm(P, hw, vt, v.b, u.att, C, hw, EPI_BF16, s));
    }
    return u_gemm(u.att, C, o, o.b, x, C, hw, EPI_RESID_F32, s);
}

// diffusers UNetMidBlock2D of the VAE: resnet, attention, resnet -- in place on x
static int vae_mid(Unet& u, const std::string& pre, float* x, int H, int W, int C, hipStream_t s) {
    U_RC(unet_resnet(u, pre + ".resnets.0", x, H, W, C, C, nullptr, x, s));
    U_RC(vae_attention(u, pre + ".attentions.0", x, H, W, C, s));
    return unet_resnet(u, pre + ".resnets.1", x, H, W, C, C, nullptr, x, s);
}

static int vae_levels(const Unet& u, const char* fn) {
    const r3g_unet_config& c = u.c;
    if (c.n_levels < 1 || c.n_levels > 4 || c.layers_per_block < 1)
        return fail(R3G_ERR_STATE, "%s: the configuration has no block structure", fn);
    if (c.in_channels < 1 || c.in_channels > 32 || c.out_channels < 1 || c.out_channels > 64)
        return fail(R3G_ERR_INVALID, "%s: latent channels must be in [1, 32], image channels in [1, 64]", fn);
    return R3G_OK;
}

// AutoencoderKL.decode: z f32 [h*w][latent] -> post_quant_conv -> Decoder.forward -> image f32 [(8h)(8w)][rup(image channels, 4)]
// (for n_levels = 4).  config: block_out_channels = the ENCODER's order (128, 256, 512, 512), in_channels = latent channels,
// out_channels = image channels.
static int vae_decode(Unet& u, const float* z, int h, int w, float* out, hipStream_t s) {
    U_RC(vae_levels(u, "r3g_aekl_decode"));
    const r3g_unet_config& c = u.c;
    const int n = c.n_levels, L = c.layers_per_block, zc = c.in_channels;
    const int* ch = c.block_out_channels;
    const int ctop = ch[n - 1];
    U_RC(u_check_shape(u, h << (n - 1), w << (n - 1), ch[0], "r3g_aekl_decode"));
    U_RC(u_check_shape(u, h, w, ctop, "r3g_aekl_decode"));
    // post_quant_conv (1x1, latent -> latent; K zero-padded to 64 in the operand and in the re-laid weight)
    ULin pq, ci;
    U_RC(u_lin(u, "post_quant_conv", true, rup(zc, 4), 64, &pq));
    U_RC(u_lin(u, "decoder.conv_in", true, ctop, 9 * 64, &ci));
    const int zp = (int)rup(zc, 4);
    U_TRY(cast_pad_launch(z, zc, u.xn, 64, h * w, zc, 64, 1.0f, s));
    U_RC(u_gemm(u.xn, 64, pq, pq.b, u.t1, zp, h * w, EPI_F32, s));
    U_TRY(cast_pad_launch(u.t1, zp, u.xn, 64, h * w, zc, 64, 1.0f, s));
    int side = 0;
    U_RC(u_conv3x3(u, u.xn, h, w, 64, 1, ci, ci.b, u.hb[side], EPI_F32, s));
    U_RC(vae_mid(u, "decoder.mid_block", u.hb[side], h, w, ctop, s));
    int hh = h, ww = w, cc = ctop;
    for (int i = 0; i < n; ++i) {
        const int cout = ch[n - 1 - i];
        const std::string pre = "decoder.up_blocks." + std::to_string(i);
        for (int j = 0; j <= L; ++j) {
            U_RC(unet_resnet(u, pre + ".resnets." + std::to_string(j), u.hb[side], hh, ww, cc, cout, nullptr, u.hb[side ^ 1], s));
            side ^= 1;
            cc = cout;
        }
        if (i < n - 1) {
            U_RC(unet_upsample(u, pre + ".upsamplers.0", u.hb[side], hh, ww, cc, u.hb[side ^ 1], s));
            side ^= 1;
            hh *= 2; ww *= 2;
        }
    }
    const float *gw, *gb;
    U_RC(u_vec(u, "decoder.conv_norm_out.weight", cc, &gw));
    U_RC(u_vec(u, "decoder.conv_norm_out.bias", cc, &gb));
    ULin co;
    U_RC(u_lin(u, "decoder.conv_out", true, (int)rup(c.out_channels, 4), 9 * cc, &co));
    U_TRY(group_norm_launch(u.hb[side], hh * ww, cc, c.groups, gw, gb, c.resnet_eps, 1, u.xn, u.gn_partial, s));
    return u_conv3x3(u, u.xn, hh, ww, cc, 1, co, co.b, out, EPI_F32, s);
}

// AutoencoderKL.encode up to the moments: image f32 [H*W][image channels] -> Encoder.forward -> quant_conv -> f32 [(H/8)(W/8)][2 latent]
// (mean | log-variance; the latent distribution's mode is the first half)
static int vae_encode(Unet& u, const float* img, int H, int W, float* out, hipStream_t s) {
    U_RC(vae_levels(u, "r3g_aekl_encode"));
    const r3g_unet_config& c = u.c;
    const int n = c.n_levels, L = c.layers_per_block, zc2 = 2 * c.in_channels;
    const int* ch = c.block_out_channels;
    if ((H % (1 << (n - 1))) || (W % (1 << (n - 1)))) return fail(R3G_ERR_INVALID, "r3g_aekl_encode: %d x %d is not divisible by %d", H, W, 1 << (n - 1));
    U_RC(u_check_shape(u, H, W, ch[0], "r3g_aekl_encode"));
    ULin ci;
    U_RC(u_lin(u, "encoder.conv_in", true, ch[0], 9 * 64, &ci));
    U_TRY(cast_pad_launch(img, c.out_channels, u.xn, 64, H * W, c.out_channels, 64, 1.0f, s));
    int side = 0;
    U_RC(u_conv3x3(u, u.xn, H, W, 64, 1, ci, ci.b, u.hb[side], EPI_F32, s));
    int hh = H, ww = W, cc = ch[0];
    for (int i = 0; i < n; ++i) {
        const std::string pre = "encoder.down_blocks." + std::to_string(i);
        for (int j = 0; j < L; ++j) {
            U_RC(unet_resnet(u, pre + ".resnets." + std::to_string(j), u.hb[side], hh, ww, cc, ch[i], nullptr, u.hb[side ^ 1], s));
            side ^= 1;
            cc = ch[i];
        }
        if (i < n - 1) {     // Downsample2D(padding 0): F.pad(x, (0, 1, 0, 1)), conv 3x3 stride 2
            ULin cv;
            U_RC(u_lin(u, pre + ".downsamplers.0.conv", true, cc, 9 * cc, &cv));
            U_TRY(f32_to_bf16_launch(u.hb[side], u.xn, (int64_t)hh * ww * cc, s));
            U_RC(u_conv3x3(u, u.xn, hh, ww, cc, 2, cv, cv.b, u.hb[side ^ 1], EPI_F32, s, 0));
            side ^= 1;
            hh /= 2; ww /= 2;
        }
    }
    U_RC(vae_mid(u, "encoder.mid_block", u.hb[side], hh, ww, cc, s));
    const float *gw, *gb;
    U_RC(u_vec(u, "encoder.conv_norm_out.weight", cc, &gw));
    U_RC(u_vec(u, "encoder.conv_norm_out.bias", cc, &gb));
    ULin co, qc;
    const int zp = (int)rup(zc2, 4);
    U_RC(u_lin(u, "encoder.conv_out", true, zp, 9 * cc, &co));
    U_RC(u_lin(u, "quant_conv", true, zp, 64, &qc));
    U_TRY(group_norm_launch(u.hb[side], hh * ww, cc, c.groups, gw, gb, c.resnet_eps, 1, u.xn, u.gn_partial, s));
    U_RC(u_conv3x3(u, u.xn, hh, ww, cc, 1, co, co.b, u.t1, EPI_F32, s));
    U_TRY(cast_pad_launch(u.t1, zp, u.xn, 64, hh * ww, zc2, 64, 1.0f, s));
    return u_gemm(u.xn, 64, qc, qc.b, out, zp, hh * ww, EPI_F32, s);
}

static void unet_free(Unet* u) {
    if (!u) return;
    if (u->arena) (void)hipFree(u->arena);
    if (u->skips) (void)hipFree(u->skips);
    for (auto& kv : u->cond)
        if (kv.second.p) (void)hipFree(kv.second.p);
    delete u;
}

static int unet_create(Ctx* ctx, const r3g_unet_config* cfg) {
    if (ctx->unet) { unet_free((Unet*)ctx->unet); ctx->unet = nullptr; }
    const r3g_unet_config& c = *cfg;
    if (c.max_hw < 1 || c.max_channels < 64 || c.max_channels % 64 || c.max_channels > 3072 || c.temb_dim < 1 || c.temb_dim % 8 ||
        c.ctx_dim % 64 || c.ctx_tokens < 1 || c.groups < 1 || c.groups > 256)
        return fail(R3G_ERR_INVALID, "r3g_unet_create: bad configuration");
    Unet* u = new Unet();
    u->c = c;
    const int64_t hw = c.max_hw, hwp = rup(hw, 128) + 128 * kMaxViews, C = c.max_channels, heads = C / 64, ckp = rup(c.ctx_tokens, 64);
    size_t off = 0;
    auto carve = [&](int64_t bytes) { size_t o = off; off += (size_t)rup(bytes, 256); return o; };
    const size_t o_h = carve(hw * C * 4), o_t1 = carve(hw * C * 4), o_xn = carve(hw * C * 2), o_col = carve(hw * 9 * C * 2),
                 o_q = carve(heads * hwp * 64 * 2), o_k = carve(heads * hwp * 64 * 2), o_v = carve(heads * hwp * 64 * 2),
                 o_att = carve(hw * C * 2), o_ff = carve(hw * 8 * C * 2), o_ff2 = carve(hw * 4 * C * 2),
                 o_ck = carve(2 * heads * ckp * 64 * 2), o_cv = carve(2 * heads * ckp * 64 * 2),   /* two contexts: the guidance pair */ o_vec = carve(4 * C * 4),
                 o_gn = carve((int64_t)kMaxViews * (256LL * 256 * 2 * 8 + 256 * 2 * 4)), o_cat = carve(hw * C * 4), o_hb0 = carve(hw * C * 4), o_hb1 = carve(hw * C * 4),
                 o_emb = carve((int64_t)(c.temb_dim + 2 * C) * 4), o_vecn = carve((int64_t)kMaxViews * C * 4),
                 o_embn = carve((int64_t)kMaxViews * c.temb_dim * 4), o_gate = carve(2 * C * 4), o_split = carve(kSplitWsElems * 4), o_zero = carve(256);
    hipError_t e = hipMalloc((void**)&u->arena, off);
    if (e != hipSuccess) { unet_free(u); return hip_fail(e, "hipMalloc(unet arena)"); }
    e = hipMemset(u->arena, 0, off);      // padded rows must start finite
    if (e != hipSuccess) { unet_free(u); return hip_fail(e, "hipMemset(unet arena)"); }
    char* a = u->arena;
    u->h = (float*)(a + o_h); u->t1 = (float*)(a + o_t1); u->xn = (uint16_t*)(a + o_xn); u->col = (uint16_t*)(a + o_col);
    u->Q = (uint16_t*)(a + o_q); u->K = (uint16_t*)(a + o_k); u->Vt = (uint16_t*)(a + o_v); u->att = (uint16_t*)(a + o_att);
    u->ff = (uint16_t*)(a + o_ff); u->ff2 = (uint16_t*)(a + o_ff2); u->ctxK = (uint16_t*)(a + o_ck); u->ctxVt = (uint16_t*)(a + o_cv);
    u->vec = (float*)(a + o_vec); u->gn_partial = (double*)(a + o_gn);
    u->catbuf = (float*)(a + o_cat); u->hb[0] = (float*)(a + o_hb0); u->hb[1] = (float*)(a + o_hb1); u->emb = (float*)(a + o_emb);
    u->vecn = (float*)(a + o_vecn); u->embn = (float*)(a + o_embn); u->gate = (float*)(a + o_gate);
    u->splitws = (float*)(a + o_split);
    u->zeros = (const uint16_t*)(a + o_zero);      // the arena is zero-filled above and nothing writes here
    ctx->unet = u;
    return R3G_OK;
}

}  // namespace r3g

using namespace r3g;

void r3g::Ctx::release_unet() {
    if (unet) unet_free((Unet*)unet);
    unet = nullptr;
}

#define NEED_UNET(fn)                                                                              \
    Unet* u = ctx ? (Unet*)reinterpret_cast<Ctx*>(ctx)->unet : nullptr;                            \
    if (!u) return fail(R3G_ERR_STATE, fn ": r3g_unet_create has not been called");

extern "C" {

int r3g_unet_create(r3g_ctx* ctx, const r3g_unet_config* cfg) {
    if (!ctx || !cfg) return fail(R3G_ERR_INVALID, "r3g_unet_create: null argument");
    return unet_create(reinterpret_cast<Ctx*>(ctx), cfg);
}

int r3g_unet_set_tensor(r3g_ctx* ctx, const char* name, const void* d_ptr, int dtype, int64_t rows, int64_t cols) {
    NEED_UNET("r3g_unet_set_tensor");
    if (!name || !d_ptr || (dtype != 0 && dtype != 1) || rows <= 0 || cols <= 0)
        return fail(R3G_ERR_INVALID, "r3g_unet_set_tensor: bad argument for '%s'", name ? name : "?");
    if (dtype == 1 && cols % 64) return fail(R3G_ERR_INVALID, "r3g_unet_set_tensor: '%s' K=%lld is not a multiple of 64", name, (long long)cols);
    u->w[name] = UTensor{d_ptr, dtype, rows, cols};
    return R3G_OK;
}

int r3g_unet_resnet(r3g_ctx* ctx, const char* prefix, const float* d_x, int height, int width, int c_in, int c_out,
                    const float* d_temb, float* d_out, void* stream) {
    NEED_UNET("r3g_unet_resnet");
    if (!prefix || !d_x || !d_temb || !d_out) return fail(R3G_ERR_INVALID, "r3g_unet_resnet: null argument");
    return unet_resnet(*u, prefix, d_x, height, width, c_in, c_out, d_temb, d_out, (hipStream_t)stream);
}

int r3g_unet_transformer(r3g_ctx* ctx, const char* prefix, float* d_x, int height, int width, int channels, const uint16_t* d_ctx,
                         int tokens, void* stream) {
    NEED_UNET("r3g_unet_transformer");
    if (!prefix || !d_x || !d_ctx) return fail(R3G_ERR_INVALID, "r3g_unet_transformer: null argument");
    return unet_transformer(*u, prefix, d_x, height, width, channels, d_ctx, tokens, (hipStream_t)stream);
}

int r3g_unet_downsample(r3g_ctx* ctx, const char* prefix, const float* d_x, int height, int width, int channels, float* d_out,
                        void* stream) {
    NEED_UNET("r3g_unet_downsample");
    if (!prefix || !d_x || !d_out) return fail(R3G_ERR_INVALID, "r3g_unet_downsample: null argument");
    return unet_downsample(*u, prefix, d_x, height, width, channels, d_out, (hipStream_t)stream);
}

int r3g_unet_down_block(r3g_ctx* ctx, const char* prefix, const float* d_x, int height, int width, int c_in, int c_out,
                        const float* d_temb, const uint16_t* d_ctx, int tokens, int layers, int add_downsample, float* d_states,
                        float* d_out, void* stream) {
    NEED_UNET("r3g_unet_down_block");
    if (!prefix || !d_x || !d_temb || !d_ctx || !d_states || layers < 1 || (add_downsample && !d_out))
        return fail(R3G_ERR_INVALID, "r3g_unet_down_block: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const std::string pre = prefix;
    const int64_t n = (int64_t)height * width * c_out;
    // diffusers CrossAttnDownBlock2D.forward: every layer's hidden state is also an output (the skip connections of the up path)
    const float* in = d_x;
    int cin = c_in;
    for (int i = 0; i < layers; ++i) {
        float* dst = d_states + i * n;
        int rc = unet_resnet(*u, pre + ".resnets." + std::to_string(i), in, height, width, cin, c_out, d_temb, dst, s);
        if (rc) return rc;
        rc = unet_transformer(*u, pre + ".attentions." + std::to_string(i), dst, height, width, c_out, d_ctx, tokens, s);
        if (rc) return rc;
        in = dst;
        cin = c_out;
    }
    if (add_downsample) return unet_downsample(*u, pre + ".downsamplers.0", in, height, width, c_out, d_out, s);
    return R3G_OK;
}

int r3g_unet_forward(r3g_ctx* ctx, const float* d_sample, int height, int width, float timestep, const uint16_t* d_ctx, int tokens,
                     float* d_out, void* stream) {
    NEED_UNET("r3g_unet_forward");
    if (!d_sample || !d_ctx || !d_out) return fail(R3G_ERR_INVALID, "r3g_unet_forward: null argument");
    u->mv_flags = 0;
    return unet_forward(*u, d_sample, height, width, timestep, d_ctx, tokens, d_out, (hipStream_t)stream);
}

// the constant gate vectors of the 2.5D attention branches (mva_scale | ref_scale) and the mode of the pass
static int mv_begin(Unet& u, int n_views, int flags, float mva_scale, float ref_scale, hipStream_t s, const char* fn) {
    if (n_views < 1 || n_views > kMaxViews) return fail(R3G_ERR_INVALID, "%s: n_views must be in [1, %d]", fn, kMaxViews);
    if (flags & ~7) return fail(R3G_ERR_INVALID, "%s: flags is a combination of 1 (write the reference states), 2 (read them) and 4 (guidance pair)", fn);
    if ((flags & 4) && ((n_views & 1) || (flags & 1)))
        return fail(R3G_ERR_INVALID, "%s: flag 4 (the samples are a guidance pair) needs an even n_views and excludes flag 1", fn);
    u.mv_flags = flags & 3; u.groups = (flags & 4) ? 2 : 1; u.mva_scale = mva_scale; u.ref_scale = ref_scale;
    uint32_t bits[2];
    std::memcpy(&bits[0], &mva_scale, 4);
    std::memcpy(&bits[1], &ref_scale, 4);
    U_TRY(hipMemsetD32Async((hipDeviceptr_t)u.gate, (int)bits[0], (size_t)u.c.max_channels, s));
    U_TRY(hipMemsetD32Async((hipDeviceptr_t)(u.gate + u.c.max_channels), (int)bits[1], (size_t)u.c.max_channels, s));
    return R3G_OK;
}

int r3g_unet_forward_mv(r3g_ctx* ctx, const float* d_sample, int height, int width, float timestep, const uint16_t* d_ctx, int tokens,
                        int n_views, const int32_t* class_labels, int flags, float mva_scale, float ref_scale, float* d_out,
                        void* stream) {
    NEED_UNET("r3g_unet_forward_mv");
    if (!d_sample || !d_ctx || !d_out) return fail(R3G_ERR_INVALID, "r3g_unet_forward_mv: null argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = mv_begin(*u, n_views, flags, mva_scale, ref_scale, s, "r3g_unet_forward_mv");
    if (rc) return rc;
    rc = unet_forward(*u, d_sample, height, width, timestep, d_ctx, tokens, d_out, s, n_views, class_labels);
    u->mv_flags = 0; u->groups = 1;
    return rc;
}

int r3g_unet_transformer_mv(r3g_ctx* ctx, const char* prefix, float* d_x, int height, int width, int channels, const uint16_t* d_ctx,
                            int tokens, int n_views, int flags, float mva_scale, float ref_scale, void* stream) {
    NEED_UNET("r3g_unet_transformer_mv");
    if (!prefix || !d_x || !d_ctx) return fail(R3G_ERR_INVALID, "r3g_unet_transformer_mv: null argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = mv_begin(*u, n_views, flags, mva_scale, ref_scale, s, "r3g_unet_transformer_mv");
    if (rc) return rc;
    rc = unet_transformer(*u, prefix, d_x, height, width, channels, d_ctx, tokens, s, n_views);
    u->mv_flags = 0; u->groups = 1;
    return rc;
}

int r3g_unet_condition(r3g_ctx* ctx, const char* prefix, const void** d_ptr, int64_t* rows, int64_t* cols) {
    NEED_UNET("r3g_unet_condition");
    if (!prefix || !d_ptr || !rows || !cols) return fail(R3G_ERR_INVALID, "r3g_unet_condition: null argument");
    auto it = u->cond.find(prefix);
    if (it == u->cond.end() || !it->second.p) return fail(R3G_ERR_STATE, "r3g_unet_condition: no pass with flag 1 has written '%s'", prefix);
    *d_ptr = it->second.p; *rows = it->second.rows; *cols = it->second.cols;
    return R3G_OK;
}

int r3g_unet_mid_block(r3g_ctx* ctx, const char* prefix, const float* d_x, int height, int width, int channels,
                       const float* d_temb, const uint16_t* d_ctx, int tokens, float* d_out, void* stream) {
    NEED_UNET("r3g_unet_mid_block");
    if (!prefix || !d_x || !d_temb || !d_ctx || !d_out) return fail(R3G_ERR_INVALID, "r3g_unet_mid_block: null argument");
    hipStream_t s = (hipStream_t)stream;
    const std::string pre = prefix;
    int rc = unet_resnet(*u, pre + ".resnets.0", d_x, height, width, channels, channels, d_temb, d_out, s);
    if (rc) return rc;
    rc = unet_transformer(*u, pre + ".attentions.0", d_out, height, width, channels, d_ctx, tokens, s);
    if (rc) return rc;
    return unet_resnet(*u, pre + ".resnets.1", d_out, height, width, channels, channels, d_temb, d_out, s);
}

int r3g_aekl_decode(r3g_ctx* ctx, const float* d_latent, int height, int width, float* d_image, void* stream) {
    NEED_UNET("r3g_aekl_decode");
    if (!d_latent || !d_image) return fail(R3G_ERR_INVALID, "r3g_aekl_decode: null argument");
    return vae_decode(*u, d_latent, height, width, d_image, (hipStream_t)stream);
}

int r3g_aekl_encode(r3g_ctx* ctx, const float* d_image, int height, int width, float* d_moments, void* stream) {
    NEED_UNET("r3g_aekl_encode");
    if (!d_image || !d_moments) return fail(R3G_ERR_INVALID, "r3g_aekl_encode: null argument");
    return vae_encode(*u, d_image, height, width, d_moments, (hipStream_t)stream);
}

int r3g_sched_model_input(const float* d_latent, int channels, const float* d_cond, int cond_channels, int64_t pixels, float sigma,
                          float* d_out, void* stream) {
    if (!d_latent || !d_cond || !d_out || channels < 1 || cond_channels < 1 || pixels < 1 || !(sigma >= 0.0f))
        return fail(R3G_ERR_INVALID, "r3g_sched_model_input: bad argument");
    hipError_t e = model_input_launch(d_latent, channels, d_cond, cond_channels, pixels, sigma, d_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "model_input_launch");
    return R3G_OK;
}

int r3g_sched_pix2pix_input(const float* d_latent, const float* d_image_latent, int channels, int64_t pixels, float sigma,
                            float* d_out, void* stream) {
    return r3g_sched_model_input(d_latent, channels, d_image_latent, channels, pixels, sigma, d_out, stream);
}

int r3g_sched_cfg_combine(const float* d_uncond, const float* d_cond, int64_t n, float guidance_scale, float* d_out, void* stream) {
    if (!d_uncond || !d_cond || !d_out || n < 1) return fail(R3G_ERR_INVALID, "r3g_sched_cfg_combine: bad argument");
    hipError_t e = cfg_combine_launch(d_uncond, d_cond, n, guidance_scale, d_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cfg_combine_launch");
    return R3G_OK;
}

int r3g_sched_euler_ancestral_step(float* d_sample, const float* d_model_out, const float* d_noise, int64_t n, float sigma_from,
                                   float sigma_to, int prediction_type, void* stream) {
    if (!d_sample || !d_model_out || !d_noise || n < 1 || (prediction_type != 0 && prediction_type != 1))
        return fail(R3G_ERR_INVALID, "r3g_sched_euler_ancestral_step: bad argument");
    if (!(sigma_from > 0.0f) || sigma_to < 0.0f || sigma_to > sigma_from)
        return fail(R3G_ERR_INVALID, "r3g_sched_euler_ancestral_step: sigmas must satisfy 0 <= sigma_to <= sigma_from, sigma_from > 0");
    hipError_t e = euler_ancestral_step_launch(d_sample, d_model_out, d_noise, n, sigma_from, sigma_to, prediction_type, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "euler_ancestral_step_launch");
    return R3G_OK;
}

}  // extern "C"
