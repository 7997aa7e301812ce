// kernels.h -- host-side launch interface of the MFMA / elementwise kernels (internal to libr3g.so)
#ifndef R3G_KERNELS_H
#define R3G_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace r3g {

// ------------------------------------------------------------------ GEMM (gemm.hip)
enum GemmEpilogue {
    EPI_BF16 = 0,            // C(bf16) = acc + bias
    EPI_BF16_GELU_TANH = 1,  // C(bf16) = gelu_tanh(acc + bias)
    EPI_BF16_GELU_ERF = 2,   // C(bf16) = gelu_erf(acc + bias)
    EPI_RESID_F32 = 3,       // C(f32) += gate[b][n] * (acc + bias)      (gate null -> 1)
    EPI_F32 = 4,             // C(f32) = acc + bias
    EPI_QKV = 5,             // split into attention operands: per-head q/k norm, Q/K [B][H][L][64], V^T [B][H][64][L]
    EPI_RESID_BF16 = 6,      // C(bf16) = bf16(C + gate[b][n] * (acc + bias)): a 16-bit residual stream (sum formed in fp32)
    EPI_FP8_GELU_ERF = 7,    // C(e4m3) = fp8(gelu_erf(acc + bias) * out_inv_scale): operand of a following fp8 GEMM (static scale)
    EPI_RESID_F16 = 8,       // C(fp16) = fp16(C + gate[b][n] * (acc + bias)): the reference's own stream type (its pipelines run in fp16)
    // Round 6: the LAST residual GEMM of the geo decoder with ln_post + output_proj folded in.  x = bf16(C + acc + bias) is formed as in
    // EPI_RESID_BF16 but NOT stored: every wave writes, per row, the statistics of its 64 columns -- {sum, sum of squared deviations
    // from the chunk mean, sum of x * lnd_gw[n]} -- to lnd_part[row][N / 64][4]; lnd_finalize_launch merges the chunks (Chan's
    // formula, fixed order) into logit = rstd * (dot - mean * sum(gw)) + const.  One problem, 256 x 256 phased kernel only.
    EPI_RESID_BF16_LND = 9,
    // Round 6, the geo decoder's ln_3 folded into the GEMMs around it (weights are static, so LN(x) W^T = rstd (x W'^T - mu c1) + c2
    // with W' = bf16(W * gamma), c1[n] = sum_k W'[n][k], c2[n] = sum_k beta[k] W[n][k] + b[n]):
    EPI_RESID_BF16_ST = 10,        // EPI_RESID_BF16 that ALSO writes the per-row chunk statistics {sum, squared deviations} to lnd_part
    EPI_BF16_GELU_ERF_LNF = 11,    // C(bf16) = gelu_erf(rstd[m] (acc - mu[m] lnf_c1[n]) + bias[n]), (mu, rstd) = lnf_stats[m]; A = the raw stream
};

enum QkNorm { QKN_NONE = 0, QKN_RMS = 1, QKN_LAYERNORM = 2 };

enum QkvLayout {
    QKV_KHD = 0,       // columns (K H D): q heads | k heads | v heads          (DiT, Dinov2 fused qkv)
    QKV_HEAD_QKV = 1,  // per head (q,k,v) interleaved, 192 columns per head    (ShapeVAE c_qkv)
    QKV_HEAD_KV = 2,   // per head (k,v) interleaved, 128 columns per head      (geo decoder c_kv)
    QKV_Q_ONLY = 3,    // q heads only                                          (geo decoder c_q)
};

// Position of key k inside a V^T row.  Within every aligned group of 16 keys the two middle 4-key blocks are swapped
// (key 8g + 4h + e sits at 8h + 4g + e), so that the 8 keys one lane feeds into a PV MFMA -- {4h..4h+3} and
// {8+4h..8+4h+3} of the group, the keys its S^T accumulator registers hold -- are one contiguous 16-byte chunk.
__host__ __device__ inline int64_t vt_key_pos(int64_t k) {
    const int64_t p = k & 15;
    return (k & ~(int64_t)15) | (p & 3) | ((p & 4) << 1) | ((p & 8) >> 1);
}

// destination of the EPI_QKV epilogue (what qkv_split_kernel produces, fused into the projection)
struct QkvEpi {
    uint16_t* Q; uint16_t* K; uint16_t* Vt;
    int Lq_pad, Lk_pad;   // allocated rows of Q and of K / V^T
    int dst_row0;         // first destination row (token offset in the joint sequence)
    int heads, layout, norm;  // QkvLayout, QkNorm
    const float *qw, *qb, *kw, *kb;
    float eps;
    float q_scale;        // q is multiplied by this (in fp32, before rounding to bf16); 0 means 1
    // optional row segments (nseg > 0): GEMM row m in [seg_m0, seg_m1) goes to attention batch seg_batch, destination
    // row seg_dst + (m - seg_m0); rows in no segment are dropped.  nseg == 0: batch = GEMM batch, row = dst_row0 + m.
    int nseg;
    int seg_m0[3], seg_m1[3], seg_batch[3], seg_dst[3];
    // nseg > 3: the segments live in device memory instead, seg_tab[i] = {m0, m1, batch, dst} as four ints, sorted by m0,
    // disjoint, and such that no window of 128 consecutive rows meets more than three of them (gemm_launch checks the host
    // copy seg_tab_host): a wave picks the three that can meet its rows and proceeds as above.  (Several objects' CFG
    // pairs in one DiT launch: 4 segments per object.)
    const int* seg_tab;
    const int* seg_tab_host;
};

// A operand of an implicit-GEMM 3 x 3 convolution (round 5): GEMM row m = output pixel (sample, oy, ox) of nb stacked samples,
// GEMM column k = (tap, input channel) -- the row of the im2col matrix that is never written.  x: bf16 rows [nb][H*W][Cin],
// Cin % 64 == 0 (a 64-wide k-tile stays inside one tap); zero: 128 bytes of zeros for taps outside the image.
struct ConvA {
    const uint16_t* x;      // null: an ordinary GEMM (GemmArgs::A)
    const uint16_t* zero;
    int H, W, Cin, stride, pad, Ho, Wo;
};

struct GemmArgs {
    const uint16_t* A;  // bf16 [batch][M][K], row stride lda, batch stride strideA (elements)
    int64_t lda, strideA;
    const uint16_t* W;  // bf16 [N][K], row stride ldw
    int64_t ldw;
    int64_t strideW;    // batch stride of W (elements): only the 128x128 kernel reads it (split-K: the batches are slices of K); 0 = shared
    const float* bias;  // [N] or null
    void* C;            // bf16 or f32 [batch][M][N], row stride ldc, batch stride strideC (elements)
    int64_t ldc, strideC;
    const float* gate;  // [batch][N] (strideGate, may be 0) or null
    int64_t strideGate;
    const uint16_t* resid_src;   // EPI_RESID_BF16 / EPI_RESID_F16: the old values are read from here (same ldc / strideC) instead of from C; null: C
    int M, N, K;        // K % 64 == 0, N % 4 == 0
    int epi;
    int batch;          // filled in by gemm_launch
    float out_inv_scale;   // EPI_FP8_GELU_ERF: 1 / (the static scale of the fp8 output)
    int wide_epilogue;  // 1: stores go through the wave-private LDS transpose (row-contiguous 16-byte accesses)
    int raster_group;   // tile columns per rasterisation group (0 = row-major), filled in by gemm_launch
    int gelu_pk;        // 1: GELU epilogues in packed fp16 (gemm_common.h), filled in by gemm_launch
    QkvEpi qkv;         // EPI_QKV only (N % 64 == 0)
    // deterministic split-K for the fp32 epilogues (EPI_F32, EPI_RESID_F32) of ONE under-filled deep-K problem (the texture UNets'
    // 3 x 3 convolutions at the coarse levels): the caller's workspace for the partial products, null = never split
    float* split_ws;
    int64_t split_ws_elems;
    ConvA conv;         // conv.x != null: A is the implicit im2col matrix of conv.x (fp32 epilogues, one problem, 128x128 kernel)
    const float* lnd_gw;   // EPI_RESID_BF16_LND: ln_post.weight[n] * output_proj.weight[n], f32 [N]
    float* lnd_part;       // EPI_RESID_BF16_LND / _ST: f32 [M][N / 64][4]
    const float* lnf_c1;   // EPI_BF16_GELU_ERF_LNF: f32 [N]
    const float* lnf_stats;   // EPI_BF16_GELU_ERF_LNF: f32 [M][2] = (mean, rstd) of the rows of A
};

hipError_t gemm_launch(const GemmArgs& p, int batch, hipStream_t s);
// two problems with the same epilogue in one launch (the second one's tiles follow the first one's); p2 may be null
hipError_t gemm_launch2(const GemmArgs& p, int batch, const GemmArgs* p2, int batch2, hipStream_t s);
void gemm_set_glds(bool on);  // staging path: LDS-DMA (default) or register-staged
void attn_set_glds(bool on);
void attn_set_ablate(int mask);   // timing-only ablation builds of the attention kernel (tools/bench_attn.py)
void attn_set_variant(int v);       // round 6: bit 0 re-stabilise test on the sum of the exponentials, bit 1 row sum on plain adds (generations 2 / 6 / 7)
void attn_set_pipelined(bool on);  // software-pipelined attention kernel (off by default: slower) vs the plain one
void gemm_set_config(int waves);   // tile kernel: 0 automatic | 4 | 8 | 9 | 10 | 11 | 12 | 13 | 16 | 32 (include/r3g.h)
void gemm_set_raster(int group);
void gemm_set_auto_rule(int rule, int num_cu);  // tile-choice rule (0: first version, 1: current); num_cu > 0 sets the CU count
void gemm_set_persistent_qkv(bool on);   // fused QKV launches on the persistent phased kernel (default on since round 6)
void gemm_set_epi_slices(bool on);   // persistent phased kernel: bf16 / fused-QKV epilogues in 64-row passes through the wave's slices of k-tile buffer 1 (default on)
void gemm_set_mixed(bool on);        // a single block's [fused QKV | MLP-in + GELU] as one persistent launch (default on)
// fused QKV projection (pq, EPI_QKV) and MLP-in + GELU(tanh) (pm) over the same rows: one launch when the grid fills the machine
hipError_t gemm_launch_qkv_mlp(const GemmArgs& pq, const GemmArgs& pm, hipStream_t s);
bool gemm_auto_takes_256(int M, int N, int K);   // the automatic tile rule sends one (M, N, K) problem to the 256 x 256 kernels
void gemm_set_conv_implicit(bool on);   // the texture models' 3 x 3 convolutions gather their A operand themselves (default on) | im2col + GEMM
bool gemm_conv_implicit();              // ... and the staging path / tile override allow it right now
void gemm_set_xcd_walk(bool on);        // persistent phased kernel: contiguous tile range per XCD (default on)
void gemm_set_splitk128(bool on);    // split-K of the 128x128 kernel where GemmArgs::split_ws allows it (default on)
int gemm_last_splitk_slices();          // the K slices the last gemm_launch / gemm_launch2 of this process actually ran with (1 = no split)
int gemm_splitk128_factor(int M, int N, int K);   // the number of K slices the rule picks for one problem (1 = no split)
void gemm_set_splitk(bool on);       // deterministic split-K over 256x256 tiles for under-filled deep-K residual GEMMs
void gemm_set_persistent_resid(int mask);   // persistent form also for the fp32 (bit 0) / bf16 (bit 1) residual epilogues
void gemm_set_persistent(bool on);   // phased kernel walks several tiles per workgroup (default on)
void gemm_set_phased(bool on);   // 256x256 tiles: phased kernel (default) or the two-stage one
void gemm_set_gelu_pk(bool on);   // GELU epilogues in packed fp16 (default on) | fp32 with v_exp_f32 / v_rcp_f32 (rounds 3-4)
void gemm_set_wide_epilogue(bool on);  // 4|8 waves per 128x128 tile, 2|3 LDS stages

// ------------------------------------------------------------------ attention (attn.hip)
constexpr int kAttnMaxEntries = 8;
struct AttnEntry {
    int lq, lk;
    int o_split, bias_key;
    int64_t o_row0, o_row_split;
    float bias_log2;
    int buf;
};
struct AttnArgs {
    const uint16_t* Q;   // bf16 [B][H][Lq_pad][64]
    const uint16_t* K;   // bf16 [B][H][Lk_pad][64]
    const uint16_t* Vt;  // bf16 [B][H][64][Lk_pad]
    uint16_t* O;         // bf16 [B][Lq][ldo] : head h at columns h*64 .. h*64+63
    int64_t ldo, strideO;  // row stride and batch stride of O (elements)
    int B, H;
    int Lq, Lq_pad;      // valid / allocated query rows (Lq_pad % 128 == 0)
    int Lk, Lk_pad;      // valid / allocated keys (Lk_pad % 64 == 0)
    int kv_batch_stride_zero;  // 1: K/V are shared by all batches (cross attention with B query chunks)
    float scale;         // softmax scale (1/sqrt(64))
    int q_prescaled;     // 1: Q already holds q * scale * log2(e) (QkvEpi::q_scale = attn_q_scale())
    // Ragged mode (B <= kAttnMaxEntries work entries; CFG with a de-duplicated unconditional context, for one or several
    // objects): entry e has its own lengths, reads Q / K / V^T of batch slot ent[e].buf, an output row map
    //   row(e, q) = q < o_split ? o_row0 + q : o_row_split + (q - o_split)     (strideO ignored)
    // and one key that stands for `2^bias_log2` identical keys (its score gets +bias_log2 in log2 units; the key must lie
    // in the last, padded key tile).  bias_key < 0: none.  Work items are issued entry by entry inside a head: list
    // the long entries first.
    int ragged;
    AttnEntry ent[kAttnMaxEntries];
};

hipError_t attention_launch(const AttnArgs& p, hipStream_t s);
// factor the producers of Q fold into it for the current attention kernel: scale * log2(e) (generation 2), or 1
float attn_q_scale(float scale);
void attn_set_wide_min(int items);     // generation 7: 256-query workgroups (generation 6) from this many work items on (default 2048)
void attn_set_prio(int v);            // generation 9: 0 no s_setprio | 1 matrix phase raised | 2 softmax phases raised
void attn_set_stamps(int on);         // generation 9: print per-phase s_memtime sums of every launch to stderr (timing experiments)
void attn_set_generation(int gen);   // 7 (default: 6 on deep grids, else 2) | 2 | 6 | 1: the first-round kernel (expects plain Q; same V^T layout)
void ln_set_rows4_min(int rows);     // automatic rule: launches of at least this many rows take 4 rows per wave (65536)
void ln_set_modes(bool on);         // LayerNorm: compile-time instantiations for the affine-only / modulation-only launches (default on)
void ln_set_rows_per_wave(int rows);  // LayerNorm / ln_dot row kernels: 0 automatic | 1 | 4 rows per wave
void ln_set_fixed_count(bool on);     // 1 (default): compile-time element counts for C = 1024 / 1536

// ------------------------------------------------------------------ elementwise / norms (elem.hip)
// y(bf16)[r][c] = ((x - mean) * rstd * (w ? w[c] : 1) + (b ? b[c] : 0)) * (1 + scale[batch][c]) + shift[batch][c]
struct LnArgs {
    const float* x; int64_t ldx;        // f32 [rows][C] (16-bit [rows][C] behind the same pointer when x_bf16 != 0)
    int x_bf16;                         // 0: f32 | 1: bf16 | 2: fp16
    uint16_t* y; int64_t ldy;           // bf16 [rows][C]
    uint8_t* y8; int64_t ldy8; float* y_scale;   // y8 != null: e4m3 [rows][C] + one scale per row INSTEAD of y (C % 256 == 0)
    int64_t x_batch_stride, y_batch_stride;  // row r lives at batch (r / rows_per_batch), local row r % rows_per_batch
    const float* w; const float* b;     // affine [C] or null
    const float* scale; const float* shift; int64_t mod_stride;  // per-batch [C] or null
    // rows in [seg2_row0, seg2_row1) use scale2 / shift2 instead (the txt-stream rows of a DiT double block, whose
    // modulation differs from the img rows around them); seg2_row1 <= seg2_row0: unused
    int seg2_row0, seg2_row1;
    const float* scale2; const float* shift2;
    int rows, C, rows_per_batch;
    float eps;
};
hipError_t layernorm_launch(const LnArgs& p, hipStream_t s);

// Split a fused projection output into attention operands, normalising q and k per head (dim 64).
struct QkvSplitArgs {
    const uint16_t* src; int64_t ld; int64_t src_batch_stride;  // bf16 [B][L][ld]
    int q_off, k_off, v_off;   // column of head 0 / dim 0 for q, k, v (-1: absent)
    int head_stride;           // column distance between consecutive heads
    uint16_t* Q; uint16_t* K; uint16_t* Vt;   // destinations (may be null when absent)
    int Lq_pad, Lk_pad;        // allocated rows of Q and of K/Vt
    int dst_row0;              // first destination row (token offset inside the joint sequence)
    int B, H, L;               // L source tokens per batch
    int norm;                  // QkNorm
    const float* qw; const float* qb; const float* kw; const float* kb;  // [64] scale / bias
    float eps;
    float q_scale;             // as QkvEpi::q_scale
};
hipError_t qkv_split_launch(const QkvSplitArgs& p, hipStream_t s);

// y[b][n] = act(x[b][:] . W[n][:] + bias[n]); x f32 [B][K], W bf16 [N][K], y f32; B <= 8
hipError_t gemv_launch(const float* x, int B, int K, const uint16_t* W, int64_t ldw, const float* bias, float* y,
                       int N, int act_silu_in, int act_silu_out, hipStream_t s);

// Many small GEMVs sharing one input x (all adaLN modulations of a DiT forward in one launch)
struct GemvJob { const uint16_t* W; const float* bias; int64_t ldw; int N; int64_t out_off; };  // y + out_off : [B][N]
hipError_t gemv_multi_launch(const float* x, int B, int K, const GemvJob* d_jobs, int njobs, float* y, int silu_in,
                             hipStream_t s);

// t == null: every batch entry uses t_scalar
hipError_t timestep_embedding_launch(const float* t, float t_scalar, int B, float time_factor, float* out /*[B][256]*/,
                                     hipStream_t s);
// out(bf16)[r][c] = in(f32)[r][c] for c < C, 0 for C <= c < Cpad
hipError_t cast_pad_launch(const float* in, int64_t ldi, uint16_t* out, int64_t ldo, int rows, int C, int Cpad,
                           float scale, hipStream_t s);
// x(f32)[b][r][c] = v(bf16/f32) broadcast helpers
hipError_t fill_rows_launch(float* dst, int64_t ld, int rows, int C, const float* row_values, hipStream_t s);
// latents += dsigma * (v_u + g (v_c - v_u));  v = [2][n] (cond first)
hipError_t cfg_euler_launch(float* latents, const float* v2, int64_t n, float guidance, float dsigma, hipStream_t s);
hipError_t nonfinite_flag_launch(const float* x, int64_t n, int* flag, hipStream_t s);   // *flag |= 1 when x holds a NaN / infinity
// swiglu: out(bf16)[r][c] = silu(in[r][c]) * in[r][F + c]
hipError_t swiglu_launch(const uint16_t* in, int64_t ldi, uint16_t* out, int64_t ldo, int rows, int F, hipStream_t s);
// Fourier features of dense grid points [start, start+count): bf16 [count][64] = (xyz, sin(x 2^k).., cos.., 0 pad)
hipError_t fourier_grid_launch(uint16_t* out, int64_t start, int count, int R, double bound, int num_freqs,
                               int include_pi, hipStream_t s);
// logits[r] = LN(x[r]) . w + b  (ln_post + output_proj fused);  x f32 [rows][C]
// EPI_RESID_BF16_LND's two small kernels (elem.hip): gw[n] = lnw[n] * w[n], consts = {sum gw, sum lnb[n] w[n] + b}; and the merge of the
// per-chunk statistics into out[row] = rstd * (dot - mean * consts[0]) + consts[1]
hipError_t lnd_prepare_launch(const float* lnw, const float* lnb, const float* w, float b, int N, float* gw, float* consts, hipStream_t s);
// ln_3 fold: W' = bf16(W * gamma) [N][K], c1[n] = sum_k W'[n][k], c2[n] = sum_k beta[k] W[n][k] + b[n];  (mean, rstd) per row from the chunk statistics
hipError_t lnf_prepare_launch(const uint16_t* w, int64_t ldw, const float* b, const float* gamma, const float* beta, int N, int K,
                              uint16_t* w2, float* c1, float* c2, hipStream_t s);
hipError_t lnf_stats_launch(const float* part, int rows, int parts, float eps, float* stats, hipStream_t s);
hipError_t lnd_finalize_launch(const float* part, int rows, int parts, float eps, const float* consts, float* out, hipStream_t s);
hipError_t ln_dot_launch(const float* x, int64_t ldx, int rows, int C, int do_ln, const float* lnw, const float* lnb, float eps,
                         const float* w, float b, float* out, hipStream_t s, int x_bf16 = 0);
// DINOv2 patch embedding im2col: image f32 [3][S][S] -> bf16 [P*P][Kpad], K = 3*ps*ps ordered (c, dy, dx)
hipError_t im2col_launch(const float* img, int S, int ps, uint16_t* out, int Kpad, hipStream_t s);
// x(f32)[r][c] = a(f32)[r][c] (+ pos[r][c]) ; assorted small helpers
hipError_t add_rows_launch(float* x, int64_t ldx, const float* pos, int64_t ldp, int rows, int C, hipStream_t s);
// FP8 (OCP e4m3) path, BASELINE.json configs[3]: row-scaled quantisation and the fp8 form of the phased GEMM
hipError_t quant_fp8_rows_launch(const uint16_t* x, int64_t ldx, int rows, int K, uint8_t* q, int64_t ldq, float* scale,
                                 hipStream_t s);
hipError_t gemm_fp8_launch(const GemmArgs& p, const float* scale_a, const float* scale_w, hipStream_t s);
hipError_t f32_to_bf16_launch(const float* in, uint16_t* out, int64_t n, hipStream_t s);

// ------------------------------------------------------------------ UNet blocks of the texture stage (conv_kernels.hip)
// bf16 [H][W][C] -> bf16 [Ho*Wo][9 C] (column (ky*3 + kx)*C + c; zero padding 1; stride 1 | 2; C % 8 == 0)
hipError_t im2col3x3_launch(const uint16_t* x, int H, int W, int C, int stride, int pad, uint16_t* out, hipStream_t s, int nb = 1);   // nb contiguous samples
// GroupNorm (+ SiLU) of nb samples of f32 rows [nb][rows][C] -> bf16, every sample normalised on its own, all of them in three
// launches; partial: workspace of nb * (group_norm_blocks(rows) * groups * 2 doubles + groups floats)
int group_norm_blocks(int rows);
hipError_t group_norm_launch(const float* x, int rows, int C, int groups, const float* gamma, const float* beta, float eps,
                             int do_silu, uint16_t* y, double* partial, hipStream_t s, int nb = 1,
                             const float* addv = nullptr, int64_t add_stride = 0);   // addv [nb][add_stride]: added to every row of its sample first
// out(bf16)[r][c] = in[r][c] * gelu_erf(in[r][F + c])   (diffusers GEGLU)
hipError_t geglu_launch(const uint16_t* in, int64_t ldi, uint16_t* out, int64_t ldo, int rows, int F, hipStream_t s);
hipError_t vec_add_launch(const float* a, const float* b, float* out, int n, hipStream_t s);
// P[r][:] = softmax(scale * S[r][:]) as bf16 (fp32 scores in; n % 4 == 0): the VAE mid block's single-head attention
hipError_t softmax_rows_launch(const float* S, int64_t lds, uint16_t* P, int64_t ldp, int rows, int n, float scale, hipStream_t s);
// sampling loops of the texture stage's diffusion pipelines: UNet input rows (latent / sqrt(sigma^2 + 1) | conditioning latents),
// classifier-free guidance, and one EulerAncestralDiscreteScheduler step in place on the sample (vpred: v_prediction)
hipError_t model_input_launch(const float* lat, int zc, const float* cond, int ic, int64_t pixels, float sigma, float* out, hipStream_t s);
hipError_t cfg_combine_launch(const float* uncond, const float* cond, int64_t n, float scale, float* out, hipStream_t s);
hipError_t euler_ancestral_step_launch(float* x, const float* model_out, const float* noise, int64_t n, float sigma_from,
                                       float sigma_to, int vpred, hipStream_t s);
// nearest 2x upsampling: f32 [H][W][C] -> bf16 [2H][2W][C]
hipError_t upsample2x_launch(const float* x, int H, int W, int C, uint16_t* y, hipStream_t s);
// diffusers Timesteps(flip_sin_to_cos, freq_shift 0): out f32 [dim] = [cos | sin]
hipError_t unet_timestep_launch(float t, int dim, float* out, hipStream_t s);

}  // namespace r3g
#endif
