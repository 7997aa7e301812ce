/*
 * r3g.h -- C ABI of libr3g.so: the MI355X-native implementation of the per-object 2D->3D
 * asset-generation hot path of 3D-RE-GEN (stage "Hunyuan_2d_to_3d").
 *
 * The reference has no FFI on this path: its boundary is the Python API of the un-vendored
 * `hy3dgen` package as used by src/2d_to_3d_models/run.py:10-17,77-84 (SURVEY.md section 8b,
 * level B3).  This header is level B4: the plain-C surface that sits directly underneath that
 * Python API.  Each entry point names the reference-side computation it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error (R3G_ERR_*); the message of the last
 *    error on the calling thread is available from r3g_last_error();
 *  - pointers named d_* are DEVICE pointers (HBM); h_* are host pointers;
 *  - `stream` is a hipStream_t passed as void* (0 = the null stream); work is enqueued on it and,
 *    unless stated otherwise, the call returns without synchronising;
 *  - no torch / C++ types cross the boundary; one r3g_ctx per process and device, not
 *    thread-safe (mirrors the reference's one-process-per-task model, run.py:108-136);
 *  - there is NO CPU fallback: without a visible gfx950 device r3g_create fails.
 */
#ifndef R3G_H
#define R3G_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R3G_VERSION 100 /* 0.1.0 */

#define R3G_OK 0
#define R3G_ERR_INVALID (-1)     /* bad argument */
#define R3G_ERR_HIP (-2)         /* HIP runtime error, see r3g_last_error() */
#define R3G_ERR_NO_DEVICE (-3)   /* no gfx950 device visible */
#define R3G_ERR_STATE (-4)       /* call order violated (e.g. emit without count) */
#define R3G_ERR_LEVEL_RANGE (-10) /* skimage: ValueError("Surface level must be within volume data range.") */
#define R3G_ERR_NO_SURFACE (-11)  /* skimage: RuntimeError("No surface found at the given iso value.") */

typedef struct r3g_ctx r3g_ctx;

int r3g_version(void);
const char* r3g_last_error(void);

/* One context per (process, device).  Replaces the implicit CUDA context the reference worker
 * gets from CUDA_VISIBLE_DEVICES (src/2d_to_3d_models/run.py:114). */
int r3g_create(int device, r3g_ctx** out);
void r3g_destroy(r3g_ctx* ctx);

/* ---- marching cubes --------------------------------------------------------------------------
 * Replaces hy3dgen MCSurfaceExtractor.run's
 *     skimage.measure.marching_cubes(grid_logit.cpu().numpy(), mc_level, method="lewiner")
 * (skimage/measure/_marching_cubes_lewiner.py:280-349; reached from run.py:77-84) plus the
 * vertex rescale and winding flip that follow it, with the grid left in HBM.
 *
 * r3g_mc_count: classify all (n0-1)(n1-1)(n2-1) cells of the C-contiguous fp32 grid
 *   d_grid[n0][n1][n2], scan, and report the mesh size.  Synchronises `stream`.
 *   use_classic != 0 selects skimage's method="lorensen" tables.
 *   Errors mirror the wrapper: R3G_ERR_LEVEL_RANGE, R3G_ERR_NO_SURFACE.
 * r3g_mc_emit: write the mesh of the preceding r3g_mc_count (same grid, still resident):
 *   d_verts float32 [nV][3], d_faces int32 [nF][3].
 *   xform == NULL  : skimage's return convention -- vertices in index space, columns
 *                    (axis0, axis1, axis2), faces reversed (gradient_direction='descent').
 *   xform != NULL  : 9 doubles {grid_size[3], bbox_size[3], bbox_min[3]}; vertices become
 *                    float32(double(v) / grid_size * bbox_size + bbox_min) exactly as upstream
 *                    (note upstream's grid_size = R+1), and reverse_faces selects the winding
 *                    (0 = after export_to_trimesh's faces[:, ::-1], i.e. outward for
 *                    positive-inside fields).
 */
int r3g_mc_count(r3g_ctx* ctx, const float* d_grid, int n0, int n1, int n2, double level, int use_classic,
                 int64_t* n_verts, int64_t* n_faces, void* stream);
int r3g_mc_emit(r3g_ctx* ctx, float* d_verts, int32_t* d_faces, const double* xform, int reverse_faces,
                void* stream);

/* ---- mesh cleaners ------------------------------------------------------
 * Replace `mesh = FloaterRemover()(mesh); mesh = DegenerateFaceRemover()(mesh); mesh = FaceReducer()(mesh)`
 * (reference src/2d_to_3d_models/run.py:93-94; upstream hy3dgen/shapegen/postprocessors.py runs pymeshlab on
 * one CPU thread) on the device buffers r3g_mc_emit filled: d_verts float32 [*n_verts][3], d_faces int32
 * [*n_faces][3].  Each call compacts both arrays in place (survivors keep their order), stores the new sizes in
 * *n_verts / *n_faces and synchronises `stream`.
 *   remove_floaters  : drops every connected component with fewer faces than max(1, floor(min_ratio * faces of the largest
 *                      component)) (MeshLab truncates the product), then unreferenced vertices.  Faces are joined through
 *                      shared EDGES, as MeshLab's face-face adjacency joins them: two parts that touch in a single vertex
 *                      are two components (r3g_set_option("floater_by_vertex", 1): joined through shared vertices, the
 *                      behaviour of rounds 1-2).
 *   remove_degenerate: drops faces with a repeated vertex index, then unreferenced vertices.
 *   reduce_faces     : no-op when *n_faces <= max_faces; otherwise quadric-error-metric edge collapse (the algorithm
 *                      class of upstream's MeshLab filter) in rounds of independent collapses: every vertex picks its
 *                      cheapest valid edge (optimal placement; link condition, no face turning by more than ~78
 *                      degrees, no slivers, boundary vertices stay on the boundary) and PROPOSES it; per round a maximal
 *                      set of proposals with pairwise disjoint closed neighbourhoods is selected (three Luby iterations on
 *                      the fixed (key, proposer) order, round 4) and collapses simultaneously, the last round is cut at the key that
 *                      meets the budget (result within 1 % below max_faces).  Closed surfaces stay closed manifolds
 *                      of the same genus.  Equivalence with MeshLab is geometric, not index-wise.
 *   cluster_faces    : vertex clustering on a uniform grid over the bounding box (first resolution
 *                      floor(sqrt(max_faces / 2.2)), shrunk by 0.9 until the face budget holds, at most 24 times): a
 *                      cluster's vertex is the mean of its members, faces that collapse or repeat an earlier face's
 *                      vertex set are dropped.  Robust on triangle soups; does not preserve topology.
 * The results are pure functions of the input: oracle/mesh_clean.py reproduces floaters / degenerate / cluster bit for
 * bit, tests/emu/qem_emu.cpp (the same per-element code in host loops) the edge collapse. */
int r3g_mesh_remove_floaters(r3g_ctx* ctx, float* d_verts, int64_t* n_verts, int32_t* d_faces, int64_t* n_faces,
                             double min_ratio, void* stream);
int r3g_mesh_remove_degenerate(r3g_ctx* ctx, float* d_verts, int64_t* n_verts, int32_t* d_faces, int64_t* n_faces,
                               void* stream);
int r3g_mesh_reduce_faces(r3g_ctx* ctx, float* d_verts, int64_t* n_verts, int32_t* d_faces, int64_t* n_faces,
                          int64_t max_faces, void* stream);
int r3g_mesh_cluster_faces(r3g_ctx* ctx, float* d_verts, int64_t* n_verts, int32_t* d_faces, int64_t* n_faces,
                           int64_t max_faces, void* stream);

/* ---- texture stage: native pieces ------------------------------------------------------------
 * SURVEY.md 8(f) rank 3.  Upstream's Hunyuan3DPaintPipeline (reference call site src/2d_to_3d_models/run.py:97, built at
 * :126-128) uses two native extensions, `custom_rasterizer` (CUDA) and `mesh_processor.cpp`, and bakes the generated views
 * into a UV texture in its MeshRender.  These are their gfx950 counterparts; the two diffusion UNets are NOT part of this
 * library.  All buffers are device pointers; images are row-major [H][W][C] float32, row 0 = top.
 *   rasterize     ~ custom_rasterizer.rasterize(pos, tri, (H, W)): pos_clip float32 [V][4], tri int32 [F][3] ->
 *                   findices int32 [H][W] (face + 1, 0 = empty), bary float32 [H][W][3] (perspective-correct).  Pixel
 *                   (ix, iy) samples (ix + 0.5, iy + 0.5) of the screen x = (x/w/2 + 1/2)(W-1) + 1/2; nearest depth wins, equal
 *                   depths go to the smaller face index (result independent of scheduling).
 *   interpolate   ~ custom_rasterizer.interpolate(attr, findices, bary, tri): out[p] = sum_k bary[p][k] attr[tri[f][k]].
 *   view_weight   baking weight of a view: view_weight * cos^power of the view-space normal, 0 below cos_threshold, on
 *                   the silhouette and where the depth jumps by more than depth_edge between 4-neighbours.
 *   bake          scatters a view (image float32 [H][W][3] in [0,1]) into acc uint64 [T][T][4] (zero it first; fixed point,
 *                   integer atomics: exact, order independent); texel = round(uv * (T-1)), v = 0 is the top row.
 *   bake_gather   the same accumulation, texel-centric (what hy3dgen.texgen uses): every texel covered in findices_uv /
 *                   bary_uv (= rasterize() of the mesh in UV space) interpolates its clip position from clip_uv float32
 *                   [Vuv][4] (this view's clip coordinates of the UV vertices), must not lie behind the view's depth
 *                   buffer (depth float32 [H][W] = interpolate() of clip z/w; tolerance depth_eps), takes the weight of the
 *                   pixel it falls into and a bilinear sample of the image.  No atomics; texture as dense as the UV raster.
 *   bake_finalize acc -> texture float32 [T][T][3], mask uint8 [T][T] (1 = painted).
 *   inpaint       ~ mesh_processor.meshVerticeInpaint + the dilation upstream does with cv2: vertex colours from painted
 *                   texels, propagation along mesh edges (weights 1/(d^2 + 1e-6), Jacobi rounds until nothing changes),
 *                   unpainted covered texels from the vertex colours (mask 2), `dilate_iters` steps into empty texels (mask
 *                   3).  findices_uv / bary_uv = rasterize() of the mesh in UV space.  Synchronises the stream. */
int r3g_tex_rasterize(r3g_ctx* ctx, const float* d_pos_clip, int64_t n_verts, const int32_t* d_tri, int64_t n_faces, int height,
                      int width, int32_t* d_findices, float* d_bary, void* stream);
int r3g_tex_interpolate(r3g_ctx* ctx, const float* d_attr, int channels, const int32_t* d_tri, const int32_t* d_findices,
                        const float* d_bary, int64_t n_pixels, float* d_out, void* stream);
int r3g_tex_view_weight(r3g_ctx* ctx, const int32_t* d_findices, const float* d_depth, const float* d_normal, int height,
                        int width, float cos_threshold, float depth_edge, float view_weight, float power, float* d_weight,
                        void* stream);
int r3g_tex_bake(r3g_ctx* ctx, const float* d_image, const float* d_weight, const int32_t* d_findices, const float* d_bary,
                 const float* d_uv, const int32_t* d_uv_tri, int64_t n_pixels, int tex_size, uint64_t* d_acc, void* stream);
int r3g_tex_bake_gather(r3g_ctx* ctx, const int32_t* d_findices_uv, const float* d_bary_uv, const float* d_clip_uv,
                        const int32_t* d_uv_tri, int tex_size, const float* d_image, const float* d_weight,
                        const int32_t* d_findices, const float* d_depth, int height, int width, float depth_eps, uint64_t* d_acc,
                        void* stream);
int r3g_tex_bake_finalize(r3g_ctx* ctx, const uint64_t* d_acc, int tex_size, float* d_texture, uint8_t* d_mask, void* stream);
int r3g_tex_inpaint(r3g_ctx* ctx, float* d_texture, uint8_t* d_mask, int tex_size, const int32_t* d_findices_uv,
                    const float* d_bary_uv, const float* d_verts, int64_t n_verts, const int32_t* d_pos_tri, const float* d_uv,
                    const int32_t* d_uv_tri, int64_t n_faces, int dilate_iters, int* rounds_out, void* stream);

/* ---- shape model (DiT + ShapeVAE + DINOv2 conditioner) -----------------------------------------
 * Replaces the modules `Hunyuan3DDiTFlowMatchingPipeline.from_pretrained` instantiates from the
 * checkpoint's config.yaml (reference call sites src/2d_to_3d_models/run.py:122-124,204-206;
 * upstream hy3dgen/shapegen/pipelines.py).  All dimensions are data; head_dim must be 64.
 */
typedef struct r3g_model_config {
    /* denoisers/hunyuan3ddit.py Hunyuan3DDiT */
    int32_t dit_in_channels, dit_context_dim, dit_hidden, dit_heads, dit_depth_double, dit_depth_single,
        dit_mlp_hidden, dit_qkv_bias;
    float dit_time_factor;
    /* autoencoders/model.py ShapeVAE (+ geo_decoder) */
    int32_t vae_num_latents, vae_embed_dim, vae_width, vae_heads, vae_layers, vae_num_freqs, vae_include_pi,
        vae_qkv_bias, vae_qk_norm, vae_mlp_ratio, vae_ln_post;
    float vae_scale_factor;
    /* conditioner.DinoImageEncoder (transformers Dinov2Model, SwiGLU FFN) */
    int32_t cond_image_size, cond_patch, cond_hidden, cond_layers, cond_heads, cond_ffn_hidden;
    float cond_ln_eps;
    /* query points per internal pass of r3g_grid_query (0 = default 131072); NOT upstream's num_chunks:
     * results do not depend on it */
    int32_t grid_chunk;
} r3g_model_config;

/* Allocate the activation arena for this configuration (replaces instantiate_from_config). */
int r3g_model_create(r3g_ctx* ctx, const r3g_model_config* cfg);
/* Register one parameter under its upstream state-dict name ("model.double_blocks.0.img_attn.qkv.weight",
 * "vae.post_kl.bias", "conditioner.main_image_encoder.model.encoder.layer.0.norm1.weight", ...).
 * dtype 0 = float32, 1 = bfloat16.  Matrices: bf16 [rows=N][cols=K], K zero-padded to a multiple of 64;
 * vectors: f32.  The memory stays owned by the caller and must outlive the context.
 * (Replaces load_state_dict of the safetensors checkpoint.) */
int r3g_model_set_tensor(r3g_ctx* ctx, const char* name, const void* d_ptr, int dtype, int64_t rows, int64_t cols);
int r3g_model_set_scalar(r3g_ctx* ctx, const char* name, float value);
/* Give back the device memory the model holds beyond its weights and its arena: the query-side cache of the geo decoder
 * (r3g_grid_query keeps the object-independent half of every grid pass resident, up to the budget of option
 * "geo_q_cache_gb", default 30 % of the device's memory).  The next r3g_grid_query builds it again.  A stage that
 * is about to start another large consumer on the same GPU (texture models, a second process) calls this first.
 * Synchronises the device.  No counterpart upstream (the reference recomputes the query side per chunk). */
int r3g_model_trim(r3g_ctx* ctx);

/* conditioner forward: d_image f32 [3][S][S] already resized/cropped/normalised (ImageEncoder.transform);
 * d_cond_out bf16 [S/14*S/14+1][cond_hidden] = Dinov2Model(...).last_hidden_state (CLS first). */
int r3g_cond_encode(r3g_ctx* ctx, const float* d_image, uint16_t* d_cond_out, void* stream);

/* Hunyuan3DDiT.forward(x, t, contexts={'main': cond}): d_x f32 [B][num_latents][in_channels], d_t f32 [B]
 * (sigma in [0,1]), d_cond bf16 [B][cond_tokens][context_dim] -> d_out f32 like d_x.  B in {1,2}.
 * n_double / n_single < 0 run every block (>= 0: only the first n, for per-block parity tests). */
int r3g_dit_forward(r3g_ctx* ctx, const float* d_x, const float* d_t, const uint16_t* d_cond, float* d_out, int batch,
                    int n_double, int n_single, void* stream);

/* The joint residual stream the preceding r3g_dit_forward(batch, n_double, n_single) left behind, i.e. the input of
 * the next block (or of final_layer): d_out f32 [batch][cond_tokens + num_latents][hidden] in upstream's order
 * cat(cond, latent).  For per-block parity tests: (stream after k+1 blocks) - (stream after k blocks) is block k's
 * branch contribution (hunyuan3ddit.py DoubleStreamBlock / SingleStreamBlock.forward). */
int r3g_dit_stream(r3g_ctx* ctx, float* d_out, int batch, void* stream);

/* The denoising loop of Hunyuan3DDiTFlowMatchingPipeline.__call__ with classifier-free guidance:
 * sigmas = linspace(0,1,steps) (+ FlowMatchEulerDiscreteScheduler shift), per step
 *   v = DiT(cat([x]*2), sigma, d_cond2);  v = v_u + g (v_c - v_u);  x += (sigma_next - sigma) v
 * d_latents f32 [num_latents][in_channels] in/out; d_cond2 bf16 [2][tokens][dim] = [cond, uncond].
 * uncond_uniform != 0 declares that all tokens of the unconditional context are identical (upstream:
 * zeros_like(cond)); they are then carried as ONE token that counts `tokens` times in the softmax -- the same
 * function with 31 % fewer rows in that batch entry (switch "cfg_dedup" of r3g_set_option turns this off). */
int r3g_flow_sample(r3g_ctx* ctx, float* d_latents, const uint16_t* d_cond2, int steps, float guidance_scale,
                    float shift, int uncond_uniform, void* stream);
/* The same loop for n_objects independent objects (upstream: the pipeline's batch dimension when `image` is a list; the
 * reference gets object-level parallelism from its worker pool, src/2d_to_3d_models/run.py:176-193): d_latents f32
 * [n_objects][num_latents][in_channels] in/out, d_cond2 bf16 [n_objects][2][tokens][dim].  With uncond_uniform the objects
 * share every DiT launch (up to 4 per launch; the GEMMs then have enough rows for 256x256 tiles on every layer); each
 * object's result is bit-identical to what r3g_flow_sample gives for it alone. */
int r3g_flow_sample_batch(r3g_ctx* ctx, float* d_latents, const uint16_t* d_cond2, int n_objects, int steps,
                          float guidance_scale, float shift, int uncond_uniform, void* stream);

/* ShapeVAE.forward(latents / scale_factor) (post_kl + transformer) and the geo decoder's K/V of the
 * result (computed once; upstream recomputes them for every chunk).  d_z_out (optional) f32
 * [num_latents][width] receives the decoded latents. */
int r3g_vae_decode(r3g_ctx* ctx, const float* d_latents, float* d_z_out, void* stream);

/* VanillaVolumeDecoder: occupancy logits of dense grid points [start, start+count) of the (R+1)^3 grid
 * (point index = (i*(R+1)+j)*(R+1)+k, coordinates np.linspace(-bound, bound, R+1)) written to
 * d_grid[start ...], fp32.  Needs a preceding r3g_vae_decode. */
int r3g_grid_query(r3g_ctx* ctx, double bound, int octree_resolution, float* d_grid, int64_t start, int64_t count,
                   void* stream);

/* ---- texture stage: UNet blocks (first slice) -------------------------------------------
 * The two diffusion models behind upstream's Hunyuan3DPaintPipeline.__call__ (reference call site
 * src/2d_to_3d_models/run.py:97; built at :126-128, :207-209) are diffusers UNet2DConditionModels on the Stable-Diffusion-2.1
 * layout.  These entry points are their building blocks on gfx950 -- ResnetBlock2D, Transformer2DModel
 * (use_linear_projection, one BasicTransformerBlock: self-attention, cross-attention over the context, GEGLU feed-forward),
 * Downsample2D / Upsample2D, the compositions CrossAttnDownBlock2D / UNetMidBlock2DCrossAttn, and the whole
 * UNet2DConditionModel.forward (r3g_unet_forward); the VAE, the schedulers, the context encoders and upstream's multiview /
 * reference attention extensions are NOT part of this library yet (the stage keeps reporting where its colours come from).
 * Activations are rows: f32 [height*width][channels] (NHWC: a pixel's channels contiguous; channels % 64 == 0, head dim
 * 64), the context bf16 [tokens][ctx_dim], the time embedding f32 [temb_dim] (the output of the UNet's time_embedding MLP).
 * Weights are registered under diffusers' state-dict names below `prefix` ("down_blocks.0", "mid_block", ...):
 *   3x3 convolutions bf16 [C_out][9 C_in] with column (ky*3 + kx)*C_in + c (a re-layout of torch's [C_out][C_in][3][3]),
 *   linear layers bf16 [N][K], vectors f32; fused projections registered as "<block>.attn1.to_qkv.weight" = rows of to_q, to_k,
 *   to_v and "<block>.attn2.to_kv.weight" = per head its 64 rows of to_k then its 64 rows of to_v (pure re-layouts, done by
 *   r3g/unet.py).  All calls enqueue on `stream` and return. */
typedef struct r3g_unet_config {
    int32_t max_hw;         /* largest height*width an entry point will see */
    int32_t max_channels;   /* largest channel count (multiple of 64, <= 2048) */
    int32_t temb_dim;       /* 1280 for SD 2.1 */
    int32_t ctx_dim;        /* cross_attention_dim: 1024 for SD 2.1 */
    int32_t ctx_tokens;     /* most context tokens (77 for CLIP text) */
    int32_t groups;         /* norm_num_groups: 32 */
    float resnet_eps;       /* GroupNorm eps of the resnets: 1e-5 (Transformer2DModel's norm uses 1e-6, its LayerNorms 1e-5) */
    /* block structure, for r3g_unet_forward only (0 levels: building blocks only): SD 2.1 = 4 levels (320, 640, 1280, 1280),
     * 2 layers per block, 4 -> 4 channels.  max_channels must cover the widest concatenation of the up path (2560) */
    int32_t n_levels, layers_per_block, in_channels, out_channels;
    int32_t block_out_channels[4];
} r3g_unet_config;
int r3g_unet_create(r3g_ctx* ctx, const r3g_unet_config* cfg);
int r3g_unet_set_tensor(r3g_ctx* ctx, const char* name, const void* d_ptr, int dtype, int64_t rows, int64_t cols);
/* ResnetBlock2D.forward(x, temb) -> d_out f32 [height*width][c_out] (may alias d_x when c_in == c_out) */
int r3g_unet_resnet(r3g_ctx* ctx, const char* prefix, const float* d_x, int height, int width, int c_in, int c_out,
                    const float* d_temb, float* d_out, void* stream);
/* Transformer2DModel.forward(x, encoder_hidden_states), in place on d_x */
int r3g_unet_transformer(r3g_ctx* ctx, const char* prefix, float* d_x, int height, int width, int channels, const uint16_t* d_ctx,
                         int tokens, void* stream);
/* Downsample2D.forward: conv 3x3, stride 2, padding 1 -> d_out f32 [(height+1)/2 * (width+1)/2][channels] */
int r3g_unet_downsample(r3g_ctx* ctx, const char* prefix, const float* d_x, int height, int width, int channels, float* d_out,
                        void* stream);
/* CrossAttnDownBlock2D.forward: `layers` x (resnet, transformer) [+ downsample].  d_states f32 [layers][height*width][c_out]
 * receives every layer's hidden state (diffusers' output_states: the skip connections); d_out the downsampled result. */
int r3g_unet_down_block(r3g_ctx* ctx, const char* prefix, const float* d_x, int height, int width, int c_in, int c_out,
                        const float* d_temb, const uint16_t* d_ctx, int tokens, int layers, int add_downsample, float* d_states,
                        float* d_out, void* stream);
/* UNet2DConditionModel.forward(sample, timestep, encoder_hidden_states) on the SD-2.1 layout (CrossAttnDownBlock2D x (n-1) +
 * DownBlock2D | UNetMidBlock2DCrossAttn | UpBlock2D + CrossAttnUpBlock2D x (n-1); conv_in / time_embedding / conv_norm_out /
 * conv_out under diffusers' names; "conv_in.weight" registered with its input channels zero-padded to 64):
 * d_sample f32 [height*width][in_channels] -> d_out f32 [height*width][out_channels]; height, width divisible by 2^(n-1). */
int r3g_unet_forward(r3g_ctx* ctx, const float* d_sample, int height, int width, float timestep, const uint16_t* d_ctx, int tokens,
                     float* d_out, void* stream);
/* ---- several samples per call and upstream's 2.5D transformer blocks: the multiview UNet of the texture stage ----
 * [UPSTREAM-RECALLED] hy3dgen/texgen/hunyuanpaint/unet/modules.py: UNet2p5DConditionModel wraps an SD-2.1 UNet2DConditionModel;
 * every BasicTransformerBlock becomes a Basic2p5DTransformerBlock with two more attentions on norm1's output --
 * attn_multiview (self-attention over the tokens of ALL views of the object as one sequence, scaled by mva_scale) and
 * attn_refview (every view's tokens attend to the normalised hidden states a separate "reference" UNet pass over the input
 * image has kept per block, scaled by ref_scale) --, conv_in takes 12 channels (latent | normal map latent | position map
 * latent) and an nn.Embedding over camera indices is added to the time embedding (class_labels).
 * Samples are stacked as rows: d_sample f32 [n_views][height*width][in_channels], d_out likewise; r3g_unet_create's max_hw
 * is then the largest n_views*height*width, ctx_tokens at least the number of reference tokens.  Extra tensors
 * (r3g_unet_set_tensor): "<...>.transformer_blocks.0.attn_multiview.to_qkv.weight" ([3C][C]: to_q | to_k | to_v),
 * ".attn_multiview.to_out.0.{weight,bias}", ".attn_refview.to_q.weight", ".attn_refview.to_kv.weight" (per head 64 k rows,
 * 64 v rows), ".attn_refview.to_out.0.{weight,bias}", "class_embedding.weight" (f32 [classes][temb_dim]), and per transformer
 * "cond:<transformer prefix>" = the reference pass's kept states (bf16 [reference tokens][C], from r3g_unet_condition of the
 * context that ran the reference pass).  Blocks whose 2.5D weights are not registered run as the plain block.
 * flags: 1 = keep norm1's output of every transformer ("w" mode, the reference pass), 2 = run attn_refview where its weights
 * and "cond:" tensor exist ("r" mode), 4 = the n_views samples are the two evaluations of one classifier-free-guidance step in
 * ONE launch set (round 5; n_views even, not with flag 1): samples [0, n_views/2) are the conditional evaluation -- context rows
 * [0, tokens) of d_ctx, attn_refview under flag 2 --, samples [n_views/2, n_views) the unconditional one -- context rows
 * [tokens, 2 tokens) of d_ctx (bf16 [2 tokens][ctx_dim]), no attn_refview --, and attn_multiview's sequence is one HALF's
 * views; every sample's result equals its two-call result up to the order of fp32 additions in the split-K convolutions.
 * max_hw must then hold n_views*height*width rows.  class_labels: host array [n_views] or NULL. */
int r3g_unet_forward_mv(r3g_ctx* ctx, const float* d_sample, int height, int width, float timestep, const uint16_t* d_ctx, int tokens,
                        int n_views, const int32_t* class_labels, int flags, float mva_scale, float ref_scale, float* d_out,
                        void* stream);
/* one Transformer2DModel with the 2.5D block, in place on d_x f32 [n_views][height*width][channels] */
int r3g_unet_transformer_mv(r3g_ctx* ctx, const char* prefix, float* d_x, int height, int width, int channels, const uint16_t* d_ctx,
                            int tokens, int n_views, int flags, float mva_scale, float ref_scale, void* stream);
/* what a pass with flag 1 kept for the transformer `prefix` ("down_blocks.0.attentions.1", "mid_block.attentions.0", ...):
 * bf16 [rows = samples*height*width][cols = channels], owned by the context, valid until its next pass with flag 1 */
int r3g_unet_condition(r3g_ctx* ctx, const char* prefix, const void** d_ptr, int64_t* rows, int64_t* cols);
/* UNetMidBlock2DCrossAttn.forward: resnet, transformer, resnet -> d_out f32 [height*width][channels] */
int r3g_unet_mid_block(r3g_ctx* ctx, const char* prefix, const float* d_x, int height, int width, int channels,
                       const float* d_temb, const uint16_t* d_ctx, int tokens, float* d_out, void* stream);

/* ---- the SD-family VAE (diffusers AutoencoderKL) on the same blocks ------------------------------
 * Replaces, inside upstream's texture pipelines (behind reference src/2d_to_3d_models/run.py:97), `vae.encode(image)` /
 * `vae.decode(latents)` of the delight and multiview diffusion models.  The context is one created by r3g_unet_create with
 * block_out_channels = the ENCODER's order (128, 256, 512, 512), layers_per_block 2, in_channels = latent channels (4),
 * out_channels = image channels (3), groups 32, resnet_eps 1e-6; weights registered with r3g_unet_set_tensor under diffusers'
 * names ("encoder.down_blocks.0.resnets.0.conv1.weight", "decoder.mid_block.attentions.0.to_q.weight", "quant_conv.weight",
 * "post_quant_conv.weight", ...), 3x3 convolutions re-laid as for the UNet, conv_in's input channels and the 1x1 convolutions'
 * K zero-padded to 64, conv_out / quant_conv / post_quant_conv rows zero-padded to a multiple of 4 (python: r3g.unet).
 * The mid block's single-head attention (head dim = channels) is two GEMMs around a row softmax; height*width of the
 * latent grid must be a multiple of 64. */
/* AutoencoderKL.decode(z).sample: d_latent f32 [height*width][latent channels] (already divided by the scaling factor) ->
 * d_image f32 [(8 height)(8 width)][4] (channels 0..2 = RGB in [-1, 1] nominally, channel 3 = 0) */
int r3g_aekl_decode(r3g_ctx* ctx, const float* d_latent, int height, int width, float* d_image, void* stream);
/* AutoencoderKL.encode(x).latent_dist parameters: d_image f32 [height*width][image channels] -> d_moments f32
 * [(height/8)(width/8)][2 latent channels] = (mean | log-variance); the distribution's mode is the mean */
int r3g_aekl_encode(r3g_ctx* ctx, const float* d_image, int height, int width, float* d_moments, void* stream);

/* ---- sampling loops of upstream's diffusion pipelines (the delighting StableDiffusionInstructPix2PixPipeline and the
 * multiview pipeline, both with an EulerAncestralDiscreteScheduler; [UPSTREAM-RECALLED] hy3dgen/texgen/utils/dehighlight_utils.py,
 * hy3dgen/texgen/hunyuanpaint/pipeline.py) around r3g_unet_forward / r3g_unet_forward_mv: the elementwise steps diffusers runs
 * between UNet evaluations.  Stateless (no context). */
/* d_out f32 [pixels][channels + cond_channels] = (d_latent[p] / sqrt(sigma^2 + 1) | d_cond[p]):
 * scheduler.scale_model_input(latents, t) and torch.cat([latents, conditioning latents], dim=1) on rows (InstructPix2Pix: the image
 * latents; the multiview pipeline: the normal- and position-map latents of the view) */
int r3g_sched_model_input(const float* d_latent, int channels, const float* d_cond, int cond_channels, int64_t pixels, float sigma,
                          float* d_out, void* stream);
/* the InstructPix2Pix case: cond_channels = channels */
int r3g_sched_pix2pix_input(const float* d_latent, const float* d_image_latent, int channels, int64_t pixels, float sigma,
                            float* d_out, void* stream);
/* classifier-free guidance: d_out[i] = d_uncond[i] + guidance_scale (d_cond[i] - d_uncond[i]) */
int r3g_sched_cfg_combine(const float* d_uncond, const float* d_cond, int64_t n, float guidance_scale, float* d_out, void* stream);
/* EulerAncestralDiscreteScheduler.step in place on d_sample f32 [n]: prediction_type 0 epsilon | 1 v_prediction;
 * sigma_up^2 = sigma_to^2 (sigma_from^2 - sigma_to^2) / sigma_from^2, sigma_down^2 = sigma_to^2 - sigma_up^2,
 * sample += (sample - x0) / sigma_from * (sigma_down - sigma_from) + d_noise * sigma_up */
int r3g_sched_euler_ancestral_step(float* d_sample, const float* d_model_out, const float* d_noise, int64_t n, float sigma_from,
                                   float sigma_to, int prediction_type, void* stream);

/* ---- single kernels, for parity tests through the ABI ------------------------------------------ */
/* C = epilogue(A[m][k] . W[n][k]^T + bias); epilogue: 0 bf16, 1 bf16 gelu(tanh), 2 bf16 gelu(erf),
 * 3 f32 C += gate*(..), 4 f32.  k % 64 == 0, n % 4 == 0. */
int r3g_op_gemm(const uint16_t* d_a, int64_t lda, const uint16_t* d_w, int64_t ldw, const float* d_bias, void* d_c,
                int64_t ldc, const float* d_gate, int m, int n, int k, int epilogue, int use_lds_dma, void* stream);
/* r3g_op_gemm for the fp32 epilogues (3, 4) with the caller's split-K workspace d_ws (ws_elems floats), as the texture UNets' 3 x 3
 * convolutions launch it (csrc/unet.cpp: u_conv3x3): a problem of at most 256 tiles of 128 x 128 with k >= 2048 runs as S slices of k
 * -- the S batches of one launch of the 128 x 128 kernel into d_ws [S][m][n] -- that a second kernel adds in the order 0 .. S-1
 * before bias / gate / residual (S = the largest divisor of k / 64 with S * tiles <= 512 and >= 8 k-steps per slice; two slices only from k = 4096).  No atomics:
 * the result is a pure function of the operands and (m, n, k).  *slices (may be null) = S, 1 = the ordinary launch. */
int r3g_op_gemm_splitk(const uint16_t* d_a, int64_t lda, const uint16_t* d_w, int64_t ldw, const float* d_bias, float* d_c, int64_t ldc,
                       const float* d_gate, int m, int n, int k, int epilogue, float* d_ws, int64_t ws_elems, int* slices,
                       void* stream);
/* softmax(Q K^T / 8) V for head_dim 64: Q bf16 [B][H][lq_pad][64] (plain q: this entry point folds the softmax scale
 * in itself), K bf16 [B][H][lk_pad][64], Vt bf16 [B][H][64][lk_pad] -> O bf16 [B][lq][H*64].
 * Vt holds V transposed with the keys of each row in the kernel's operand order: inside every aligned group of 16 keys
 * the two middle 4-key blocks are swapped (key 8g+4h+e at position 8h+4g+e; python: r3g.layout.make_vt). */
int r3g_op_attention(const uint16_t* d_q, const uint16_t* d_k, const uint16_t* d_vt, uint16_t* d_o, int batch, int heads,
                     int lq, int lq_pad, int lk, int lk_pad, int shared_kv, int use_lds_dma, void* stream);
/* FP8 operands (BASELINE.json configs[3]).  quant_fp8: bf16 [rows][k] -> OCP e4m3 bytes [rows][k] + one fp32 scale per row
 * (amax / 448; round to nearest even, saturating).  gemm_fp8: C = epilogue((scale_a[m] scale_w[n]) sum_k a8[m][k] w8[n][k] +
 * bias) on v_mfma_scale_f32_16x16x128_f8f6f4 (twice the bf16 matrix rate); k % 256 == 0, lda / ldw in bytes and multiples
 * of 16, epilogue one of 0, 1, 2 (bf16 out), 3 (fp32 residual), 6 (bf16 residual).  Test / benchmark hooks; the model uses
 * the same kernels for the geo decoder under r3g_set_option("geo_fp8", 1). */
int r3g_op_quant_fp8(const uint16_t* d_x, int64_t ldx, int rows, int k, uint8_t* d_q, int64_t ldq, float* d_scale, void* stream);
int r3g_op_gemm_fp8(const uint8_t* d_a8, int64_t lda, const float* d_scale_a, const uint8_t* d_w8, int64_t ldw,
                    const float* d_scale_w, const float* d_bias, void* d_c, int64_t ldc, const float* d_gate, int m, int n,
                    int k, int epilogue, void* stream);
/* Per-kernel-family timing with HIP events on the launch stream (bench.py's roofline leg).  Families, in order:
 * 0 gemm, 1 attention, 2 layernorm, 3 qkv_split, 4 gemv, 5 elementwise, 6 mc_classify, 7 mc_other, 8 mesh
 * cleaners (n >= 9).  work = algorithmic FLOPs (0,1,4) / bytes (6,8) summed over the launches since r3g_prof_enable(1). */
int r3g_prof_enable(int on);
int r3g_prof_read(int64_t* counts, double* ms, double* work, int n);
/* per family: algorithmic bytes (operands read once, result written once) of the launches whose `work` is FLOPs */
int r3g_prof_read_bytes(double* bytes, int n);
/* A/B switches for tests and ablations.  Default 1: "fuse_qkv" (QKV split/norm/transpose in the projection epilogue
 * vs a separate kernel), "batch_mods" (all adaLN modulations of a forward in one GEMV launch), "lds_dma"
 * (= r3g_set_staging), "cfg_dedup" (one weighted token for a uniform unconditional context), "group_streams" (img and
 * txt stream of a double block in one GEMM / LayerNorm launch), "geo_q_cache" (1: the part of the geo decoder that does not depend on the object -- Fourier features, query_proj, ln_1, c_q,
 * q-norm of every grid point -- stays resident in HBM after its first evaluation, 2 x 34.8 GB at 257^3, and is read instead of
 * recomputed; bit-identical; skipped by itself when the memory is not there), "skip_zero_step" (the DiT evaluation of a step with d_sigma = 0 --
 * upstream's last step -- is skipped: x += 0 * v), "overlap_mlp" (0 default | 1: MLP half of a single block's linear1
 * on a second stream beside the attention kernel), "gemm_wide_epilogue" (stores through the LDS transpose).
 * Tuning: "gemm_waves" (0 auto | 4 | 8 | 9 = 256x256 two-stage | 10 = 256x128 | 11 = 256x256 phased | 12 = phased,
 * persistent grid | 13 = phased, deterministic split-K over two workgroups per tile | 16 | 32 = deep ring), "gemm_raster" (-1 auto | tile columns per rasterisation group), "gemm_phased"
 * (1: 256x256 tiles on the phased counted-vmcnt kernel), "gemm_persistent" (1: its persistent form for bf16 outputs
 * with more tiles than CUs), "gemm_xcd_walk" (1 default: in the persistent form an XCD's workgroups walk ONE contiguous range of the
 * rasterised tile order (consecutive steps touch neighbouring A rows) -- geo c_fc 1 109 -> 1 045 us, L2 <-> fabric bytes unchanged; bit-identical | 0: rounds 2-4's walk), "gemm_persistent_resid" (0 | bit 0: the fp32, bit 1: the bf16 read-modify-write epilogue on the persistent form as well), "gemm_num_cu" (CUs the tile rules assume, default 256), "gemm_auto_rule" (2: the current
 * tile-choice rule | 1: round 2's | 0: round 1's first version), "attn_generation" (7 default: 6 where its
 * 256-query workgroups make four rounds of the device, otherwise 2 | 2 = four waves of 32 queries | 6 = four waves of 64
 * queries, bit-identical to 2 | 1 = the first-round kernel | 3, 4, 5 = pipelined / 8-wave variants), "attn_wide_min" (2048: work items of 256 queries from which attn_generation 7 takes generation 6), "ln_rows" (0 automatic | 1 | 4 rows per wave in the
 * LayerNorm / ln_dot row kernels), "ln_rows4_min" (65536: launches of at least this many rows take 4 rows per wave under the automatic rule), "ln_fixed" (1: their instantiations with a compile-time row length for C = 1024 / 1536), "ln_modes" (1, round 6: for C = 1024 the affine-only and the modulation-only launches take instantiations with that decided at compile time -- bit-identical | 0: the generic kernel), "attn_pipelined" (0), "attn_ablate"
 * (timing-only masks, results are garbage), "floater_by_vertex" (0), "mc_rows" (4 | 8 | 16 | 32 node rows per wave in the marching-cubes row
 * kernel), "mc_deferred" (1: tiling selection batched per wave | 0: round 1's per-row kernel), "geo_resid_bf16" (1:
 * 16-bit residual stream in the geo decoder block), "geo_fp8" (0 default | 1: the geo decoder's c_q and MLP GEMMs on e4m3
 * operands | 2: MLP only | 3: c_q only -- a different precision, NOT result-preserving), "gemm_splitk" (0), "conv_implicit" (1 default: the 3 x 3 convolutions of the texture models run as implicit GEMMs -- the 128 x 128
 * kernel gathers the nine shifted rows of its A operand from the activation rows itself, bit-identical to the GEMM over the im2col
 * matrix | 0: im2col + GEMM, rounds 3-4), "gemm_splitk128" (1 default: the 3 x 3 convolutions of the texture models split a deep k over
 * several workgroups where their grid would fill less than half of the chip, r3g_op_gemm_splitk | 0: one workgroup per tile walks all of k, rounds 2-4 -- another
 * order of the fp32 additions, not bit-preserving), "geo_q_cache_gb" (the
 * budget of geo_q_cache in GiB; < 0, the default: 30 % of the device's memory; a grid that needs more gets a prefix of its passes
 * cached), "dit_resid_f16" (1 default: the residual stream of the DiT's de-duplicated CFG path in fp16 -- the reference's own
 * activation type -- | 0: in fp32, rounds 1-3; NOT bit-preserving: 50-step latents at full depth 3.3e-3 against 3.0e-3 from the
 * fp32 oracle, tests/test_cfg1_golden_gpu.py; -21 ms per object), "gelu_pk" (1 default: the GELU epilogues of the GEMMs evaluate x S(x) with S in packed fp16 -- csrc/gemm_common.h: a
 * degree-6 polynomial on v_pk_fma_f16, |error of S| <= 7.5e-4 as evaluated in fp16 (6.5e-4 tanh / 7.2e-4 erf measured; the fit alone 1.2e-4), +1.2 % on the rel-L2 error behind the bf16 rounding of the output |
 * 0: rounds 3-4's fp32 forms with v_exp_f32 / v_rcp_f32; NOT bit-preserving), attn_generation 9 (opt-in, round 5: the phased 12-wave kernel -- three groups of four
 * waves one phase apart, a wave's matrix phase under the two softmax phases of its SIMD neighbours; bit-identical to generations 2
 * and 6, measured slower: profiles/r05_attention_phases.md), "attn_prio" (generation 9: 0 no s_setprio | 1 its matrix phase at priority 1 | 2 its softmax phases), "attn_stamps" (0 | 1: attn_generation 9 prints the s_memtime ticks of its
 * phases, work and barrier wait, per launch to stderr -- timing experiments, synchronous), "dit_f16_guard" (1 default: the final latents of a launch group that ran
 * on the fp16 stream are checked for NaN / infinity and a group that overflowed runs again on the fp32 stream, with a line on
 * stderr | 0: no check),
 * "attn_variant" (round 6; generations 2 / 6 / 7: 1 default -- the FAST pass takes no maximum after the first key block and tests
 * the sum of a lane's 16 exponentials against 2^16 into a sticky flag; a workgroup whose valid queries set it runs its query tile
 * again with variant 0's body | 0 = rounds 2-5.  Bit-identical to variant 0 unless variant 0 would have moved its stabiliser where
 * the fast pass does not: then equal within the bf16 rounding of P),
 * "geo_ln3_fold" (1 default, round 6 | 0: the geo decoder's ln_3 as its own launch writing a normalised bf16 copy of the stream, rounds
 * 1-5; folded, c_proj's epilogue also writes the rows' chunk statistics, and c_fc runs on the raw stream with W' = bf16(W gamma) and
 * rstd (acc - mean c1) + c2 in front of its GELU -- the same function without the bf16 rounding of the normalised operand: logits move
 * by ~1e-3 of their scale, inside the 1e-2 tolerance against the fp32 oracle),
 * "geo_lnd_fused" (1 default, round 6 | 0: the geo decoder's ln_post + output_proj as their own launch over the stored residual
 * stream, rounds 1-5; fused, the last residual GEMM's epilogue writes per-row statistics of its 64-column chunks instead of the
 * stream and a small kernel merges them -- same function, another summation order: logits equal to ~1e-6 of their scale),
 * "gemm_epi_slices" (1 default, round 6 | 0: the persistent phased kernel's bf16 / fused-QKV epilogues in 32-row passes through 4 KiB of
 * extra scratch per wave instead of 64-row passes through the wave's own staging slices of the idle k-tile buffer), "gemm_mixed" (1
 * default, round 6 | 0: a DiT single block's fused QKV projection and MLP-in + GELU projection as two launches instead of one
 * persistent launch over both problems' tiles), "gemm_persistent_qkv" (1 default since round 6 | 0: fused QKV launches with more 256x256 tiles than CUs -- the double
 * blocks' img + txt pair -- one tile per workgroup instead of the persistent phased kernel; with round 4's 32-row epilogue passes the
 * persistent form was 17 ms per object slower, with "gemm_epi_slices" it is 4 ms faster, profiles/r06_ab.md), "flow_first_step" / "flow_last_step" (0 / -1: r3g_flow_sample runs steps [first, last) of its schedule; consecutive
 * segments continuing on each other's latents are the same launches as one call -- how tests read the latents after 10, 20, ...
 * of 50 steps).  None of them changes a
 * result bit, except fuse_qkv / batch_mods / cfg_dedup (different summation order, same function) and attn_generation
 * (different rounding points inside the softmax). */
int r3g_set_option(const char* name, int value);
/* Process-wide event counters (round 6).  "dit_f16_fallbacks": launch groups of r3g_flow_sample_batch whose fp16 residual stream
 * produced non-finite latents and that therefore ran a second time on the fp32 stream ("dit_f16_guard"; such a group costs twice its
 * time -- bench.py and the stage report carry the count so that a slow run says why).  "dit_groups": launch groups run so far.
 * r3g_flow_sample_batch is SYNCHRONOUS while the guard is on (one 4-byte read-back per group) and must not be captured into a
 * hipGraph then; with "dit_f16_guard" 0 or "dit_resid_f16" 0 it only enqueues work.  Unknown name: R3G_ERR_INVALID. */
int r3g_get_counter(const char* name, int64_t* value);
/* operand staging of the MFMA kernels: 1 = LDS-DMA (global_load_lds, default), 0 = through registers */
int r3g_set_staging(int use_lds_dma);

#ifdef __cplusplus
}
#endif
#endif
