// r3g_api.cpp -- C ABI (include/r3g.h) over the HIP kernels.  Compiled with hipcc.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/r3g.h"
#include "r3g_ctx.h"
#include "mesh_kernels.h"
#include "tex_kernels.h"

namespace r3g {
static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(R3G_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

int Ctx::reserve(char** buf, size_t* have, size_t need, const char* what) {
    if (*have >= need) return R3G_OK;
    if (*buf) {
        hipError_t e = hipFree(*buf);
        *buf = nullptr;
        *have = 0;
        if (e != hipSuccess) return hip_fail(e, what);
    }
    hipError_t e = hipMalloc((void**)buf, need);
    if (e != hipSuccess) return hip_fail(e, what);
    *have = need;
    return R3G_OK;
}
}  // namespace r3g

using namespace r3g;

extern "C" {

int r3g_version(void) { return R3G_VERSION; }
const char* r3g_last_error(void) { return g_err; }

int r3g_create(int device, r3g_ctx** out) {
    if (!out) return fail(R3G_ERR_INVALID, "r3g_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(R3G_ERR_NO_DEVICE, "r3g_create: no HIP device visible (%s); libr3g has no CPU path",
                    e == hipSuccess ? "count=0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(R3G_ERR_INVALID, "r3g_create: device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(R3G_ERR_NO_DEVICE, "r3g_create: device %d is %s, libr3g is built for gfx950 only", device,
                    prop.gcnArchName);
    e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    Ctx* c = new (std::nothrow) Ctx();
    if (!c) return fail(R3G_ERR_HIP, "out of host memory");
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    gemm_set_auto_rule(-1, c->num_cu);
    e = hipHostMalloc((void**)&c->h_small, 64, hipHostMallocDefault);
    if (e != hipSuccess) {
        delete c;
        return hip_fail(e, "hipHostMalloc");
    }
    *out = reinterpret_cast<r3g_ctx*>(c);
    return R3G_OK;
}

void r3g_destroy(r3g_ctx* ctx) {
    if (!ctx) return;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    (void)hipSetDevice(c->device);
    if (c->mc_ws) (void)hipFree(c->mc_ws);
    if (c->mesh_ws) (void)hipFree(c->mesh_ws);
    if (c->tex_ws) (void)hipFree(c->tex_ws);
    if (c->h_small) (void)hipHostFree(c->h_small);
    c->release_model();
    c->release_unet();
    delete c;
}

int r3g_mc_count(r3g_ctx* ctx, const float* d_grid, int n0, int n1, int n2, double level, int use_classic,
                 int64_t* n_verts, int64_t* n_faces, void* stream) {
    if (!ctx || !d_grid || !n_verts || !n_faces) return fail(R3G_ERR_INVALID, "r3g_mc_count: null argument");
    if (n0 < 2 || n1 < 2 || n2 < 2) return fail(R3G_ERR_INVALID, "Input array must be at least 2x2x2.");
    const uint64_t nnodes = (uint64_t)n0 * n1 * n2;
    if (nnodes * 3 >= (1ull << 31)) return fail(R3G_ERR_INVALID, "r3g_mc_count: grid too large for int32 vertex ids");
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    hipStream_t s = (hipStream_t)stream;
    c->mc_counted = false;
    McWorkspaceLayout lay;
    const size_t need = mc_workspace_bytes(n0, n1, n2, &lay);
    int rc = c->reserve(&c->mc_ws, &c->mc_ws_bytes, need, "hipMalloc(mc workspace)");
    if (rc) return rc;
    hipError_t e = mc_count_launch(d_grid, n0, n1, n2, level, use_classic, c->mc_ws, lay, s);
    if (e != hipSuccess) return hip_fail(e, "mc_count_launch");
    e = hipMemcpyAsync(c->h_small, c->mc_ws + lay.off_small, 64, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(mc totals)");
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize(mc count)");
    const unsigned status = *(const unsigned*)c->h_small;
    const unsigned long long nv = ((const unsigned long long*)c->h_small)[2];
    const unsigned long long nf = ((const unsigned long long*)c->h_small)[3];
    *n_verts = (int64_t)nv;
    *n_faces = (int64_t)nf;
    // numpy: level < vol.min() or level > vol.max(); a NaN anywhere makes both comparisons false
    if (!(status & 4u) && (!(status & 1u) || !(status & 2u)))
        return fail(R3G_ERR_LEVEL_RANGE, "Surface level must be within volume data range.");
    if (nv == 0) return fail(R3G_ERR_NO_SURFACE, "No surface found at the given iso value.");
    c->mc_lay = lay;
    c->mc_lay.nnz = (uint32_t)((const unsigned long long*)c->h_small)[4];
    c->mc_grid = d_grid;
    c->mc_n[0] = n0; c->mc_n[1] = n1; c->mc_n[2] = n2;
    c->mc_level = level;
    c->mc_counted = true;
    return R3G_OK;
}

int r3g_mc_emit(r3g_ctx* ctx, float* d_verts, int32_t* d_faces, const double* xform, int reverse_faces,
                void* stream) {
    if (!ctx || !d_verts || !d_faces) return fail(R3G_ERR_INVALID, "r3g_mc_emit: null argument");
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c->mc_counted) return fail(R3G_ERR_STATE, "r3g_mc_emit: no successful r3g_mc_count precedes this call");
    hipError_t e = mc_emit_launch(c->mc_grid, c->mc_n[0], c->mc_n[1], c->mc_n[2], c->mc_level, c->mc_ws, c->mc_lay,
                                  d_verts, d_faces, xform, reverse_faces, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "mc_emit_launch");
    return R3G_OK;
}

// ---- mesh cleaners -------------------------------------------------------------------------------------------
static int mesh_args(const char* who, r3g_ctx* ctx, float* d_verts, int64_t* n_verts, int32_t* d_faces,
                     int64_t* n_faces) {
    if (!ctx || !n_verts || !n_faces) return fail(R3G_ERR_INVALID, "%s: null argument", who);
    if (*n_verts < 0 || *n_faces < 0 || *n_verts >= (1ll << 31) || *n_faces >= (1ll << 30))
        return fail(R3G_ERR_INVALID, "%s: mesh size out of range", who);
    if ((*n_verts && !d_verts) || (*n_faces && !d_faces)) return fail(R3G_ERR_INVALID, "%s: null buffer", who);
    return R3G_OK;
}

int r3g_mesh_remove_floaters(r3g_ctx* ctx, float* d_verts, int64_t* n_verts, int32_t* d_faces, int64_t* n_faces,
                             double min_ratio, void* stream) {
    int rc = mesh_args("r3g_mesh_remove_floaters", ctx, d_verts, n_verts, d_faces, n_faces);
    if (rc) return rc;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    rc = c->reserve(&c->mesh_ws, &c->mesh_ws_bytes, mesh_workspace_bytes(*n_verts, *n_faces, 0), "hipMalloc(mesh workspace)");
    if (rc) return rc;
    hipError_t e = mesh_remove_floaters(c->mesh_ws, c->mesh_ws_bytes, (unsigned*)c->h_small, d_verts, n_verts, d_faces,
                                        n_faces, min_ratio, (hipStream_t)stream);
    return e == hipSuccess ? R3G_OK : hip_fail(e, "mesh_remove_floaters");
}

int r3g_mesh_remove_degenerate(r3g_ctx* ctx, float* d_verts, int64_t* n_verts, int32_t* d_faces, int64_t* n_faces,
                               void* stream) {
    int rc = mesh_args("r3g_mesh_remove_degenerate", ctx, d_verts, n_verts, d_faces, n_faces);
    if (rc) return rc;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    rc = c->reserve(&c->mesh_ws, &c->mesh_ws_bytes, mesh_workspace_bytes(*n_verts, *n_faces, 0), "hipMalloc(mesh workspace)");
    if (rc) return rc;
    hipError_t e = mesh_remove_degenerate(c->mesh_ws, c->mesh_ws_bytes, (unsigned*)c->h_small, d_verts, n_verts, d_faces,
                                          n_faces, (hipStream_t)stream);
    return e == hipSuccess ? R3G_OK : hip_fail(e, "mesh_remove_degenerate");
}

int r3g_mesh_reduce_faces(r3g_ctx* ctx, float* d_verts, int64_t* n_verts, int32_t* d_faces, int64_t* n_faces,
                          int64_t max_faces, void* stream) {
    int rc = mesh_args("r3g_mesh_reduce_faces", ctx, d_verts, n_verts, d_faces, n_faces);
    if (rc) return rc;
    if (max_faces < 1 || max_faces > 200000000) return fail(R3G_ERR_INVALID, "r3g_mesh_reduce_faces: max_faces out of range");
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    rc = c->reserve(&c->mesh_ws, &c->mesh_ws_bytes, mesh_workspace_bytes(*n_verts, *n_faces, 0), "hipMalloc(mesh workspace)");
    if (rc) return rc;
    hipError_t e = mesh_reduce_faces(c->mesh_ws, c->mesh_ws_bytes, (unsigned*)c->h_small, d_verts, n_verts, d_faces,
                                     n_faces, max_faces, nullptr, (hipStream_t)stream);
    return e == hipSuccess ? R3G_OK : hip_fail(e, "mesh_reduce_faces");
}

int r3g_mesh_cluster_faces(r3g_ctx* ctx, float* d_verts, int64_t* n_verts, int32_t* d_faces, int64_t* n_faces,
                           int64_t max_faces, void* stream) {
    int rc = mesh_args("r3g_mesh_cluster_faces", ctx, d_verts, n_verts, d_faces, n_faces);
    if (rc) return rc;
    if (max_faces < 1 || max_faces > 200000000) return fail(R3G_ERR_INVALID, "r3g_mesh_cluster_faces: max_faces out of range");
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    const int64_t r = mesh_reduce_initial_res(max_faces);
    rc = c->reserve(&c->mesh_ws, &c->mesh_ws_bytes, mesh_workspace_bytes(*n_verts, *n_faces, r * r * r),
                    "hipMalloc(mesh workspace)");
    if (rc) return rc;
    hipError_t e = mesh_cluster_faces(c->mesh_ws, c->mesh_ws_bytes, (unsigned*)c->h_small, d_verts, n_verts, d_faces,
                                      n_faces, max_faces, (hipStream_t)stream);
    return e == hipSuccess ? R3G_OK : hip_fail(e, "mesh_cluster_faces");
}

// ---- texture stage ------------------------------------------------------------------------------------------------
int r3g_tex_rasterize(r3g_ctx* ctx, const float* d_pos_clip, int64_t n_verts, const int32_t* d_tri, int64_t n_faces, int height,
                      int width, int32_t* d_findices, float* d_bary, void* stream) {
    if (!ctx || !d_findices || !d_bary) return fail(R3G_ERR_INVALID, "r3g_tex_rasterize: null argument");
    if (height < 1 || width < 1 || height > 16384 || width > 16384)
        return fail(R3G_ERR_INVALID, "r3g_tex_rasterize: image size out of range");
    if (n_verts < 0 || n_faces < 0 || n_faces >= (1ll << 30) || (n_faces && (!d_pos_clip || !d_tri)))
        return fail(R3G_ERR_INVALID, "r3g_tex_rasterize: bad mesh arguments");
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    int rc = c->reserve(&c->tex_ws, &c->tex_ws_bytes, 8 * (size_t)height * width, "hipMalloc(z-buffer)");
    if (rc) return rc;
    hipError_t e = tex_rasterize(d_pos_clip, d_tri, (int)n_faces, height, width, (unsigned long long*)c->tex_ws, d_findices,
                                 d_bary, (hipStream_t)stream);
    return e == hipSuccess ? R3G_OK : hip_fail(e, "tex_rasterize");
}

int r3g_tex_interpolate(r3g_ctx* ctx, const float* d_attr, int channels, const int32_t* d_tri, const int32_t* d_findices,
                        const float* d_bary, int64_t n_pixels, float* d_out, void* stream) {
    if (!ctx || !d_attr || !d_tri || !d_findices || !d_bary || !d_out) return fail(R3G_ERR_INVALID, "r3g_tex_interpolate: null argument");
    if (channels < 1 || channels > 64 || n_pixels < 0) return fail(R3G_ERR_INVALID, "r3g_tex_interpolate: bad sizes");
    hipError_t e = tex_interpolate(d_attr, channels, d_tri, d_findices, d_bary, n_pixels, d_out, (hipStream_t)stream);
    return e == hipSuccess ? R3G_OK : hip_fail(e, "tex_interpolate");
}

int r3g_tex_view_weight(r3g_ctx* ctx, const int32_t* d_findices, const float* d_depth, const float* d_normal, int height,
                        int width, float cos_threshold, float depth_edge, float view_weight, float power, float* d_weight,
                        void* stream) {
    if (!ctx || !d_findices || !d_depth || !d_normal || !d_weight) return fail(R3G_ERR_INVALID, "r3g_tex_view_weight: null argument");
    if (height < 1 || width < 1) return fail(R3G_ERR_INVALID, "r3g_tex_view_weight: bad sizes");
    hipError_t e = tex_view_weight(d_findices, d_depth, d_normal, height, width, cos_threshold, depth_edge, view_weight, power,
                                   d_weight, (hipStream_t)stream);
    return e == hipSuccess ? R3G_OK : hip_fail(e, "tex_view_weight");
}

int r3g_tex_bake(r3g_ctx* ctx, const float* d_image, const float* d_weight, const int32_t* d_findices, const float* d_bary,
                 const float* d_uv, const int32_t* d_uv_tri, int64_t n_pixels, int tex_size, uint64_t* d_acc, void* stream) {
    if (!ctx || !d_image || !d_weight || !d_findices || !d_bary || !d_uv || !d_uv_tri || !d_acc)
        return fail(R3G_ERR_INVALID, "r3g_tex_bake: null argument");
    if (tex_size < 1 || tex_size > 16384 || n_pixels < 0) return fail(R3G_ERR_INVALID, "r3g_tex_bake: bad sizes");
    hipError_t e = tex_bake(d_image, d_weight, d_findices, d_bary, d_uv, d_uv_tri, n_pixels, tex_size, (unsigned long long*)d_acc,
                            (hipStream_t)stream);
    return e == hipSuccess ? R3G_OK : hip_fail(e, "tex_bake");
}

int r3g_tex_bake_gather(r3g_ctx* ctx, const int32_t* d_findices_uv, const float* d_bary_uv, const float* d_clip_uv,
                        const int32_t* d_uv_tri, int tex_size, const float* d_image, const float* d_weight,
                        const int32_t* d_findices, const float* d_depth, int height, int width, float depth_eps, uint64_t* d_acc,
                        void* stream) {
    if (!ctx || !d_findices_uv || !d_bary_uv || !d_clip_uv || !d_uv_tri || !d_image || !d_weight || !d_findices || !d_depth || !d_acc)
        return fail(R3G_ERR_INVALID, "r3g_tex_bake_gather: null argument");
    if (tex_size < 1 || tex_size > 16384 || height < 1 || width < 1) return fail(R3G_ERR_INVALID, "r3g_tex_bake_gather: bad sizes");
    hipError_t e = tex_bake_gather(d_findices_uv, d_bary_uv, d_clip_uv, d_uv_tri, tex_size, d_image, d_weight, d_findices, d_depth,
                                   height, width, depth_eps, (unsigned long long*)d_acc, (hipStream_t)stream);
    return e == hipSuccess ? R3G_OK : hip_fail(e, "tex_bake_gather");
}

int r3g_tex_bake_finalize(r3g_ctx* ctx, const uint64_t* d_acc, int tex_size, float* d_texture, uint8_t* d_mask, void* stream) {
    if (!ctx || !d_acc || !d_texture || !d_mask) return fail(R3G_ERR_INVALID, "r3g_tex_bake_finalize: null argument");
    if (tex_size < 1 || tex_size > 16384) return fail(R3G_ERR_INVALID, "r3g_tex_bake_finalize: bad sizes");
    hipError_t e = tex_bake_finalize((const unsigned long long*)d_acc, tex_size, d_texture, d_mask, (hipStream_t)stream);
    return e == hipSuccess ? R3G_OK : hip_fail(e, "tex_bake_finalize");
}

int r3g_tex_inpaint(r3g_ctx* ctx, float* d_texture, uint8_t* d_mask, int tex_size, const int32_t* d_findices_uv,
                    const float* d_bary_uv, const float* d_verts, int64_t n_verts, const int32_t* d_pos_tri, const float* d_uv,
                    const int32_t* d_uv_tri, int64_t n_faces, int dilate_iters, int* rounds_out, void* stream) {
    if (!ctx || !d_texture || !d_mask || !d_findices_uv || !d_bary_uv || !d_verts || !d_pos_tri || !d_uv || !d_uv_tri)
        return fail(R3G_ERR_INVALID, "r3g_tex_inpaint: null argument");
    if (tex_size < 1 || tex_size > 16384 || n_verts < 1 || n_faces < 1 || n_verts >= (1ll << 31) || n_faces >= (1ll << 30) ||
        dilate_iters < 0 || dilate_iters > 4096)
        return fail(R3G_ERR_INVALID, "r3g_tex_inpaint: bad sizes");
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    int rc = c->reserve(&c->tex_ws, &c->tex_ws_bytes, tex_inpaint_workspace(n_verts, tex_size), "hipMalloc(inpaint workspace)");
    if (rc) return rc;
    hipError_t e = tex_inpaint(c->tex_ws, (unsigned*)c->h_small, d_texture, d_mask, tex_size, d_findices_uv, d_bary_uv, d_verts,
                               n_verts, d_pos_tri, d_uv, d_uv_tri, n_faces, dilate_iters, rounds_out, (hipStream_t)stream);
    return e == hipSuccess ? R3G_OK : hip_fail(e, "tex_inpaint");
}

}  // extern "C"
