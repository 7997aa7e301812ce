// tex_kernels.h -- host-side interface of tex_kernels.hip (internal to libr3g.so): the native pieces of the texture stage
#ifndef R3G_TEX_KERNELS_H
#define R3G_TEX_KERNELS_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace r3g {

// z-buffer rasteriser.  zbuf: H*W 64-bit words of workspace.
hipError_t tex_rasterize(const float* pos4, const int32_t* tri, int F, int H, int W, unsigned long long* zbuf,
                         int32_t* findices, float* bary, hipStream_t s);
hipError_t tex_interpolate(const float* attr, int C, const int32_t* tri, const int32_t* findices, const float* bary,
                           int64_t npix, float* out, hipStream_t s);
hipError_t tex_view_weight(const int32_t* findices, const float* depth, const float* normal, int H, int W,
                           float cos_thresh, float depth_edge, float view_weight, float power, float* weight,
                           hipStream_t s);
hipError_t tex_bake(const float* image, const float* weight, const int32_t* findices, const float* bary, const float* uv,
                    const int32_t* uv_tri, int64_t npix, int T, unsigned long long* acc, hipStream_t s);
hipError_t tex_bake_gather(const int32_t* findices_uv, const float* bary_uv, const float* clip_uv, const int32_t* uv_tri, int T,
                           const float* image, const float* weight, const int32_t* findices, const float* depth, int H, int W,
                           float depth_eps, unsigned long long* acc, hipStream_t s);
hipError_t tex_bake_finalize(const unsigned long long* acc, int T, float* tex, uint8_t* mask, hipStream_t s);

// bytes of workspace tex_inpaint needs for V vertices and a T x T texture
size_t tex_inpaint_workspace(int64_t V, int T);
// vertex colours from the painted texels -> propagation over mesh edges -> unpainted covered texels from the vertices ->
// dilation into the gutter.  Synchronises the stream (the propagation loop reads a flag).  *rounds_out (optional).
hipError_t tex_inpaint(char* ws, unsigned* h_flag, float* tex, uint8_t* mask, int T, const int32_t* findices_uv,
                       const float* bary_uv, const float* verts, int64_t V, const int32_t* pos_tri, const float* uv,
                       const int32_t* uv_tri, int64_t F, int dilate_iters, int* rounds_out, hipStream_t s);

}  // namespace r3g
#endif
