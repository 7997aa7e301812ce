// tex_kernels.hip -- native pieces of the texture stage on gfx950 (HBM / atomic-bound pixel and texel work, no MFMA).
//
// upstream's Hunyuan3DPaintPipeline (reference call site src/2d_to_3d_models/run.py:97,126-128)
// leans on two native extensions, `custom_rasterizer` (CUDA: rasterize / interpolate) and `mesh_processor.cpp`
// (meshVerticeInpaint), plus the texture baking of its MeshRender.  Neither source is in /root/reference (the Hunyuan3D-2
// submodule is empty), so these kernels follow the published behaviour as recalled and are checked against a numpy
// restatement written for this repo (tex_ref.py in the test oracle directory; parity unpinned).
//
//   tex_rasterize      clip-space triangles -> per pixel {face id + 1, perspective-correct barycentrics}; z-buffer by a 64-bit
//                      atomicMin on (depth bits << 32 | face id + 1): nearest wins, ties go to the smaller face id, so the
//                      image does not depend on scheduling.  One wave per face, lanes over its bounding box.
//   tex_interpolate    per-pixel attribute = sum of barycentric * per-corner attribute (any index array: positions, uv).
//   tex_view_weight    per-pixel baking weight: view_weight * cos^power, zero below a cosine threshold and on depth edges.
//   tex_bake_gather    texel-centric baking (the pipeline's default): each covered texel projects itself into the view,
//                      tests visibility against the view's depth buffer and samples the image bilinearly; no atomics.
//   tex_bake           scatter a view's colours into texel accumulators (64-bit fixed point, integer atomics: exact and
//                      order independent) ; tex_bake_finalize divides.
//   tex_inpaint        vertex colours from painted texels (smallest corner id wins) -> propagation over mesh edges with
//                      inverse-square-distance weights, one Jacobi round per launch -> unpainted covered texels from the
//                      vertex colours -> dilation into the gutter between charts.
// Built with -ffp-contract=off so that the float32 arithmetic is reproducible operation by operation.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tex_kernels.h"

#pragma clang fp contract(off)

namespace r3g {
namespace {

constexpr unsigned long long kEmpty = ~0ull;

struct ScreenTri {
    float x[3], y[3], z[3], w[3];
    float area;
    bool ok;
};

__device__ __forceinline__ ScreenTri screen_tri(const float* __restrict__ pos4, const int32_t* __restrict__ tri, int f, int H,
                                                int W) {
    ScreenTri t;
    t.ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* p = pos4 + 4 * (int64_t)tri[3 * (int64_t)f + k];
        const float w = p[3];
        t.ok = t.ok && w > 0.f;
        t.x[k] = (p[0] / w * 0.5f + 0.5f) * (float)(W - 1) + 0.5f;
        t.y[k] = (p[1] / w * 0.5f + 0.5f) * (float)(H - 1) + 0.5f;
        t.z[k] = p[2] / w * 0.49999f + 0.5f;
        t.w[k] = w;
    }
    t.area = (t.x[1] - t.x[0]) * (t.y[2] - t.y[0]) - (t.x[2] - t.x[0]) * (t.y[1] - t.y[0]);
    t.ok = t.ok && t.area != 0.f && t.area == t.area;
    return t;
}

// screen-space barycentrics of the pixel centre (px, py); returns false outside
__device__ __forceinline__ bool bary_at(const ScreenTri& t, float px, float py, float (&b)[3]) {
    b[0] = ((t.x[1] - px) * (t.y[2] - py) - (t.x[2] - px) * (t.y[1] - py)) / t.area;
    b[1] = ((t.x[2] - px) * (t.y[0] - py) - (t.x[0] - px) * (t.y[2] - py)) / t.area;
    b[2] = 1.f - b[0] - b[1];
    return b[0] >= 0.f && b[1] >= 0.f && b[2] >= 0.f;
}

__global__ __launch_bounds__(256) void fill_u64(unsigned long long* p, int64_t n, unsigned long long v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ __launch_bounds__(64) void raster_faces(const float* __restrict__ pos4, const int32_t* __restrict__ tri, int F,
                                                   int H, int W, unsigned long long* __restrict__ zbuf) {
    const int f = blockIdx.x;
    if (f >= F) return;
    const ScreenTri t = screen_tri(pos4, tri, f, H, W);
    if (!t.ok) return;
    const float minx = fminf(t.x[0], fminf(t.x[1], t.x[2])), maxx = fmaxf(t.x[0], fmaxf(t.x[1], t.x[2]));
    const float miny = fminf(t.y[0], fminf(t.y[1], t.y[2])), maxy = fmaxf(t.y[0], fmaxf(t.y[1], t.y[2]));
    if (!(maxx >= 0.f && maxy >= 0.f && minx <= (float)W && miny <= (float)H)) return;
    const int ix0 = max(0, (int)floorf(minx) - 1), ix1 = min(W - 1, (int)floorf(maxx) + 1);
    const int iy0 = max(0, (int)floorf(miny) - 1), iy1 = min(H - 1, (int)floorf(maxy) + 1);
    if (ix1 < ix0 || iy1 < iy0) return;
    const int bw = ix1 - ix0 + 1;
    const int64_t n = (int64_t)bw * (iy1 - iy0 + 1);
    for (int64_t i = threadIdx.x; i < n; i += 64) {
        const int ix = ix0 + (int)(i % bw), iy = iy0 + (int)(i / bw);
        float b[3];
        if (!bary_at(t, (float)ix + 0.5f, (float)iy + 0.5f, b)) continue;
        const float depth = b[0] * t.z[0] + b[1] * t.z[1] + b[2] * t.z[2];
        if (!(depth >= 0.f && depth <= 1.f)) continue;
        const unsigned long long token = ((unsigned long long)__float_as_uint(depth) << 32) | (unsigned)(f + 1);
        atomicMin(&zbuf[(int64_t)iy * W + ix], token);
    }
}

__global__ __launch_bounds__(256) void raster_resolve(const float* __restrict__ pos4, const int32_t* __restrict__ tri, int H, int W,
                                                      const unsigned long long* __restrict__ zbuf,
                                                      int32_t* __restrict__ findices, float* __restrict__ bary) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const unsigned long long token = zbuf[i];
    float o[3] = {0.f, 0.f, 0.f};
    int id = 0;
    if (token != kEmpty) {
        id = (int)(unsigned)(token & 0xFFFFFFFFull);
        const ScreenTri t = screen_tri(pos4, tri, id - 1, H, W);
        float b[3];
        (void)bary_at(t, (float)(i % W) + 0.5f, (float)(i / W) + 0.5f, b);
        // perspective correction: divide by clip w, renormalise
        const float c0 = b[0] / t.w[0], c1 = b[1] / t.w[1], c2 = b[2] / t.w[2];
        const float s = c0 + c1 + c2;
        o[0] = c0 / s; o[1] = c1 / s; o[2] = c2 / s;
    }
    findices[i] = id;
    bary[3 * i] = o[0]; bary[3 * i + 1] = o[1]; bary[3 * i + 2] = o[2];
}

__global__ __launch_bounds__(256) void interpolate_kernel(const float* __restrict__ attr, int C, const int32_t* __restrict__ tri,
                                                          const int32_t* __restrict__ findices, const float* __restrict__ bary,
                                                          int64_t npix, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int id = findices[i];
    if (id <= 0) {
        for (int c = 0; c < C; ++c) out[i * C + c] = 0.f;
        return;
    }
    const int32_t* t = tri + 3 * (int64_t)(id - 1);
    const float b0 = bary[3 * i], b1 = bary[3 * i + 1], b2 = bary[3 * i + 2];
    const float *a0 = attr + (int64_t)t[0] * C, *a1 = attr + (int64_t)t[1] * C, *a2 = attr + (int64_t)t[2] * C;
    for (int c = 0; c < C; ++c) out[i * C + c] = b0 * a0[c] + b1 * a1[c] + b2 * a2[c];
}

__global__ __launch_bounds__(256) void view_weight_kernel(const int32_t* __restrict__ findices, const float* __restrict__ depth,
                                                          const float* __restrict__ normal, int H, int W, float cos_thresh,
                                                          float depth_edge, float view_weight, float power,
                                                          float* __restrict__ weight) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    float wgt = 0.f;
    if (findices[i] > 0) {
        const float nx = normal[3 * i], ny = normal[3 * i + 1], nz = normal[3 * i + 2];
        const float len = sqrtf(nx * nx + ny * ny + nz * nz);
        const float cosv = len > 0.f ? nz / len : 0.f;
        bool keep = cosv >= cos_thresh;
        const int x = (int)(i % W), y = (int)(i / W);
        const float d = depth[i];
        const int dx[4] = {-1, 1, 0, 0}, dy[4] = {0, 0, -1, 1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = x + dx[k], yy = y + dy[k];
            if (xx < 0 || yy < 0 || xx >= W || yy >= H) { keep = false; continue; }
            const int64_t j = (int64_t)yy * W + xx;
            if (findices[j] <= 0 || fabsf(depth[j] - d) > depth_edge) keep = false;
        }
        if (keep) wgt = view_weight * powf(cosv, power);
    }
    weight[i] = wgt;
}

__device__ __forceinline__ int texel_of(float u, int T) {
    const int t = (int)(u * (float)(T - 1) + 0.5f);
    return t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
}
__device__ __forceinline__ unsigned q16(float c) {
    const float cc = c < 0.f ? 0.f : (c > 1.f ? 1.f : c);
    return (unsigned)(cc * 65536.f + 0.5f);
}

__global__ __launch_bounds__(256) void bake_kernel(const float* __restrict__ image, const float* __restrict__ weight,
                                                   const int32_t* __restrict__ findices, const float* __restrict__ bary,
                                                   const float* __restrict__ uv, const int32_t* __restrict__ uv_tri, int64_t npix,
                                                   int T, unsigned long long* __restrict__ acc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int id = findices[i];
    const float w = weight[i];
    if (id <= 0 || !(w > 0.f)) return;
    const unsigned wq = (unsigned)(fminf(w, 65535.f) * 65536.f + 0.5f) >> 0;
    if (wq == 0u) return;
    const int32_t* t = uv_tri + 3 * (int64_t)(id - 1);
    const float b0 = bary[3 * i], b1 = bary[3 * i + 1], b2 = bary[3 * i + 2];
    const float u = b0 * uv[2 * (int64_t)t[0]] + b1 * uv[2 * (int64_t)t[1]] + b2 * uv[2 * (int64_t)t[2]];
    const float v = b0 * uv[2 * (int64_t)t[0] + 1] + b1 * uv[2 * (int64_t)t[1] + 1] + b2 * uv[2 * (int64_t)t[2] + 1];
    const int64_t tex = (int64_t)texel_of(v, T) * T + texel_of(u, T);
#pragma unroll
    for (int c = 0; c < 3; ++c) atomicAdd(&acc[4 * tex + c], (unsigned long long)wq * q16(image[3 * i + c]));
    atomicAdd(&acc[4 * tex + 3], (unsigned long long)wq);
}

// Texel-centric baking: every covered texel finds its own place in the view (its clip position comes from the UV-space
// rasterisation of the mesh), checks that it is the surface the view sees there (z-buffer depth of the nearest pixel),
// takes that pixel's baking weight and a bilinear sample of the image.  One thread per texel: no atomics, and the texture
// is as dense as the UV raster, however coarse the view is.
__global__ __launch_bounds__(256) void bake_gather_kernel(const int32_t* __restrict__ findices_uv, const float* __restrict__ bary_uv,
                                                          const float* __restrict__ clip_uv, const int32_t* __restrict__ uv_tri,
                                                          int64_t ntexel, const float* __restrict__ image,
                                                          const float* __restrict__ weight, const int32_t* __restrict__ findices,
                                                          const float* __restrict__ depth, int H, int W, float depth_eps,
                                                          unsigned long long* __restrict__ acc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ntexel) return;
    const int id = findices_uv[i];
    if (id <= 0) return;
    const int32_t* t = uv_tri + 3 * (int64_t)(id - 1);
    const float b0 = bary_uv[3 * i], b1 = bary_uv[3 * i + 1], b2 = bary_uv[3 * i + 2];
    float p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        p[k] = b0 * clip_uv[4 * (int64_t)t[0] + k] + b1 * clip_uv[4 * (int64_t)t[1] + k] + b2 * clip_uv[4 * (int64_t)t[2] + k];
    if (!(p[3] > 0.f)) return;
    const float sx = (p[0] / p[3] * 0.5f + 0.5f) * (float)(W - 1) + 0.5f;
    const float sy = (p[1] / p[3] * 0.5f + 0.5f) * (float)(H - 1) + 0.5f;
    const float z = p[2] / p[3];
    const int px = (int)floorf(sx), py = (int)floorf(sy);
    if (px < 0 || py < 0 || px >= W || py >= H) return;
    const int64_t pix = (int64_t)py * W + px;
    const float w = weight[pix];
    if (findices[pix] <= 0 || !(w > 0.f) || z > depth[pix] + depth_eps) return;
    const unsigned wq = (unsigned)(fminf(w, 65535.f) * 65536.f + 0.5f);
    if (wq == 0u) return;
    // bilinear sample at the pixel-centre coordinates (sx - 0.5, sy - 0.5), clamped at the border
    const float fx = sx - 0.5f, fy = sy - 0.5f;
    int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
    const float ax = fx - (float)x0, ay = fy - (float)y0;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float c00 = image[3 * ((int64_t)y0 * W + x0) + c], c01 = image[3 * ((int64_t)y0 * W + x1) + c];
        const float c10 = image[3 * ((int64_t)y1 * W + x0) + c], c11 = image[3 * ((int64_t)y1 * W + x1) + c];
        const float top = c00 + (c01 - c00) * ax, bot = c10 + (c11 - c10) * ax;
        acc[4 * i + c] += (unsigned long long)wq * q16(top + (bot - top) * ay);
    }
    acc[4 * i + 3] += (unsigned long long)wq;
}

__global__ __launch_bounds__(256) void bake_finalize_kernel(const unsigned long long* __restrict__ acc, int64_t n, float* __restrict__ tex,
                                                            uint8_t* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long w = acc[4 * i + 3];
    for (int c = 0; c < 3; ++c) tex[3 * i + c] = w ? (float)((double)acc[4 * i + c] / ((double)w * 65536.0)) : 0.f;
    mask[i] = w ? 1 : 0;
}

// ---- inpainting ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_u32(unsigned* p, int64_t n, unsigned v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// The texel a face corner takes its vertex colour from: the corner's UV pulled a quarter of the way towards the centroid of
// its chart triangle.  (The texel nearest to the corner itself usually lies OUTSIDE the triangle -- with one chart per face
// its centre is beyond the chart's edge, unpainted -- so most corners gave no seed and the vertex colours came from a few
// texels spread by propagation.)  fp32, products rounded before the sums (-ffp-contract=off): the numpy restatement used by the tests repeats it.
__device__ __forceinline__ int64_t corner_texel(const float* uv, const int32_t* uv_tri, int64_t corner, int T) {
    const int64_t f3 = corner - corner % 3;
    const int64_t j = uv_tri[corner], j0 = uv_tri[f3], j1 = uv_tri[f3 + 1], j2 = uv_tri[f3 + 2];
    const float third = 1.0f / 3.0f;
    const float cu = ((uv[2 * j0] + uv[2 * j1]) + uv[2 * j2]) * third;
    const float cv = ((uv[2 * j0 + 1] + uv[2 * j1 + 1]) + uv[2 * j2 + 1]) * third;
    const float pu = uv[2 * j] * 0.75f + cu * 0.25f;
    const float pv = uv[2 * j + 1] * 0.75f + cv * 0.25f;
    return (int64_t)texel_of(pv, T) * T + texel_of(pu, T);
}

__global__ __launch_bounds__(256) void vertex_owner_kernel(const uint8_t* __restrict__ mask, int T, const float* __restrict__ uv,
                                                           const int32_t* __restrict__ uv_tri, const int32_t* __restrict__ pos_tri,
                                                           int64_t ncorner, unsigned* __restrict__ owner) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= ncorner) return;
    if (mask[corner_texel(uv, uv_tri, c, T)]) atomicMin(&owner[pos_tri[c]], (unsigned)c);
}

__global__ __launch_bounds__(256) void vertex_gather_kernel(const float* __restrict__ tex, int T, const float* __restrict__ uv,
                                                            const int32_t* __restrict__ uv_tri, const unsigned* __restrict__ owner,
                                                            int64_t V, float* __restrict__ vcolor, uint8_t* __restrict__ vmask) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const unsigned c = owner[v];
    float r = 0.f, g = 0.f, b = 0.f;
    if (c != 0xFFFFFFFFu) {
        const int64_t t = corner_texel(uv, uv_tri, c, T);
        r = tex[3 * t]; g = tex[3 * t + 1]; b = tex[3 * t + 2];
    }
    vcolor[3 * v] = r; vcolor[3 * v + 1] = g; vcolor[3 * v + 2] = b;
    vmask[v] = c != 0xFFFFFFFFu ? 1 : 0;
}

// every directed edge (a <- b) of every face: a coloured b hands its colour to an uncoloured a, weight 1 / (|ab|^2 + eps)
__global__ __launch_bounds__(256) void propagate_edges_kernel(const float* __restrict__ verts, const int32_t* __restrict__ pos_tri,
                                                              int64_t ncorner, const float* __restrict__ vcolor,
                                                              const uint8_t* __restrict__ vmask, unsigned long long* __restrict__ acc) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= ncorner) return;
    const int64_t f3 = c - c % 3;
    const int a = pos_tri[c];
    if (vmask[a]) return;
#pragma unroll
    for (int o = 1; o < 3; ++o) {
        const int b = pos_tri[f3 + (c - f3 + o) % 3];
        if (!vmask[b] || b == a) continue;
        const float dx = verts[3 * (int64_t)a] - verts[3 * (int64_t)b], dy = verts[3 * (int64_t)a + 1] - verts[3 * (int64_t)b + 1],
                    dz = verts[3 * (int64_t)a + 2] - verts[3 * (int64_t)b + 2];
        const float w = 1.f / (dx * dx + dy * dy + dz * dz + 1e-6f);
        const unsigned long long wq = (unsigned long long)(w * 1024.f + 0.5f);
        if (!wq) continue;
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(&acc[4 * (int64_t)a + k], wq * q16(vcolor[3 * (int64_t)b + k]));
        atomicAdd(&acc[4 * (int64_t)a + 3], wq);
    }
}

__global__ __launch_bounds__(256) void propagate_commit_kernel(int64_t V, unsigned long long* __restrict__ acc, float* __restrict__ vcolor,
                                                               const uint8_t* __restrict__ vmask, uint8_t* __restrict__ vmask_next,
                                                               unsigned* __restrict__ flag) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    uint8_t m = vmask[v];
    const unsigned long long w = acc[4 * v + 3];
    if (!m && w) {
        for (int k = 0; k < 3; ++k) vcolor[3 * v + k] = (float)((double)acc[4 * v + k] / ((double)w * 65536.0));
        m = 1;
        *flag = 1u;   // benign race: every writer stores the same value
    }
    vmask_next[v] = m;
    acc[4 * v] = acc[4 * v + 1] = acc[4 * v + 2] = acc[4 * v + 3] = 0ull;
}

__global__ __launch_bounds__(256) void fill_texels_kernel(const int32_t* __restrict__ findices_uv, const float* __restrict__ bary_uv,
                                                          const int32_t* __restrict__ pos_tri, const float* __restrict__ vcolor,
                                                          const uint8_t* __restrict__ vmask, int64_t n, float* __restrict__ tex,
                                                          uint8_t* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || mask[i]) return;
    const int id = findices_uv[i];
    if (id <= 0) return;
    const int32_t* t = pos_tri + 3 * (int64_t)(id - 1);
    float s = 0.f, r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!vmask[t[k]]) continue;
        const float w = bary_uv[3 * i + k];
        s += w;
        r += w * vcolor[3 * (int64_t)t[k]]; g += w * vcolor[3 * (int64_t)t[k] + 1]; b += w * vcolor[3 * (int64_t)t[k] + 2];
    }
    if (!(s > 0.f)) return;
    tex[3 * i] = r / s; tex[3 * i + 1] = g / s; tex[3 * i + 2] = b / s;
    mask[i] = 2;
}

// one Jacobi step of the dilation: an empty texel takes the mean of its filled 8-neighbours (state of the previous step)
__global__ __launch_bounds__(256) void dilate_kernel(const float* __restrict__ tex, const uint8_t* __restrict__ mask, int T,
                                                     float* __restrict__ tex_out, uint8_t* __restrict__ mask_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)T * T) return;
    float r = tex[3 * i], g = tex[3 * i + 1], b = tex[3 * i + 2];
    uint8_t m = mask[i];
    if (!m) {
        const int x = (int)(i % T), y = (int)(i / T);
        float s = 0.f, ar = 0.f, ag = 0.f, ab = 0.f;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx, yy = y + dy;
                if ((dx | dy) == 0 || xx < 0 || yy < 0 || xx >= T || yy >= T) continue;
                const int64_t j = (int64_t)yy * T + xx;
                if (!mask[j]) continue;
                s += 1.f; ar += tex[3 * j]; ag += tex[3 * j + 1]; ab += tex[3 * j + 2];
            }
        if (s > 0.f) { r = ar / s; g = ag / s; b = ab / s; m = 3; }
    }
    tex_out[3 * i] = r; tex_out[3 * i + 1] = g; tex_out[3 * i + 2] = b;
    mask_out[i] = m;
}

inline dim3 blocks(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

hipError_t tex_rasterize(const float* pos4, const int32_t* tri, int F, int H, int W, unsigned long long* zbuf,
                         int32_t* findices, float* bary, hipStream_t s) {
    const int64_t n = (int64_t)H * W;
    hipLaunchKernelGGL(fill_u64, blocks(n), dim3(256), 0, s, zbuf, n, kEmpty);
    if (F > 0) hipLaunchKernelGGL(raster_faces, dim3(F), dim3(64), 0, s, pos4, tri, F, H, W, zbuf);
    hipLaunchKernelGGL(raster_resolve, blocks(n), dim3(256), 0, s, pos4, tri, H, W, zbuf, findices, bary);
    return hipGetLastError();
}

hipError_t tex_interpolate(const float* attr, int C, const int32_t* tri, const int32_t* findices, const float* bary,
                           int64_t npix, float* out, hipStream_t s) {
    hipLaunchKernelGGL(interpolate_kernel, blocks(npix), dim3(256), 0, s, attr, C, tri, findices, bary, npix, out);
    return hipGetLastError();
}

hipError_t tex_view_weight(const int32_t* findices, const float* depth, const float* normal, int H, int W,
                           float cos_thresh, float depth_edge, float view_weight, float power, float* weight,
                           hipStream_t s) {
    hipLaunchKernelGGL(view_weight_kernel, blocks((int64_t)H * W), dim3(256), 0, s, findices, depth, normal, H, W, cos_thresh,
                       depth_edge, view_weight, power, weight);
    return hipGetLastError();
}

hipError_t tex_bake(const float* image, const float* weight, const int32_t* findices, const float* bary, const float* uv,
                    const int32_t* uv_tri, int64_t npix, int T, unsigned long long* acc, hipStream_t s) {
    hipLaunchKernelGGL(bake_kernel, blocks(npix), dim3(256), 0, s, image, weight, findices, bary, uv, uv_tri, npix, T, acc);
    return hipGetLastError();
}

hipError_t tex_bake_gather(const int32_t* findices_uv, const float* bary_uv, const float* clip_uv, const int32_t* uv_tri, int T,
                           const float* image, const float* weight, const int32_t* findices, const float* depth, int H, int W,
                           float depth_eps, unsigned long long* acc, hipStream_t s) {
    const int64_t n = (int64_t)T * T;
    hipLaunchKernelGGL(bake_gather_kernel, blocks(n), dim3(256), 0, s, findices_uv, bary_uv, clip_uv, uv_tri, n, image, weight,
                       findices, depth, H, W, depth_eps, acc);
    return hipGetLastError();
}

hipError_t tex_bake_finalize(const unsigned long long* acc, int T, float* tex, uint8_t* mask, hipStream_t s) {
    const int64_t n = (int64_t)T * T;
    hipLaunchKernelGGL(bake_finalize_kernel, blocks(n), dim3(256), 0, s, acc, n, tex, mask);
    return hipGetLastError();
}

namespace {
struct InpaintLayout { size_t owner, vcolor, vmask0, vmask1, acc, tex_tmp, mask_tmp, flag, total; };
InpaintLayout inpaint_layout(int64_t V, int T) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    InpaintLayout l;
    size_t o = 0;
    l.owner = o;    o += al(4 * (size_t)V);
    l.vcolor = o;   o += al(12 * (size_t)V);
    l.vmask0 = o;   o += al((size_t)V);
    l.vmask1 = o;   o += al((size_t)V);
    l.acc = o;      o += al(32 * (size_t)V);
    l.tex_tmp = o;  o += al(12 * (size_t)T * T);
    l.mask_tmp = o; o += al((size_t)T * T);
    l.flag = o;     o += 256;
    l.total = o;
    return l;
}
}  // namespace

size_t tex_inpaint_workspace(int64_t V, int T) { return inpaint_layout(V, T).total; }

hipError_t tex_inpaint(char* ws, unsigned* h_flag, float* tex, uint8_t* mask, int T, const int32_t* findices_uv,
                       const float* bary_uv, const float* verts, int64_t V, const int32_t* pos_tri, const float* uv,
                       const int32_t* uv_tri, int64_t F, int dilate_iters, int* rounds_out, hipStream_t s) {
    const InpaintLayout l = inpaint_layout(V, T);
    unsigned* owner = (unsigned*)(ws + l.owner);
    float* vcolor = (float*)(ws + l.vcolor);
    uint8_t* vm[2] = {(uint8_t*)(ws + l.vmask0), (uint8_t*)(ws + l.vmask1)};
    unsigned long long* acc = (unsigned long long*)(ws + l.acc);
    unsigned* flag = (unsigned*)(ws + l.flag);
    const int64_t nc = 3 * F, nt = (int64_t)T * T;
    hipError_t e;
    hipLaunchKernelGGL(fill_u32, blocks(V), dim3(256), 0, s, owner, V, 0xFFFFFFFFu);
    hipLaunchKernelGGL(vertex_owner_kernel, blocks(nc), dim3(256), 0, s, mask, T, uv, uv_tri, pos_tri, nc, owner);
    hipLaunchKernelGGL(vertex_gather_kernel, blocks(V), dim3(256), 0, s, tex, T, uv, uv_tri, owner, V, vcolor, vm[0]);
    if ((e = hipMemsetAsync(acc, 0, 32 * (size_t)V, s)) != hipSuccess) return e;
    // Jacobi rounds until nothing changes.  A round after convergence is a no-op, so the convergence flag is read back
    // once per batch of kBatch rounds (one host synchronisation per batch instead of per round); every round of a batch
    // has a flag of its own, so the number of rounds that changed something is still exact.
    constexpr int kBatch = 8;
    int cur = 0, rounds = 0;
    bool done = false;
    while (!done && rounds < 8192) {
        if ((e = hipMemsetAsync(flag, 0, 4 * kBatch, s)) != hipSuccess) return e;
        for (int k = 0; k < kBatch; ++k) {
            hipLaunchKernelGGL(propagate_edges_kernel, blocks(nc), dim3(256), 0, s, verts, pos_tri, nc, vcolor, vm[cur], acc);
            hipLaunchKernelGGL(propagate_commit_kernel, blocks(V), dim3(256), 0, s, V, acc, vcolor, vm[cur], vm[cur ^ 1], flag + k);
            cur ^= 1;
        }
        if ((e = hipMemcpyAsync(h_flag, flag, 4 * kBatch, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        for (int k = 0; k < kBatch; ++k) {
            if (h_flag[k] == 0u) { done = true; break; }
            ++rounds;
        }
    }
    if (rounds_out) *rounds_out = rounds;
    hipLaunchKernelGGL(fill_texels_kernel, blocks(nt), dim3(256), 0, s, findices_uv, bary_uv, pos_tri, vcolor, vm[cur], nt, tex, mask);
    float* tt[2] = {tex, (float*)(ws + l.tex_tmp)};
    uint8_t* mm[2] = {mask, (uint8_t*)(ws + l.mask_tmp)};
    int side = 0;
    for (int i = 0; i < dilate_iters; ++i) {
        hipLaunchKernelGGL(dilate_kernel, blocks(nt), dim3(256), 0, s, tt[side], mm[side], T, tt[side ^ 1], mm[side ^ 1]);
        side ^= 1;
    }
    if (side) {
        if ((e = hipMemcpyAsync(tex, tt[1], 12 * (size_t)nt, hipMemcpyDeviceToDevice, s)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(mask, mm[1], (size_t)nt, hipMemcpyDeviceToDevice, s)) != hipSuccess) return e;
    }
    return hipGetLastError();
}

}  // namespace r3g
