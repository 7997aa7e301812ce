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

#ifdef __cplusplus
}
#endif
#endif
