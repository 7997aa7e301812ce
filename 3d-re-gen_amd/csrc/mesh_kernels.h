// mesh_kernels.h -- host-side interface of mesh_kernels.hip (internal to libr3g.so)
#ifndef R3G_MESH_KERNELS_H
#define R3G_MESH_KERNELS_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace r3g {

// workspace for any of the three cleaners on a mesh of nv vertices / nf faces (max_cells: clustering grid, 0 if unused)
size_t mesh_workspace_bytes(int64_t nv, int64_t nf, int64_t max_cells);
int mesh_reduce_initial_res(int64_t max_faces);
void mesh_set_floater_by_vertex(bool on);   // 0 (default): faces joined through shared edges (MeshLab) | 1: through shared vertices

// All three compact verts / faces in place (survivors keep their order) and update *nv_io / *nf_io.
// They synchronise the stream internally (sizes travel through the pinned h_small, >= 32 bytes).
hipError_t mesh_remove_floaters(char* ws, size_t ws_bytes, unsigned* h_small, float* verts, int64_t* nv_io,
                                int32_t* faces, int64_t* nf_io, double min_ratio, hipStream_t s);
hipError_t mesh_remove_degenerate(char* ws, size_t ws_bytes, unsigned* h_small, float* verts, int64_t* nv_io,
                                  int32_t* faces, int64_t* nf_io, hipStream_t s);
// quadric-error-metric edge collapse (the algorithm class upstream's FaceReducer uses); *rounds_out (optional) = rounds
hipError_t mesh_reduce_faces(char* ws, size_t ws_bytes, unsigned* h_small, float* verts, int64_t* nv_io,
                             int32_t* faces, int64_t* nf_io, int64_t max_faces, int* rounds_out, hipStream_t s);
// uniform-grid vertex clustering (round 1's reducer; kept as the bit-exactly restated alternative)
hipError_t mesh_cluster_faces(char* ws, size_t ws_bytes, unsigned* h_small, float* verts, int64_t* nv_io,
                              int32_t* faces, int64_t* nf_io, int64_t max_faces, hipStream_t s);

}  // namespace r3g
#endif
