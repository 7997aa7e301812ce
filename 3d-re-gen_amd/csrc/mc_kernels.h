// mc_kernels.h -- host-side launch interface of mc_kernels.hip (internal to libr3g.so)
#ifndef R3G_MC_KERNELS_H
#define R3G_MC_KERNELS_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace r3g {

struct McWorkspaceLayout {
    uint32_t nblk, nchunk, ncells;
    uint32_t nnz;                     // non-empty blocks: filled in by the caller from totals[2] after the count pass
    uint64_t off_small, small_bytes;  // status(u32) @0, totals {nV, nF, nNZ} (3 x u64) @16, chunk sums @64
    uint64_t zero_bytes;              // bytes from off_small that mc_count_launch clears
    uint64_t off_blk, off_blkoff, off_nz, off_act, off_etab;
};

void mc_set_deferred(bool on);         // row kernel: tiling selection batched per wave (default) or per row
void mc_set_rows_per_wave(int rows);  // node rows a wave marches through in the row kernel (4|8|16|32)
size_t mc_workspace_bytes(int n0, int n1, int n2, McWorkspaceLayout* lay);

// K1 + K2.  After the stream drains: status word at ws+off_small, totals {nV, nF, nNZ} at +16.
hipError_t mc_count_launch(const float* grid, int n0, int n1, int n2, double level, int classic, char* ws,
                           const McWorkspaceLayout& lay, hipStream_t stream);

// K3 + K4.  xf9 = {grid_size[3], bbox_size[3], bbox_min[3]} or null (index-space vertices).
hipError_t mc_emit_launch(const float* grid, int n0, int n1, int n2, double level, char* ws,
                          const McWorkspaceLayout& lay, float* verts, int32_t* faces, const double* xf9,
                          int reversed, hipStream_t stream);

}  // namespace r3g
#endif
