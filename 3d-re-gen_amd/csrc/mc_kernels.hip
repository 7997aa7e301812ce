// mc_kernels.hip -- Lewiner marching cubes on gfx950 (HBM-bound byte/index work, no MFMA).
//
// Drop-in for the CPU call `skimage.measure.marching_cubes(grid.cpu().numpy(), mc_level,
// method="lewiner")` made by hy3dgen MCSurfaceExtractor.run on the reference hot path
// (src/2d_to_3d_models/run.py:77-84): the (R+1)^3 fp32 grid stays in HBM (no 68 MB D2H) and the
// mesh comes out with the sequential kernel's exact vertex numbering.
//
// Launch structure (cells are linearised in scan order, axis 2 fastest; 256 cells per block):
//   K1 mc_classify : 1 thread/cell, 8 coalesced row reads -> tiling (fp64 ambiguity tests only in
//                    active lanes), wave64 shuffle scan + LDS across the 4 waves -> per-block
//                    compacted records {tiling, counts, in-block prefix}, block sums, chunk sums.
//   K2 mc_scan     : exclusive scan of the block sums (one workgroup per 1024-block chunk).
//   K3 mc_vertices : 1 thread per ACTIVE cell: fp64 interpolation, float32 store, edge->id table.
//   K4 mc_faces    : 1 thread per ACTIVE cell: triangle corners -> ids (own rank or table lookup).
// Only K1 touches the whole grid: algorithmic traffic = one grid read + one mesh write.
//
// Built with -ffp-contract=off: the ambiguity tests and interpolation must not be fused.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define R3G_DEV static __device__ __forceinline__
#define R3G_LUT_QUAL static __device__ const
#include "mc_cell.h"
#include "mc_kernels.h"
#include "prof.h"

#pragma clang fp contract(off)

using namespace r3g_mc;

namespace {

constexpr int kBlock = 256;
constexpr int kChunk = 1024;  // blocks per scan chunk

__device__ __forceinline__ void cell_coords(uint32_t c, int cx, int cy, int& x, int& y, int& z) {
    const uint32_t row = c / (uint32_t)cx;
    x = (int)(c - row * (uint32_t)cx);
    z = (int)(row / (uint32_t)cy);
    y = (int)(row - (uint32_t)z * (uint32_t)cy);
}

// inclusive scan of a packed 3x16-bit counter across the 64 lanes of a wave
__device__ __forceinline__ unsigned long long wave_inclusive_scan(unsigned long long v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

__global__ __launch_bounds__(kBlock) void mc_classify(const float* __restrict__ grid, int nx, int ny, int cx,
                                                      int cy, uint32_t ncells, double level, int classic,
                                                      uint2* __restrict__ act, uint4* __restrict__ blk,
                                                      unsigned long long* __restrict__ chunk_sums,
                                                      unsigned* __restrict__ status) {
    __shared__ unsigned long long wave_tot[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t b = blockIdx.x;
    const uint32_t c = b * kBlock + tid;
    unsigned rec = 0, flags = 0;
    int x = 0, y = 0, z = 0, index = 0;
    if (c < ncells) {
        // coordinates: one (wave-uniform) division for the block's first cell, then at most a few wraps per thread
        int x0, y0, z0;
        cell_coords(b * kBlock, cx, cy, x0, y0, z0);
        x = x0 + tid; y = y0; z = z0;
        if (cx >= kBlock) {
            if (x >= cx) { x -= cx; if (++y >= cy) { y = 0; ++z; } }
        } else {
            cell_coords(c, cx, cy, x, y, z);
        }
        flags = load_signs(grid, nx, ny, x, y, z, level, &index);
    }
    const bool active = index != 0 && index != 255;
    // range flags: one global atomic per wave, and only while it would still change the status word
    {
        const unsigned long long le = __ballot(flags & R3G_MC_FLAG_LE), ge = __ballot(flags & R3G_MC_FLAG_GE),
                                 nn = __ballot(flags & R3G_MC_FLAG_NAN);
        const unsigned wf = (le ? R3G_MC_FLAG_LE : 0u) | (ge ? R3G_MC_FLAG_GE : 0u) | (nn ? R3G_MC_FLAG_NAN : 0u);
        if (lane == 0 && (wf & ~*(volatile unsigned*)status)) atomicOr(status, wf);
    }
    // most blocks contain no surface cell: they publish zeros and leave before the tiling / scan work
    if (!__syncthreads_or(active ? 1 : 0)) {
        if (tid == 0) blk[b] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    if (active) {
        double v[8];
        int idx2;
        load_corners(grid, nx, ny, x, y, z, level, v, &idx2);
        rec = classify_cell(v, index, classic != 0, x, y, z);
    }
    // packed counters: [0..15] new vertices, [16..31] triangles, [32..47] active cells
    const unsigned long long mine = (unsigned long long)((rec >> 20) & 0xFu) |
                                    ((unsigned long long)((rec >> 16) & 0xFu) << 16) |
                                    ((unsigned long long)(rec ? 1u : 0u) << 32);
    const unsigned long long incl = wave_inclusive_scan(mine, lane);
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    unsigned long long base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        const unsigned long long t = wave_tot[w];
        if (w < wid) base += t;
        total += t;
    }
    const unsigned long long excl = base + incl - mine;
    if (rec) {
        const unsigned vloc = (unsigned)(excl & 0xFFFFu), tloc = (unsigned)((excl >> 16) & 0xFFFFu);
        const unsigned arank = (unsigned)((excl >> 32) & 0xFFFFu);
        act[(size_t)b * kBlock + arank] = make_uint2(rec, (unsigned)tid | (vloc << 8) | (tloc << 20));
    }
    if (tid == 0) {
        const unsigned sv = (unsigned)(total & 0xFFFFu), st = (unsigned)((total >> 16) & 0xFFFFu);
        const unsigned sa = (unsigned)((total >> 32) & 0xFFFFu);
        blk[b] = make_uint4(sv, st, sa, 0u);
        if (sv | st) atomicAdd(&chunk_sums[b / kChunk], (unsigned long long)sv | ((unsigned long long)st << 32));
    }
}

// One workgroup per chunk of 1024 block sums.  base = sum of all earlier chunks (<= a few hundred
// values), then an exclusive scan inside the chunk.  The last chunk publishes the totals.
__global__ __launch_bounds__(kChunk) void mc_scan(const uint4* __restrict__ blk, uint32_t nblk,
                                                  const unsigned long long* __restrict__ chunk_sums,
                                                  uint2* __restrict__ blkoff, unsigned long long* __restrict__ totals) {
    __shared__ unsigned long long s_red[kChunk / 64];
    __shared__ unsigned long long s_wave[kChunk / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t ch = blockIdx.x;
    // both halves are < 2^32 in total for any grid the API admits, so packed 2x32 adds cannot carry
    unsigned long long part = 0;
    for (uint32_t i = tid; i < ch; i += kChunk) part += chunk_sums[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
    if (lane == 0) s_red[wid] = part;
    const uint32_t bi = ch * kChunk + tid;
    unsigned long long mine = 0;
    if (bi < nblk) {
        const uint4 s = blk[bi];
        mine = (unsigned long long)s.x | ((unsigned long long)s.y << 32);
    }
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long n = __shfl_up(incl, d, 64);
        if (lane >= d) incl += n;
    }
    if (lane == 63) s_wave[wid] = incl;
    __syncthreads();
    unsigned long long base = 0;
#pragma unroll
    for (int w = 0; w < kChunk / 64; ++w) {
        base += s_red[w];
        if (w < wid) base += s_wave[w];
    }
    const unsigned long long excl = base + incl - mine;
    if (bi < nblk) blkoff[bi] = make_uint2((unsigned)(excl & 0xFFFFFFFFull), (unsigned)(excl >> 32));
    if (bi == nblk - 1) {
        const unsigned long long tot = excl + mine;
        totals[0] = tot & 0xFFFFFFFFull;
        totals[1] = tot >> 32;
    }
}

__global__ __launch_bounds__(kBlock) void mc_vertices(const float* __restrict__ grid, int nx, int ny, int cx, int cy,
                                                      double level, const uint2* __restrict__ act,
                                                      const uint4* __restrict__ blk, const uint2* __restrict__ blkoff,
                                                      int32_t* __restrict__ etab, float* __restrict__ verts,
                                                      Xform xf, int use_xf) {
    const uint32_t b = blockIdx.x;
    const unsigned tid = threadIdx.x;
    if (tid >= blk[b].z) return;
    const uint2 a = act[(size_t)b * kBlock + tid];
    const uint32_t c = b * kBlock + (a.y & 0xFFu);
    int x, y, z, index;
    cell_coords(c, cx, cy, x, y, z);
    double v[8];
    load_corners(grid, nx, ny, x, y, z, level, v, &index);
    emit_cell_vertices(a.x, blkoff[b].x + ((a.y >> 8) & 0xFFFu), v, x, y, z, nx, ny, etab, verts,
                       use_xf ? &xf : nullptr);
}

__global__ __launch_bounds__(kBlock) void mc_faces(int nx, int ny, int cx, int cy, const uint2* __restrict__ act,
                                                   const uint4* __restrict__ blk, const uint2* __restrict__ blkoff,
                                                   const int32_t* __restrict__ etab, int32_t* __restrict__ faces,
                                                   int reversed) {
    const uint32_t b = blockIdx.x;
    const unsigned tid = threadIdx.x;
    if (tid >= blk[b].z) return;
    const uint2 a = act[(size_t)b * kBlock + tid];
    const uint32_t c = b * kBlock + (a.y & 0xFFu);
    int x, y, z;
    cell_coords(c, cx, cy, x, y, z);
    const uint2 off = blkoff[b];
    emit_cell_faces(a.x, off.x + ((a.y >> 8) & 0xFFFu), off.y + (a.y >> 20), x, y, z, nx, ny, etab, faces,
                    reversed != 0);
}

}  // namespace

namespace r3g {

size_t mc_workspace_bytes(int n0, int n1, int n2, McWorkspaceLayout* lay) {
    const uint64_t ncells = (uint64_t)(n0 - 1) * (n1 - 1) * (n2 - 1);
    const uint64_t nblk = (ncells + kBlock - 1) / kBlock;
    const uint64_t nchunk = (nblk + kChunk - 1) / kChunk;
    const uint64_t nnodes = (uint64_t)n0 * n1 * n2;
    auto align = [](uint64_t v) { return (v + 255) & ~(uint64_t)255; };
    uint64_t o = 0;
    lay->nblk = (uint32_t)nblk;
    lay->nchunk = (uint32_t)nchunk;
    lay->ncells = (uint32_t)ncells;
    lay->off_small = o;  // [status u32 | pad | totals 2xu64 | chunk_sums u64 x nchunk]  (zeroed per call)
    lay->small_bytes = align(32 + 8 * nchunk);
    o += lay->small_bytes;
    lay->off_blk = o;    o += align(16 * nblk);
    lay->off_blkoff = o; o += align(8 * nblk);
    lay->off_act = o;    o += align(8 * nblk * kBlock);
    lay->off_etab = o;   o += align(12 * nnodes);
    return (size_t)o;
}

hipError_t mc_count_launch(const float* grid, int n0, int n1, int n2, double level, int classic, char* ws,
                           const McWorkspaceLayout& lay, hipStream_t stream) {
    const int nx = n2, ny = n1, cx = n2 - 1, cy = n1 - 1;
    hipError_t e = hipMemsetAsync(ws + lay.off_small, 0, lay.small_bytes, stream);
    if (e != hipSuccess) return e;
    unsigned* status = (unsigned*)(ws + lay.off_small);
    unsigned long long* totals = (unsigned long long*)(ws + lay.off_small + 16);
    unsigned long long* chunk_sums = (unsigned long long*)(ws + lay.off_small + 32);
    {
    ProfScope ps(PC_MC_CLASSIFY, 4.0 * (double)n0 * n1 * n2, stream);
    hipLaunchKernelGGL(mc_classify, dim3(lay.nblk), dim3(kBlock), 0, stream, grid, nx, ny, cx, cy, lay.ncells, level,
                       classic, (uint2*)(ws + lay.off_act), (uint4*)(ws + lay.off_blk), chunk_sums, status);
    }
    ProfScope ps2(PC_MC_OTHER, 0.0, stream);
    hipLaunchKernelGGL(mc_scan, dim3(lay.nchunk), dim3(kChunk), 0, stream, (const uint4*)(ws + lay.off_blk), lay.nblk,
                       chunk_sums, (uint2*)(ws + lay.off_blkoff), totals);
    return hipGetLastError();
}

hipError_t mc_emit_launch(const float* grid, int n0, int n1, int n2, double level, char* ws,
                          const McWorkspaceLayout& lay, float* verts, int32_t* faces, const double* xf9,
                          int reversed, hipStream_t stream) {
    (void)n0;
    const int nx = n2, ny = n1, cx = n2 - 1, cy = n1 - 1;
    Xform xf;
    for (int i = 0; i < 3; ++i) {
        xf.grid_size[i] = xf9 ? xf9[i] : 1.0;
        xf.bbox_size[i] = xf9 ? xf9[3 + i] : 1.0;
        xf.bbox_min[i] = xf9 ? xf9[6 + i] : 0.0;
    }
    ProfScope ps(PC_MC_OTHER, 0.0, stream);
    hipLaunchKernelGGL(mc_vertices, dim3(lay.nblk), dim3(kBlock), 0, stream, grid, nx, ny, cx, cy, level,
                       (const uint2*)(ws + lay.off_act), (const uint4*)(ws + lay.off_blk),
                       (const uint2*)(ws + lay.off_blkoff), (int32_t*)(ws + lay.off_etab), verts, xf, xf9 ? 1 : 0);
    hipLaunchKernelGGL(mc_faces, dim3(lay.nblk), dim3(kBlock), 0, stream, nx, ny, cx, cy,
                       (const uint2*)(ws + lay.off_act), (const uint4*)(ws + lay.off_blk),
                       (const uint2*)(ws + lay.off_blkoff), (const int32_t*)(ws + lay.off_etab), faces, reversed);
    return hipGetLastError();
}

}  // namespace r3g
