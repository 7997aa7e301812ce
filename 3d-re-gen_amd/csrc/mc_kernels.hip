// mc_kernels.hip -- Lewiner marching cubes on gfx950 (HBM-bound byte/index work, no MFMA).
//
// Drop-in for the CPU call `skimage.measure.marching_cubes(grid.cpu().numpy(), mc_level,
// method="lewiner")` made by hy3dgen MCSurfaceExtractor.run on the reference hot path
// (src/2d_to_3d_models/run.py:77-84): the (R+1)^3 fp32 grid stays in HBM (no 68 MB D2H) and the
// mesh comes out with the sequential kernel's exact vertex numbering.
//
// Launch structure (cells are linearised in scan order, axis 2 fastest; 256 cells per block):
//   K1 classify    : sign field -> tiling (fp64 ambiguity tests only in active lanes) -> per-block
//                    compacted records {tiling, counts, in-block prefix}, block sums, chunk sums.
//       mc_classify_rows (row length % 256 == 0, e.g. the 257^3 grid): one WAVE per 256-cell block, 4 cells
//                    per lane, marching 16 rows along axis 1 so that every node row is loaded once per wave
//                    as dwordx4 + dword, next rows in flight while the current ones are tested (float-domain
//                    sign test).  The few active cells of a row are packed into consecutive lanes through
//                    LDS before the tiling selection; the in-block scan is a wave64 shuffle scan (no barrier).
//       mc_classify      (any other shape): one thread per cell, 8 coalesced row reads, LDS across 4 waves.
//   K2 mc_scan     : exclusive scan of the block sums (one workgroup per 1024-block chunk) + the
//                    compacted list of non-empty blocks.
//   K3 mc_vertices : 1 thread per ACTIVE cell: fp64 interpolation, float32 store, edge->id table.
//   K4 mc_faces    : 1 thread per ACTIVE cell: triangle corners -> ids (own rank or table lookup).
//                    K3/K4 are launched over the non-empty blocks only.
// Only K1 touches the whole grid: algorithmic traffic = one grid read + one mesh write.
//
// Built with -ffp-contract=off: the ambiguity tests and interpolation must not be fused.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define R3G_DEV static __device__ __forceinline__
#define R3G_HOSTDEV static __host__ __device__ __forceinline__
#define R3G_LUT_QUAL static __device__ const
#include "mc_cell.h"
#include "mc_kernels.h"
#include "prof.h"

#pragma clang fp contract(off)

using namespace r3g_mc;

namespace {

constexpr int kBlock = 256;
constexpr int kChunk = 1024;  // blocks per scan chunk
constexpr int kEmitThreads = 64;  // K3 / K4: one wave per non-empty block

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
                                                      unsigned* __restrict__ chunk_nz,
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
        if (sa) atomicAdd(&chunk_nz[b / kChunk], 1u);
    }
}

// ---- row kernel ------------------------------------------------------------------------------------------------
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // node rows of a 257-wide grid are only 4-B aligned

struct Row5 { f4u a; float e; };  // the 5 nodes above a lane's 4 cells
__device__ __forceinline__ Row5 row_load(const float* r) {
    Row5 v;
    v.a = *(const f4u*)r;
    v.e = r[4];
    return v;
}
__device__ __forceinline__ unsigned row_mask(const Row5& r, float lo, int exact, unsigned* flags) {
    unsigned m = 0;
    m |= node_greater(r.a.x, lo, exact, flags) ? 1u : 0u;
    m |= node_greater(r.a.y, lo, exact, flags) ? 2u : 0u;
    m |= node_greater(r.a.z, lo, exact, flags) ? 4u : 0u;
    m |= node_greater(r.a.w, lo, exact, flags) ? 8u : 0u;
    m |= node_greater(r.e, lo, exact, flags) ? 16u : 0u;
    return m;
}

// blk[] is pre-zeroed by the launcher: blocks without a surface cell write nothing at all.
template <int kRowsPerWave>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 8)))
void mc_classify_rows(const float* __restrict__ grid, int nx, int ny, int cx, int cy, int cz, float lo, int lo_exact,
                      double level, int classic, uint2* __restrict__ act, uint4* __restrict__ blk,
                      unsigned long long* __restrict__ chunk_sums, unsigned* __restrict__ chunk_nz,
                      unsigned* __restrict__ status) {
    __shared__ unsigned char s_slot[kBlock / 64][256];
    __shared__ __attribute__((aligned(16))) unsigned s_rec[kBlock / 64][256];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t segs = (uint32_t)cx >> 8;                       // 256-cell blocks per row
    const uint32_t ychunks = ((uint32_t)cy + kRowsPerWave - 1) / kRowsPerWave;
    const uint32_t task = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);  // wave-uniform
    if (task >= segs * ychunks * (uint32_t)cz) return;
    const uint32_t seg = task % segs, t2 = task / segs;
    const int z = (int)(t2 / ychunks), y0 = (int)(t2 % ychunks) * kRowsPerWave;
    const int y1 = min(cy, y0 + kRowsPerWave);
    const int xb = (int)seg * 256 + lane * 4;
    const int64_t sz = (int64_t)nx * ny;
    const float* p = grid + (int64_t)z * sz + (int64_t)y0 * nx + xb;
    unsigned flags = 0;
    unsigned m0, m1;
    {
        const Row5 r0 = row_load(p), r1 = row_load(p + sz);
        m0 = row_mask(r0, lo, lo_exact, &flags);
        m1 = row_mask(r1, lo, lo_exact, &flags);
    }
    Row5 q0 = row_load(p + nx), q1 = row_load(p + sz + nx);
    for (int y = y0; y < y1; ++y) {
        Row5 f0 = q0, f1 = q1;
        if (y + 1 < y1) {   // rows y+2 go in flight before rows y+1 are consumed
            f0 = row_load(p + 2 * (int64_t)nx);
            f1 = row_load(p + sz + 2 * (int64_t)nx);
        }
        const unsigned n0 = row_mask(q0, lo, lo_exact, &flags), n1 = row_mask(q1, lo, lo_exact, &flags);
        unsigned abits = 0;   // which of the lane's 4 cells straddle the level
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned idx = ((m0 >> c) & 1u) | (((m0 >> (c + 1)) & 1u) << 1) | (((n0 >> (c + 1)) & 1u) << 2) |
                                 (((n0 >> c) & 1u) << 3) | (((m1 >> c) & 1u) << 4) | (((m1 >> (c + 1)) & 1u) << 5) |
                                 (((n1 >> (c + 1)) & 1u) << 6) | (((n1 >> c) & 1u) << 7);
            abits |= (idx != 0u && idx != 255u) ? 1u << c : 0u;
        }
        if (__any(abits != 0u)) {
            // The tiling code is long and only a few cells of a row are active: pack the active cells into
            // consecutive lanes through LDS (wave-local, no barrier), classify, and hand the records back.
            const unsigned cnt = __popc(abits);
            unsigned incl_c = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned n = __shfl_up(incl_c, d, 64);
                if (lane >= d) incl_c += n;
            }
            const unsigned n_active = __shfl(incl_c, 63, 64);
            unsigned slot = incl_c - cnt;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (abits & (1u << c)) s_slot[wid][slot++] = (unsigned char)(lane * 4 + c);
            reinterpret_cast<uint4*>(s_rec[wid])[lane] = make_uint4(0u, 0u, 0u, 0u);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (unsigned i = lane; i < n_active; i += 64) {
                const int id = s_slot[wid][i];
                double v[8];
                int index;
                const int x = (int)seg * 256 + id;
                load_corners(grid, nx, ny, x, y, z, level, v, &index);
                s_rec[wid][id] = classify_cell(v, index, classic != 0, x, y, z);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint4 rr = reinterpret_cast<const uint4*>(s_rec[wid])[lane];
            const unsigned rec[4] = {rr.x, rr.y, rr.z, rr.w};
            unsigned long long mine = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                mine += (unsigned long long)((rec[c] >> 20) & 0xFu) |
                        ((unsigned long long)((rec[c] >> 16) & 0xFu) << 16) |
                        ((unsigned long long)(rec[c] ? 1u : 0u) << 32);
            const unsigned long long incl = wave_inclusive_scan(mine, lane);
            unsigned long long excl = incl - mine;
            const uint32_t b = ((uint32_t)z * (uint32_t)cy + (uint32_t)y) * segs + seg;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (rec[c]) {
                    const unsigned vloc = (unsigned)(excl & 0xFFFFu), tloc = (unsigned)((excl >> 16) & 0xFFFFu);
                    const unsigned arank = (unsigned)((excl >> 32) & 0xFFFFu);
                    act[(size_t)b * kBlock + arank] =
                        make_uint2(rec[c], (unsigned)(lane * 4 + c) | (vloc << 8) | (tloc << 20));
                    excl += (unsigned long long)((rec[c] >> 20) & 0xFu) |
                            ((unsigned long long)((rec[c] >> 16) & 0xFu) << 16) | (1ull << 32);
                }
            }
            if (lane == 63) {
                const unsigned sv = (unsigned)(incl & 0xFFFFu), st = (unsigned)((incl >> 16) & 0xFFFFu);
                const unsigned sa = (unsigned)((incl >> 32) & 0xFFFFu);
                if (sa) {
                    blk[b] = make_uint4(sv, st, sa, 0u);
                    if (sv | st)
                        atomicAdd(&chunk_sums[b / kChunk], (unsigned long long)sv | ((unsigned long long)st << 32));
                    atomicAdd(&chunk_nz[b / kChunk], 1u);
                }
            }
        }
        m0 = n0; m1 = n1;
        q0 = f0; q1 = f1;
        p += nx;
    }
    const unsigned long long le = __ballot(flags & R3G_MC_FLAG_LE), ge = __ballot(flags & R3G_MC_FLAG_GE),
                             nn = __ballot(flags & R3G_MC_FLAG_NAN);
    const unsigned wf = (le ? R3G_MC_FLAG_LE : 0u) | (ge ? R3G_MC_FLAG_GE : 0u) | (nn ? R3G_MC_FLAG_NAN : 0u);
    if (lane == 0 && (wf & ~*(volatile unsigned*)status)) atomicOr(status, wf);
}

// One workgroup per chunk of 1024 block sums.  base = sum of all earlier chunks (<= a few hundred
// values), then an exclusive scan inside the chunk.  The last chunk publishes the totals.
__global__ __launch_bounds__(kChunk) void mc_scan(const uint4* __restrict__ blk, uint32_t nblk,
                                                  const unsigned long long* __restrict__ chunk_sums,
                                                  const unsigned* __restrict__ chunk_nz,
                                                  uint2* __restrict__ blkoff, uint32_t* __restrict__ nzlist,
                                                  unsigned long long* __restrict__ totals) {
    __shared__ unsigned long long s_red[kChunk / 64];
    __shared__ unsigned long long s_wave[kChunk / 64];
    __shared__ unsigned s_red_nz[kChunk / 64];
    __shared__ unsigned s_wave_nz[kChunk / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t ch = blockIdx.x;
    // both halves are < 2^32 in total for any grid the API admits, so packed 2x32 adds cannot carry
    unsigned long long part = 0;
    unsigned part_nz = 0;
    for (uint32_t i = tid; i < ch; i += kChunk) {
        part += chunk_sums[i];
        part_nz += chunk_nz[i];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        part += __shfl_xor(part, d, 64);
        part_nz += __shfl_xor(part_nz, d, 64);
    }
    if (lane == 0) { s_red[wid] = part; s_red_nz[wid] = part_nz; }
    const uint32_t bi = ch * kChunk + tid;
    unsigned long long mine = 0;
    unsigned mine_nz = 0;
    if (bi < nblk) {
        const uint4 s = blk[bi];
        mine = (unsigned long long)s.x | ((unsigned long long)s.y << 32);
        mine_nz = s.z ? 1u : 0u;
    }
    unsigned long long incl = mine;
    unsigned incl_nz = mine_nz;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long n = __shfl_up(incl, d, 64);
        const unsigned nn = __shfl_up(incl_nz, d, 64);
        if (lane >= d) { incl += n; incl_nz += nn; }
    }
    if (lane == 63) { s_wave[wid] = incl; s_wave_nz[wid] = incl_nz; }
    __syncthreads();
    unsigned long long base = 0;
    unsigned base_nz = 0;
#pragma unroll
    for (int w = 0; w < kChunk / 64; ++w) {
        base += s_red[w];
        base_nz += s_red_nz[w];
        if (w < wid) { base += s_wave[w]; base_nz += s_wave_nz[w]; }
    }
    const unsigned long long excl = base + incl - mine;
    const unsigned excl_nz = base_nz + incl_nz - mine_nz;
    if (bi < nblk) blkoff[bi] = make_uint2((unsigned)(excl & 0xFFFFFFFFull), (unsigned)(excl >> 32));
    if (mine_nz) nzlist[excl_nz] = bi;
    if (bi == nblk - 1) {
        const unsigned long long tot = excl + mine;
        totals[0] = tot & 0xFFFFFFFFull;
        totals[1] = tot >> 32;
        totals[2] = excl_nz + mine_nz;
    }
}

__global__ __launch_bounds__(kEmitThreads) void mc_vertices(const float* __restrict__ grid, int nx, int ny, int cx, int cy,
                                                      double level, const uint2* __restrict__ act,
                                                      const uint4* __restrict__ blk, const uint2* __restrict__ blkoff,
                                                      const uint32_t* __restrict__ nzlist,
                                                      int32_t* __restrict__ etab, float* __restrict__ verts,
                                                      Xform xf, int use_xf) {
    // one wave per non-empty block: a block of 256 cells holds ~10-30 active cells (at most 256: the loop)
    const uint32_t b = nzlist[blockIdx.x];
    const unsigned n_act = blk[b].z;
    const uint32_t voff = blkoff[b].x;
    for (unsigned tid = threadIdx.x; tid < n_act; tid += kEmitThreads) {
        const uint2 a = act[(size_t)b * kBlock + tid];
        const uint32_t c = b * kBlock + (a.y & 0xFFu);
        int x, y, z;
        cell_coords(c, cx, cy, x, y, z);
        emit_cell_vertices(a.x, voff + ((a.y >> 8) & 0xFFFu), grid, level, x, y, z, nx, ny, etab, verts, xf, use_xf != 0);
    }
}

__global__ __launch_bounds__(kEmitThreads) void mc_faces(int nx, int ny, int cx, int cy, const uint2* __restrict__ act,
                                                   const uint4* __restrict__ blk, const uint2* __restrict__ blkoff,
                                                   const uint32_t* __restrict__ nzlist,
                                                   const int32_t* __restrict__ etab, int32_t* __restrict__ faces,
                                                   int reversed) {
    const uint32_t b = nzlist[blockIdx.x];
    const unsigned n_act = blk[b].z;
    const uint2 off = blkoff[b];
    for (unsigned tid = threadIdx.x; tid < n_act; tid += kEmitThreads) {
        const uint2 a = act[(size_t)b * kBlock + tid];
        const uint32_t c = b * kBlock + (a.y & 0xFFu);
        int x, y, z;
        cell_coords(c, cx, cy, x, y, z);
        emit_cell_faces(a.x, off.x + ((a.y >> 8) & 0xFFFu), off.y + (a.y >> 20), x, y, z, nx, ny, etab, faces,
                        reversed != 0);
    }
}

}  // namespace

namespace r3g {

static int g_rows_per_wave = 16;
void mc_set_rows_per_wave(int rows) { g_rows_per_wave = rows == 4 || rows == 8 || rows == 32 ? rows : 16; }

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
    // [status u32 | pad | totals 3xu64 @16 | chunk_sums u64 x nchunk @64 | chunk_nz u32 x nchunk], then the block
    // sums: this whole prefix is zeroed per call (one memset)
    lay->off_small = o;
    lay->small_bytes = align(64 + 12 * nchunk);
    o += lay->small_bytes;
    lay->off_blk = o;    o += align(16 * nblk);
    lay->zero_bytes = o - lay->off_small;
    lay->off_blkoff = o; o += align(8 * nblk);
    lay->off_nz = o;     o += align(4 * nblk);
    lay->nnz = 0;
    lay->off_act = o;    o += align(8 * nblk * kBlock);
    lay->off_etab = o;   o += align(12 * nnodes);
    return (size_t)o;
}

hipError_t mc_count_launch(const float* grid, int n0, int n1, int n2, double level, int classic, char* ws,
                           const McWorkspaceLayout& lay, hipStream_t stream) {
    const int nx = n2, ny = n1, cx = n2 - 1, cy = n1 - 1, cz = n0 - 1;
    hipError_t e = hipMemsetAsync(ws + lay.off_small, 0, lay.zero_bytes, stream);
    if (e != hipSuccess) return e;
    unsigned* status = (unsigned*)(ws + lay.off_small);
    unsigned long long* totals = (unsigned long long*)(ws + lay.off_small + 16);
    unsigned long long* chunk_sums = (unsigned long long*)(ws + lay.off_small + 64);
    unsigned* chunk_nz = (unsigned*)(ws + lay.off_small + 64 + 8 * (size_t)lay.nchunk);
    {
    ProfScope ps(PC_MC_CLASSIFY, 4.0 * (double)n0 * n1 * n2, stream);
    if (cx % kBlock == 0) {
        float lo;
        int exact;
        level_floor(level, &lo, &exact);
        const int rows = g_rows_per_wave;
        const uint32_t tasks = (uint32_t)(cx / kBlock) * (uint32_t)((cy + rows - 1) / rows) * (uint32_t)cz;
        auto kern = rows == 4 ? mc_classify_rows<4> : rows == 8 ? mc_classify_rows<8> : rows == 32 ? mc_classify_rows<32>
                                                                                                : mc_classify_rows<16>;
        hipLaunchKernelGGL(kern, dim3((tasks + 3) / 4), dim3(kBlock), 0, stream, grid, nx, ny, cx, cy, cz,
                           lo, exact, level, classic, (uint2*)(ws + lay.off_act), (uint4*)(ws + lay.off_blk),
                           chunk_sums, chunk_nz, status);
    } else {
        hipLaunchKernelGGL(mc_classify, dim3(lay.nblk), dim3(kBlock), 0, stream, grid, nx, ny, cx, cy, lay.ncells,
                           level, classic, (uint2*)(ws + lay.off_act), (uint4*)(ws + lay.off_blk), chunk_sums,
                           chunk_nz, status);
    }
    }
    ProfScope ps2(PC_MC_OTHER, 0.0, stream);
    hipLaunchKernelGGL(mc_scan, dim3(lay.nchunk), dim3(kChunk), 0, stream, (const uint4*)(ws + lay.off_blk), lay.nblk,
                       chunk_sums, chunk_nz, (uint2*)(ws + lay.off_blkoff), (uint32_t*)(ws + lay.off_nz), totals);
    return hipGetLastError();
}

hipError_t mc_emit_launch(const float* grid, int n0, int n1, int n2, double level, char* ws,
                          const McWorkspaceLayout& lay, float* verts, int32_t* faces, const double* xf9,
                          int reversed, hipStream_t stream) {
    (void)n0;
    const int nx = n2, ny = n1, cx = n2 - 1, cy = n1 - 1;
    if (lay.nnz == 0) return hipSuccess;
    Xform xf;
    for (int i = 0; i < 3; ++i) {
        xf.grid_size[i] = xf9 ? xf9[i] : 1.0;
        xf.bbox_size[i] = xf9 ? xf9[3 + i] : 1.0;
        xf.bbox_min[i] = xf9 ? xf9[6 + i] : 0.0;
    }
    ProfScope ps(PC_MC_OTHER, 0.0, stream);
    const uint32_t* nz = (const uint32_t*)(ws + lay.off_nz);
    hipLaunchKernelGGL(mc_vertices, dim3(lay.nnz), dim3(kEmitThreads), 0, stream, grid, nx, ny, cx, cy, level,
                       (const uint2*)(ws + lay.off_act), (const uint4*)(ws + lay.off_blk),
                       (const uint2*)(ws + lay.off_blkoff), nz, (int32_t*)(ws + lay.off_etab), verts, xf, xf9 ? 1 : 0);
    hipLaunchKernelGGL(mc_faces, dim3(lay.nnz), dim3(kEmitThreads), 0, stream, nx, ny, cx, cy,
                       (const uint2*)(ws + lay.off_act), (const uint4*)(ws + lay.off_blk),
                       (const uint2*)(ws + lay.off_blkoff), nz, (const int32_t*)(ws + lay.off_etab), faces, reversed);
    return hipGetLastError();
}

}  // namespace r3g
