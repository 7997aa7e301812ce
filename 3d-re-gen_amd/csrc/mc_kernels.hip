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
//       mc_classify_rows2 (row length % 256 == 0, e.g. the 257^3 grid; default): one WAVE per 256-cell block, 4
//                    cells per lane, marching 16 rows along axis 1 so that every node row is loaded once per wave
//                    as dwordx4 + dword, next rows in flight while the current ones are tested.  The sign test is
//                    v_cmp into wave-wide SGPR masks and the "does this cell straddle the level" logic is 64-bit
//                    scalar arithmetic on them (~30 VALU + ~65 SALU instructions per 256-cell row); active cells
//                    are appended to a per-wave list in LDS and the long tiling selection runs once over the list
//                    with all lanes busy; in-row prefix sums by a segmented wave scan; ONE pair of chunk-total
//                    atomics per wave.  Workgroups are numbered so that each XCD owns a slab of z.
//       mc_classify_rows  (option mc_deferred=0): round 1's variant, per-lane bit masks and the tiling selection
//                    per row; kept as the cross-check (same outputs bit for bit).
//       mc_classify      (any other shape): one thread per cell, 8 coalesced row reads, LDS across 4 waves.
//   K2 mc_scan     : exclusive scan of the block sums (one workgroup per 1024-block chunk) + the
//                    compacted list of non-empty blocks.
//   K3 mc_vertices : 1 thread per ACTIVE cell: fp64 interpolation, float32 store, edge->id table.
//   K4 mc_faces    : 1 thread per ACTIVE cell: triangle corners -> ids (own rank or table lookup).
//                    K3/K4 are launched over the non-empty blocks only, 8 consecutive ones per 256-thread workgroup (24 measured slower: too few workgroups).
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
// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Every node slice is read twice (as the lower
// and as the upper face of a cell layer): giving each XCD a contiguous range of task numbers (= a slab of z) makes the
// second read an L2 hit instead of a second trip over the fabric.
__device__ __forceinline__ uint32_t xcd_slab_block(uint32_t bid, uint32_t nb) {
    const uint32_t xcd = bid & 7u, local = bid >> 3, q = nb >> 3, r = nb & 7u;
    return xcd * q + min(xcd, r) + local;
}
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
    const uint32_t task = xcd_slab_block(blockIdx.x, gridDim.x) * (kBlock / 64) + (threadIdx.x >> 6);  // wave-uniform
    if (task >= segs * ychunks * (uint32_t)cz) return;
    const uint32_t seg = task % segs, t2 = task / segs;
    const int z = (int)(t2 / ychunks), y0 = (int)(t2 % ychunks) * kRowsPerWave;
    const int y1 = min(cy, y0 + kRowsPerWave);
    const int xb = (int)seg * 256 + lane * 4;
    const int64_t sz = (int64_t)nx * ny;
    const float* p = grid + (int64_t)z * sz + (int64_t)y0 * nx + xb;
    unsigned flags = 0;
    uint32_t acc_chunk = 0xFFFFFFFFu;
    unsigned long long acc_sum = 0;
    unsigned acc_nz = 0;
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
                    // chunk totals: one atomic pair per wave and chunk, not per row (the ~60 chunk counters of a 257^3
                    // grid otherwise take ~50k same-address atomics)
                    if (acc_chunk != b / kChunk) {
                        if (acc_nz) {
                            if (acc_sum) atomicAdd(&chunk_sums[acc_chunk], acc_sum);
                            atomicAdd(&chunk_nz[acc_chunk], acc_nz);
                        }
                        acc_chunk = b / kChunk; acc_sum = 0; acc_nz = 0;
                    }
                    acc_sum += (unsigned long long)sv | ((unsigned long long)st << 32);
                    acc_nz += 1u;
                }
            }
        }
        m0 = n0; m1 = n1;
        q0 = f0; q1 = f1;
        p += nx;
    }
    if (acc_nz) {   // lane 63 only
        if (acc_sum) atomicAdd(&chunk_sums[acc_chunk], acc_sum);
        atomicAdd(&chunk_nz[acc_chunk], acc_nz);
    }
    const unsigned long long le = __ballot(flags & R3G_MC_FLAG_LE), ge = __ballot(flags & R3G_MC_FLAG_GE),
                             nn = __ballot(flags & R3G_MC_FLAG_NAN);
    const unsigned wf = (le ? R3G_MC_FLAG_LE : 0u) | (ge ? R3G_MC_FLAG_GE : 0u) | (nn ? R3G_MC_FLAG_NAN : 0u);
    if (lane == 0 && (wf & ~*(volatile unsigned*)status)) atomicOr(status, wf);
}

// ---- row kernel, second version: the tiling selection is DEFERRED.  While a wave streams its node rows it only appends
// the cells that straddle the level to a list in LDS (position inside the 256-cell row block + row number); the long,
// latency-bound selection code (fp64 ambiguity tests, chains of dependent table loads) then runs ONCE over the whole
// list, 64 cells at a time, instead of once per row with one or two busy lanes -- a smooth surface crosses ~40 % of
// the rows of a 257^3 grid but activates only ~20 cells per 16 rows.  Records, in-row prefix sums and block totals are
// produced by a segmented wave scan over the list (rows are contiguous in it).  Same outputs as mc_classify_rows.
constexpr int kListCap = 1024;   // entries per wave; a row adds at most 256

template <int kRowsPerWave>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8)))
void mc_classify_rows2(const float* __restrict__ grid, int nx, int ny, int cx, int cy, int cz, float lo, int lo_exact,
                       double level, int classic, uint2* __restrict__ act, uint4* __restrict__ blk,
                       unsigned long long* __restrict__ chunk_sums, unsigned* __restrict__ chunk_nz,
                       unsigned* __restrict__ status) {
    __shared__ unsigned short s_cell[kBlock / 64][kListCap];          // (row - y0) << 8 | cell
    __shared__ unsigned s_rec[kBlock / 64][kListCap];
    __shared__ unsigned long long s_rowbase[kBlock / 64][kRowsPerWave + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t segs = (uint32_t)cx >> 8;
    const uint32_t ychunks = ((uint32_t)cy + kRowsPerWave - 1) / kRowsPerWave;
    const uint32_t task = xcd_slab_block(blockIdx.x, gridDim.x) * (kBlock / 64) + (threadIdx.x >> 6);
    if (task >= segs * ychunks * (uint32_t)cz) return;
    const uint32_t seg = task % segs, t2 = task / segs;
    const int z = (int)(t2 / ychunks), y0 = (int)(t2 % ychunks) * kRowsPerWave;
    const int y1 = min(cy, y0 + kRowsPerWave);
    const int xb = (int)seg * 256 + lane * 4;
    const int64_t sz = (int64_t)nx * ny;
    const float* p = grid + (int64_t)z * sz + (int64_t)y0 * nx + xb;
    unsigned flags = 0;
    unsigned n_list = 0;   // wave-uniform
    uint32_t acc_chunk = 0xFFFFFFFFu;   // per-lane partial chunk totals, reduced to one atomic pair per wave at the end
    unsigned long long acc_sum = 0;
    unsigned acc_nz = 0;

    auto flush = [&]() {
        if (n_list == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 1. tiling selection for every listed cell, 64 at a time
        for (unsigned i = lane; i < n_list; i += 64) {
            const unsigned e = s_cell[wid][i];
            const int x = (int)seg * 256 + (int)(e & 0xFFu), y = y0 + (int)(e >> 8);
            double v[8];
            int index;
            load_corners(grid, nx, ny, x, y, z, level, v, &index);
            s_rec[wid][i] = classify_cell(v, index, classic != 0, x, y, z);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 2. segmented exclusive scan (segments = rows): packed counters [0..15] vertices, [16..31] triangles, [32..47] cells
        unsigned long long carry = 0;
        for (unsigned base = 0; base < n_list; base += 64) {
            const unsigned i = base + lane;
            const bool in = i < n_list;
            const unsigned rec = in ? s_rec[wid][i] : 0u;
            const unsigned row = in ? (unsigned)(s_cell[wid][i] >> 8) : 0xFFFFu;
            const unsigned long long mine = (unsigned long long)((rec >> 20) & 0xFu) |
                                            ((unsigned long long)((rec >> 16) & 0xFu) << 16) |
                                            ((unsigned long long)(rec ? 1u : 0u) << 32);
            const unsigned long long incl = wave_inclusive_scan(mine, lane) + carry;
            const unsigned long long gex = incl - mine;
            // the first entry of a row publishes the running total at the row's start
            const bool head = in && (i == 0 || (unsigned)(s_cell[wid][i - 1] >> 8) != row);
            if (head) s_rowbase[wid][row] = gex;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (in) {
                const unsigned long long excl = gex - s_rowbase[wid][row];
                const uint32_t b = ((uint32_t)z * (uint32_t)cy + (uint32_t)(y0 + (int)row)) * segs + seg;
                if (rec) {
                    const unsigned vloc = (unsigned)(excl & 0xFFFFu), tloc = (unsigned)((excl >> 16) & 0xFFFFu);
                    const unsigned arank = (unsigned)((excl >> 32) & 0xFFFFu);
                    act[(size_t)b * kBlock + arank] =
                        make_uint2(rec, (unsigned)(s_cell[wid][i] & 0xFFu) | (vloc << 8) | (tloc << 20));
                }
                const bool last = i + 1 == n_list || (unsigned)(s_cell[wid][i + 1] >> 8) != row;
                if (last) {
                    const unsigned long long tot = excl + mine;
                    const unsigned sv = (unsigned)(tot & 0xFFFFu), st = (unsigned)((tot >> 16) & 0xFFFFu);
                    const unsigned sa = (unsigned)((tot >> 32) & 0xFFFFu);
                    if (sa) {
                        blk[b] = make_uint4(sv, st, sa, 0u);
                        if (acc_nz && acc_chunk != b / kChunk) {
                            if (acc_sum) atomicAdd(&chunk_sums[acc_chunk], acc_sum);
                            atomicAdd(&chunk_nz[acc_chunk], acc_nz);
                            acc_sum = 0; acc_nz = 0;
                        }
                        acc_chunk = b / kChunk;
                        acc_sum += (unsigned long long)sv | ((unsigned long long)st << 32);
                        acc_nz += 1u;
                    }
                }
            }
            carry = __shfl(incl, 63, 64);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        n_list = 0;
    };

    // Streaming part.  A node row is turned into five wave-wide masks (one per node of a lane's 4-cell group: bit L of
    // G[k] <=> node 4L+k > level) by v_cmp into SGPR pairs; everything that decides whether a cell straddles the level
    // is then 64-bit SCALAR arithmetic on those masks, shared by the rows above / below and the two slices -- about
    // 30 vector and 40 scalar instructions per row of 256 cells.  (The per-lane bit-twiddling this replaces took ~200
    // vector instructions per row and made the kernel VALU-bound at 1.4 TB/s.)
    float vmin = __builtin_inff(), vmax = -__builtin_inff();   // range flags: min / max per lane, NaNs by v_cmp_u
    unsigned long long nan_mask = 0;
    struct RowMasks { unsigned long long o[4], a[4]; };       // per cell column c: OR / AND over nodes c, c+1 of both slices
    auto masks_of = [&](const Row5& r0, const Row5& r1) -> RowMasks {
        vmax = fmaxf(fmaxf(vmax, fmaxf(r0.a.x, r0.a.y)), fmaxf(fmaxf(r0.a.z, r0.a.w), r0.e));
        vmax = fmaxf(fmaxf(vmax, fmaxf(r1.a.x, r1.a.y)), fmaxf(fmaxf(r1.a.z, r1.a.w), r1.e));
        vmin = fminf(fminf(vmin, fminf(r0.a.x, r0.a.y)), fminf(fminf(r0.a.z, r0.a.w), r0.e));
        vmin = fminf(fminf(vmin, fminf(r1.a.x, r1.a.y)), fminf(fminf(r1.a.z, r1.a.w), r1.e));
        nan_mask |= __ballot(__builtin_isunordered(r0.a.x, r0.a.y)) | __ballot(__builtin_isunordered(r0.a.z, r0.a.w)) |
                    __ballot(__builtin_isunordered(r1.a.x, r1.a.y)) | __ballot(__builtin_isunordered(r1.a.z, r1.a.w)) |
                    __ballot(__builtin_isunordered(r0.e, r1.e));
        const unsigned long long g0[5] = {__ballot(r0.a.x > lo), __ballot(r0.a.y > lo), __ballot(r0.a.z > lo),
                                          __ballot(r0.a.w > lo), __ballot(r0.e > lo)};
        const unsigned long long g1[5] = {__ballot(r1.a.x > lo), __ballot(r1.a.y > lo), __ballot(r1.a.z > lo),
                                          __ballot(r1.a.w > lo), __ballot(r1.e > lo)};
        RowMasks m;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            m.o[c] = g0[c] | g0[c + 1] | g1[c] | g1[c + 1];
            m.a[c] = g0[c] & g0[c + 1] & g1[c] & g1[c + 1];
        }
        return m;
    };
    RowMasks mp = masks_of(row_load(p), row_load(p + sz));
    Row5 q0 = row_load(p + nx), q1 = row_load(p + sz + nx);
    for (int y = y0; y < y1; ++y) {
        Row5 f0 = q0, f1 = q1;
        if (y + 1 < y1) {   // rows y+2 go in flight before rows y+1 are consumed
            f0 = row_load(p + 2 * (int64_t)nx);
            f1 = row_load(p + sz + 2 * (int64_t)nx);
        }
        const RowMasks mn = masks_of(q0, q1);
        unsigned long long actm[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) actm[c] = (mp.o[c] | mn.o[c]) & ~(mp.a[c] & mn.a[c]);
        if ((actm[0] | actm[1] | actm[2] | actm[3]) != 0ull) {
            const unsigned n_active = (unsigned)(__popcll(actm[0]) + __popcll(actm[1]) + __popcll(actm[2]) + __popcll(actm[3]));
            if (n_list + n_active > (unsigned)kListCap) flush();
            // list position of a lane's first active cell: active cells in the lanes below (cells are ordered by x)
            unsigned slot = n_list;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                slot += __builtin_amdgcn_mbcnt_hi((unsigned)(actm[c] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)actm[c], 0u));
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if ((actm[c] >> lane) & 1ull) s_cell[wid][slot++] = (unsigned short)(((unsigned)(y - y0) << 8) | (unsigned)(lane * 4 + c));
            n_list += n_active;
        }
        mp = mn;
        q0 = f0; q1 = f1;
        p += nx;
    }
    {
        if (vmin <= lo) flags |= R3G_MC_FLAG_LE;
        if (vmax > lo || (lo_exact && vmax == lo)) flags |= R3G_MC_FLAG_GE;
        if (nan_mask) flags |= R3G_MC_FLAG_NAN;
    }
    flush();
    {
        // all rows of a wave normally fall into one chunk: reduce over the wave, one lane adds
        const unsigned long long have = __ballot(acc_nz != 0u);
        if (have) {
            const uint32_t c0 = (uint32_t)__shfl((int)acc_chunk, __ffsll((long long)have) - 1, 64);
            if (__all(acc_nz == 0u || acc_chunk == c0)) {
                unsigned long long ts = acc_sum;
                unsigned tn = acc_nz;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    ts += (unsigned long long)__shfl_xor((long long)ts, d, 64);
                    tn += (unsigned)__shfl_xor((int)tn, d, 64);
                }
                if (lane == 0) {
                    if (ts) atomicAdd(&chunk_sums[c0], ts);
                    atomicAdd(&chunk_nz[c0], tn);
                }
            } else if (acc_nz) {
                if (acc_sum) atomicAdd(&chunk_sums[acc_chunk], acc_sum);
                atomicAdd(&chunk_nz[acc_chunk], acc_nz);
            }
        }
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

// K3 / K4 geometry (round 5): a workgroup of 256 threads takes kEmitGroup (8) CONSECUTIVE non-empty blocks and spreads their active
// cells over its threads (a prefix over the blocks' active counts in LDS, a short linear search).  Rounds 1-4 gave every non-empty
// block a 64-thread workgroup of its own: a block of a smooth surface holds ~11 active cells, so five lanes in six idled and an
// object-sized mesh took 19 855 workgroups per kernel -- 45 + 25 us for 113 k vertices / 226 k faces (profiles/r05_mc_timeline.md).
// Every output position comes from the scan (block offset + the record's in-block prefix), so the mesh is the same bit for bit.
constexpr int kEmitGroup = 8;
constexpr int kEmitWg = 256;

// the calling thread's (block, index of an active cell in it) for item `i` of the workgroup's cells, or false when i is past them
struct EmitMap {
    unsigned pre[kEmitGroup + 1];
    uint32_t blk_id[kEmitGroup];
};
__device__ __forceinline__ void emit_map_build(EmitMap* m, const uint4* __restrict__ blk, const uint32_t* __restrict__ nzlist,
                                               uint32_t nnz) {
    if (threadIdx.x == 0) {
        unsigned run = 0;
#pragma unroll
        for (int j = 0; j < kEmitGroup; ++j) {
            const uint32_t g = blockIdx.x * kEmitGroup + j;
            m->pre[j] = run;
            if (g < nnz) {
                const uint32_t b = nzlist[g];
                m->blk_id[j] = b;
                run += blk[b].z;
            } else {
                m->blk_id[j] = 0;
            }
        }
        m->pre[kEmitGroup] = run;
    }
    __syncthreads();
}
__device__ __forceinline__ bool emit_map_find(const EmitMap* m, unsigned i, uint32_t* b, unsigned* local) {
    if (i >= m->pre[kEmitGroup]) return false;
    int j = 0;
#pragma unroll
    for (int k = 1; k < kEmitGroup; ++k) j += (i >= m->pre[k]) ? 1 : 0;
    *b = m->blk_id[j];
    *local = i - m->pre[j];
    return true;
}

__global__ __launch_bounds__(kEmitWg) void mc_vertices(const float* __restrict__ grid, int nx, int ny, int cx, int cy,
                                                      double level, const uint2* __restrict__ act,
                                                      const uint4* __restrict__ blk, const uint2* __restrict__ blkoff,
                                                      const uint32_t* __restrict__ nzlist, uint32_t nnz,
                                                      int32_t* __restrict__ etab, float* __restrict__ verts,
                                                      Xform xf, int use_xf) {
    __shared__ EmitMap map;
    emit_map_build(&map, blk, nzlist, nnz);
    for (unsigned i = threadIdx.x;; i += kEmitWg) {
        uint32_t b;
        unsigned tid;
        if (!emit_map_find(&map, i, &b, &tid)) break;
        const uint32_t voff = blkoff[b].x;
        const uint2 a = act[(size_t)b * kBlock + tid];
        const uint32_t c = b * kBlock + (a.y & 0xFFu);
        int x, y, z;
        cell_coords(c, cx, cy, x, y, z);
        emit_cell_vertices(a.x, voff + ((a.y >> 8) & 0xFFFu), grid, level, x, y, z, nx, ny, etab, verts, xf, use_xf != 0);
    }
}

__global__ __launch_bounds__(kEmitWg) void mc_faces(int nx, int ny, int cx, int cy, const uint2* __restrict__ act,
                                                   const uint4* __restrict__ blk, const uint2* __restrict__ blkoff,
                                                   const uint32_t* __restrict__ nzlist, uint32_t nnz,
                                                   const int32_t* __restrict__ etab, int32_t* __restrict__ faces,
                                                   int reversed) {
    __shared__ EmitMap map;
    emit_map_build(&map, blk, nzlist, nnz);
    for (unsigned i = threadIdx.x;; i += kEmitWg) {
        uint32_t b;
        unsigned tid;
        if (!emit_map_find(&map, i, &b, &tid)) break;
        const uint2 off = blkoff[b];
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
static bool g_deferred = true;   // row kernel with the tiling selection batched over all rows of a wave
void mc_set_deferred(bool on) { g_deferred = on; }
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
        if (g_deferred) {
            kern = rows == 4 ? mc_classify_rows2<4> : rows == 8 ? mc_classify_rows2<8> : rows == 32 ? mc_classify_rows2<32>
                                                                                                   : mc_classify_rows2<16>;
        }
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
    const uint32_t wgs = (lay.nnz + kEmitGroup - 1) / kEmitGroup;
    hipLaunchKernelGGL(mc_vertices, dim3(wgs), dim3(kEmitWg), 0, stream, grid, nx, ny, cx, cy, level,
                       (const uint2*)(ws + lay.off_act), (const uint4*)(ws + lay.off_blk),
                       (const uint2*)(ws + lay.off_blkoff), nz, (uint32_t)lay.nnz, (int32_t*)(ws + lay.off_etab), verts, xf, xf9 ? 1 : 0);
    hipLaunchKernelGGL(mc_faces, dim3(wgs), dim3(kEmitWg), 0, stream, nx, ny, cx, cy,
                       (const uint2*)(ws + lay.off_act), (const uint4*)(ws + lay.off_blk),
                       (const uint2*)(ws + lay.off_blkoff), nz, (uint32_t)lay.nnz, (const int32_t*)(ws + lay.off_etab), faces, reversed);
    return hipGetLastError();
}

}  // namespace r3g
