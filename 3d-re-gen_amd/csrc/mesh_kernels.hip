// mesh_kernels.hip -- mesh cleaners on device buffers (HBM-bound index work, gfx950).
//
// Drop-ins for the three cleaners the reference stage applies to every marching-cubes mesh
// (src/2d_to_3d_models/run.py:93-94: FloaterRemover, DegenerateFaceRemover, FaceReducer from
// hy3dgen.shapegen.postprocessors -- pymeshlab on one CPU thread upstream).  SURVEY.md section 8(f) rank 1.
// The mesh stays where marching cubes left it (verts float32 [V][3], faces int32 [F][3] in HBM); every
// operation is a handful of streaming kernels:
//   floaters  : lock-free union-find over the face edges (roots always link to the smaller vertex id, so a
//               component's label is its smallest vertex id, independent of scheduling), faces per component by
//               integer atomics, threshold, order-preserving compaction (chunked exclusive scan).
//   degenerate: flag faces with a repeated index, same compaction.
//   reduce    : vertex clustering on a uniform grid: cell of a vertex in fp64 exactly as the host restatement,
//               occupied cells ranked by an exclusive scan (= sorted unique keys), faces remapped, degenerate and
//               duplicate faces dropped (open-addressing table that keeps the smallest face index of every
//               vertex-set), cluster positions = mean of the members accumulated in 2^-32 fixed point with
//               integer atomics (exact, order independent).  The caller shrinks the grid until the face budget
//               is met.
// Every result is a pure function of the input (no float atomics, no order dependence): the parity tests compare
// bit-for-bit with the numpy restatement the tests hold (mesh_clean).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mesh_kernels.h"
#include "prof.h"

#pragma clang fp contract(off)

#define R3G_QEM_HD static __host__ __device__ __forceinline__
#define R3G_QEM_LAMBDA __device__
#include "qem_driver.h"

namespace r3g {
namespace {

constexpr int kT = 256;
constexpr int kItems = 8;                 // per thread in the scan kernels
constexpr int kTile = kT * kItems;        // 2048 elements per block

inline unsigned nblocks(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// ------------------------------------------------------------------ exclusive scan of 32-bit counts (flags)
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* total) {
    __shared__ unsigned s_wave[kT / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned n = __shfl_up(incl, d, 64);
        if (lane >= d) incl += n;
    }
    if (lane == 63) s_wave[wid] = incl;
    __syncthreads();
    unsigned base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kT / 64; ++w) {
        const unsigned t = s_wave[w];
        if (w < wid) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(kT) void scan_block_sums(const unsigned* __restrict__ in, int64_t n,
                                                      unsigned* __restrict__ bsum) {
    const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < kItems; ++i)
        if (base + i < n) s += in[base + i];
    unsigned tot;
    (void)block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// one workgroup: exclusive scan of the block sums in place; total -> *total_out
__global__ __launch_bounds__(kT) void scan_of_sums(unsigned* __restrict__ bsum, unsigned nb,
                                                   unsigned* __restrict__ total_out) {
    unsigned carry = 0;
    for (unsigned start = 0; start < nb; start += kT) {
        const unsigned i = start + threadIdx.x;
        const unsigned v = i < nb ? bsum[i] : 0u;
        unsigned tot;
        const unsigned ex = block_exclusive_scan(v, &tot);
        if (i < nb) bsum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(kT) void scan_apply(const unsigned* __restrict__ in, int64_t n,
                                                 const unsigned* __restrict__ bsum, unsigned* __restrict__ out) {
    const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
    unsigned v[kItems], s = 0;
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        v[i] = base + i < n ? in[base + i] : 0u;
        s += v[i];
    }
    unsigned tot;
    unsigned ex = bsum[blockIdx.x] + block_exclusive_scan(s, &tot);
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
    }
}

// ------------------------------------------------------------------ union-find over face edges
__device__ __forceinline__ int uf_find(const int* __restrict__ parent, int x) {
    int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}

// find with path halving: a non-root slot is re-pointed at its grandparent (still an ancestor with a smaller id, so
// concurrent finds and links stay correct; a root's slot is only ever changed by the CAS in uf_union)
__device__ __forceinline__ int uf_find_halving(int* __restrict__ parent, int x) {
    int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
        p = gp;
    }
    return x;
}

__device__ __forceinline__ void uf_union(int* __restrict__ parent, int a, int b) {
    for (;;) {
        a = uf_find_halving(parent, a);
        b = uf_find_halving(parent, b);
        if (a == b) return;
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        // the larger root goes under the smaller one; only a root's own slot is ever CASed
        if (atomicCAS(&parent[hi], hi, lo) == hi) return;
    }
}

__global__ void iota_kernel(int* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (int)i;
}

__global__ void uf_hook_kernel(const int32_t* __restrict__ faces, int64_t nf, int* __restrict__ parent) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const int a = faces[3 * i], b = faces[3 * i + 1], c = faces[3 * i + 2];
    uf_union(parent, a, b);
    uf_union(parent, a, c);
}

// ---- faces joined through shared EDGES (MeshLab's face-face adjacency: two parts that touch in a single vertex are two
// components).  Pass 1: every undirected edge (min vertex, max vertex) goes into an open-addressing table, its slot keeps the
// SMALLEST face id that has the edge (atomicMin: independent of scheduling).  Pass 2: every face unites itself with the
// owner of each of its edges -- all faces around an edge (two, or more on a non-manifold edge) end up in one set.
constexpr unsigned long long kEdgeEmpty = ~0ull;
__device__ __forceinline__ unsigned long long edge_key(int a, int b) {
    const unsigned lo = (unsigned)(a < b ? a : b), hi = (unsigned)(a < b ? b : a);
    return ((unsigned long long)lo << 32) | hi;
}
__device__ __forceinline__ uint64_t edge_hash(unsigned long long k) {   // splitmix64 finaliser
    k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
    k ^= k >> 27; k *= 0x94d049bb133111ebull;
    return k ^ (k >> 31);
}
__global__ void edge_insert_kernel(const int32_t* __restrict__ faces, int64_t nf, unsigned long long* __restrict__ keys,
                                   int* __restrict__ owner, uint64_t mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const int v[3] = {faces[3 * i], faces[3 * i + 1], faces[3 * i + 2]};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int a = v[e], b = v[(e + 1) % 3];
        if (a == b) continue;                       // a collapsed side of a degenerate face joins nothing
        const unsigned long long k = edge_key(a, b);
        uint64_t slot = edge_hash(k) & mask;
        for (;;) {
            const unsigned long long prev = atomicCAS(&keys[slot], kEdgeEmpty, k);
            if (prev == kEdgeEmpty || prev == k) { atomicMin(&owner[slot], (int)i); break; }
            slot = (slot + 1) & mask;
        }
    }
}
__global__ void edge_union_kernel(const int32_t* __restrict__ faces, int64_t nf, const unsigned long long* __restrict__ keys,
                                  const int* __restrict__ owner, uint64_t mask, int* __restrict__ parent) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const int v[3] = {faces[3 * i], faces[3 * i + 1], faces[3 * i + 2]};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int a = v[e], b = v[(e + 1) % 3];
        if (a == b) continue;
        const unsigned long long k = edge_key(a, b);
        uint64_t slot = edge_hash(k) & mask;
        while (keys[slot] != k) slot = (slot + 1) & mask;     // present: pass 1 put it there
        const int o = owner[slot];
        if (o != (int)i) uf_union(parent, (int)i, o);
    }
}

// roots never change here (nothing links any more), so every chain ends in its final root
__global__ void uf_flatten_kernel(int* __restrict__ parent, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int r = uf_find(parent, (int)i);
        __hip_atomic_store(&parent[i], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Faces per component.  Most faces of a wave belong to the same (large) component, and same-address atomics
// serialise in L2: lanes with equal roots are merged first (one atomic per distinct root per wave).
// BY_FACE: root[] is indexed by the face (edge-joined components), otherwise by the face's first vertex
template <bool BY_FACE>
__global__ void comp_count_kernel(const int32_t* __restrict__ faces, int64_t nf, const int* __restrict__ root,
                                  unsigned* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int r = i < nf ? root[BY_FACE ? i : (int64_t)faces[3 * i]] : -1;
    unsigned long long todo = __ballot(r >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lr = __shfl(r, leader, 64);
        const unsigned long long same = __ballot(r == lr) & todo;
        if (lane == leader) atomicAdd(&cnt[lr], (unsigned)__popcll(same));
        todo &= ~same;
    }
}

__global__ void max_u32_kernel(const unsigned* __restrict__ v, int64_t n, unsigned* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned m = i < n ? v[i] : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m > *(volatile unsigned*)out) atomicMax(out, m);   // skip atomics that cannot win
}

// keep[f] = faces-of-component >= max(1, (unsigned)(min_ratio * largest)): MeshLab's small-component selection removes the
// components with fewer faces than the TRUNCATED product (compute_selection_by_small_disconnected_components_per_face)
template <bool BY_FACE>
__global__ void floater_flag_kernel(const int32_t* __restrict__ faces, int64_t nf, const int* __restrict__ root,
                                    const unsigned* __restrict__ cnt, const unsigned* __restrict__ largest,
                                    double min_ratio, unsigned* __restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const double t = floor(min_ratio * (double)*largest);
    const unsigned thr = t < 1.0 ? 1u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t);
    keep[i] = cnt[root[BY_FACE ? i : (int64_t)faces[3 * i]]] >= thr ? 1u : 0u;
}

__global__ void nondegenerate_flag_kernel(const int32_t* __restrict__ faces, int64_t nf, unsigned* __restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const int a = faces[3 * i], b = faces[3 * i + 1], c = faces[3 * i + 2];
    keep[i] = (a != b && b != c && a != c) ? 1u : 0u;
}

// ------------------------------------------------------------------ compaction
__global__ void mark_used_kernel(const int32_t* __restrict__ faces, int64_t nf, const unsigned* __restrict__ keep,
                                 unsigned* __restrict__ used) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf || (keep && !keep[i])) return;
    used[faces[3 * i]] = 1u;
    used[faces[3 * i + 1]] = 1u;
    used[faces[3 * i + 2]] = 1u;
}

// faces kept -> position fpos[i], vertex ids through vpos
__global__ void scatter_faces_kernel(const int32_t* __restrict__ faces, int64_t nf, const unsigned* __restrict__ keep,
                                     const unsigned* __restrict__ fpos, const unsigned* __restrict__ vpos,
                                     int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf || !keep[i]) return;
    const int64_t o = 3 * (int64_t)fpos[i];
    out[o] = (int32_t)vpos[faces[3 * i]];
    out[o + 1] = (int32_t)vpos[faces[3 * i + 1]];
    out[o + 2] = (int32_t)vpos[faces[3 * i + 2]];
}

__global__ void scatter_verts_kernel(const float* __restrict__ verts, int64_t nv, const unsigned* __restrict__ used,
                                     const unsigned* __restrict__ vpos, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv || !used[i]) return;
    const int64_t o = 3 * (int64_t)vpos[i];
    out[o] = verts[3 * i];
    out[o + 1] = verts[3 * i + 1];
    out[o + 2] = verts[3 * i + 2];
}

// ------------------------------------------------------------------ vertex clustering
__device__ __forceinline__ unsigned f32_sortable(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float f32_unsortable(unsigned s) {
    const unsigned u = (s & 0x80000000u) ? (s & 0x7FFFFFFFu) : ~s;
    union { unsigned u; float f; } c;
    c.u = u;
    return c.f;
}

// bbox[0..2] = min, bbox[3..5] = max (sortable-uint encoding; init min = 0xFFFFFFFF, max = 0)
__global__ void bbox_kernel(const float* __restrict__ verts, int64_t nv, unsigned* __restrict__ bbox) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    if (i < nv) {
#pragma unroll
        for (int a = 0; a < 3; ++a) lo[a] = hi[a] = f32_sortable(verts[3 * i + a]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const unsigned l = __shfl_xor(lo[a], d, 64), h = __shfl_xor(hi[a], d, 64);
            lo[a] = l < lo[a] ? l : lo[a];
            hi[a] = h > hi[a] ? h : hi[a];
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {   // same-address atomics serialise: only the ones that still improve the box
            if (lo[a] < *(volatile unsigned*)&bbox[a]) atomicMin(&bbox[a], lo[a]);
            if (hi[a] > *(volatile unsigned*)&bbox[3 + a]) atomicMax(&bbox[3 + a], hi[a]);
        }
    }
}

struct ClusterGrid {
    double lo[3];
    double extent;
    int res;
};

// key[v] = cell of vertex v; occ[key] = 1
__global__ void cluster_key_kernel(const float* __restrict__ verts, int64_t nv, ClusterGrid g, int* __restrict__ key,
                                   unsigned* __restrict__ occ) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    int64_t c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // numpy: floor((v - lo) / extent * res).astype(int64).clip(0, res - 1), v held in float64
        const double t = floor(((double)verts[3 * i + a] - g.lo[a]) / g.extent * (double)g.res);
        int64_t q = (int64_t)t;
        q = q < 0 ? 0 : (q > g.res - 1 ? g.res - 1 : q);
        c[a] = q;
    }
    const int k = (int)((c[0] * g.res + c[1]) * g.res + c[2]);
    key[i] = k;
    occ[k] = 1u;
}

__global__ void cluster_assign_kernel(const int* __restrict__ key, int64_t nv, const unsigned* __restrict__ rank,
                                      int* __restrict__ inv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nv) inv[i] = (int)rank[key[i]];
}

struct Tri { int a, b, c; };
__device__ __forceinline__ Tri sorted_tri(const int32_t* __restrict__ faces, const int* __restrict__ inv, int64_t i) {
    int a = inv[faces[3 * i]], b = inv[faces[3 * i + 1]], c = inv[faces[3 * i + 2]];
    if (a > b) { const int t = a; a = b; b = t; }
    if (b > c) { const int t = b; b = c; c = t; }
    if (a > b) { const int t = a; a = b; b = t; }
    Tri r = {a, b, c};
    return r;
}
__device__ __forceinline__ unsigned tri_hash(const Tri& t) {
    unsigned long long h = (unsigned long long)(unsigned)t.a * 0x9E3779B97F4A7C15ull;
    h ^= (unsigned long long)(unsigned)t.b * 0xC2B2AE3D27D4EB4Full + (h >> 29);
    h ^= (unsigned long long)(unsigned)t.c * 0x165667B19E3779F9ull + (h >> 31);
    h ^= h >> 32;
    return (unsigned)h;
}

// table slot = smallest index among the non-degenerate faces that share one (unordered) cluster triple
__global__ void dedup_insert_kernel(const int32_t* __restrict__ faces, int64_t nf, const int* __restrict__ inv,
                                    int* __restrict__ table, unsigned mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const Tri t = sorted_tri(faces, inv, i);
    if (t.a == t.b || t.b == t.c) return;
    unsigned p = tri_hash(t) & mask;
    for (;;) {
        int cur = __hip_atomic_load(&table[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur < 0) {
            cur = atomicCAS(&table[p], -1, (int)i);
            if (cur < 0) return;
        }
        const Tri o = sorted_tri(faces, inv, cur);   // every face that ever sits in this slot has the same triple
        if (o.a == t.a && o.b == t.b && o.c == t.c) {
            atomicMin(&table[p], (int)i);
            return;
        }
        p = (p + 1) & mask;
    }
}

__global__ void dedup_flag_kernel(const int32_t* __restrict__ faces, int64_t nf, const int* __restrict__ inv,
                                  const int* __restrict__ table, unsigned mask, unsigned* __restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const Tri t = sorted_tri(faces, inv, i);
    unsigned k = 0;
    if (t.a != t.b && t.b != t.c) {
        unsigned p = tri_hash(t) & mask;
        for (;;) {
            const int cur = table[p];
            if (cur < 0) break;   // cannot happen after the insert pass
            const Tri o = sorted_tri(faces, inv, cur);
            if (o.a == t.a && o.b == t.b && o.c == t.c) {
                k = cur == (int)i ? 1u : 0u;
                break;
            }
            p = (p + 1) & mask;
        }
    }
    keep[i] = k;
}

// faces kept -> cluster ids (original corner order), clusters referenced -> used
__global__ void cluster_faces_kernel(const int32_t* __restrict__ faces, int64_t nf, const int* __restrict__ inv,
                                     const unsigned* __restrict__ keep, const unsigned* __restrict__ fpos,
                                     int32_t* __restrict__ out, unsigned* __restrict__ used) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf || !keep[i]) return;
    const int64_t o = 3 * (int64_t)fpos[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = inv[faces[3 * i + j]];
        out[o + j] = c;
        used[c] = 1u;
    }
}

// sums[c][a] += rint(v * 2^32) (exact integer accumulation), cnt[c] += 1
__global__ void cluster_sum_kernel(const float* __restrict__ verts, int64_t nv, const int* __restrict__ inv,
                                   long long* __restrict__ sums, unsigned* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const int c = inv[i];
    atomicAdd(&cnt[c], 1u);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const long long q = (long long)rint((double)verts[3 * i + a] * 4294967296.0);
        atomicAdd((unsigned long long*)&sums[3 * (int64_t)c + a], (unsigned long long)q);
    }
}

// out vertex (compacted position vpos[c]) = float32( sum / 2^32 / count )
__global__ void cluster_mean_kernel(const long long* __restrict__ sums, const unsigned* __restrict__ cnt, int64_t nc,
                                    const unsigned* __restrict__ used, const unsigned* __restrict__ vpos,
                                    float* __restrict__ out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc || !used[c]) return;
    const int64_t o = 3 * (int64_t)vpos[c];
#pragma unroll
    for (int a = 0; a < 3; ++a) out[o + a] = (float)((double)sums[3 * c + a] / 4294967296.0 / (double)cnt[c]);
}

__global__ void remap_faces_kernel(int32_t* __restrict__ faces, int64_t n3, const unsigned* __restrict__ vpos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) faces[i] = (int32_t)vpos[faces[i]];
}

// ------------------------------------------------------------------ host-side helpers
struct Arena {
    char* base;
    size_t off, cap;
    template <typename T>
    T* take(int64_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base + off);
        off += sizeof(T) * (size_t)(n > 0 ? n : 1);
        return p;
    }
};

#define R3G_HIP(x)                         \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return e_;   \
    } while (0)

// out[i] = sum of in[0..i), *d_total = sum of all; scratch: block sums
hipError_t exclusive_scan(const unsigned* in, int64_t n, unsigned* out, unsigned* bsum, unsigned* d_total,
                          hipStream_t s) {
    const unsigned nb = nblocks(n, kTile);
    hipLaunchKernelGGL(scan_block_sums, dim3(nb), dim3(kT), 0, s, in, n, bsum);
    hipLaunchKernelGGL(scan_of_sums, dim3(1), dim3(kT), 0, s, bsum, nb, d_total);
    hipLaunchKernelGGL(scan_apply, dim3(nb), dim3(kT), 0, s, in, n, (const unsigned*)bsum, out);
    return hipGetLastError();
}

size_t scan_scratch_elems(int64_t n) { return (size_t)nblocks(n, kTile) + 1; }

}  // namespace

size_t mesh_workspace_bytes(int64_t nv, int64_t nf, int64_t max_cells) {
    const int64_t m = nv > nf ? nv : nf;
    const int64_t big = m > max_cells ? m : max_cells;
    int64_t table = 1;
    while (table < 2 * nf) table <<= 1;
    size_t b = 0;
    auto add = [&](size_t bytes) { b += (bytes + 255) & ~(size_t)255; };
    add(4 * (size_t)nv);           // parent / key
    add(4 * (size_t)nv);           // cnt / inv
    add(4 * (size_t)nf);           // keep
    add(4 * (size_t)nf);           // fpos
    add(4 * (size_t)big);          // used / occ
    add(4 * (size_t)big);          // vpos / rank
    add(4 * scan_scratch_elems(big));
    add(12 * (size_t)nf);          // faces out
    add(12 * (size_t)nv);          // verts out
    add(4 * (size_t)table);        // dedup table
    add(24 * (size_t)nv);          // cluster sums
    add(4 * (size_t)nv);           // cluster counts
    add(256);                      // small results
    {   // floater removal through shared edges: face parents / counts + the edge table (released before the compaction)
        int64_t cap = 64;
        while (cap < 6 * nf) cap <<= 1;
        size_t e = 0;
        auto adde = [&](size_t bytes) { e += (bytes + 255) & ~(size_t)255; };
        adde(64); adde(4 * (size_t)nf); adde(4 * (size_t)nf); adde(4 * (size_t)nf); adde(8 * (size_t)cap); adde(4 * (size_t)cap);
        if (e > b) b = e;
    }
    // the edge-collapse decimator (qem_driver.h Buffers)
    size_t q = 0;
    auto addq = [&](size_t bytes) { q += (bytes + 255) & ~(size_t)255; };
    addq(256);
    addq(4 * (size_t)nv); addq(4 * (size_t)(nv + 1)); addq(12 * (size_t)nf); addq(80 * (size_t)nv); addq((size_t)nv);
    addq(4 * (size_t)nv); addq(8 * (size_t)nv); addq(4 * (size_t)nv); addq(8 * (size_t)nv); addq(4 * (size_t)nv);
    addq(4 * (size_t)nv); addq(4 * (size_t)m); addq(4 * (size_t)m); addq(12 * (size_t)nf); addq(12 * (size_t)nv);
    addq(4 * (size_t)nv); addq(8 * (size_t)nv); addq(4 * (size_t)nv); addq(4 * scan_scratch_elems(3 * nf > m ? 3 * nf : m));
    return (b > q ? b : q) + 4096;
}

// Shared tail: drop the faces with keep == 0 and the vertices no kept face references (order preserved).
static hipError_t compact_mesh(Arena& ar, float* verts, int64_t nv, int32_t* faces, int64_t nf, const unsigned* keep,
                               unsigned* d_small, unsigned* h_small, int64_t* nv_out, int64_t* nf_out,
                               hipStream_t s) {
    unsigned* fpos = ar.take<unsigned>(nf);
    unsigned* used = ar.take<unsigned>(nv);
    unsigned* vpos = ar.take<unsigned>(nv);
    const int64_t m = nv > nf ? nv : nf;
    unsigned* bsum = ar.take<unsigned>((int64_t)scan_scratch_elems(m));
    int32_t* fout = ar.take<int32_t>(3 * nf);
    float* vout = ar.take<float>(3 * nv);
    R3G_HIP(hipMemsetAsync(used, 0, 4 * (size_t)nv, s));
    R3G_HIP(exclusive_scan(keep, nf, fpos, bsum, d_small + 0, s));
    hipLaunchKernelGGL(mark_used_kernel, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf, keep, used);
    R3G_HIP(exclusive_scan(used, nv, vpos, bsum, d_small + 1, s));
    hipLaunchKernelGGL(scatter_faces_kernel, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf, keep,
                       (const unsigned*)fpos, (const unsigned*)vpos, fout);
    hipLaunchKernelGGL(scatter_verts_kernel, dim3(nblocks(nv, kT)), dim3(kT), 0, s, (const float*)verts, nv,
                       (const unsigned*)used, (const unsigned*)vpos, vout);
    R3G_HIP(hipGetLastError());
    R3G_HIP(hipMemcpyAsync(h_small, d_small, 8, hipMemcpyDeviceToHost, s));
    R3G_HIP(hipStreamSynchronize(s));
    *nf_out = h_small[0];
    *nv_out = h_small[1];
    R3G_HIP(hipMemcpyAsync(faces, fout, 12 * (size_t)*nf_out, hipMemcpyDeviceToDevice, s));
    R3G_HIP(hipMemcpyAsync(verts, vout, 12 * (size_t)*nv_out, hipMemcpyDeviceToDevice, s));
    return hipSuccess;
}

static bool g_floater_by_vertex = false;   // option "floater_by_vertex": rounds 1-2 joined components through shared vertices
void mesh_set_floater_by_vertex(bool on) { g_floater_by_vertex = on; }

static int64_t edge_table_slots(int64_t nf) {
    int64_t cap = 64;
    while (cap < 6 * nf) cap <<= 1;      // 3 nf edge insertions at most: load factor <= 0.5
    return cap;
}

hipError_t mesh_remove_floaters(char* ws, size_t ws_bytes, unsigned* h_small, float* verts, int64_t* nv_io,
                                int32_t* faces, int64_t* nf_io, double min_ratio, hipStream_t s) {
    const int64_t nv = *nv_io, nf = *nf_io;
    if (nv == 0 || nf == 0) return hipSuccess;
    ProfScope ps(PC_MESH, 12.0 * (double)(nv + nf), s);
    Arena ar = {ws, 0, ws_bytes};
    unsigned* d_small = ar.take<unsigned>(16);
    R3G_HIP(hipMemsetAsync(d_small, 0, 64, s));
    unsigned* keep = ar.take<unsigned>(nf);
    if (g_floater_by_vertex) {
        // round 1 / 2 semantics: components joined through shared vertices
        int* parent = ar.take<int>(nv);
        unsigned* cnt = ar.take<unsigned>(nv);
        R3G_HIP(hipMemsetAsync(cnt, 0, 4 * (size_t)nv, s));
        hipLaunchKernelGGL(iota_kernel, dim3(nblocks(nv, kT)), dim3(kT), 0, s, parent, nv);
        hipLaunchKernelGGL(uf_hook_kernel, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf, parent);
        hipLaunchKernelGGL(uf_flatten_kernel, dim3(nblocks(nv, kT)), dim3(kT), 0, s, parent, nv);
        hipLaunchKernelGGL(comp_count_kernel<false>, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf,
                           (const int*)parent, cnt);
        hipLaunchKernelGGL(max_u32_kernel, dim3(nblocks(nv, kT)), dim3(kT), 0, s, (const unsigned*)cnt, nv, d_small + 2);
        hipLaunchKernelGGL(floater_flag_kernel<false>, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf,
                           (const int*)parent, (const unsigned*)cnt, (const unsigned*)(d_small + 2), min_ratio, keep);
    } else {
        // MeshLab's semantics: faces joined through shared edges (union-find over FACES, edge -> smallest face table)
        const int64_t cap = edge_table_slots(nf);
        int* parent = ar.take<int>(nf);
        unsigned* cnt = ar.take<unsigned>(nf);
        unsigned long long* keys = ar.take<unsigned long long>(cap);
        int* owner = ar.take<int>(cap);
        if (ar.off > ar.cap) return hipErrorOutOfMemory;
        R3G_HIP(hipMemsetAsync(cnt, 0, 4 * (size_t)nf, s));
        R3G_HIP(hipMemsetAsync(keys, 0xFF, 8 * (size_t)cap, s));
        R3G_HIP(hipMemsetAsync(owner, 0x7F, 4 * (size_t)cap, s));      // 0x7F7F7F7F: above every face id
        hipLaunchKernelGGL(iota_kernel, dim3(nblocks(nf, kT)), dim3(kT), 0, s, parent, nf);
        hipLaunchKernelGGL(edge_insert_kernel, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf, keys, owner,
                           (uint64_t)(cap - 1));
        hipLaunchKernelGGL(edge_union_kernel, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf,
                           (const unsigned long long*)keys, (const int*)owner, (uint64_t)(cap - 1), parent);
        hipLaunchKernelGGL(uf_flatten_kernel, dim3(nblocks(nf, kT)), dim3(kT), 0, s, parent, nf);
        hipLaunchKernelGGL(comp_count_kernel<true>, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf,
                           (const int*)parent, cnt);
        hipLaunchKernelGGL(max_u32_kernel, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const unsigned*)cnt, nf, d_small + 2);
        hipLaunchKernelGGL(floater_flag_kernel<true>, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf,
                           (const int*)parent, (const unsigned*)cnt, (const unsigned*)(d_small + 2), min_ratio, keep);
        // the tables are dead from here on: compact_mesh may reuse their space
        ar.off = (size_t)(reinterpret_cast<char*>(parent) - ar.base);
    }
    R3G_HIP(hipGetLastError());
    return compact_mesh(ar, verts, nv, faces, nf, keep, d_small, h_small, nv_io, nf_io, s);
}

hipError_t mesh_remove_degenerate(char* ws, size_t ws_bytes, unsigned* h_small, float* verts, int64_t* nv_io,
                                  int32_t* faces, int64_t* nf_io, hipStream_t s) {
    const int64_t nv = *nv_io, nf = *nf_io;
    if (nv == 0 || nf == 0) return hipSuccess;
    ProfScope ps(PC_MESH, 12.0 * (double)(nv + nf), s);
    Arena ar = {ws, 0, ws_bytes};
    unsigned* d_small = ar.take<unsigned>(16);
    unsigned* keep = ar.take<unsigned>(nf);
    R3G_HIP(hipMemsetAsync(d_small, 0, 64, s));
    hipLaunchKernelGGL(nondegenerate_flag_kernel, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf, keep);
    R3G_HIP(hipGetLastError());
    return compact_mesh(ar, verts, nv, faces, nf, keep, d_small, h_small, nv_io, nf_io, s);
}

int mesh_reduce_initial_res(int64_t max_faces) {
    // a closed surface crossing an r^3 grid has ~2.2 r^2 faces
    int r = (int)sqrt((double)max_faces / 2.2);
    return r < 4 ? 4 : r;
}

hipError_t mesh_cluster_faces(char* ws, size_t ws_bytes, unsigned* h_small, float* verts, int64_t* nv_io,
                              int32_t* faces, int64_t* nf_io, int64_t max_faces, hipStream_t s) {
    const int64_t nv = *nv_io, nf = *nf_io;
    if (nv == 0 || nf == 0 || nf <= max_faces) return hipSuccess;
    ProfScope ps(PC_MESH, 12.0 * (double)(nv + nf), s);
    Arena ar = {ws, 0, ws_bytes};
    unsigned* d_small = ar.take<unsigned>(16);
    int res = mesh_reduce_initial_res(max_faces);
    const int64_t max_cells = (int64_t)res * res * res;
    const int64_t big = (nv > nf ? nv : nf) > max_cells ? (nv > nf ? nv : nf) : max_cells;
    int* key = ar.take<int>(nv);
    int* inv = ar.take<int>(nv);
    unsigned* keep = ar.take<unsigned>(nf);
    unsigned* fpos = ar.take<unsigned>(nf);
    unsigned* occ = ar.take<unsigned>(big);    // occupied cells, later: referenced clusters
    unsigned* rank = ar.take<unsigned>(big);   // cell -> cluster id, later: cluster -> output vertex
    unsigned* bsum = ar.take<unsigned>((int64_t)scan_scratch_elems(big));
    int32_t* fout = ar.take<int32_t>(3 * nf);
    float* vout = ar.take<float>(3 * nv);
    int64_t tsize = 1;
    while (tsize < 2 * nf) tsize <<= 1;
    int* table = ar.take<int>(tsize);
    long long* sums = ar.take<long long>(3 * nv);
    unsigned* ccnt = ar.take<unsigned>(nv);

    // bounding box
    {
        const unsigned init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
        R3G_HIP(hipMemcpyAsync(d_small + 4, init, sizeof(init), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(bbox_kernel, dim3(nblocks(nv, kT)), dim3(kT), 0, s, (const float*)verts, nv, d_small + 4);
        R3G_HIP(hipMemcpyAsync(h_small, d_small + 4, 24, hipMemcpyDeviceToHost, s));
        R3G_HIP(hipStreamSynchronize(s));
    }
    ClusterGrid g;
    double ext = 0.0;
    for (int a = 0; a < 3; ++a) {
        g.lo[a] = (double)f32_unsortable(h_small[a]);
        const double d = (double)f32_unsortable(h_small[3 + a]) - g.lo[a];
        ext = d > ext ? d : ext;
    }
    g.extent = ext > 1e-12 ? ext : 1e-12;

    int64_t ncl = 0, nkeep = 0;
    for (int iter = 0; iter < 24; ++iter) {
        g.res = res;
        const int64_t cells = (int64_t)res * res * res;
        R3G_HIP(hipMemsetAsync(occ, 0, 4 * (size_t)cells, s));
        R3G_HIP(hipMemsetAsync(table, 0xFF, 4 * (size_t)tsize, s));
        hipLaunchKernelGGL(cluster_key_kernel, dim3(nblocks(nv, kT)), dim3(kT), 0, s, (const float*)verts, nv, g, key, occ);
        R3G_HIP(exclusive_scan(occ, cells, rank, bsum, d_small + 0, s));
        hipLaunchKernelGGL(cluster_assign_kernel, dim3(nblocks(nv, kT)), dim3(kT), 0, s, (const int*)key, nv,
                           (const unsigned*)rank, inv);
        hipLaunchKernelGGL(dedup_insert_kernel, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf,
                           (const int*)inv, table, (unsigned)(tsize - 1));
        hipLaunchKernelGGL(dedup_flag_kernel, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf,
                           (const int*)inv, (const int*)table, (unsigned)(tsize - 1), keep);
        R3G_HIP(exclusive_scan(keep, nf, fpos, bsum, d_small + 1, s));
        R3G_HIP(hipMemcpyAsync(h_small, d_small, 8, hipMemcpyDeviceToHost, s));
        R3G_HIP(hipStreamSynchronize(s));
        ncl = h_small[0];
        nkeep = h_small[1];
        if (nkeep <= max_faces) break;
        const int next = (int)((double)res * 0.9);
        res = next < 2 ? 2 : next;
    }
    // cluster positions and the compacted output
    R3G_HIP(hipMemsetAsync(sums, 0, 24 * (size_t)ncl, s));
    R3G_HIP(hipMemsetAsync(ccnt, 0, 4 * (size_t)ncl, s));
    R3G_HIP(hipMemsetAsync(occ, 0, 4 * (size_t)ncl, s));
    hipLaunchKernelGGL(cluster_sum_kernel, dim3(nblocks(nv, kT)), dim3(kT), 0, s, (const float*)verts, nv,
                       (const int*)inv, sums, ccnt);
    hipLaunchKernelGGL(cluster_faces_kernel, dim3(nblocks(nf, kT)), dim3(kT), 0, s, (const int32_t*)faces, nf,
                       (const int*)inv, (const unsigned*)keep, (const unsigned*)fpos, fout, occ);
    R3G_HIP(exclusive_scan(occ, ncl, rank, bsum, d_small + 2, s));
    hipLaunchKernelGGL(cluster_mean_kernel, dim3(nblocks(ncl, kT)), dim3(kT), 0, s, (const long long*)sums,
                       (const unsigned*)ccnt, ncl, (const unsigned*)occ, (const unsigned*)rank, vout);
    hipLaunchKernelGGL(remap_faces_kernel, dim3(nblocks(3 * nkeep, kT)), dim3(kT), 0, s, fout, 3 * nkeep,
                       (const unsigned*)rank);
    R3G_HIP(hipGetLastError());
    R3G_HIP(hipMemcpyAsync(h_small, d_small + 2, 4, hipMemcpyDeviceToHost, s));
    R3G_HIP(hipStreamSynchronize(s));
    *nf_io = nkeep;
    *nv_io = h_small[0];
    R3G_HIP(hipMemcpyAsync(faces, fout, 12 * (size_t)*nf_io, hipMemcpyDeviceToDevice, s));
    R3G_HIP(hipMemcpyAsync(verts, vout, 12 * (size_t)*nv_io, hipMemcpyDeviceToDevice, s));
    return hipSuccess;
}

// ------------------------------------------------------------------ quadric edge collapse (qem_core.h / qem_driver.h)
namespace {

template <class F>
__global__ __launch_bounds__(kT) void parfor_kernel(F f, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i < n) f(i);
}

__global__ __launch_bounds__(kT) void sum_if_kernel(const uint32_t* __restrict__ w, const uint64_t* __restrict__ key,
                                                     int64_t n, uint64_t thr, unsigned long long* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    unsigned long long v = 0;
    if (i < n && w[i] && key[i] <= thr) v = w[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

struct HipBackend {
    hipStream_t s;
    unsigned* bsum;
    unsigned* d_small;     // >= 16 bytes
    unsigned* h_small;     // pinned
    hipError_t err = hipSuccess;
    struct Atomics {
        __device__ static uint32_t inc(uint32_t* p) { return atomicAdd(p, 1u); }
        __device__ static void min64(uint64_t* p, uint64_t v) { atomicMin((unsigned long long*)p, (unsigned long long)v); }
        __device__ static void min32(int32_t* p, int32_t v) { atomicMin(p, v); }
    };
    void note(hipError_t e) { if (err == hipSuccess && e != hipSuccess) err = e; }
    template <class F>
    void parfor(int64_t n, F f) {
        if (n <= 0) return;
        hipLaunchKernelGGL(parfor_kernel<F>, dim3(nblocks(n, kT)), dim3(kT), 0, s, f, n);
    }
    uint32_t scan(const uint32_t* in, int64_t n, uint32_t* out) {
        if (n <= 0) return 0;
        note(exclusive_scan(in, n, out, bsum, d_small, s));
        note(hipMemcpyAsync(h_small, d_small, 4, hipMemcpyDeviceToHost, s));
        note(hipStreamSynchronize(s));
        return h_small[0];
    }
    uint64_t sum_if(const uint32_t* w, const uint64_t* key, int64_t n, uint64_t thr) {
        note(hipMemsetAsync(d_small + 2, 0, 8, s));
        hipLaunchKernelGGL(sum_if_kernel, dim3(nblocks(n, kT)), dim3(kT), 0, s, w, key, n, thr,
                           (unsigned long long*)(d_small + 2));
        note(hipMemcpyAsync(h_small + 2, d_small + 2, 8, hipMemcpyDeviceToHost, s));
        note(hipStreamSynchronize(s));
        return *(const unsigned long long*)(h_small + 2);
    }
    void zero(void* p, size_t bytes) { if (bytes) note(hipMemsetAsync(p, 0, bytes, s)); }
    void copy(void* dst, const void* src, size_t bytes) {
        if (bytes) note(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    }
};

}  // namespace

// FaceReducer: quadric-error-metric edge collapse down to <= max_faces faces (see qem_core.h)
hipError_t mesh_reduce_faces(char* ws, size_t ws_bytes, unsigned* h_small, float* verts, int64_t* nv_io,
                             int32_t* faces, int64_t* nf_io, int64_t max_faces, int* rounds_out, hipStream_t s) {
    const int64_t nv = *nv_io, nf = *nf_io;
    if (rounds_out) *rounds_out = 0;
    if (nv == 0 || nf == 0 || nf <= max_faces) return hipSuccess;
    ProfScope ps(PC_MESH, 12.0 * (double)(nv + nf), s);
    Arena ar = {ws, 0, ws_bytes};
    const int64_t m = nv > nf ? nv : nf;
    HipBackend be{s, nullptr, ar.take<unsigned>(16), h_small};
    r3g_qem::Buffers b;
    b.verts = verts; b.faces = faces;
    b.deg = ar.take<uint32_t>(nv); b.off = ar.take<uint32_t>(nv + 1); b.adj = ar.take<int32_t>(3 * nf);
    b.quad = ar.take<double>(10 * nv); b.bnd = ar.take<uint8_t>(nv); b.partner = ar.take<int32_t>(nv);
    b.key = ar.take<uint64_t>(nv); b.mark_lo = ar.take<int32_t>(nv); b.mark_key = ar.take<uint64_t>(nv);
    b.sel = ar.take<uint32_t>(nv); b.remap = ar.take<int32_t>(nv); b.keep = ar.take<uint32_t>(m);
    b.pos = ar.take<uint32_t>(m); b.faces_tmp = ar.take<int32_t>(3 * nf); b.verts_tmp = ar.take<float>(3 * nv);
    b.used = ar.take<uint32_t>(nv);
    b.inkey = ar.take<uint64_t>(nv); b.inwho = ar.take<int32_t>(nv);
    be.bsum = ar.take<unsigned>((int64_t)scan_scratch_elems(3 * nf > m ? 3 * nf : m));
    if (ar.off > ws_bytes) return hipErrorOutOfMemory;
    const r3g_qem::Result r = r3g_qem::decimate(be, b, nv, nf, max_faces);
    R3G_HIP(be.err);
    R3G_HIP(hipGetLastError());
    R3G_HIP(hipStreamSynchronize(s));
    *nv_io = r.nv;
    *nf_io = r.nf;
    if (rounds_out) *rounds_out = r.rounds;
    return hipSuccess;
}

}  // namespace r3g
