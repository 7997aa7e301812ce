// qem_driver.h -- the round loop of the edge-collapse decimator, written once against a small "backend" interface
// (parallel-for over an index range, exclusive scan of 32-bit flags, integer atomics, sum reduction read back to the host).
// qem.hip instantiates it with HIP kernels on device buffers; tests/emu/qem_emu.cpp with host loops (test only).
#ifndef R3G_QEM_DRIVER_H
#define R3G_QEM_DRIVER_H
#include <stdint.h>

#include "qem_core.h"

namespace r3g_qem {

struct Buffers {               // all sized for the INPUT mesh (nv vertices, nf faces)
    float* verts;              // [nv][3]  in / out
    int32_t* faces;            // [nf][3]  in / out
    uint32_t* deg;             // [nv]     faces per vertex, then insertion cursor
    uint32_t* off;             // [nv + 1]
    int32_t* adj;              // [3 nf]
    double* quad;              // [nv][10]
    uint8_t* bnd;              // [nv]
    int32_t* partner;          // [nv]
    uint64_t* key;             // [nv]
    int32_t* mark_lo;          // [nv]
    uint64_t* mark_key;        // [nv]
    uint32_t* sel;             // [nv]     faces removed by the surviving collapse whose lower endpoint this is (0: none)
    int32_t* remap;            // [nv]
    uint32_t* keep;            // [max(nv, nf)] flags
    uint32_t* pos;             // [max(nv, nf)] scan result
    int32_t* faces_tmp;        // [nf][3]
    float* verts_tmp;          // [nv][3]
    uint32_t* used;            // [nv]     proposal state during a round's selection; vertex flags at the end
    uint64_t* inkey;           // [nv]     smallest key among the undecided proposals that END at this vertex
    int32_t* inwho;            // [nv]     ... and the smallest proposer id among those with that key
};

struct Result { int64_t nv, nf; int rounds; int64_t stalled_at; };

// Backend interface (B):
//   template <class F> void parfor(int64_t n, F f)            f(i) for i in [0, n)
//   uint32_t scan(const uint32_t* in, int64_t n, uint32_t* out)     exclusive scan, returns the total
//   uint64_t sum_if(const uint32_t* w, const uint64_t* key, int64_t n, uint64_t thr)   sum of w[i] with w[i] && key[i] <= thr
//   void zero(void* p, size_t bytes);  void copy(void* dst, const void* src, size_t bytes)
//   static Atomics::inc(uint32_t* p) -> old value, Atomics::min64(uint64_t*, uint64_t), Atomics::min32(int32_t*, int32_t)
//                                                                       (callable from inside parfor bodies)
template <class B>
Result decimate(B& be, const Buffers& b, int64_t nv, int64_t nf, int64_t max_faces, int max_rounds = 400) {
    using AT = typename B::Atomics;
    Result res{nv, nf, 0, -1};
    float* verts = b.verts;
    int32_t* faces = b.faces;

    // drop faces with a repeated index (they would break the adjacency walks)
    auto compact_faces = [&](int64_t n) -> int64_t {
        const uint32_t kept = be.scan(b.keep, n, b.pos);
        const uint32_t* keep = b.keep;
        const uint32_t* pos = b.pos;
        const int32_t* src = faces;
        int32_t* dst = b.faces_tmp;
        be.parfor(n, [=] R3G_QEM_LAMBDA(int64_t i) {
            if (keep[i]) {
                dst[3 * (int64_t)pos[i]] = src[3 * i];
                dst[3 * (int64_t)pos[i] + 1] = src[3 * i + 1];
                dst[3 * (int64_t)pos[i] + 2] = src[3 * i + 2];
            }
        });
        be.copy(faces, b.faces_tmp, 12 * (size_t)kept);
        return kept;
    };
    {
        uint32_t* keep = b.keep;
        const int32_t* f = faces;
        be.parfor(nf, [=] R3G_QEM_LAMBDA(int64_t i) {
            keep[i] = (f[3 * i] != f[3 * i + 1] && f[3 * i + 1] != f[3 * i + 2] && f[3 * i] != f[3 * i + 2]) ? 1u : 0u;
        });
        nf = compact_faces(nf);
    }

    auto build_adjacency = [&]() {
        be.zero(b.deg, 4 * (size_t)nv);
        uint32_t* deg = b.deg;
        const int32_t* f = faces;
        be.parfor(3 * nf, [=] R3G_QEM_LAMBDA(int64_t i) { AT::inc(&deg[f[i]]); });
        const uint32_t total = be.scan(b.deg, nv, b.off);
        uint32_t* off = b.off;
        be.parfor(1, [=] R3G_QEM_LAMBDA(int64_t) { off[nv] = total; });
        be.zero(b.deg, 4 * (size_t)nv);
        int32_t* adj = b.adj;
        be.parfor(3 * nf, [=] R3G_QEM_LAMBDA(int64_t i) {
            const int v = f[i];
            adj[off[v] + AT::inc(&deg[v])] = (int32_t)(i / 3);
        });
        // a deterministic order: ascending face ids (insertion sort, the lists are short)
        be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t v) {
            const uint32_t s = off[v], e = off[v + 1];
            for (uint32_t i = s + 1; i < e; ++i) {
                const int32_t x = adj[i];
                uint32_t j = i;
                while (j > s && adj[j - 1] > x) { adj[j] = adj[j - 1]; --j; }
                adj[j] = x;
            }
        });
    };

    build_adjacency();
    MeshView mv{verts, faces, b.off, b.adj, b.quad, b.bnd, 0};
    {
        double* quad = b.quad;
        uint8_t* bnd = b.bnd;
        be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t v) { vertex_quadric(mv, (int)v, quad + 10 * v, bnd + v); });
    }

    while (nf > max_faces && res.rounds < max_rounds) {
        // 1. cheapest valid edge per vertex: vertex v PROPOSES the collapse (v, partner[v]).  The edge was checked from both
        //    ends (link condition, both rings' orientation), so every proposal is a valid collapse of the mesh as it is now.
        int32_t* partner = b.partner;
        uint64_t* key = b.key;
        int32_t* cand_hi = b.mark_lo;        // [lo] of a SELECTED collapse: its higher endpoint
        uint64_t* cand_key = b.mark_key;     // [lo] its key (read by the budget threshold and the apply step)
        uint32_t* sel = b.sel;
        uint32_t* state = b.used;            // per proposer: 0 no proposal / out, 1 undecided
        uint32_t* taken = b.keep;            // per vertex: endpoint of a selected collapse
        uint32_t* verdict = b.pos;           // per proposer, one iteration: 1 selected now
        uint64_t* inkey = b.inkey;
        int32_t* inwho = b.inwho;
        {
            int32_t* remap = b.remap;
            be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t v) {
                best_partner(mv, (int)v, partner + v, key + v);
                sel[v] = 0;
                taken[v] = 0;
                remap[v] = (int32_t)v;
            });
            // an edge both of whose endpoints propose it is ONE proposal: the lower endpoint's
            be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t v) {
                const int32_t u = partner[v];
                state[v] = (u >= 0 && !(u < v && partner[u] == (int32_t)v)) ? 1u : 0u;
            });
        }
        // 2. a maximal set of proposals with pairwise disjoint closed neighbourhoods, smallest (key, proposer) first: a
        //    proposal is selected when no undecided proposal that touches the one-rings of its endpoints is smaller, and it is
        //    out as soon as a selected one does (Luby's rounds on a fixed total order: the result is a pure function of the
        //    mesh; three iterations reach the maximal set to within a round on the test meshes).  Proposals that END at a vertex w are seen through two atomic minima at w (key, then proposer id): taking
        //    the minimum is order-independent.  Round 3 selected only MUTUAL choices that were local minima in one pass:
        //    ~6 % of the faces per round, 52 rounds and 163 of 194 ms in step 1 on the bench's object; see
        //    profiles/r04_cleaners_kernel_time.md.
        for (int it = 0; it < kSelectIterations; ++it) {
            if (it > 0)   // proposals next to a collapse selected in an earlier iteration are out
                be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t c) {
                    if (state[c] == 1u && proposal_touches_taken(mv, (int)c, partner[c], taken)) state[c] = 0u;
                });
            be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t v) { inkey[v] = kNoKey; inwho[v] = kNoProposer; });
            be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t c) {
                if (state[c] == 1u) AT::min64(&inkey[partner[c]], key[c]);
            });
            be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t c) {
                if (state[c] == 1u && inkey[partner[c]] == key[c]) AT::min32(&inwho[partner[c]], (int32_t)c);
            });
            be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t c) {
                verdict[c] = (state[c] == 1u && proposal_is_smallest(mv, (int)c, partner, key, state, inkey, inwho)) ? 1u : 0u;
            });
            be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t c) {
                if (state[c] != 1u || !verdict[c]) return;
                const int32_t p = partner[c];
                const int32_t lo = (int32_t)c < p ? (int32_t)c : p, hi = (int32_t)c < p ? p : (int32_t)c;
                sel[lo] = (uint32_t)shared_faces(mv, lo, hi);
                cand_hi[lo] = hi;
                cand_key[lo] = key[c];
                taken[c] = 1u;
                taken[p] = 1u;
                state[c] = 0u;
            });
        }
        // 3. how many faces would go?  Keep the cheapest collapses only when the budget is nearly reached.
        const uint64_t removable = be.sum_if(b.sel, b.mark_key, nv, kNoKey);
        if (removable == 0) {
            // nothing can collapse under the current shape rules: relax them step by step (topology rules never)
            if (mv.relax < 2) { ++mv.relax; continue; }
            res.stalled_at = nf;
            break;
        }
        const uint64_t need = (uint64_t)(nf - max_faces);
        uint64_t thr = kNoKey;
        if (removable > need) {   // smallest key threshold whose collapses remove >= need faces
            uint64_t lo = 0, hi = kNoKey - 1;
            while (lo < hi) {
                const uint64_t mid = lo + (hi - lo) / 2;
                if (be.sum_if(b.sel, b.mark_key, nv, mid) >= need) hi = mid; else lo = mid + 1;
            }
            thr = lo;
        }
        // 4. apply: the lower endpoint keeps the merged vertex
        {
            int32_t* remap = b.remap;
            double* quad = b.quad;
            be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t v) {
                if (!sel[v] || cand_key[v] > thr) return;
                const int u = cand_hi[v];
                const Collapse c = edge_collapse(mv, (int)v, u, placement_mode(mv, (int)v, u));
                verts[3 * v] = (float)c.pos.x; verts[3 * v + 1] = (float)c.pos.y; verts[3 * v + 2] = (float)c.pos.z;
                for (int k = 0; k < 10; ++k) quad[10 * v + k] += quad[10 * (int64_t)u + k];
                remap[u] = (int32_t)v;
            });
            uint32_t* keep = b.keep;
            be.parfor(nf, [=] R3G_QEM_LAMBDA(int64_t i) {
                const int32_t a = remap[faces[3 * i]], bb = remap[faces[3 * i + 1]], c = remap[faces[3 * i + 2]];
                faces[3 * i] = a; faces[3 * i + 1] = bb; faces[3 * i + 2] = c;
                keep[i] = (a != bb && bb != c && a != c) ? 1u : 0u;
            });
        }
        nf = compact_faces(nf);
        ++res.rounds;
        if (nf <= max_faces) break;
        // next round: adjacency of the new face list, boundary flags from it
        build_adjacency();
        {
            uint8_t* bnd = b.bnd;
            be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t v) {
                uint8_t is = 0;
                for (uint32_t i = mv.off[v]; i < mv.off[v + 1] && !is; ++i) {
                    const int32_t* f = mv.faces + 3 * mv.adj[i];
                    for (int k = 0; k < 3; ++k)
                        if (f[k] != (int32_t)v && shared_faces(mv, (int)v, f[k]) == 1) is = 1;
                }
                bnd[v] = is;
            });
        }
    }

    // drop the vertices nothing references any more (order preserved) and re-index the faces
    {
        be.zero(b.used, 4 * (size_t)nv);
        uint32_t* used = b.used;
        const int32_t* f = faces;
        be.parfor(3 * nf, [=] R3G_QEM_LAMBDA(int64_t i) { used[f[i]] = 1u; });
        const uint32_t nv_out = be.scan(b.used, nv, b.pos);
        const uint32_t* pos = b.pos;
        float* vt = b.verts_tmp;
        be.parfor(nv, [=] R3G_QEM_LAMBDA(int64_t v) {
            if (used[v]) {
                vt[3 * (int64_t)pos[v]] = verts[3 * v];
                vt[3 * (int64_t)pos[v] + 1] = verts[3 * v + 1];
                vt[3 * (int64_t)pos[v] + 2] = verts[3 * v + 2];
            }
        });
        int32_t* fw = faces;
        be.parfor(3 * nf, [=] R3G_QEM_LAMBDA(int64_t i) { fw[i] = (int32_t)pos[fw[i]]; });
        be.copy(verts, b.verts_tmp, 12 * (size_t)nv_out);
        res.nv = nv_out;
    }
    res.nf = nf;
    return res;
}

}  // namespace r3g_qem
#endif
