// qem_core.h -- per-element bodies of the quadric-error-metric edge-collapse decimator (FaceReducer).
//
// Replaces hy3dgen.shapegen.postprocessors.FaceReducer as applied by the reference stage to every mesh
// (src/2d_to_3d_models/run.py:93-94; upstream: MeshLab meshing_decimation_quadric_edge_collapse with
// targetfacenum = 40000, preserveboundary / preservenormal / preservetopology on, optimalplacement on) and trimesh's
// simplify_quadric_decimation in the optional remesh step (run.py:47-49).  Bit parity with MeshLab's sequential
// priority queue is not a goal: the contract is geometric -- target face count, boundaries and
// topology preserved, no flipped normals, small distance to the input surface (tests/test_qem_*.py).
//
// Parallel formulation: ROUNDS of independent collapses.  In a round
//   1. every vertex picks its cheapest VALID incident edge (quadric error of the optimal position; valid = link
//      condition, no normal flip of any surrounding face, boundary rules);
//   2. every vertex PROPOSES its pick; a maximal set of proposals whose endpoints' closed one-rings are pairwise disjoint
//      is selected, smallest (cost, hash, id) first (a few rounds of Luby's algorithm on that fixed order), so selected
//      collapses can be applied simultaneously with the validity they were checked for;
//   3. survivors are applied: the lower vertex id keeps the merged vertex (position = the optimum, quadric = sum), faces
//      are re-indexed, collapsed faces dropped, the face list compacted in order.
// Everything is a pure function of the input mesh: the vertex-face adjacency is sorted, quadrics are gathered in
// adjacency order (no floating-point atomics), ties are broken by ids.  This header holds the per-vertex / per-face
// bodies; qem.hip wraps them in kernels, tests/emu/qem_emu.cpp runs the same bodies in host loops (test only).
#ifndef R3G_QEM_CORE_H
#define R3G_QEM_CORE_H
#include <stdint.h>

#ifndef R3G_QEM_HD
#define R3G_QEM_HD static inline
#endif

namespace r3g_qem {

struct MeshView {
    const float* verts;      // [nv][3] current positions
    const int32_t* faces;    // [nf][3]
    const uint32_t* off;     // [nv+1] adjacency offsets
    const int32_t* adj;      // face ids, sorted ascending per vertex
    const double* quad;      // [nv][10]: a2 ab ac ad b2 bc bd c2 cd d2
    const uint8_t* bnd;      // [nv] 1: the vertex lies on a boundary edge
    int relax;               // 0: all shape rules; 1: slivers allowed; 2: faces may turn up to 90 degrees (stalled runs only)
};

constexpr double kFlipCos = 0.2;          // a face may not turn by more than acos(0.2) ~ 78 degrees
constexpr double kMinShape = 1e-4;        // (2 area)^2 / (sum of squared edges)^2 of a new face: 1/12 when equilateral
constexpr double kBoundaryWeight = 1e3;   // weight of the boundary-preserving constraint planes (x edge length^2)
constexpr uint64_t kNoKey = ~0ull;
constexpr int kPinFree = -1, kPinEndpoint = -2;   // placement modes of edge_collapse (>= 0: pinned to that vertex)

struct D3 { double x, y, z; };
R3G_QEM_HD D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
R3G_QEM_HD D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
R3G_QEM_HD double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
R3G_QEM_HD D3 vert(const float* v, int i) { return {(double)v[3 * i], (double)v[3 * i + 1], (double)v[3 * i + 2]}; }
R3G_QEM_HD double qsqrt(double x) { return __builtin_sqrt(x); }

// quadric of the plane through p with unit normal n, weight w, added to q[10]
R3G_QEM_HD void add_plane(double* q, D3 n, D3 p, double w) {
    const double d = -dot(n, p);
    q[0] += w * n.x * n.x; q[1] += w * n.x * n.y; q[2] += w * n.x * n.z; q[3] += w * n.x * d;
    q[4] += w * n.y * n.y; q[5] += w * n.y * n.z; q[6] += w * n.y * d;
    q[7] += w * n.z * n.z; q[8] += w * n.z * d;
    q[9] += w * d * d;
}

R3G_QEM_HD bool face_has(const int32_t* f, int v) { return f[0] == v || f[1] == v || f[2] == v; }

// number of faces incident to a that also contain b (adjacency of a)
R3G_QEM_HD int shared_faces(const MeshView& m, int a, int b) {
    int n = 0;
    for (uint32_t i = m.off[a]; i < m.off[a + 1]; ++i) n += face_has(m.faces + 3 * m.adj[i], b) ? 1 : 0;
    return n;
}

// ---- initial quadric of vertex v: area-weighted face planes + constraint planes along boundary edges
R3G_QEM_HD void vertex_quadric(const MeshView& m, int v, double* q, uint8_t* is_bnd) {
    for (int k = 0; k < 10; ++k) q[k] = 0.0;
    uint8_t b = 0;
    for (uint32_t i = m.off[v]; i < m.off[v + 1]; ++i) {
        const int32_t* f = m.faces + 3 * m.adj[i];
        const D3 p0 = vert(m.verts, f[0]), p1 = vert(m.verts, f[1]), p2 = vert(m.verts, f[2]);
        const D3 n = cross(sub(p1, p0), sub(p2, p0));
        const double len = qsqrt(dot(n, n));
        if (!(len > 0.0)) continue;
        const D3 un = {n.x / len, n.y / len, n.z / len};
        add_plane(q, un, p0, 0.5 * len);
        // the two edges of this face at v: boundary when no other face shares them
        for (int e = 0; e < 3; ++e) {
            const int a = f[e], c = f[(e + 1) % 3];
            if (a != v && c != v) continue;
            const int other = a == v ? c : a;
            if (shared_faces(m, v, other) != 1) continue;
            b = 1;
            const D3 pa = vert(m.verts, a), pc = vert(m.verts, c);
            const D3 ed = sub(pc, pa);
            D3 bn = cross(ed, un);                       // in the face plane, perpendicular to the edge
            const double bl = qsqrt(dot(bn, bn));
            if (!(bl > 0.0)) continue;
            bn = {bn.x / bl, bn.y / bl, bn.z / bl};
            add_plane(q, bn, pa, kBoundaryWeight * dot(ed, ed));
        }
    }
    *is_bnd = b;
}

// ---- cost and position of collapsing the edge (lo, hi), lo < hi: a symmetric function of the unordered pair
struct Collapse { double cost; D3 pos; };

R3G_QEM_HD double quadric_error(const double* q, D3 p) {
    return q[0] * p.x * p.x + 2.0 * q[1] * p.x * p.y + 2.0 * q[2] * p.x * p.z + 2.0 * q[3] * p.x + q[4] * p.y * p.y +
           2.0 * q[5] * p.y * p.z + 2.0 * q[6] * p.y + q[7] * p.z * p.z + 2.0 * q[8] * p.z + q[9];
}

// pin >= 0: the merged vertex stays at vertex `pin` (an interior vertex collapsing INTO a boundary vertex);
// kPinEndpoint: at the cheaper of the two endpoints (two boundary vertices); kPinFree: at the quadric's optimum
R3G_QEM_HD Collapse edge_collapse(const MeshView& m, int lo, int hi, int pin) {
    double q[10];
    for (int k = 0; k < 10; ++k) q[k] = m.quad[10 * (int64_t)lo + k] + m.quad[10 * (int64_t)hi + k];
    const D3 pl = vert(m.verts, lo), ph = vert(m.verts, hi);
    if (pin >= 0 || pin == kPinEndpoint) {
        Collapse c;
        if (pin >= 0) {
            c.pos = pin == lo ? pl : ph;
            c.cost = quadric_error(q, c.pos);
        } else {       // a boundary edge: the merged vertex is one of the two endpoints (it stays ON the outline)
            const double cl = quadric_error(q, pl), ch = quadric_error(q, ph);
            c.pos = ch < cl ? ph : pl;
            c.cost = ch < cl ? ch : cl;
        }
        if (!(c.cost > 0.0)) c.cost = 0.0;
        return c;
    }
    const D3 mid = {0.5 * (pl.x + ph.x), 0.5 * (pl.y + ph.y), 0.5 * (pl.z + ph.z)};
    const D3 e = sub(ph, pl);
    const double elen2 = dot(e, e);
    Collapse best;
    bool have = false;
    // optimal placement: A x = -b
    const double a = q[0], b = q[1], c = q[2], d = q[4], ee = q[5], f = q[7];
    const double det = a * (d * f - ee * ee) - b * (b * f - ee * c) + c * (b * ee - d * c);
    const double tr = a + d + f;
    if (det > 1e-9 * tr * tr * tr || det < -1e-9 * tr * tr * tr) {
        const double bx = -q[3], by = -q[6], bz = -q[8];
        const D3 x = {(bx * (d * f - ee * ee) - b * (by * f - ee * bz) + c * (by * ee - d * bz)) / det,
                      (a * (by * f - bz * ee) - bx * (b * f - ee * c) + c * (b * bz - by * c)) / det,
                      (a * (d * bz - ee * by) - b * (b * bz - by * c) + bx * (b * ee - d * c)) / det};
        const D3 off = sub(x, mid);
        if (dot(off, off) <= 4.0 * elen2 && x.x == x.x && x.y == x.y && x.z == x.z) {
            best.pos = x;
            best.cost = quadric_error(q, x);
            have = true;
        }
    }
    if (!have) {   // singular system (flat or straight neighbourhood): the best of the endpoints and the midpoint
        const D3 cand[3] = {mid, pl, ph};
        for (int k = 0; k < 3; ++k) {
            const double cst = quadric_error(q, cand[k]);
            if (!have || cst < best.cost) { best.cost = cst; best.pos = cand[k]; have = true; }
        }
    }
    if (!(best.cost > 0.0)) best.cost = 0.0;
    // the stored position is float32: evaluate validity with exactly what will be stored
    best.pos = {(double)(float)best.pos.x, (double)(float)best.pos.y, (double)(float)best.pos.z};
    return best;
}

// every face around `v` that does not contain `other` keeps its orientation when v moves to p
R3G_QEM_HD bool ring_keeps_orientation(const MeshView& m, int v, int other, D3 p) {
    for (uint32_t i = m.off[v]; i < m.off[v + 1]; ++i) {
        const int32_t* f = m.faces + 3 * m.adj[i];
        if (face_has(f, other)) continue;
        D3 q0 = vert(m.verts, f[0]), q1 = vert(m.verts, f[1]), q2 = vert(m.verts, f[2]);
        const D3 nb = cross(sub(q1, q0), sub(q2, q0));
        if (f[0] == v) q0 = p; else if (f[1] == v) q1 = p; else q2 = p;
        const D3 na = cross(sub(q1, q0), sub(q2, q0));
        const double lb = dot(nb, nb), la = dot(na, na);
        if (!(lb > 0.0)) continue;                         // already a zero-area face (marching cubes keeps them)
        if (!(la > 0.0)) return false;                     // the face would degenerate
        const double dn = dot(nb, na);
        if (!(dn > 0.0) || (m.relax < 2 && dn * dn < kFlipCos * kFlipCos * lb * la)) return false;
        if (m.relax >= 1) continue;
        // no slivers: (2 area)^2 against the squared edge lengths (an equilateral triangle gives 1/3 of their sum squared)
        const D3 e0 = sub(q1, q0), e1 = sub(q2, q1), e2 = sub(q0, q2);
        const double l2 = dot(e0, e0) + dot(e1, e1) + dot(e2, e2);
        if (la < kMinShape * l2 * l2) return false;
    }
    return true;
}

// link condition: the vertices adjacent to both endpoints are exactly the apexes of the faces on the edge, and the
// edge carries one (boundary) or two faces; two boundary vertices may only merge along a boundary edge
R3G_QEM_HD bool link_condition(const MeshView& m, int v, int u) {
    const int ns = shared_faces(m, v, u);
    if (ns != 1 && ns != 2) return false;
    if (m.bnd[v] && m.bnd[u] && ns != 1) return false;
    int common = 0;
    for (uint32_t i = m.off[v]; i < m.off[v + 1]; ++i) {
        const int32_t* f = m.faces + 3 * m.adj[i];
        for (int k = 0; k < 3; ++k) {
            const int w = f[k];
            if (w == v || w == u) continue;
            bool seen = false;                              // count a neighbour once: at its first face around v
            for (uint32_t j = m.off[v]; j < i && !seen; ++j) seen = face_has(m.faces + 3 * m.adj[j], w);
            if (!seen && shared_faces(m, u, w) > 0) ++common;
        }
    }
    return common == ns;
}

R3G_QEM_HD uint32_t mix32(uint32_t a, uint32_t b) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u + (a << 6) + (a >> 2));
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

R3G_QEM_HD uint32_t float_bits(float x) {
    union { float f; uint32_t u; } c;
    c.f = x;
    return c.u;
}

// priority key of an edge: smaller = collapse first.  cost >= 0, so the float bits order like the values.
R3G_QEM_HD uint64_t edge_key(double cost, int lo, int hi) {
    return ((uint64_t)float_bits((float)cost) << 32) | mix32((uint32_t)lo, (uint32_t)hi);
}

// placement of the merged vertex of edge (v, u): symmetric in its arguments
R3G_QEM_HD int placement_mode(const MeshView& m, int v, int u) {
    if (m.bnd[v] != m.bnd[u]) return m.bnd[v] ? v : u;
    return m.bnd[v] ? kPinEndpoint : kPinFree;
}

// ---- step 1: the cheapest valid edge at vertex v -> partner (or -1) and key
// = the neighbour with the smallest (key, id) among those whose collapse passes the link condition and keeps both rings'
// orientation.  The keys of all neighbours are cheap (the neighbour's quadric and a 3x3 system); the validity checks are
// what costs (the link condition walks the rings of the ring).  So: keys first, then the checks in ascending key order
// until one passes -- on a smooth mesh the first.  (Round 3 ran the checks whenever a neighbour improved on the best so
// far, in adjacency order: H(n) ~ 2.5 times per vertex at valence 6.  Same result, by definition.)
R3G_QEM_HD bool first_face_of_neighbour(const MeshView& m, int v, uint32_t i, int u) {
    for (uint32_t j = m.off[v]; j < i; ++j) {
        const int32_t* g = m.faces + 3 * m.adj[j];
        if (g[0] == u || g[1] == u || g[2] == u) return false;
    }
    return true;
}

R3G_QEM_HD bool collapse_is_valid(const MeshView& m, int v, int u, D3 pos) {
    return link_condition(m, v, u) && ring_keeps_orientation(m, v, u, pos) && ring_keeps_orientation(m, u, v, pos);
}

R3G_QEM_HD void best_partner(const MeshView& m, int v, int32_t* partner, uint64_t* key) {
    // an interior vertex may merge INTO a boundary vertex (which stays put); two boundary vertices merge freely along a
    // boundary edge (the constraint planes in their quadrics keep the outline), never across the interior: placement_mode
    uint64_t floor_key = 0;      // candidates at or below (floor_key, floor_u) have been tried and were invalid
    int32_t floor_u = -1;
    for (;;) {
        // the smallest (key, id) above the floor; nothing is stored per neighbour (a scan is a few gathers and a 3x3 system
        // each; a list of keys would cost the kernel a third of its waves in registers)
        int32_t bu = -1;
        uint64_t bk = kNoKey;
        for (uint32_t i = m.off[v]; i < m.off[v + 1]; ++i) {
            const int32_t* f = m.faces + 3 * m.adj[i];
            for (int k = 0; k < 3; ++k) {
                const int u = f[k];
                // a neighbour shows up once per face around the edge (v, u): take it at the first of them
                if (u == v || !first_face_of_neighbour(m, v, i, u)) continue;
                const int lo = v < u ? v : u, hi = v < u ? u : v;
                const Collapse c = edge_collapse(m, lo, hi, placement_mode(m, v, u));
                const uint64_t ky = edge_key(c.cost, lo, hi);
                if (floor_u >= 0 && (ky < floor_key || (ky == floor_key && u <= floor_u))) continue;   // tried already
                if (bu >= 0 && (ky > bk || (ky == bk && u >= bu))) continue;
                bk = ky;
                bu = u;
            }
        }
        if (bu < 0) break;                                   // every neighbour tried
        // (the position again: keeping it across the scan costs the kernel a wave per SIMD in registers)
        const D3 bpos = edge_collapse(m, v < bu ? v : bu, v < bu ? bu : v, placement_mode(m, v, bu)).pos;
        if (collapse_is_valid(m, v, bu, bpos)) {
            *partner = bu;
            *key = bk;
            return;
        }
        floor_key = bk;
        floor_u = bu;
    }
    *partner = -1;
    *key = kNoKey;
}

// ---- step 2: the selection among the proposals (vertex c proposes the edge (c, partner[c])).
// Order of proposals: (key, proposer id), total (a vertex proposes at most one edge).  Two proposals CONFLICT when an
// endpoint of one lies in the closed one-ring of an endpoint of the other (adjacency is symmetric, so they see each other).
// An iteration: undecided proposals that touch a `taken` vertex (an endpoint of a selected collapse) are out; of the rest,
// those smaller than every conflicting undecided proposal are selected.  Selected collapses therefore have pairwise
// disjoint closed neighbourhoods.  Proposals that END at a vertex w are seen through inkey / inwho[w], the smallest of them.
constexpr int32_t kNoProposer = 0x7FFFFFFF;
constexpr int kSelectIterations = 3;

R3G_QEM_HD bool proposal_less(uint64_t ka, int32_t a, uint64_t kb, int32_t b) { return ka < kb || (ka == kb && a < b); }

R3G_QEM_HD bool proposal_touches_taken(const MeshView& m, int c, int p, const uint32_t* taken) {
    const int ends[2] = {c, p};
    for (int s = 0; s < 2; ++s)
        for (uint32_t i = m.off[ends[s]]; i < m.off[ends[s] + 1]; ++i) {
            const int32_t* f = m.faces + 3 * m.adj[i];
            if (taken[f[0]] | taken[f[1]] | taken[f[2]]) return true;
        }
    return false;
}

R3G_QEM_HD bool proposal_is_smallest(const MeshView& m, int c, const int32_t* partner, const uint64_t* key,
                                     const uint32_t* state, const uint64_t* inkey, const int32_t* inwho) {
    const int ends[2] = {c, partner[c]};
    const uint64_t kc = key[c];
    for (int s = 0; s < 2; ++s) {
        const int a = ends[s];
        for (uint32_t i = m.off[a]; i < m.off[a + 1]; ++i) {
            const int32_t* f = m.faces + 3 * m.adj[i];
            for (int k = 0; k < 3; ++k) {
                const int w = f[k];
                if (w != c && state[w] == 1u && proposal_less(key[w], w, kc, c)) return false;
                const int32_t x = inwho[w];
                if (x != kNoProposer && x != c && proposal_less(inkey[w], x, kc, c)) return false;
            }
        }
    }
    return true;
}

}  // namespace r3g_qem
#endif
