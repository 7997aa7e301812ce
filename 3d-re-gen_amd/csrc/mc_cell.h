// mc_cell.h -- per-cell logic of the Lewiner marching-cubes extractor (product code).
//
// Replaces, on the GPU, the per-cell body of skimage's Lewiner kernel that the reference path
// reaches through hy3dgen MCSurfaceExtractor.run -> skimage.measure.marching_cubes(grid, level,
// method="lewiner") (reference call site src/2d_to_3d_models/run.py:77-84; wrapper semantics
// skimage/measure/_marching_cubes_lewiner.py:280-349).
//
// Parallel formulation (what makes the sequential vertex numbering reproducible in parallel):
//  * a vertex lives on a unique grid edge (node, axis) or at a cell centre ("edge 12");
//  * the sequential kernel creates it in the FIRST cell, in scan order (axis0 outer, axis2
//    inner), whose tiling references that edge; every cell sharing a sign-change edge does
//    reference it, so that cell is the "owner" (x, max(y-1,0), max(z-1,0)) for an x-edge etc.;
//  * vertex id = exclusive scan, in scan order, of per-cell counts of owned edges + the rank of
//    the edge among the owner's owned edges in its triangle-emission order.
// All ambiguity tests and the interpolation are evaluated in double exactly as the sequential
// kernel does (compile with -ffp-contract=off), so faces AND vertices are bit-identical.
//
// This header is shared by mc_kernels.hip (device) and by tests/emu (host emulation of the
// launch structure, test-only).  It never includes anything from oracle/.
#ifndef R3G_MC_CELL_H
#define R3G_MC_CELL_H

#include <stdint.h>

#ifndef R3G_DEV
#define R3G_DEV static inline
#endif
#ifndef R3G_HOSTDEV
#define R3G_HOSTDEV static inline
#endif
#include "mc_luts.h"

#if defined(__clang__)
#define R3G_UNROLL _Pragma("unroll")
#else
#define R3G_UNROLL
#endif

#define R3G_MC_EPS 2.220446049250313e-16

namespace r3g_mc {

struct Tiling {
    int off;  // offset of the first edge index in R3G_MC_TRI
    int nt;   // triangle count
};

#define R3G_T2(NAME, cfg) (R3G_MC_OFF_##NAME + (cfg) * R3G_MC_ROW_##NAME)
#define R3G_T3(NAME, cfg, sub) (R3G_MC_OFF_##NAME + ((cfg) * R3G_MC_MID_##NAME + (sub)) * R3G_MC_ROW_##NAME)

R3G_DEV double dabs(double a) { return a < 0 ? -a : a; }

// Does the ambiguous face contain part of the surface (Lewiner "test_face").
R3G_DEV bool test_face(const double* v, int face) {
    const int af = face < 0 ? -face : face;
    double A, B, C, D;
    switch (af) {
        case 1: A = v[0]; B = v[4]; C = v[5]; D = v[1]; break;
        case 2: A = v[1]; B = v[5]; C = v[6]; D = v[2]; break;
        case 3: A = v[2]; B = v[6]; C = v[7]; D = v[3]; break;
        case 4: A = v[3]; B = v[7]; C = v[4]; D = v[0]; break;
        case 5: A = v[0]; B = v[3]; C = v[2]; D = v[1]; break;
        case 6: A = v[4]; B = v[7]; C = v[6]; D = v[5]; break;
        default: A = B = C = D = 0.0; break;
    }
    const double acbd = A * C - B * D;
    if (acbd > -R3G_MC_EPS && acbd < R3G_MC_EPS) return face >= 0;
    return (double)face * A * acbd >= 0.0;
}

// Lewiner "test_interior" with the compiled skimage kernel's fall-through behaviour
// (cases 5/10 return false when their sub-condition fails).
R3G_DEV bool test_interior(const double* v, int mc_case, int config, int subconfig, int s) {
    double t, At = 0.0, Bt = 0.0, Ct = 0.0, Dt = 0.0;
    if (mc_case == 4 || mc_case == 10) {
        const double a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        const double b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        t = -b / (2 * a + R3G_MC_EPS);
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else {
        int edge = -1;
        if (mc_case == 6) edge = R3G_MC_TEST6[config][2];
        else if (mc_case == 7) edge = R3G_MC_TEST7[config][4];
        else if (mc_case == 12) edge = R3G_MC_TEST12[config][3];
        else if (mc_case == 13) edge = R3G_MC_TRI[R3G_T3(TILING13_5_1, config, subconfig)];
        if (edge >= 0 && edge < 12) {
            // reference edge (p,q); B, C, D walk the three edges parallel to it.  Packed as 8 nibbles.
            // rows: p q B0 B1 C0 C1 D0 D1
            const unsigned long long ROWS[12] = {
                0x01327645ull, 0x12034756ull, 0x23105467ull, 0x30216574ull, 0x45763201ull, 0x56470312ull,
                0x67541023ull, 0x74652130ull, 0x04372615ull, 0x15043726ull, 0x26150437ull, 0x37261504ull};
            const unsigned long long r = ROWS[edge];
#define R3G_NIB(i) ((int)((r >> (4 * (7 - (i)))) & 0xF))
            const double vp = v[R3G_NIB(0)], vq = v[R3G_NIB(1)];
            t = vp / (vp - vq + R3G_MC_EPS);
            At = 0;
            Bt = v[R3G_NIB(2)] + (v[R3G_NIB(3)] - v[R3G_NIB(2)]) * t;
            Ct = v[R3G_NIB(4)] + (v[R3G_NIB(5)] - v[R3G_NIB(4)]) * t;
            Dt = v[R3G_NIB(6)] + (v[R3G_NIB(7)] - v[R3G_NIB(6)]) * t;
#undef R3G_NIB
        }
    }
    int test = 0;
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    switch (test) {
        case 5: return (At * Ct - Bt * Dt < R3G_MC_EPS) ? (s > 0) : false;
        case 10: return (At * Ct - Bt * Dt >= R3G_MC_EPS) ? (s > 0) : false;
        case 7: case 11: case 13: case 14: case 15: return s < 0;
        default: return s > 0;  // 0,1,2,3,4,6,8,9,12
    }
}

// Choose the tiling of one active cell.  v[] = corner values minus level in Lewiner numbering,
// index = sign bit-field (bit k set iff v[k] > 0).
R3G_DEV Tiling select_tiling(const double* v, int index, bool classic) {
    Tiling r;
    r.off = 0;
    r.nt = 0;
    if (classic) {
        r.off = R3G_MC_OFF_CASESCLASSIC + index * 16;
        while (r.nt < 5 && R3G_MC_TRI[r.off + 3 * r.nt] != -1) ++r.nt;
        return r;
    }
    const int mc_case = R3G_MC_CASES[index][0];
    const int c = R3G_MC_CASES[index][1];
    int sub = 0;
    switch (mc_case) {
        case 1: r.off = R3G_T2(TILING1, c); r.nt = 1; break;
        case 2: r.off = R3G_T2(TILING2, c); r.nt = 2; break;
        case 3:
            if (test_face(v, R3G_MC_TEST3[c])) { r.off = R3G_T2(TILING3_2, c); r.nt = 4; }
            else { r.off = R3G_T2(TILING3_1, c); r.nt = 2; }
            break;
        case 4:
            if (test_interior(v, 4, c, 0, R3G_MC_TEST4[c])) { r.off = R3G_T2(TILING4_1, c); r.nt = 2; }
            else { r.off = R3G_T2(TILING4_2, c); r.nt = 6; }
            break;
        case 5: r.off = R3G_T2(TILING5, c); r.nt = 3; break;
        case 6:
            if (test_face(v, R3G_MC_TEST6[c][0])) { r.off = R3G_T2(TILING6_2, c); r.nt = 5; }
            else if (test_interior(v, 6, c, 0, R3G_MC_TEST6[c][1])) { r.off = R3G_T2(TILING6_1_1, c); r.nt = 3; }
            else { r.off = R3G_T2(TILING6_1_2, c); r.nt = 9; }
            break;
        case 7:
            if (test_face(v, R3G_MC_TEST7[c][0])) sub += 1;
            if (test_face(v, R3G_MC_TEST7[c][1])) sub += 2;
            if (test_face(v, R3G_MC_TEST7[c][2])) sub += 4;
            switch (sub) {
                case 0: r.off = R3G_T2(TILING7_1, c); r.nt = 3; break;
                case 1: r.off = R3G_T3(TILING7_2, c, 0); r.nt = 5; break;
                case 2: r.off = R3G_T3(TILING7_2, c, 1); r.nt = 5; break;
                case 3: r.off = R3G_T3(TILING7_3, c, 0); r.nt = 9; break;
                case 4: r.off = R3G_T3(TILING7_2, c, 2); r.nt = 5; break;
                case 5: r.off = R3G_T3(TILING7_3, c, 1); r.nt = 9; break;
                case 6: r.off = R3G_T3(TILING7_3, c, 2); r.nt = 9; break;
                default:
                    if (test_interior(v, 7, c, 0, R3G_MC_TEST7[c][3])) { r.off = R3G_T2(TILING7_4_2, c); r.nt = 9; }
                    else { r.off = R3G_T2(TILING7_4_1, c); r.nt = 5; }
                    break;
            }
            break;
        case 8: r.off = R3G_T2(TILING8, c); r.nt = 2; break;
        case 9: r.off = R3G_T2(TILING9, c); r.nt = 4; break;
        case 10:
            if (test_face(v, R3G_MC_TEST10[c][0])) {
                if (test_face(v, R3G_MC_TEST10[c][1])) { r.off = R3G_T2(TILING10_1_1_, c); r.nt = 4; }
                else { r.off = R3G_T2(TILING10_2, c); r.nt = 8; }
            } else if (test_face(v, R3G_MC_TEST10[c][1])) { r.off = R3G_T2(TILING10_2_, c); r.nt = 8; }
            else if (test_interior(v, 10, c, 0, R3G_MC_TEST10[c][2])) { r.off = R3G_T2(TILING10_1_1, c); r.nt = 4; }
            else { r.off = R3G_T2(TILING10_1_2, c); r.nt = 8; }
            break;
        case 11: r.off = R3G_T2(TILING11, c); r.nt = 4; break;
        case 12:
            if (test_face(v, R3G_MC_TEST12[c][0])) {
                if (test_face(v, R3G_MC_TEST12[c][1])) { r.off = R3G_T2(TILING12_1_1_, c); r.nt = 4; }
                else { r.off = R3G_T2(TILING12_2, c); r.nt = 8; }
            } else if (test_face(v, R3G_MC_TEST12[c][1])) { r.off = R3G_T2(TILING12_2_, c); r.nt = 8; }
            else if (test_interior(v, 12, c, 0, R3G_MC_TEST12[c][2])) { r.off = R3G_T2(TILING12_1_1, c); r.nt = 4; }
            else { r.off = R3G_T2(TILING12_1_2, c); r.nt = 8; }
            break;
        case 13: {
            for (int k = 0; k < 6; ++k)
                if (test_face(v, R3G_MC_TEST13[c][k])) sub += 1 << k;
            sub = R3G_MC_SUBCONFIG13[sub];
            if (sub == 0) { r.off = R3G_T2(TILING13_1, c); r.nt = 4; }
            else if (sub <= 6) { r.off = R3G_T3(TILING13_2, c, sub - 1); r.nt = 6; }
            else if (sub <= 18) { r.off = R3G_T3(TILING13_3, c, sub - 7); r.nt = 10; }
            else if (sub <= 22) { r.off = R3G_T3(TILING13_4, c, sub - 19); r.nt = 12; }
            else if (sub <= 26) {
                const int k = sub - 23;
                if (test_interior(v, 13, c, k, R3G_MC_TEST13[c][6])) { r.off = R3G_T3(TILING13_5_1, c, k); r.nt = 6; }
                else { r.off = R3G_T3(TILING13_5_2, c, k); r.nt = 10; }
            } else if (sub <= 38) { r.off = R3G_T3(TILING13_3_, c, sub - 27); r.nt = 10; }
            else if (sub <= 44) { r.off = R3G_T3(TILING13_2_, c, sub - 39); r.nt = 6; }
            else if (sub == 45) { r.off = R3G_T2(TILING13_1_, c); r.nt = 4; }
            break;  // anything else: "impossible case 13", emits nothing
        }
        case 14: r.off = R3G_T2(TILING14, c); r.nt = 4; break;
        default: break;
    }
    return r;
}

// 13-bit mask of the local edges (bit 12 = centre vertex) whose vertex THIS cell creates.
R3G_DEV unsigned owned_mask(int x, int y, int z) {
    const bool x0 = x == 0, y0 = y == 0, z0 = z == 0;
    unsigned m = (1u << 6) | (1u << 5) | (1u << 10) | (1u << 12);  // far edges + centre: always
    if (y0 && z0) m |= 1u << 0;
    if (z0) m |= (1u << 2) | (1u << 1);
    if (y0) m |= (1u << 4) | (1u << 9);
    if (x0 && z0) m |= 1u << 3;
    if (x0) m |= (1u << 7) | (1u << 11);
    if (x0 && y0) m |= 1u << 8;
    return m;
}

// Number of vertices this cell creates = distinct owned edges its tiling references.
// 12 consecutive tiling entries (4 triangles) starting at off + base, packed 4 bits each.  The loads are
// unconditional (index clamped to the tiling's last entry) so that they are all in flight together: the
// per-cell loops below then run out of registers instead of waiting for one table byte per iteration.
R3G_DEV uint64_t load_tri12(int off, int base, int n) {
    uint64_t packed = 0;
    R3G_UNROLL
    for (int j = 0; j < 12; ++j) {
        const int i = base + j < n ? base + j : n - 1;
        packed |= (uint64_t)(R3G_MC_TRI[off + i] & 0xF) << (4 * j);
    }
    return packed;
}

R3G_DEV int count_new_vertices(const Tiling& t, unsigned owned) {
    unsigned seen = 0;
    const int n = 3 * t.nt;
    for (int base = 0; base < n; base += 12) {
        const uint64_t tri = load_tri12(t.off, base, n);
        R3G_UNROLL
        for (int j = 0; j < 12; ++j)
            if (base + j < n) seen |= 1u << (unsigned)((tri >> (4 * j)) & 0xF);
    }
    seen &= owned;
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(seen);
#else
    return __builtin_popcount(seen);
#endif
}

// Global edge slot of local edge e (0..11) of cell (x,y,z): 3*node + axis, node = (z*ny + y)*nx + x.
// axis 0: edge along x (array axis 2), 1: along y (axis 1), 2: along z (axis 0).
R3G_DEV int64_t edge_slot(int e, int x, int y, int z, int nx, int ny) {
    // per local edge: dx, dy, dz of the edge's base node and its axis, packed 2 bits each
    //            e:   0  1  2  3  4  5  6  7  8  9 10 11
    const int DX[12] = {0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0};
    const int DY[12] = {0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 1};
    const int DZ[12] = {0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0};
    const int AX[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
    const int64_t node = ((int64_t)(z + DZ[e]) * ny + (y + DY[e])) * nx + (x + DX[e]);
    return 3 * node + AX[e];
}

// Interpolated position (x, y, z order = array axes 2, 1, 0) of the vertex on local edge e,
// in double, exactly as the sequential kernel: inverse-|value| weights, then x + fx/ff.
R3G_DEV void edge_vertex_ab(double a, double b, int e, int x, int y, int z, double* out) {
    const int dx1 = R3G_MC_EDGE_DX[e][0], dx2 = R3G_MC_EDGE_DX[e][1];
    const int dy1 = R3G_MC_EDGE_DY[e][0], dy2 = R3G_MC_EDGE_DY[e][1];
    const int dz1 = R3G_MC_EDGE_DZ[e][0], dz2 = R3G_MC_EDGE_DZ[e][1];
    const double w1 = 1.0 / (R3G_MC_EPS + dabs(a));
    const double w2 = 1.0 / (R3G_MC_EPS + dabs(b));
    double fx = 0.0, fy = 0.0, fz = 0.0, ff = 0.0;
    fx += (double)dx1 * w1; fy += (double)dy1 * w1; fz += (double)dz1 * w1; ff += w1;
    fx += (double)dx2 * w2; fy += (double)dy2 * w2; fz += (double)dz2 * w2; ff += w2;
    out[0] = (double)x + 1.0 * fx / ff;
    out[1] = (double)y + 1.0 * fy / ff;
    out[2] = (double)z + 1.0 * fz / ff;
}

// Cell-centre vertex ("edge 12"): inverse-|value| weighted centre of mass of the 8 corners.
R3G_DEV void center_vertex(const double* v, int x, int y, int z, double* out) {
    const double DX[8] = {0, 1, 1, 0, 0, 1, 1, 0};
    const double DY[8] = {0, 0, 1, 1, 0, 0, 1, 1};
    const double DZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    double fx = 0.0, fy = 0.0, fz = 0.0, ff = 0.0;
    R3G_UNROLL
    for (int k = 0; k < 8; ++k) {
        const double w = 1.0 / (R3G_MC_EPS + dabs(v[k]));
        fx += DX[k] * w;
        fy += DY[k] * w;
        fz += DZ[k] * w;
        ff += w;
    }
    out[0] = (double)x + 1.0 * fx / ff;
    out[1] = (double)y + 1.0 * fy / ff;
    out[2] = (double)z + 1.0 * fz / ff;
}

// Output transform of one vertex.  pos = kernel (x,y,z); the wrapper returns (z,y,x) float32.
// With use_xf the upstream hy3dgen rescale is applied on the float32 value in double:
//   v / grid_size * bbox_size + bbox_min   (surface_extractors.MCSurfaceExtractor.run)
struct Xform {
    double grid_size[3];  // per OUTPUT column (axis0, axis1, axis2)
    double bbox_size[3];
    double bbox_min[3];
};

R3G_DEV void store_vertex(float* dst, const double* pos, const Xform& xf, bool use_xf) {
    const float o0 = (float)pos[2], o1 = (float)pos[1], o2 = (float)pos[0];
    if (use_xf) {
        dst[0] = (float)((double)o0 / xf.grid_size[0] * xf.bbox_size[0] + xf.bbox_min[0]);
        dst[1] = (float)((double)o1 / xf.grid_size[1] * xf.bbox_size[1] + xf.bbox_min[1]);
        dst[2] = (float)((double)o2 / xf.grid_size[2] * xf.bbox_size[2] + xf.bbox_min[2]);
    } else {
        dst[0] = o0;
        dst[1] = o1;
        dst[2] = o2;
    }
}

// ---------------------------------------------------------------------------------------------
// Per-cell bodies of the three passes (shared by the HIP kernels and the host emulation).
//
// Pass 1 "classify": one thread per cell -> packed record, or 0 when the cell emits nothing.
//   record: bits 0..15 tiling offset, 16..19 triangle count, 20..23 new-vertex count.
// Pass 3 "vertices": the owner computes its vertices and publishes their ids in the edge table.
// Pass 4 "faces": every triangle corner resolves to an id (own rank or edge-table lookup).
// ---------------------------------------------------------------------------------------------
#define R3G_MC_FLAG_LE 1u   // some sample <= level
#define R3G_MC_FLAG_GE 2u   // some sample >= level
#define R3G_MC_FLAG_NAN 4u  // some sample is NaN

R3G_DEV unsigned load_corners(const float* g, int nx, int ny, int x, int y, int z, double level,
                              double* v, int* index) {
    const int64_t sy = nx, sz = (int64_t)nx * ny;
    const float* p = g + (int64_t)z * sz + (int64_t)y * sy + x;
    float f[8];
    f[0] = p[0]; f[1] = p[1]; f[2] = p[sy + 1]; f[3] = p[sy];
    f[4] = p[sz]; f[5] = p[sz + 1]; f[6] = p[sz + sy + 1]; f[7] = p[sz + sy];
    unsigned flags = 0;
    int idx = 0;
    R3G_UNROLL
    for (int k = 0; k < 8; ++k) {
        const double d = (double)f[k];
        if (d <= level) flags |= R3G_MC_FLAG_LE;
        if (d >= level) flags |= R3G_MC_FLAG_GE;
        if (f[k] != f[k]) flags |= R3G_MC_FLAG_NAN;
        v[k] = d - level;
        if (v[k] > 0.0) idx |= 1 << k;
    }
    *index = idx;
    return flags;
}

// Sign field and range flags of a cell without forming the (value - level) doubles: (double)f - level > 0 is exactly
// (double)f > level (IEEE subtraction of distinct doubles never rounds to zero), so inactive cells cost 8 compares.
R3G_DEV unsigned load_signs(const float* g, int nx, int ny, int x, int y, int z, double level, int* index) {
    const int64_t sy = nx, sz = (int64_t)nx * ny;
    const float* p = g + (int64_t)z * sz + (int64_t)y * sy + x;
    float f[8];
    f[0] = p[0]; f[1] = p[1]; f[2] = p[sy + 1]; f[3] = p[sy];
    f[4] = p[sz]; f[5] = p[sz + 1]; f[6] = p[sz + sy + 1]; f[7] = p[sz + sy];
    int idx = 0;
    unsigned flags = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double d = (double)f[k];
        if (d > level) idx |= 1 << k;
        else if (d <= level) flags |= R3G_MC_FLAG_LE;   // not greater and ordered
        else flags |= R3G_MC_FLAG_NAN;
        if (d == level) flags |= R3G_MC_FLAG_GE;
    }
    if (idx) flags |= R3G_MC_FLAG_GE;
    *index = idx;
    return flags;
}

R3G_DEV unsigned classify_cell(const double* v, int index, bool classic, int x, int y, int z) {
    if (index == 0 || index == 255) return 0;
    const Tiling t = select_tiling(v, index, classic);
    if (t.nt == 0) return 0;
    const int nv = count_new_vertices(t, owned_mask(x, y, z));
    return (unsigned)t.off | ((unsigned)t.nt << 16) | ((unsigned)nv << 20);
}

// Float-domain form of the sign test used by the row kernel: with lo = the largest float <= level,
// (double)f > level  <=>  f > lo   and   (double)f <= level  <=>  f <= lo   (no float lies in (lo, level]),
// and (double)f == level  <=>  level is a float and f == lo.
R3G_HOSTDEV void level_floor(double level, float* lo, int* exact) {
    float lf = (float)level;
    if ((double)lf > level) {
        // step to the next float towards -inf (written out: no libm in device code)
        union { float f; uint32_t u; } c;
        c.f = lf;
        if (lf > 0.0f) c.u -= 1u;
        else if (lf == 0.0f) c.u = 0x80000001u;
        else c.u += 1u;
        lf = c.f;
    }
    *lo = lf;
    *exact = ((double)lf == level) ? 1 : 0;
}

R3G_DEV bool node_greater(float f, float lo, int exact, unsigned* flags) {
    const bool gt = f > lo;
    if (!gt) *flags |= (f <= lo) ? R3G_MC_FLAG_LE : R3G_MC_FLAG_NAN;
    if (gt || (exact && f == lo)) *flags |= R3G_MC_FLAG_GE;
    return gt;
}

// Vertices of one active cell.  Edge endpoints are read from the grid per edge (two L1-resident
// floats) instead of indexing a corner array dynamically, which would live in scratch memory.
R3G_DEV void emit_cell_vertices(unsigned rec, unsigned vbase, const float* g, double level, int x, int y, int z,
                                int nx, int ny, int32_t* etab, float* verts, const Xform& xf, bool use_xf) {
    const int off = (int)(rec & 0xFFFFu), n = 3 * (int)((rec >> 16) & 0xFu);
    const unsigned owned = owned_mask(x, y, z);
    const int64_t sy = nx, sz = (int64_t)nx * ny;
    const float* p = g + (int64_t)z * sz + (int64_t)y * sy + x;
    unsigned seen = 0, id = vbase;
    uint64_t tri = 0;
    for (int i = 0; i < n; ++i) {
        if (i % 12 == 0) tri = load_tri12(off, i, n);
        const int e = (int)((tri >> (4 * (i % 12))) & 0xF);
        const unsigned bit = 1u << e;
        if ((seen & bit) || !(owned & bit)) continue;
        seen |= bit;
        double pos[3];
        if (e == 12) {
            double v[8];
            int index;
            load_corners(g, nx, ny, x, y, z, level, v, &index);
            center_vertex(v, x, y, z, pos);
        } else {
            const int dx1 = R3G_MC_EDGE_DX[e][0], dx2 = R3G_MC_EDGE_DX[e][1];
            const int dy1 = R3G_MC_EDGE_DY[e][0], dy2 = R3G_MC_EDGE_DY[e][1];
            const int dz1 = R3G_MC_EDGE_DZ[e][0], dz2 = R3G_MC_EDGE_DZ[e][1];
            const double a = (double)p[dz1 * sz + dy1 * sy + dx1] - level;
            const double b = (double)p[dz2 * sz + dy2 * sy + dx2] - level;
            edge_vertex_ab(a, b, e, x, y, z, pos);
            etab[edge_slot(e, x, y, z, nx, ny)] = (int32_t)id;
        }
        store_vertex(verts + 3 * (int64_t)id, pos, xf, use_xf);
        ++id;
    }
}

R3G_DEV void emit_cell_faces(unsigned rec, unsigned vbase, unsigned tbase, int x, int y, int z, int nx, int ny,
                             const int32_t* etab, int32_t* faces, bool reversed) {
    const int off = (int)(rec & 0xFFFFu), nt = (int)((rec >> 16) & 0xFu);
    const unsigned owned = owned_mask(x, y, z);
    unsigned seen = 0, next = 0;
    unsigned long long ranks = 0;  // 4 bits per local edge: rank among this cell's new vertices
    uint64_t packed = 0;
    for (int t = 0; t < nt; ++t) {
        int32_t tri[3];
        if (t % 4 == 0) packed = load_tri12(off, 3 * t, 3 * nt);
        R3G_UNROLL
        for (int j = 0; j < 3; ++j) {
            const int e = (int)((packed >> (4 * (3 * (t % 4) + j))) & 0xF);
            const unsigned bit = 1u << e;
            if (owned & bit) {
                if (!(seen & bit)) {
                    seen |= bit;
                    ranks |= (unsigned long long)next << (4 * e);
                    ++next;
                }
                tri[j] = (int32_t)(vbase + (unsigned)((ranks >> (4 * e)) & 0xF));
            } else {
                tri[j] = etab[edge_slot(e, x, y, z, nx, ny)];
            }
        }
        int32_t* dst = faces + 3 * ((int64_t)tbase + t);
        if (reversed) { dst[0] = tri[2]; dst[1] = tri[1]; dst[2] = tri[0]; }
        else { dst[0] = tri[0]; dst[1] = tri[1]; dst[2] = tri[2]; }
    }
}

}  // namespace r3g_mc
#endif
