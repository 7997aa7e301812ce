/*
 * oracle/mc_lewiner.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Sequential CPU restatement of the marching-cubes extractor that the reference's
 * stage "Hunyuan_2d_to_3d" reaches through hy3dgen:
 *   reference call site : src/2d_to_3d_models/run.py:77-84  (pipeline_shapegen(...))
 *   upstream            : hy3dgen/shapegen/models/autoencoders/surface_extractors.py
 *                         MCSurfaceExtractor.run -> skimage.measure.marching_cubes(
 *                         grid, mc_level, method="lewiner")          [un-vendored]
 *   algorithm           : scikit-image (requirements.txt:17, >=0.24.0; the build
 *                         container has 0.18.3) skimage/measure/
 *                         _marching_cubes_lewiner.py:280-349 (wrapper, present as
 *                         source) + _marching_cubes_lewiner_cy (compiled, source
 *                         not shipped: restated here from the published Lewiner
 *                         2003 algorithm and validated bit-for-bit against the
 *                         compiled module -- see tests/test_mc_oracle.py and
 *                         tools/make_mc_golden.py).
 *
 * Parity status: PINNED.  Checked against golden vectors A-D of SURVEY.md 4.3
 * (recorded from skimage 0.18.3) and against live skimage runs on random volumes
 * in the build container.
 *
 * Conventions reproduced (all from the wrapper, file cited above):
 *   - volume is C-contiguous float32 [n0][n1][n2]; "z" = axis 0 is the outermost
 *     scan axis, "x" = axis 2 the innermost (:292, kernel loop order);
 *   - a corner is inside iff (double)value - level > 0.0 (strict);
 *   - vertices are returned as (axis0, axis1, axis2) = fliplr of the kernel's
 *     (x, y, z) (:330); faces are fliplr'd for gradient_direction='descent'
 *     (:335-338); allow_degenerate=True so nothing is removed (:345);
 *   - vertex interpolation and all ambiguity tests are done in double, the stored
 *     vertex is float32;
 *   - vertices are shared through the unique grid edge they lie on (rolling
 *     two-layer table with 4 slots per cell: x-edge, y-edge, z-edge, centre).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mc_luts_oracle.h"

/* skimage: `cdef double FLT_EPSILON = np.spacing(1.0)` i.e. DBL_EPSILON */
#define MC_EPS 2.220446049250313e-16

typedef struct {
    int nx, ny, nz; /* nx = size of axis 2 (fastest) */
    int x, y, z;
    double v[8];     /* corner values minus level, Lewiner corner numbering */
    int index;
    int c12_done;
    double c12[3];
    int *layer1, *layer2;
    float* verts;
    int64_t nverts, cap_verts;
    int32_t* faces;
    int64_t nfaceidx, cap_faces;
} mc_cell;

static int mc_grow_verts(mc_cell* c) {
    if (c->nverts + 1 > c->cap_verts) {
        int64_t ncap = c->cap_verts ? c->cap_verts * 2 : 1024;
        float* p = (float*)realloc(c->verts, (size_t)ncap * 3 * sizeof(float));
        if (!p) return -1;
        c->verts = p;
        c->cap_verts = ncap;
    }
    return 0;
}

static int mc_grow_faces(mc_cell* c) {
    if (c->nfaceidx + 1 > c->cap_faces) {
        int64_t ncap = c->cap_faces ? c->cap_faces * 2 : 3072;
        int32_t* p = (int32_t*)realloc(c->faces, (size_t)ncap * sizeof(int32_t));
        if (!p) return -1;
        c->faces = p;
        c->cap_faces = ncap;
    }
    return 0;
}

/* slot of the unique grid edge (or centre) that cell-local edge `vi` lies on */
static int* mc_slot(mc_cell* c, int vi) {
    int i = c->nx * c->y + c->x;
    int j = 0;
    int* layer;
    if (vi < 8) {
        if (vi < 4) {
            layer = c->layer1;
        } else {
            vi -= 4;
            layer = c->layer2;
        }
        if (vi == 1) { i += 1; j = 1; }
        else if (vi == 2) { i += c->nx; }
        else if (vi == 3) { j = 1; }
    } else if (vi < 12) {
        layer = c->layer1;
        j = 2;
        if (vi == 9) i += 1;
        else if (vi == 10) i += c->nx + 1;
        else if (vi == 11) i += c->nx;
    } else {
        layer = c->layer1;
        j = 3;
    }
    return &layer[4 * i + j];
}

static void mc_center(mc_cell* c) {
    double w[8];
    for (int k = 0; k < 8; ++k) w[k] = 1.0 / (MC_EPS + fabs(c->v[k]));
    double fx = 0.0, fy = 0.0, fz = 0.0, ff = 0.0;
    /* corner k -> (dx,dy,dz): 0(0,0,0) 1(1,0,0) 2(1,1,0) 3(0,1,0) 4(0,0,1) 5(1,0,1) 6(1,1,1) 7(0,1,1) */
    static const double DX[8] = {0, 1, 1, 0, 0, 1, 1, 0};
    static const double DY[8] = {0, 0, 1, 1, 0, 0, 1, 1};
    static const double DZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    for (int k = 0; k < 8; ++k) {
        fx += DX[k] * w[k];
        fy += DY[k] * w[k];
        fz += DZ[k] * w[k];
        ff += w[k];
    }
    c->c12[0] = (double)c->x + 1.0 * fx / ff;
    c->c12[1] = (double)c->y + 1.0 * fy / ff;
    c->c12[2] = (double)c->z + 1.0 * fz / ff;
    c->c12_done = 1;
}

static int mc_add_vertex(mc_cell* c, double x, double y, double z) {
    if (mc_grow_verts(c)) return -1;
    float* p = c->verts + 3 * c->nverts;
    p[0] = (float)x;
    p[1] = (float)y;
    p[2] = (float)z;
    return (int)(c->nverts++);
}

static int mc_emit(mc_cell* c, int vi) {
    int* slot = mc_slot(c, vi);
    int id = *slot;
    if (vi == 12) {
        if (!c->c12_done) mc_center(c);
        if (id < 0) {
            id = mc_add_vertex(c, c->c12[0], c->c12[1], c->c12[2]);
            if (id < 0) return -1;
            *slot = id;
        }
    } else if (id < 0) {
        int dx1 = R3G_MC_EDGE_DX[vi][0], dx2 = R3G_MC_EDGE_DX[vi][1];
        int dy1 = R3G_MC_EDGE_DY[vi][0], dy2 = R3G_MC_EDGE_DY[vi][1];
        int dz1 = R3G_MC_EDGE_DZ[vi][0], dz2 = R3G_MC_EDGE_DZ[vi][1];
        /* values indexed by dz*4+dy*2+dx; Lewiner corner numbering swaps 2<->3, 6<->7 */
        static const int REMAP[8] = {0, 1, 3, 2, 4, 5, 7, 6};
        double a = c->v[REMAP[dz1 * 4 + dy1 * 2 + dx1]];
        double b = c->v[REMAP[dz2 * 4 + dy2 * 2 + dx2]];
        double w1 = 1.0 / (MC_EPS + fabs(a));
        double w2 = 1.0 / (MC_EPS + fabs(b));
        double fx = 0.0, fy = 0.0, fz = 0.0, ff = 0.0;
        fx += (double)dx1 * w1; fy += (double)dy1 * w1; fz += (double)dz1 * w1; ff += w1;
        fx += (double)dx2 * w2; fy += (double)dy2 * w2; fz += (double)dz2 * w2; ff += w2;
        id = mc_add_vertex(c, (double)c->x + 1.0 * fx / ff, (double)c->y + 1.0 * fy / ff,
                           (double)c->z + 1.0 * fz / ff);
        if (id < 0) return -1;
        *slot = id;
    }
    if (mc_grow_faces(c)) return -1;
    c->faces[c->nfaceidx++] = id;
    return 0;
}

static int mc_add_triangles(mc_cell* c, int off, int nt) {
    for (int i = 0; i < 3 * nt; ++i)
        if (mc_emit(c, R3G_MC_TRI[off + i])) return -1;
    return 0;
}

/* Lewiner test_face: does the ambiguous face contain part of the surface */
static int mc_test_face(const mc_cell* c, int face) {
    int af = face < 0 ? -face : face;
    const double* v = c->v;
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
    double acbd = A * C - B * D;
    if (acbd > -MC_EPS && acbd < MC_EPS) return face >= 0;
    return (double)face * A * acbd >= 0.0;
}

/* Lewiner test_interior */
static int mc_test_internal(const mc_cell* c, int mc_case, int config, int subconfig, int s) {
    const double* v = c->v;
    double t, At = 0.0, Bt = 0.0, Ct = 0.0, Dt = 0.0, a, b;
    int test = 0, edge = -1;
    if (mc_case == 4 || mc_case == 10) {
        a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        t = -b / (2 * a + MC_EPS);
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else {
        if (mc_case == 6) edge = R3G_MC_TEST6[config][2];
        else if (mc_case == 7) edge = R3G_MC_TEST7[config][4];
        else if (mc_case == 12) edge = R3G_MC_TEST12[config][3];
        else if (mc_case == 13)
            edge = R3G_MC_TRI[R3G_MC_OFF_TILING13_5_1 +
                              (config * R3G_MC_MID_TILING13_5_1 + subconfig) * R3G_MC_ROW_TILING13_5_1];
        /* reference-edge table: t on edge (p,q); B,C,D interpolate three parallel edges */
        static const int E[12][8] = {
            /* p  q   B0 B1 C0 C1 D0 D1 */
            {0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7},
            {3, 0, 2, 1, 6, 5, 7, 4}, {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2},
            {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0}, {0, 4, 3, 7, 2, 6, 1, 5},
            {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};
        if (edge >= 0 && edge < 12) {
            const int* e = E[edge];
            t = v[e[0]] / (v[e[0]] - v[e[1]] + MC_EPS);
            At = 0;
            Bt = v[e[2]] + (v[e[3]] - v[e[2]]) * t;
            Ct = v[e[4]] + (v[e[5]] - v[e[4]]) * t;
            Dt = v[e[6]] + (v[e[7]] - v[e[6]]) * t;
        }
    }
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    switch (test) {
        case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
        /* Lewiner's C++ falls through to `return s<0` when the 5/10 condition fails; the
         * compiled skimage kernel returns 0 there (its if-chain ends without a return) --
         * measured against the compiled module, see tests/test_mc_oracle.py::test_case4_table */
        case 5: return (At * Ct - Bt * Dt < MC_EPS) ? (s > 0) : 0;
        case 10: return (At * Ct - Bt * Dt >= MC_EPS) ? (s > 0) : 0;
        case 7: case 11: case 13: case 14: case 15: return s < 0;
    }
    return 0;
}

#define T2(NAME, cfg) (R3G_MC_OFF_##NAME + (cfg) * R3G_MC_ROW_##NAME)
#define T3(NAME, cfg, sub) (R3G_MC_OFF_##NAME + ((cfg) * R3G_MC_MID_##NAME + (sub)) * R3G_MC_ROW_##NAME)

/* the "big switch": choose the tiling of an ambiguous-aware Lewiner case */
static int mc_big_switch(mc_cell* c, int mc_case, int config) {
    int sub = 0;
    switch (mc_case) {
        case 1: return mc_add_triangles(c, T2(TILING1, config), 1);
        case 2: return mc_add_triangles(c, T2(TILING2, config), 2);
        case 3:
            if (mc_test_face(c, R3G_MC_TEST3[config])) return mc_add_triangles(c, T2(TILING3_2, config), 4);
            return mc_add_triangles(c, T2(TILING3_1, config), 2);
        case 4:
            if (mc_test_internal(c, 4, config, sub, R3G_MC_TEST4[config]))
                return mc_add_triangles(c, T2(TILING4_1, config), 2);
            return mc_add_triangles(c, T2(TILING4_2, config), 6);
        case 5: return mc_add_triangles(c, T2(TILING5, config), 3);
        case 6:
            if (mc_test_face(c, R3G_MC_TEST6[config][0])) return mc_add_triangles(c, T2(TILING6_2, config), 5);
            if (mc_test_internal(c, 6, config, sub, R3G_MC_TEST6[config][1]))
                return mc_add_triangles(c, T2(TILING6_1_1, config), 3);
            return mc_add_triangles(c, T2(TILING6_1_2, config), 9);
        case 7:
            if (mc_test_face(c, R3G_MC_TEST7[config][0])) sub += 1;
            if (mc_test_face(c, R3G_MC_TEST7[config][1])) sub += 2;
            if (mc_test_face(c, R3G_MC_TEST7[config][2])) sub += 4;
            switch (sub) {
                case 0: return mc_add_triangles(c, T2(TILING7_1, config), 3);
                case 1: return mc_add_triangles(c, T3(TILING7_2, config, 0), 5);
                case 2: return mc_add_triangles(c, T3(TILING7_2, config, 1), 5);
                case 3: return mc_add_triangles(c, T3(TILING7_3, config, 0), 9);
                case 4: return mc_add_triangles(c, T3(TILING7_2, config, 2), 5);
                case 5: return mc_add_triangles(c, T3(TILING7_3, config, 1), 9);
                case 6: return mc_add_triangles(c, T3(TILING7_3, config, 2), 9);
                default:
                    if (mc_test_internal(c, 7, config, sub, R3G_MC_TEST7[config][3]))
                        return mc_add_triangles(c, T2(TILING7_4_2, config), 9);
                    return mc_add_triangles(c, T2(TILING7_4_1, config), 5);
            }
        case 8: return mc_add_triangles(c, T2(TILING8, config), 2);
        case 9: return mc_add_triangles(c, T2(TILING9, config), 4);
        case 10:
            if (mc_test_face(c, R3G_MC_TEST10[config][0])) {
                if (mc_test_face(c, R3G_MC_TEST10[config][1]))
                    return mc_add_triangles(c, T2(TILING10_1_1_, config), 4);
                return mc_add_triangles(c, T2(TILING10_2, config), 8);
            }
            if (mc_test_face(c, R3G_MC_TEST10[config][1])) return mc_add_triangles(c, T2(TILING10_2_, config), 8);
            if (mc_test_internal(c, 10, config, sub, R3G_MC_TEST10[config][2]))
                return mc_add_triangles(c, T2(TILING10_1_1, config), 4);
            return mc_add_triangles(c, T2(TILING10_1_2, config), 8);
        case 11: return mc_add_triangles(c, T2(TILING11, config), 4);
        case 12:
            if (mc_test_face(c, R3G_MC_TEST12[config][0])) {
                if (mc_test_face(c, R3G_MC_TEST12[config][1]))
                    return mc_add_triangles(c, T2(TILING12_1_1_, config), 4);
                return mc_add_triangles(c, T2(TILING12_2, config), 8);
            }
            if (mc_test_face(c, R3G_MC_TEST12[config][1])) return mc_add_triangles(c, T2(TILING12_2_, config), 8);
            if (mc_test_internal(c, 12, config, sub, R3G_MC_TEST12[config][2]))
                return mc_add_triangles(c, T2(TILING12_1_1, config), 4);
            return mc_add_triangles(c, T2(TILING12_1_2, config), 8);
        case 13:
            for (int k = 0; k < 6; ++k)
                if (mc_test_face(c, R3G_MC_TEST13[config][k])) sub += 1 << k;
            sub = R3G_MC_SUBCONFIG13[sub];
            if (sub == 0) return mc_add_triangles(c, T2(TILING13_1, config), 4);
            if (sub <= 6) return mc_add_triangles(c, T3(TILING13_2, config, sub - 1), 6);
            if (sub <= 18) return mc_add_triangles(c, T3(TILING13_3, config, sub - 7), 10);
            if (sub <= 22) return mc_add_triangles(c, T3(TILING13_4, config, sub - 19), 12);
            if (sub <= 26) {
                int k = sub - 23;
                if (mc_test_internal(c, 13, config, k, R3G_MC_TEST13[config][6]))
                    return mc_add_triangles(c, T3(TILING13_5_1, config, k), 6);
                return mc_add_triangles(c, T3(TILING13_5_2, config, k), 10);
            }
            if (sub <= 38) return mc_add_triangles(c, T3(TILING13_3_, config, sub - 27), 10);
            if (sub <= 44) return mc_add_triangles(c, T3(TILING13_2_, config, sub - 39), 6);
            if (sub == 45) return mc_add_triangles(c, T2(TILING13_1_, config), 4);
            return 0; /* "impossible case 13": emits nothing */
        case 14: return mc_add_triangles(c, T2(TILING14, config), 4);
    }
    return 0;
}

/*
 * Returns 0 on success, -1 out of memory, -2 bad shape.  Outputs are malloc'd (free with
 * r3g_oracle_free): verts float32 [nv][3] in (axis0,axis1,axis2) order, faces int32 [nf][3]
 * in skimage's returned ('descent' = reversed) winding.  nv==0 is returned as success here;
 * the Python bridge raises skimage's RuntimeError('No surface found ...') for it.
 */
int r3g_oracle_mc(const float* vol, int n0, int n1, int n2, double level, int use_classic,
                  float** verts_out, int32_t** faces_out, int64_t* nv_out, int64_t* nf_out) {
    if (n0 < 2 || n1 < 2 || n2 < 2) return -2;
    mc_cell c;
    memset(&c, 0, sizeof c);
    c.nx = n2; c.ny = n1; c.nz = n0;
    size_t ls = (size_t)c.nx * c.ny * 4;
    c.layer1 = (int*)malloc(ls * sizeof(int));
    c.layer2 = (int*)malloc(ls * sizeof(int));
    if (!c.layer1 || !c.layer2) { free(c.layer1); free(c.layer2); return -1; }
    for (size_t i = 0; i < ls; ++i) c.layer1[i] = c.layer2[i] = -1;
    int rc = 0;
    const size_t sy = (size_t)n2, sz = (size_t)n1 * n2;
    for (int z = 0; z < n0 - 1 && !rc; ++z) {
        /* new_z_value: swap layers, clear the upper one */
        int* t = c.layer1; c.layer1 = c.layer2; c.layer2 = t;
        for (size_t i = 0; i < ls; ++i) c.layer2[i] = -1;
        for (int y = 0; y < n1 - 1 && !rc; ++y) {
            for (int x = 0; x < n2 - 1; ++x) {
                const float* p = vol + z * sz + y * sy + x;
                c.x = x; c.y = y; c.z = z;
                c.v[0] = (double)p[0] - level;
                c.v[1] = (double)p[1] - level;
                c.v[2] = (double)p[sy + 1] - level;
                c.v[3] = (double)p[sy] - level;
                c.v[4] = (double)p[sz] - level;
                c.v[5] = (double)p[sz + 1] - level;
                c.v[6] = (double)p[sz + sy + 1] - level;
                c.v[7] = (double)p[sz + sy] - level;
                int idx = 0;
                for (int k = 0; k < 8; ++k)
                    if (c.v[k] > 0.0) idx |= 1 << k;
                c.index = idx;
                c.c12_done = 0;
                if (use_classic) {
                    int nt = 0;
                    while (R3G_MC_TRI[R3G_MC_OFF_CASESCLASSIC + idx * 16 + 3 * nt] != -1) ++nt;
                    if (nt > 0) rc = mc_add_triangles(&c, R3G_MC_OFF_CASESCLASSIC + idx * 16, nt);
                } else {
                    int mc_case = R3G_MC_CASES[idx][0];
                    if (mc_case > 0) rc = mc_big_switch(&c, mc_case, R3G_MC_CASES[idx][1]);
                }
                if (rc) break;
            }
        }
    }
    free(c.layer1);
    free(c.layer2);
    if (rc) { free(c.verts); free(c.faces); return rc; }
    /* wrapper post-processing: fliplr(vertices), fliplr(faces) */
    for (int64_t i = 0; i < c.nverts; ++i) {
        float t = c.verts[3 * i];
        c.verts[3 * i] = c.verts[3 * i + 2];
        c.verts[3 * i + 2] = t;
    }
    int64_t nf = c.nfaceidx / 3;
    for (int64_t i = 0; i < nf; ++i) {
        int32_t t = c.faces[3 * i];
        c.faces[3 * i] = c.faces[3 * i + 2];
        c.faces[3 * i + 2] = t;
    }
    *verts_out = c.verts;
    *faces_out = c.faces;
    *nv_out = c.nverts;
    *nf_out = nf;
    return 0;
}

void r3g_oracle_free(void* p) { free(p); }
