// tests/emu/mc_emu.cpp -- TEST-ONLY host emulation of the HIP marching-cubes launch structure.
//
// Runs the product's per-cell bodies (3d-re-gen_amd/csrc/mc_cell.h) through the same four
// passes as mc_kernels.hip -- classify+block compaction, block-offset scan, vertices+edge table,
// faces -- with the GPU's 256-cell blocks replaced by loops.  It lets the CPU test-suite check
// the ownership / ranking / numbering logic against the oracle without a GPU.  It is not part
// of the product library and is never a fallback: libr3g.so has no CPU path.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#define R3G_DEV static inline
#include "mc_cell.h"

using namespace r3g_mc;

extern "C" int r3g_emu_mc(const float* grid, int n0, int n1, int n2, double level, int classic,
                          const double* xf9, int reversed, float** verts_out, int32_t** faces_out,
                          int64_t* nV, int64_t* nF, unsigned* flags_out) {
    const int nx = n2, ny = n1, nz = n0;
    const int cx = nx - 1, cy = ny - 1, cz = nz - 1;
    if (cx < 1 || cy < 1 || cz < 1) return -2;
    const int64_t ncells = (int64_t)cx * cy * cz;
    const int64_t nblk = (ncells + 255) / 256;
    std::vector<uint32_t> rec(nblk * 256), loc(nblk * 256);
    std::vector<uint32_t> blkV(nblk), blkT(nblk), blkA(nblk);
    unsigned flags = 0;
    // pass 1: classify, in-block exclusive scan, compaction
    for (int64_t b = 0; b < nblk; ++b) {
        uint32_t sv = 0, st = 0, sa = 0;
        for (int t = 0; t < 256; ++t) {
            const int64_t c = b * 256 + t;
            if (c >= ncells) break;
            const int x = (int)(c % cx), y = (int)((c / cx) % cy), z = (int)(c / ((int64_t)cx * cy));
            double v[8];
            int index;
            const unsigned f1 = load_corners(grid, nx, ny, x, y, z, level, v, &index);
            int index2;
            const unsigned f2 = load_signs(grid, nx, ny, x, y, z, level, &index2);  // the kernel's fast path
            if (f1 != f2 || index != index2) return -4;
            {   // the row kernel's float-domain sign test must agree as well
                float lo; int exact;
                level_floor(level, &lo, &exact);
                const int64_t sy = nx, sz = (int64_t)nx * ny;
                const float* p = grid + z * sz + y * sy + x;
                const float f[8] = {p[0], p[1], p[sy + 1], p[sy], p[sz], p[sz + 1], p[sz + sy + 1], p[sz + sy]};
                unsigned f3 = 0; int index3 = 0;
                for (int k = 0; k < 8; ++k) if (node_greater(f[k], lo, exact, &f3)) index3 |= 1 << k;
                if (f3 != f1 || index3 != index) return -5;
            }
            flags |= f1;
            const unsigned r = classify_cell(v, index, classic != 0, x, y, z);
            if (r) {
                rec[b * 256 + sa] = r;
                loc[b * 256 + sa] = (uint32_t)t | (sv << 8) | (st << 20);
                sv += (r >> 20) & 0xF;
                st += (r >> 16) & 0xF;
                ++sa;
            }
        }
        blkV[b] = sv; blkT[b] = st; blkA[b] = sa;
    }
    // pass 2: exclusive scan of block sums
    std::vector<uint32_t> offV(nblk), offT(nblk);
    uint64_t tv = 0, tt = 0;
    for (int64_t b = 0; b < nblk; ++b) { offV[b] = (uint32_t)tv; offT[b] = (uint32_t)tt; tv += blkV[b]; tt += blkT[b]; }
    *nV = (int64_t)tv; *nF = (int64_t)tt; *flags_out = flags;
    float* verts = (float*)malloc(sizeof(float) * 3 * (tv ? tv : 1));
    int32_t* faces = (int32_t*)malloc(sizeof(int32_t) * 3 * (tt ? tt : 1));
    std::vector<int32_t> etab((size_t)3 * nx * ny * nz, -1);  // -1 only to catch bugs; the GPU table is uninitialised
    Xform xf;
    if (xf9) for (int i = 0; i < 3; ++i) { xf.grid_size[i] = xf9[i]; xf.bbox_size[i] = xf9[3 + i]; xf.bbox_min[i] = xf9[6 + i]; }
    // pass 3: vertices
    for (int64_t b = 0; b < nblk; ++b)
        for (uint32_t a = 0; a < blkA[b]; ++a) {
            const uint32_t r = rec[b * 256 + a], l = loc[b * 256 + a];
            const int64_t c = b * 256 + (l & 0xFF);
            const int x = (int)(c % cx), y = (int)((c / cx) % cy), z = (int)(c / ((int64_t)cx * cy));
            emit_cell_vertices(r, offV[b] + ((l >> 8) & 0xFFF), grid, level, x, y, z, nx, ny, etab.data(), verts, xf, xf9 != nullptr);
        }
    // pass 4: faces
    int bad = 0;
    for (int64_t b = 0; b < nblk; ++b)
        for (uint32_t a = 0; a < blkA[b]; ++a) {
            const uint32_t r = rec[b * 256 + a], l = loc[b * 256 + a];
            const int64_t c = b * 256 + (l & 0xFF);
            const int x = (int)(c % cx), y = (int)((c / cx) % cy), z = (int)(c / ((int64_t)cx * cy));
            emit_cell_faces(r, offV[b] + ((l >> 8) & 0xFFF), offT[b] + (l >> 20), x, y, z, nx, ny, etab.data(), faces, reversed != 0);
        }
    for (uint64_t i = 0; i < 3 * tt; ++i) if (faces[i] < 0 || (uint64_t)faces[i] >= tv) ++bad;
    *verts_out = verts; *faces_out = faces;
    return bad ? -3 : 0;
}

extern "C" void r3g_emu_free(void* p) { free(p); }
