// tests/emu/qem_emu.cpp -- TEST-ONLY host run of the edge-collapse decimator: the product's per-element bodies
// (3d-re-gen_amd/csrc/qem_core.h) and round loop (qem_driver.h) with the HIP kernels replaced by loops.  It lets the CPU
// test-suite check the geometric contract without a GPU and gives the GPU tests a bit-exact expectation (the algorithm
// is a pure function of the input).  It is not part of the product library and is never a fallback.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#define R3G_QEM_HD static inline
#define R3G_QEM_LAMBDA
#include "qem_driver.h"

namespace {

struct HostBackend {
    struct Atomics {
        static uint32_t inc(uint32_t* p) { return (*p)++; }
        static void min64(uint64_t* p, uint64_t v) { if (v < *p) *p = v; }
        static void min32(int32_t* p, int32_t v) { if (v < *p) *p = v; }
    };
    template <class F>
    void parfor(int64_t n, F f) {
        for (int64_t i = 0; i < n; ++i) f(i);
    }
    uint32_t scan(const uint32_t* in, int64_t n, uint32_t* out) {
        uint32_t acc = 0;
        for (int64_t i = 0; i < n; ++i) { const uint32_t v = in[i]; out[i] = acc; acc += v; }
        return acc;
    }
    uint64_t sum_if(const uint32_t* w, const uint64_t* key, int64_t n, uint64_t thr) {
        uint64_t acc = 0;
        for (int64_t i = 0; i < n; ++i) if (w[i] && key[i] <= thr) acc += w[i];
        return acc;
    }
    void zero(void* p, size_t bytes) { memset(p, 0, bytes); }
    void copy(void* dst, const void* src, size_t bytes) { memmove(dst, src, bytes); }
};

}  // namespace

extern "C" int r3g_emu_qem(float* verts, int64_t* nv_io, int32_t* faces, int64_t* nf_io, int64_t max_faces, int* rounds) {
    const int64_t nv = *nv_io, nf = *nf_io;
    if (nv <= 0 || nf <= 0 || nf <= max_faces) { if (rounds) *rounds = 0; return 0; }
    const int64_t m = nv > nf ? nv : nf;
    std::vector<uint32_t> deg(nv), off(nv + 1), sel(nv), keep(m), pos(m), used(nv);
    std::vector<int32_t> adj(3 * nf), partner(nv), mark_lo(nv), remap(nv), faces_tmp(3 * nf);
    std::vector<double> quad(10 * nv);
    std::vector<uint8_t> bnd(nv);
    std::vector<uint64_t> key(nv), mark_key(nv), inkey(nv);
    std::vector<int32_t> inwho(nv);
    std::vector<float> verts_tmp(3 * nv);
    r3g_qem::Buffers b{verts, faces, deg.data(), off.data(), adj.data(), quad.data(), bnd.data(), partner.data(), key.data(),
                       mark_lo.data(), mark_key.data(), sel.data(), remap.data(), keep.data(), pos.data(), faces_tmp.data(),
                       verts_tmp.data(), used.data(), inkey.data(), inwho.data()};
    HostBackend be;
    const r3g_qem::Result r = r3g_qem::decimate(be, b, nv, nf, max_faces);
    *nv_io = r.nv;
    *nf_io = r.nf;
    if (rounds) *rounds = r.rounds;
    return 0;
}
