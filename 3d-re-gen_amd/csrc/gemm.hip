// gemm.hip -- bf16 MFMA GEMM with fused epilogues for gfx950.
//
//   C[b][m][n] (+)= epi( sum_k A[b][m][k] * W[n][k] + bias[n] )
//
// A: activations, bf16 row-major (lda), optional batch stride.  W: nn.Linear weight [N][K] bf16 (K contiguous;
// a column slab of a wider matrix is addressed through ldw).  fp32 accumulation on
// v_mfma_f32_16x16x32_bf16.  This one kernel carries every dense projection of the hot path
// (upstream hunyuan3ddit.py DoubleStreamBlock/SingleStreamBlock/LastLayer Linear layers,
// autoencoders/attention_blocks.py c_qkv/c_q/c_kv/c_proj/MLP, Dinov2 projections), which the
// reference executes as cuBLAS GEMM + separate bias/GELU/gate/residual elementwise passes.
//
// Structure: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles.  Operands are
// staged HBM->LDS with 16-byte LDS-DMA (global_load_lds_dwordx4), double-buffered: the loads of
// k-tile t+1 are issued before the MFMAs of tile t and waited for (vmcnt(0) + barrier) after them.
// LDS image is lane-linear per DMA instruction; the bank-conflict swizzle is applied on the SOURCE
// chunk index (chunk ^ ((row>>1)&7)) and on the fragment reads, so ds_read_b128 is conflict-free.
// The MFMA operands are swapped (W fragment as "A", activation fragment as "B") so that each lane's
// 4 accumulator registers are 4 consecutive n of one row m: epilogue loads/stores are 8/16 bytes.
// Workgroup ids are remapped so that the 8 XCDs each own a contiguous run of tiles (L2 locality).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "gemm_common.h"
#include "wave_sum.h"
#include "kernels.h"
#include "prof.h"

namespace r3g {
namespace {

constexpr int BK = 64;

// compile-time loop (the bodies are separate inlined calls: no reliance on the loop unroller for large bodies)
template <int B, int E, int S, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + S, E, S>(f);
    }
}

// one operand tile: ROWS rows x 64 k, in pieces of 1 KiB (8 rows), spread over NW waves
template <bool GLDS, int NW, int ROWS>
__device__ __forceinline__ void stage_tile(const uint16_t* __restrict__ src, int64_t ld, int row0, int rows_valid,
                                           int k0, char* lds_tile, int wid, int lane, int tid) {
    if constexpr (GLDS) {
#pragma unroll
        for (int i = 0; i < ROWS / 8 / NW; ++i) {
            const int piece = wid * (ROWS / 8 / NW) + i;  // 1 KiB piece = 8 tile rows
            const int row = piece * 8 + (lane >> 3);
            const int kc = (lane & 7) ^ ((row >> 1) & 7);
            int gr = row0 + row;
            gr = gr < rows_valid ? gr : rows_valid - 1;
            const uint16_t* g = src + (int64_t)gr * ld + k0 + kc * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024), 16, 0, 0);
        }
    } else {
        uint4 v[ROWS / 8 / NW];
#pragma unroll
        for (int i = 0; i < ROWS / 8 / NW; ++i) {
            const int c = i * (NW * 64) + tid;
            const int row = c >> 3, kc = c & 7;
            int gr = row0 + row;
            gr = gr < rows_valid ? gr : rows_valid - 1;
            v[i] = *reinterpret_cast<const uint4*>(src + (int64_t)gr * ld + k0 + kc * 8);
        }
#pragma unroll
        for (int i = 0; i < ROWS / 8 / NW; ++i) {
            const int c = i * (NW * 64) + tid;
            const int row = c >> 3, kc = c & 7;
            *reinterpret_cast<uint4*>(lds_tile + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4)) = v[i];
        }
    }
}

// Epilogue shared by all GEMM kernels.  acc[j][i]: wave-local sub-tile (n group j of 16 columns, m group i of 16 rows);
// lane holds row m = m0 + wr*WROWS + i*16 + (lane&15), columns n0 + wc*64 + j*16 + (lane>>4)*4 + {0..3}.
// lds_wave: wave-private LDS scratch of WROWS x 128 bytes (free once the k-loop's last barrier has passed), or null.
// PI: 16-row groups per pass through the scratch (PI * 16 rows x 128 bytes of LDS per wave; PI = MI: one pass).
// VM0: s_waitcnt vmcnt(0) before the first global store (the persistent kernel issues the next tile's LDS-DMA before this
// epilogue and wants it landed, but must not wait for the epilogue's own stores afterwards).
// SLICE: byte distance between the 16-row groups (2 KiB each) of a scratch made of 128-byte rows.  2048 = one contiguous block
// (every kernel but the persistent one); the persistent kernel (round 6) passes 16384: the four groups of a 64-row pass are the
// wave's OWN 2 KiB staging slices in the four regions of k-tile buffer 1, which is idle between the last k-tile of a tile and
// the staging of the next tile's k-tile 1 -- and which only this wave's LDS-DMA overwrites, so no workgroup barrier is owed.
// Round 6: the scratch pointer is an LDS-address-space pointer (and "no scratch" a flag of its own).  As a generic `char*` that could
// be null, every access went through a flat -> local conversion with a null check and its lane address was rebuilt from scratch --
// ~14 vector instructions per ds_write, each write in a basic block of its own (hipcc -S), ~500-900 per tile: as much as the
// arithmetic of the epilogue.  With the address space known the compiler keeps one lane base and folds the rest into offsets.
#define R3G_LDS __attribute__((address_space(3)))
typedef R3G_LDS char* LdsP;
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));   // (HIP's uint2 / uint4 are class types: they cannot be copied out of an
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));   // address-space-qualified object; plain vector types can)
__device__ __forceinline__ uint4 lds_ld4(const R3G_LDS char* a) { const u32x4v v = *reinterpret_cast<const R3G_LDS u32x4v*>(a); return make_uint4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ uint2 lds_ld2(const R3G_LDS char* a) { const u32x2v v = *reinterpret_cast<const R3G_LDS u32x2v*>(a); return make_uint2(v[0], v[1]); }
__device__ __forceinline__ void lds_st2(R3G_LDS char* a, uint2 v) { *reinterpret_cast<R3G_LDS u32x2v*>(a) = (u32x2v){v.x, v.y}; }
__device__ __forceinline__ LdsP lds_ptr(char* generic_smem_ptr) { return (LdsP)generic_smem_ptr; }

template <int EPI, int MI, bool SMALLREG, int PI = MI, bool VM0 = false, int SLICE = 2048>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[4][MI], int m0, int n0, int batch, int wr,
                                              int wc, int lane, LdsP lds_wave = nullptr, bool have_lds = false,
                                              const f32x4* bias_pre = nullptr) {
    constexpr int WROWS = MI * 16;
    static_assert(MI % PI == 0, "scratch passes");
    static_assert(SLICE == 2048 || ((EPI == EPI_QKV || EPI == EPI_BF16 || EPI == EPI_BF16_GELU_TANH || EPI == EPI_BF16_GELU_ERF || EPI == EPI_BF16_GELU_ERF_LNF) && PI == 4),
                  "sliced scratch: 64-row passes of 128-byte rows");
    // byte offset of row R of a scratch of 128-byte rows
    auto roff = [](int R) __attribute__((always_inline)) { return SLICE == 2048 ? R * 128 : (R >> 4) * SLICE + (R & 15) * 128; };
    if constexpr (EPI == EPI_QKV) {
        // The wave's 64 columns are exactly one head of q, k or v.  Lane holds, for row m = .. + i*16 + (lane&15),
        // head dims d = j*16 + (lane>>4)*4 + {0..3}; the other dims of that row sit in lanes lane^16, lane^32, lane^48.
        const QkvEpi& e = p.qkv;
        const int g = (n0 + wc * 64) >> 6;
        if (n0 + wc * 64 >= p.N) {
            if (VM0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the wait every wave owes the next tile's first k-tile)
            return;
        }
        int type, head;  // 0 q, 1 k, 2 v
        if (e.layout == QKV_KHD) { type = g / e.heads; head = g - type * e.heads; }
        else if (e.layout == QKV_HEAD_QKV) { head = g / 3; type = g - head * 3; }
        else if (e.layout == QKV_HEAD_KV) { head = g >> 1; type = 1 + (g & 1); }
        else { type = 0; head = g; }
        const int dbase = (lane >> 4) << 2;
        // row segments: the three from the kernel argument, or (nseg > 3) the three of the device table that can meet this
        // wave's rows -- the table is sorted and no 128-row window meets more than three (checked at launch)
        int sg_n = e.nseg, sg_m0[3], sg_m1[3], sg_b[3], sg_d[3];
#pragma unroll
        for (int sgi = 0; sgi < 3; ++sgi) { sg_m0[sgi] = e.seg_m0[sgi]; sg_m1[sgi] = e.seg_m1[sgi]; sg_b[sgi] = e.seg_batch[sgi]; sg_d[sgi] = e.seg_dst[sgi]; }
        if (e.nseg > 3) {
            // lane i holds segment i (nseg <= 64); the segments that end at or before the wave's first row are counted by a
            // ballot (the table is sorted), the next three are broadcast from their lanes
            const int mw_u = __builtin_amdgcn_readfirstlane(m0 + wr * WROWS);
            const int4* tab = reinterpret_cast<const int4*>(e.seg_tab);
            int4 t = make_int4(0, 0, 0, 0);
            if (lane < e.nseg) t = tab[lane];
            const int first = __popcll(__ballot(lane < e.nseg && t.y <= mw_u));
            sg_n = 3;
#pragma unroll
            for (int sgi = 0; sgi < 3; ++sgi) {
                const int idx = first + sgi;
                const bool ok = idx < e.nseg;
                const int li = ok ? idx : 0;
                sg_m0[sgi] = ok ? __builtin_amdgcn_readlane(t.x, li) : 0;
                sg_m1[sgi] = ok ? __builtin_amdgcn_readlane(t.y, li) : 0;
                sg_b[sgi] = __builtin_amdgcn_readlane(t.z, li);
                sg_d[sgi] = __builtin_amdgcn_readlane(t.w, li);
            }
        }
        const float* nw = type == 0 ? e.qw : e.kw;
        const float* nb = type == 0 ? e.qb : e.kb;
        const bool do_norm = type < 2 && e.norm != QKN_NONE;
        // the per-column constants of the wave's 64 columns, loaded ONCE: inside the row loop the compiler has to re-load them
        // after every store (the pointers may alias for all it knows) and waits for them eight times per tile
        f32x4 bias4[4], nw4[4], nb4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bias4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            nw4[j] = (f32x4){1.f, 1.f, 1.f, 1.f};
            nb4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias4[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wc * 64 + j * 16 + dbase);
            if (do_norm && nw) nw4[j] = *reinterpret_cast<const f32x4*>(nw + j * 16 + dbase);
            if (do_norm && e.norm == QKN_LAYERNORM && nb) nb4[j] = *reinterpret_cast<const f32x4*>(nb + j * 16 + dbase);
        }
        constexpr int PROWS = PI * 16;
        // one pass of PI 16-row groups through the wave's scratch (PROWS x 128 bytes): rows ip*16 .. of the wave's sub-tile
        auto flush_pass = [&](const int ip) __attribute__((always_inline)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (VM0 && ip == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // destination of tile row `rr` (token m): attention batch and row, or -1 when the row is dropped
            auto map_row = [&](int m, int& ob, int64_t& drow) {
                ob = batch;
                drow = (int64_t)e.dst_row0 + m;
                if (m >= p.M) return false;
                if (sg_n == 0) return true;
                bool hit = false;
#pragma unroll
                for (int sgi = 0; sgi < 3; ++sgi)
                    if (sgi < sg_n && m >= sg_m0[sgi] && m < sg_m1[sgi]) {
                        hit = true;
                        ob = sg_b[sgi];
                        drow = (int64_t)sg_d[sgi] + (m - sg_m0[sgi]);
                    }
                return hit;
            };
            const int mw = m0 + wr * WROWS + ip * 16;
            // Round 6: the common case -- the pass's PROWS rows lie inside ONE segment (or there are no segments) and inside M -- is
            // decided once per pass on the scalar unit; every store address is then base + row * stride, where the general path maps
            // each row through up to three segments and multiplies 64-bit products per store (~30 vector instructions x 16 stores per
            // tile: as much as the norm arithmetic itself).  Same addresses, same data.
            const int mw_u = __builtin_amdgcn_readfirstlane(mw);
            bool whole = mw_u + PROWS <= p.M;
            int w_ob = batch;
            int64_t w_drow = (int64_t)e.dst_row0 + mw_u;
            if (sg_n > 0) {
                bool hit = false;
#pragma unroll
                for (int sgi = 0; sgi < 3; ++sgi)
                    if (sgi < sg_n && mw_u >= sg_m0[sgi] && mw_u + PROWS <= sg_m1[sgi]) {
                        hit = true;
                        w_ob = sg_b[sgi];
                        w_drow = (int64_t)sg_d[sgi] + (mw_u - sg_m0[sgi]);
                    }
                whole = whole && hit;
            }
            if (whole && type < 2) {
                const int cc = lane & 7;
                uint16_t* base = (type == 0 ? e.Q + (((int64_t)w_ob * e.heads + head) * e.Lq_pad + w_drow) * 64
                                            : e.K + (((int64_t)w_ob * e.heads + head) * e.Lk_pad + w_drow) * 64) + cc * 8;
#pragma unroll
                for (int t = 0; t < PROWS / 8; ++t) {
                    const int rr = t * 8 + (lane >> 3);
                    const uint4 dv = lds_ld4(lds_wave + roff(rr) + ((cc ^ (rr & 7)) << 4));
                    *reinterpret_cast<uint4*>(base + rr * 64) = dv;
                }
            } else if (whole && type == 2 && ((w_drow & 15) | ((int64_t)e.Lk_pad & 7)) == 0) {
                // (PROWS is a multiple of 16 and the segment keeps the rows consecutive: every 16-token group stays whole)
                constexpr int CPR = PROWS / 8;
                const int64_t lk = e.Lk_pad;
                uint16_t* vbase = e.Vt + (((int64_t)w_ob * e.heads + head) * 64) * lk + w_drow;
#pragma unroll
                for (int t = 0; t < 64 / (64 / CPR); ++t) {
                    const int d = t * (64 / CPR) + lane / CPR, ck = lane % CPR;
                    const int u = ck >> 1, h8 = (ck & 1) * 8;
                    const int sw = (d >> 2) & (CPR - 1);
                    const R3G_LDS char* drow_lds = lds_wave + (PROWS * 2 == 128 ? roff(d) : d * (PROWS * 2));
                    const uint2 lo = lds_ld2(drow_lds + (((2 * u) ^ sw) << 4) + h8);
                    const uint2 hi = lds_ld2(drow_lds + (((2 * u + 1) ^ sw) << 4) + h8);
                    *reinterpret_cast<uint4*>(vbase + (int64_t)d * lk + 16 * u + h8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            } else if (type < 2) {
                // general form (a pass that meets a segment boundary or the end of M): every row mapped on its own.  Rare, so the
                // loops stay rolled: unrolled, the two general forms were three quarters of the kernel's 35 000 instructions.
                // 8 tokens x 128 contiguous bytes per store instruction (a token's 64 dims are one cache line)
                const int cc = lane & 7;
#pragma unroll 1
                for (int t = 0; t < PROWS / 8; ++t) {
                    const int rr = t * 8 + (lane >> 3);
                    int ob;
                    int64_t drow;
                    if (!map_row(mw + rr, ob, drow)) continue;
                    const uint4 dv = lds_ld4(lds_wave + roff(rr) + ((cc ^ (rr & 7)) << 4));
                    uint16_t* base = (type == 0 ? e.Q + (((int64_t)ob * e.heads + head) * e.Lq_pad + drow) * 64
                                                : e.K + (((int64_t)ob * e.heads + head) * e.Lk_pad + drow) * 64);
                    *reinterpret_cast<uint4*>(base + cc * 8) = dv;
                }
            } else {
                // V^T[dim][key position]: a lane takes the 8 tokens that are contiguous in the destination (vt_key_pos:
                // tokens 16u + 4h + {0..3} and 16u + 8 + 4h + {0..3} = positions 16u + 8h + {0..7}, 16 bytes) when the
                // whole 16-token group stays together there, otherwise token by token (segment boundaries, ragged ends)
                constexpr int CPR = PROWS / 8;   // 16-byte chunks per dim row of the LDS image
#pragma unroll 1
                for (int t = 0; t < 64 / (64 / CPR); ++t) {
                    const int d = t * (64 / CPR) + lane / CPR, ck = lane % CPR;
                    const int u = ck >> 1, h8 = (ck & 1) * 8;
                    const int sw = (d >> 2) & (CPR - 1);
                    const R3G_LDS char* drow_lds = lds_wave + (PROWS * 2 == 128 ? roff(d) : d * (PROWS * 2));
                    const uint2 lo = lds_ld2(drow_lds + (((2 * u) ^ sw) << 4) + h8);
                    const uint2 hi = lds_ld2(drow_lds + (((2 * u + 1) ^ sw) << 4) + h8);
                    const int mfirst = mw + 16 * u;
                    int ob0, ob15;
                    int64_t dr0, dr15;
                    const bool ok0 = map_row(mfirst, ob0, dr0), ok15 = map_row(mfirst + 15, ob15, dr15);
                    const int64_t lk = e.Lk_pad;
                    if (ok0 && ok15 && ob0 == ob15 && dr15 == dr0 + 15 && ((dr0 & 15) | (lk & 7)) == 0) {
                        uint16_t* dst = e.Vt + (((int64_t)ob0 * e.heads + head) * 64 + d) * lk + dr0 + h8;
                        *reinterpret_cast<uint4*>(dst) = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    } else {
                        const uint32_t w4[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll 1
                        for (int k = 0; k < 8; ++k) {
                            const int tk = 16 * u + (k < 4 ? (h8 >> 1) + k : 8 + (h8 >> 1) + (k - 4));
                            int ob;
                            int64_t dr;
                            if (map_row(mw + tk, ob, dr))
                                e.Vt[(((int64_t)ob * e.heads + head) * 64 + d) * lk + vt_key_pos(dr)] =
                                    (uint16_t)((k < 2 ? w4[0] : k < 4 ? w4[1] : k < 6 ? w4[2] : w4[3]) >> ((k & 1) * 16));   // (selects: no dynamic register index)
                        }
                    }
                }
            }
            if (PI != MI) {      // the next pass reuses the scratch
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        };
        // passes of PI row groups: the groups of a pass are staged in the scratch (or stored directly), then the pass leaves.
        // (round 6: "through the scratch or not" is decided once, outside the unrolled loops)
        auto all_passes = [&](auto WIDEC) __attribute__((always_inline)) {
        constexpr bool kWide = decltype(WIDEC)::value;
        static_for<0, MI, PI>([&](auto IPC) __attribute__((always_inline)) {
        constexpr int ip = decltype(IPC)::value;
#pragma unroll
        for (int ii = 0; ii < PI; ++ii) {
            const int i = ip + ii;
            const int m = m0 + wr * WROWS + i * 16 + (lane & 15);
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[j][i];
                if (p.bias) v[j] += bias4[j];
            }
            // sums over the four lanes that share a row (lane ^ 16, lane ^ 32): v_permlane16_swap / v_permlane32_swap on the vector
            // pipe instead of two ds_bpermute round trips through LDS per row group (round 6; a + b in the same order: same bits)
            auto row4_sum = [](float x) __attribute__((always_inline)) {
                typedef __attribute__((ext_vector_type(2))) unsigned int u2;
                const u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
                const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
                const u2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
                return __uint_as_float(b[0]) + __uint_as_float(b[1]);
            };
            if (do_norm && e.norm == QKN_RMS) {
                // RMS norm (the DiT's q / k): no mean, no additive term -- round 6 takes them out of the arithmetic instead of
                // subtracting a zero and adding a zero vector per value (32 of ~100 vector instructions per 16-row group);
                // (v * r) * w in the general form's order: the same bits
                float s2 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) s2 += v[j][c] * v[j][c];
                s2 = row4_sum(s2);
                const float r = rsqrtf(s2 * (1.f / 64.f) + e.eps);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] * r * nw4[j];
            } else if (do_norm) {
                float mean = 0.f;
                if (e.norm == QKN_LAYERNORM) {
                    float s1 = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) s1 += v[j][0] + v[j][1] + v[j][2] + v[j][3];
                    mean = row4_sum(s1) * (1.f / 64.f);
                }
                float s2 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const float d = v[j][c] - mean; s2 += d * d; }
                s2 = row4_sum(s2);
                const float r = rsqrtf(s2 * (1.f / 64.f) + e.eps);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (v[j] - mean) * r * nw4[j] + nb4[j];
            }
            if (type == 0 && e.q_scale != 0.f) {   // softmax scale * log2(e) folded into q before the bf16 rounding
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= e.q_scale;
            }
            if constexpr (kWide) {
                // stage the normalised tile in the wave's LDS scratch: Q / K as [token][64 dims] (16-byte chunks
                // swizzled by the token), V as [dim][token] (chunks of 8 tokens swizzled by the dim group)
                const int r = ii * 16 + (lane & 15), q = lane >> 4;
                if (type < 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        uint2 pk;
                        pk.x = pack_bf16(v[j][0], v[j][1]);
                        pk.y = pack_bf16(v[j][2], v[j][3]);
                        const int chunk = (2 * j + (q >> 1)) ^ (r & 7);
                        lds_st2(lds_wave + roff(r) + chunk * 16 + (q & 1) * 8, pk);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int d = j * 16 + q * 4 + c;
                            const int chunk = (r >> 3) ^ ((d >> 2) & (PROWS / 8 - 1));
                            *reinterpret_cast<R3G_LDS uint16_t*>(lds_wave + (PROWS * 2 == 128 ? roff(d) : d * (PROWS * 2)) + chunk * 16 + (r & 7) * 2) =
                                (uint16_t)(pack_bf16(v[j][c], 0.f) & 0xFFFFu);
                        }
                }
                continue;
            }
            if (m >= p.M) continue;
            int64_t drow = (int64_t)e.dst_row0 + m;
            int ob = batch;
            if (sg_n > 0) {
                bool hit = false;
#pragma unroll
                for (int sgi = 0; sgi < 3; ++sgi)
                    if (sgi < sg_n && m >= sg_m0[sgi] && m < sg_m1[sgi]) {
                        hit = true;
                        ob = sg_b[sgi];
                        drow = (int64_t)sg_d[sgi] + (m - sg_m0[sgi]);
                    }
                if (!hit) continue;
            }
            if (type < 2) {
                uint16_t* base = (type == 0 ? e.Q + (((int64_t)ob * e.heads + head) * e.Lq_pad + drow) * 64
                                            : e.K + (((int64_t)ob * e.heads + head) * e.Lk_pad + drow) * 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint2 pk;
                    pk.x = pack_bf16(v[j][0], v[j][1]);
                    pk.y = pack_bf16(v[j][2], v[j][3]);
                    *reinterpret_cast<uint2*>(base + j * 16 + dbase) = pk;
                }
            } else {
                // V^T[d][key position]: one token per lane (2-byte stores; the LDS-transposed path below is the fast one)
                uint16_t* base = e.Vt + (((int64_t)ob * e.heads + head) * 64) * (int64_t)e.Lk_pad + vt_key_pos(drow);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        base[(int64_t)(j * 16 + dbase + c) * e.Lk_pad] = (uint16_t)(pack_bf16(v[j][c], 0.f) & 0xFFFFu);
            }
        }
        if constexpr (kWide) flush_pass(ip);
        });
        };
        if (have_lds) all_passes(std::true_type{});
        else all_passes(std::false_type{});
        return;
    }
    // epilogue: lane holds, for sub-tile (j,i): row m = .. + (lane&15), cols n = .. + (lane>>4)*4 + {0..3}.
    // All loads (bias, gate, old residual values) are issued before the first store so that the stores are not
    // serialised behind per-iteration waits.
    const int mrow = m0 + wr * WROWS + (lane & 15);
    const int ncol = n0 + wc * 64 + ((lane >> 4) << 2);
    const float* gate = p.gate ? p.gate + (int64_t)batch * p.strideGate : nullptr;
    if constexpr (EPI == EPI_FP8_GELU_ERF) {
        // e4m3 output with a static scale: the operand of the fp8 GEMM that follows (the MLP hidden of the geo decoder)
        uint8_t* C8 = reinterpret_cast<uint8_t*>(p.C) + (int64_t)batch * p.strideC;
        const float inv = p.out_inv_scale;
        const bool wide = have_lds && PI == MI && (p.N & 15) == 0 && (p.ldc & 15) == 0 && (p.strideC & 15) == 0 &&
                          (reinterpret_cast<uintptr_t>(p.C) & 15) == 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ncol + j * 16;
            f32x4 bj = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.bias) bj = *reinterpret_cast<const f32x4*>(p.bias + (n < p.N ? n : p.N - 4));
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const f32x4 v = gelu_erf4<false>(acc[j][i] + bj) * inv;
                int w = 0;
                w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
                const int r = i * 16 + (lane & 15), m = mrow + i * 16;
                if (wide) *reinterpret_cast<R3G_LDS uint32_t*>(lds_wave + r * 64 + ((j ^ (r & 3)) << 4) + ((lane >> 4) << 2)) = (uint32_t)w;
                else if (n < p.N && m < p.M) *reinterpret_cast<uint32_t*>(C8 + (int64_t)m * p.ldc + n) = (uint32_t)w;
            }
        }
        if (wide) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int cc = lane & 3, n = n0 + wc * 64 + cc * 16;
#pragma unroll
            for (int t = 0; t < WROWS / 16; ++t) {       // 16 rows x 64 contiguous bytes per store instruction
                const int rr = t * 16 + (lane >> 2);
                const int m = m0 + wr * WROWS + rr;
                const uint4 d = lds_ld4(lds_wave + rr * 64 + ((cc ^ (rr & 3)) << 4));
                if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(C8 + (int64_t)m * p.ldc + n) = d;
            }
        }
        return;
    }
    if constexpr (EPI == EPI_RESID_BF16 || EPI == EPI_RESID_F16 || EPI == EPI_RESID_BF16_LND || EPI == EPI_RESID_BF16_ST) {
        constexpr bool F16 = EPI == EPI_RESID_F16;
        constexpr bool ST = EPI == EPI_RESID_BF16_ST;     // stored AND per-row chunk statistics (kernels.h)
        constexpr bool LND = EPI == EPI_RESID_BF16_LND || ST;   // LND proper: the new values are not stored, statistics + dot instead
        // bf16 residual stream: x = bf16(x + gate * (acc + bias)), the sum formed in fp32.
        uint16_t* X = reinterpret_cast<uint16_t*>(p.C) + (int64_t)batch * p.strideC;
        const uint16_t* XR = p.resid_src ? p.resid_src + (int64_t)batch * p.strideC : X;   // where the old values come from
        f32x4 bj[4], gj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = ncol + j * 16;
            const int nbc = nb < p.N ? nb : p.N - 4;
            bj[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            gj[j] = (f32x4){1.f, 1.f, 1.f, 1.f};
            if (p.bias) bj[j] = *reinterpret_cast<const f32x4*>(p.bias + nbc);
            if (gate) gj[j] = *reinterpret_cast<const f32x4*>(gate + nbc);
        }
        const bool wide = have_lds && PI >= 2 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (p.strideC & 7) == 0 &&
                          (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && (reinterpret_cast<uintptr_t>(XR) & 15) == 0;
        if (wide) {
            // passes of RP rows x 64 columns of fp32 through the wave's scratch (256-byte rows, 16-byte chunks swizzled by
            // the row); the read side owns 8 consecutive columns of a row: one 16-byte load of the old values, one
            // 16-byte store -- 8 rows x 128 contiguous bytes per instruction, whole cache lines both ways
            constexpr int RP = PI * 8;          // rows per pass (PI * 16 rows x 128 B of scratch = RP rows x 256 B)
            constexpr int GP = PI / 2;          // 16-row groups per pass
            const int cc = lane & 7;
            const int n = n0 + wc * 64 + cc * 8;
            f32x4 gw0 = (f32x4){0.f, 0.f, 0.f, 0.f}, gw1 = gw0;
            if constexpr (LND && !ST) {
                const int nc = n < p.N ? n : p.N - 8;
                gw0 = *reinterpret_cast<const f32x4*>(p.lnd_gw + nc);
                gw1 = *reinterpret_cast<const f32x4*>(p.lnd_gw + nc + 4);
            }
            // (round 6) the wave's block inside the matrix: unclamped loads, unpredicated stores, one address product per tile
            const bool full = m0 + wr * WROWS + WROWS <= p.M && n0 + wc * 64 + 64 <= p.N;      // wave-uniform
            const int64_t row0 = (int64_t)(m0 + wr * WROWS + (lane >> 3)) * p.ldc + n, step8 = 8 * (int64_t)p.ldc;
#pragma unroll
            for (int ip = 0; ip < MI; ip += GP) {
                uint4 old[RP / 8];
                if (full) {
#pragma unroll
                    for (int t = 0; t < RP / 8; ++t)
                        old[t] = *reinterpret_cast<const uint4*>(XR + row0 + (int64_t)(ip * 2 + t) * step8);
                } else {
#pragma unroll
                    for (int t = 0; t < RP / 8; ++t) {
                        const int m = m0 + wr * WROWS + ip * 16 + t * 8 + (lane >> 3);
                        const int mc = m < p.M ? m : p.M - 1;
                        const int nc = n < p.N ? n : p.N - 8;
                        old[t] = *reinterpret_cast<const uint4*>(XR + (int64_t)mc * p.ldc + nc);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ii = 0; ii < GP; ++ii) {
                        const int r = ii * 16 + (lane & 15);
                        const int chunk = (j * 4 + (lane >> 4)) ^ (r & 15);
                        *reinterpret_cast<R3G_LDS f32x4*>(lds_wave + r * 256 + chunk * 16) = gj[j] * (acc[j][ip + ii] + bj[j]);
                    }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                auto rows_out = [&](auto FULLC) __attribute__((always_inline)) {
                constexpr bool kFull = decltype(FULLC)::value;
#pragma unroll
                for (int t = 0; t < RP / 8; ++t) {
                    const int rr = t * 8 + (lane >> 3);
                    const int m = m0 + wr * WROWS + ip * 16 + rr;
                    const f32x4 d0 = *reinterpret_cast<const R3G_LDS f32x4*>(lds_wave + rr * 256 + (((2 * cc) ^ (rr & 15)) << 4));
                    const f32x4 d1 = *reinterpret_cast<const R3G_LDS f32x4*>(lds_wave + rr * 256 + (((2 * cc + 1) ^ (rr & 15)) << 4));
                    const uint32_t o[4] = {old[t].x, old[t].y, old[t].z, old[t].w};
                    uint32_t q[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x2 ov = unpack16<F16>(o[k]);
                        const f32x4& d = k < 2 ? d0 : d1;
                        q[k] = pack16<F16>(ov[0] + d[(k & 1) * 2], ov[1] + d[(k & 1) * 2 + 1]);
                    }
                    if constexpr (LND) {
                        // the row's 64 columns of this wave sit in the 8 lanes that share lane >> 3: chunk sum, then the squared
                        // deviations from the chunk mean and the dot product with gamma * w, each reduced over those lanes
                        float v[8];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f32x2 u = unpack16<false>(q[k]);
                            v[2 * k] = u[0];
                            v[2 * k + 1] = u[1];
                        }
                        // (round 6: the three butterfly steps over the 8 lanes of a row by DPP -- csrc/wave_sum.h -- where hipcc had put nine
                        // ds_bpermute round trips per 8 rows: 288 per tile)
                        float s1 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                        s1 += lane_xor<1>(s1);
                        s1 += lane_xor<2>(s1);
                        s1 += lane_xor<4>(s1);
                        const float mu = s1 * (1.0f / 64.0f);
                        float m2 = 0.f, s3 = 0.f;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float dv = v[k] - mu;
                            m2 = fmaf(dv, dv, m2);
                            s3 = fmaf(v[k], k < 4 ? gw0[k] : gw1[k - 4], s3);
                        }
                        m2 += lane_xor<1>(m2); s3 += lane_xor<1>(s3);
                        m2 += lane_xor<2>(m2); s3 += lane_xor<2>(s3);
                        m2 += lane_xor<4>(m2); s3 += lane_xor<4>(s3);
                        if (cc == 0 && m < p.M && n < p.N)
                            *reinterpret_cast<f32x4*>(p.lnd_part + ((int64_t)m * (p.N >> 6) + ((n0 >> 6) + wc)) * 4) = (f32x4){s1, m2, s3, 0.f};
                    }
                    if constexpr (!LND || ST) {
                        if constexpr (kFull) *reinterpret_cast<uint4*>(X + row0 + (int64_t)(ip * 2 + t) * step8) = make_uint4(q[0], q[1], q[2], q[3]);
                        else if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(X + (int64_t)m * p.ldc + n) = make_uint4(q[0], q[1], q[2], q[3]);
                    }
                }
                };
                if (full) rows_out(std::true_type{});
                else rows_out(std::false_type{});
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            return;
        }
        if constexpr (LND) return;   // (the launcher only admits launches that take the wide path)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint2 old[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int n = ncol + j * 16, m = mrow + i * 16;
                const int nc = n < p.N ? n : p.N - 4, mc = m < p.M ? m : p.M - 1;
                old[i] = *reinterpret_cast<const uint2*>(XR + (int64_t)mc * p.ldc + nc);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int n = ncol + j * 16, m = mrow + i * 16;
                const f32x4 d = gj[j] * (acc[j][i] + bj[j]);
                uint2 pk;
                const f32x2 ox = unpack16<F16>(old[i].x), oy = unpack16<F16>(old[i].y);
                pk.x = pack16<F16>(ox[0] + d[0], ox[1] + d[1]);
                pk.y = pack16<F16>(oy[0] + d[2], oy[1] + d[3]);
                if (n < p.N && m < p.M) *reinterpret_cast<uint2*>(X + (int64_t)m * p.ldc + n) = pk;
            }
        }
        return;
    }
    if constexpr (EPI == EPI_RESID_F32) {
        float* X = reinterpret_cast<float*>(p.C) + (int64_t)batch * p.strideC;
        const bool wide = have_lds && (p.ldc & 3) == 0 && (p.strideC & 3) == 0 &&
                          (reinterpret_cast<uintptr_t>(p.C) & 15) == 0;
        if (wide) {
            // Read-modify-write through a wave-private LDS transpose, 32 columns (two 16-column groups) per pass: the
            // global accesses become 8 rows x 128 contiguous bytes per instruction instead of 16 rows x 64 bytes.
            const int cc = lane & 7;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                f32x4 bj[2], gj[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int nb = ncol + (2 * jp + jj) * 16;
                    const int nbc = nb < p.N ? nb : p.N - 4;
                    bj[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    gj[jj] = (f32x4){1.f, 1.f, 1.f, 1.f};
                    if (p.bias) bj[jj] = *reinterpret_cast<const f32x4*>(p.bias + nbc);
                    if (gate) gj[jj] = *reinterpret_cast<const f32x4*>(gate + nbc);
                }
                const int n = n0 + wc * 64 + jp * 32 + cc * 4;
                const int nc = n < p.N ? n : p.N - 4;
#pragma unroll
                for (int ip = 0; ip < MI; ip += PI) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int ii = 0; ii < PI; ++ii) {
                            const int r = ii * 16 + (lane & 15);
                            const int chunk = (jj * 4 + (lane >> 4)) ^ (r & 7);
                            *reinterpret_cast<R3G_LDS f32x4*>(lds_wave + r * 128 + chunk * 16) = gj[jj] * (acc[2 * jp + jj][ip + ii] + bj[jj]);
                        }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (VM0 && jp == 0 && ip == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int t0 = 0; t0 < PI * 2; t0 += 4) {
                        f32x4 old[4];
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) {
                            const int m = m0 + wr * WROWS + ip * 16 + (t0 + tt) * 8 + (lane >> 3);
                            const int mc = m < p.M ? m : p.M - 1;
                            old[tt] = *reinterpret_cast<const f32x4*>(X + (int64_t)mc * p.ldc + nc);
                        }
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) {
                            const int rr = (t0 + tt) * 8 + (lane >> 3);
                            const int m = m0 + wr * WROWS + ip * 16 + rr;
                            const f32x4 d = *reinterpret_cast<const R3G_LDS f32x4*>(lds_wave + rr * 128 + ((cc ^ (rr & 7)) << 4));
                            if (n < p.N && m < p.M) *reinterpret_cast<f32x4*>(X + (int64_t)m * p.ldc + n) = old[tt] + d;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            return;
        }
    }
    f32x4 biasv[4], gatev[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = ncol + j * 16;
        const int nc = n < p.N ? n : p.N - 4;  // clamped: loads are unconditional, only stores are predicated
        biasv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gatev[j] = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (bias_pre) biasv[j] = bias_pre[j];     // loaded by the caller before it issued other memory traffic
        else if (p.bias) biasv[j] = *reinterpret_cast<const f32x4*>(p.bias + nc);
        if (EPI == EPI_RESID_F32 && gate) gatev[j] = *reinterpret_cast<const f32x4*>(gate + nc);
    }
    const int64_t cbase = (int64_t)batch * p.strideC;
    if constexpr (EPI == EPI_RESID_F32) {
        float* X = reinterpret_cast<float*>(p.C) + cbase;   // (VM0: the loads of the old values are the wait)
        // 16-wave tiles run at a 128-VGPR budget: read-modify-write one column group at a time there
        constexpr int JG = SMALLREG ? 1 : 4;
#pragma unroll
        for (int j0 = 0; j0 < 4; j0 += JG) {
            f32x4 old[JG][MI];
#pragma unroll
            for (int jj = 0; jj < JG; ++jj)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int n = ncol + (j0 + jj) * 16, m = mrow + i * 16;
                    const int nc = n < p.N ? n : p.N - 4, mc = m < p.M ? m : p.M - 1;
                    old[jj][i] = *reinterpret_cast<const f32x4*>(X + (int64_t)mc * p.ldc + nc);
                }
#pragma unroll
            for (int jj = 0; jj < JG; ++jj)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int j = j0 + jj;
                    const int n = ncol + j * 16, m = mrow + i * 16;
                    const f32x4 o = old[jj][i] + gatev[j] * (acc[j][i] + biasv[j]);
                    if (n < p.N && m < p.M) *reinterpret_cast<f32x4*>(X + (int64_t)m * p.ldc + n) = o;
                }
        }
    } else {
        // bf16 outputs go through a wave-private LDS transpose when the layout allows 16-byte stores: a store
        // instruction then writes 8 rows x 128 contiguous bytes (whole cache lines) instead of 16 rows x 32 bytes
        const bool wide = EPI != EPI_F32 && have_lds && (p.N & 7) == 0 && (p.ldc & 7) == 0 &&
                          (p.strideC & 7) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0;
        if (VM0 && !wide) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // LNF (round 6): the LayerNorm of A's rows folded into this GEMM -- v = rstd[m] (acc - mean[m] c1[n]) + c2[n] (c2 = p.bias)
        constexpr bool LNF = EPI == EPI_BF16_GELU_ERF_LNF;
        f32x4 c1v[4];
        f32x2 st[MI];
        if constexpr (LNF) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = ncol + j * 16;
                c1v[j] = *reinterpret_cast<const f32x4*>(p.lnf_c1 + (n < p.N ? n : p.N - 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = mrow + i * 16;
                st[i] = *reinterpret_cast<const f32x2*>(p.lnf_stats + 2 * (int64_t)(m < p.M ? m : p.M - 1));
            }
        }
        // PK: the GELU epilogues in packed fp16 (gemm_common.h; run-time option "gelu_pk", one wave-uniform branch per tile)
        // WIDEC: the LDS-transposed form, decided ONCE per tile (round 6: as a run-time test inside the unrolled loops it cut the
        // epilogue into one basic block per LDS write)
        auto body = [&](auto PKC, auto WIDEC) __attribute__((always_inline)) {
        constexpr bool PK = decltype(PKC)::value;
        constexpr bool kWide = decltype(WIDEC)::value && EPI != EPI_F32;
        static_assert(MI % 2 == 0 && PI % 2 == 0, "register groups are taken two at a time");
        // two register groups (rows i, i + 1 of column group j) at a time: with the packed-fp16 GELU their four pair chains run in
        // lockstep (gemm_common.h gelu_pk_sn)
        auto value2 = [&](int j, int i, f32x4 (&v)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v[h] = acc[j][i + h] + biasv[j];
                if constexpr (LNF) v[h] = st[i + h][1] * (acc[j][i + h] - st[i + h][0] * c1v[j]) + biasv[j];
            }
            constexpr bool kTanh = EPI == EPI_BF16_GELU_TANH, kErf = EPI == EPI_BF16_GELU_ERF || LNF;
            if constexpr ((kTanh || kErf) && PK) {
                gelu_pk4n<kErf, 2>(v);
            } else if constexpr (kTanh) {
                v[0] = gelu_tanh4<false>(v[0]); v[1] = gelu_tanh4<false>(v[1]);
            } else if constexpr (kErf) {
                v[0] = gelu_erf4<false>(v[0]); v[1] = gelu_erf4<false>(v[1]);
            }
        };
        if constexpr (kWide) {
            // lane parts of the scratch addresses: row (lane & 15) of a 16-row group, 8-byte half (q & 1) of chunk (2 j + (q >> 1)) ^ (row & 7)
            const int q = lane >> 4, l7 = lane & 7;
            const LdsP wbase = lds_wave + (lane & 15) * 128 + (q & 1) * 8;
            int wch[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wch[j] = ((2 * j + (q >> 1)) ^ l7) << 4;
            const int cc = lane & 7;
            const LdsP rbase = lds_wave + (lane >> 3) * 128 + ((cc ^ ((lane >> 3) & 7)) << 4);   // row t * 8 + (lane >> 3): (rr & 7) == (lane >> 3)
            uint16_t* C = reinterpret_cast<uint16_t*>(p.C) + cbase;
            const int n = n0 + wc * 64 + cc * 8;
            const bool full = m0 + wr * WROWS + WROWS <= p.M && n0 + wc * 64 + 64 <= p.N;      // wave-uniform
            uint16_t* const crow = C + (int64_t)(m0 + wr * WROWS + (lane >> 3)) * p.ldc + n;    // row (lane >> 3) of the wave's block
            const int64_t step8 = 8 * (int64_t)p.ldc;
#pragma unroll
            for (int ip = 0; ip < MI; ip += PI) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ii = 0; ii < PI; ii += 2) {
                        f32x4 v[2];
                        value2(j, ip + ii, v);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            uint2 pk;
                            pk.x = pack_bf16(v[h][0], v[h][1]);
                            pk.y = pack_bf16(v[h][2], v[h][3]);
                            lds_st2(wbase + roff((ii + h) * 16) + wch[j], pk);
                        }
                    }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (VM0 && ip == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (full) {      // the wave's 128 x 64 block lies inside the matrix: no predicates, one address product per tile
                    // all LDS reads of the pass first, then the stores (round 6: hipcc kept two reads in flight and every store
                    // waited out an LDS round trip -- s_memtime put the epilogue of the wave row that stores second at 9.6 k ticks)
                    uint4 d[PI * 2];
#pragma unroll
                    for (int t = 0; t < PI * 2; ++t) d[t] = lds_ld4(rbase + roff(t * 8));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < PI * 2; ++t)
                        *reinterpret_cast<uint4*>(crow + (int64_t)(ip * 2 + t) * step8) = d[t];
                } else {
#pragma unroll
                    for (int t = 0; t < PI * 2; ++t) {
                        const int m = m0 + wr * WROWS + ip * 16 + t * 8 + (lane >> 3);
                        const uint4 d = lds_ld4(rbase + roff(t * 8));
                        if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(C + (int64_t)m * p.ldc + n) = d;
                    }
                }
                if (PI != MI) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i2 = 0; i2 < MI; i2 += 2) {
                    f32x4 v2[2];
                    value2(j, i2, v2);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int i = i2 + h;
                        const int n = ncol + j * 16, m = mrow + i * 16;
                        const f32x4 v = v2[h];
                        if (n < p.N && m < p.M) {
                            const int64_t off = cbase + (int64_t)m * p.ldc + n;
                            if (EPI == EPI_F32) {
                                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + off) = v;
                            } else {
                                uint2 pk;
                                pk.x = pack_bf16(v[0], v[1]);
                                pk.y = pack_bf16(v[2], v[3]);
                                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.C) + off) = pk;
                            }
                        }
                    }
                }
        }
        };
        auto body_w = [&](auto PKC) __attribute__((always_inline)) {
            if (wide) body(PKC, std::true_type{});
            else body(PKC, std::false_type{});
        };
        if constexpr (EPI == EPI_BF16_GELU_TANH || EPI == EPI_BF16_GELU_ERF || EPI == EPI_BF16_GELU_ERF_LNF) {
            if (p.gelu_pk) body_w(std::true_type{});
            else body_w(std::false_type{});
        } else {
            body_w(std::false_type{});
        }
    }
}

// NW = 4: waves 2(m) x 2(n), 64x64 per wave.  NW = 8: waves 4(m) x 2(n), 32x64 per wave (a wave always owns 64
// whole columns = one attention head for the EPI_QKV epilogue).  NS = LDS stages (2: wait for everything each k-tile;
// 3: LDS-DMA of tile t+2 stays in flight across the barrier -- counted vmcnt + raw s_barrier).
// Two problems can share one launch (pb.M > 0): the tiles of pb follow those of pa in the grid.  The two streams of a
// DiT double block run the same layer shapes on different weights; the short one (1371 context rows = 88 tiles of
// 128x128) would leave two thirds of the CUs idle on its own.
template <int EPI, bool GLDS, int NW, int BIG>
__global__ __launch_bounds__(NW * 64) void gemm_kernel(GemmArgs pa, GemmArgs pb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // BIG = 1: 256x256 tile (16 waves of 64x64, or 8 waves of 128x64); BIG = 0: 128x128 (4 waves 64x64 / 8 waves 32x64)
    //       BIG = 2: 256x128 tile (8 waves of 64x64): 1/3 less staging traffic than 128x128, finer grid than 256x256
    constexpr int WN = BIG == 1 ? 4 : 2;      // waves along n (64 columns each)
    constexpr int WM = NW / WN;               // waves along m
    constexpr int BM = BIG ? 256 : 128, BN = WN * 64;
    constexpr int MI = BM / WM / 16;          // 16-row sub-tiles per wave
    constexpr int WROWS = MI * 16;
    constexpr int TILE_A = BM * BK * 2, TILE_W = BN * BK * 2, STAGE_BYTES = TILE_A + TILE_W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware bijective remap of the 1-D workgroup id (block b runs on XCD b % 8)
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int wg_all = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int tiles_a = ((pa.N + BN - 1) / BN) * ((pa.M + BM - 1) / BM) * pa.batch;
    const bool second = wg_all >= tiles_a;                 // wave-uniform
    // the selected problem is read straight from the kernarg segment (scalar loads at a uniform offset); selecting
    // between the two by-value structs would make the compiler copy both to scratch
    typedef const char __attribute__((address_space(4))) * kernarg_ptr;
    kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kSecond = (sizeof(GemmArgs) + alignof(GemmArgs) - 1) / alignof(GemmArgs) * alignof(GemmArgs);
    const GemmArgs& p = *(const GemmArgs*)(const GemmArgs __attribute__((address_space(4)))*)(ka + (second ? kSecond : 0));
    (void)pb;
    const int wg = second ? wg_all - tiles_a : wg_all;
    // L2-friendly rasterisation: tiles are walked in groups of GN tile-columns, row-major inside a group, so the ~64
    // tiles resident on one XCD span ~8 tile-rows x 8 tile-columns (A and W panels of a group stay in the 4 MiB L2).
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    // automatic: 4-column groups for 256x256 tiles and for wide 128x128 grids (>= 16 tile columns; run 32: +2..7 %)
    const int rg = p.raster_group < 0 ? ((BIG == 1 || tiles_n >= 16) ? 4 : 0) : p.raster_group;
    const int GN = rg > 0 ? rg : tiles_n;  // 0: plain row-major tile order
    const int rows_all = tiles_m * p.batch;  // (batch, tm) flattened
    const int per_group = rows_all * GN;
    const int group = wg / per_group;
    const int within = wg - group * per_group;
    const int gn_cur = (tiles_n - group * GN) < GN ? (tiles_n - group * GN) : GN;
    const int rowi = within / gn_cur;
    const int tn = group * GN + (within - rowi * gn_cur);
    const int batch = rowi / tiles_m;
    const int tm = rowi - batch * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const uint16_t* A = p.A + (int64_t)batch * p.strideA;
    const uint16_t* W = p.W + (int64_t)batch * p.strideW;
    const int wr = wid / WN, wc = wid % WN;

    f32x4 acc[4][MI];  // [j: n sub-tile][i: m sub-tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    auto stage = [&](int t, int slot) {
        char* base = smem + slot * STAGE_BYTES;
        stage_tile<GLDS, NW, BM>(A, p.lda, m0, p.M, t * BK, base, wid, lane, tid);
        stage_tile<GLDS, NW, BN>(W, p.ldw, n0, p.N, t * BK, base + TILE_A, wid, lane, tid);
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment read offsets (bytes) inside a tile, for the two 32-wide k-steps of a BK=64 tile
    int offA[MI], offB[4];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int rowA = wr * WROWS + i * 16 + (lane & 15);
        offA[i] = rowA * 128 + ((((lane >> 4)) ^ ((rowA >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rowB = wc * 64 + i * 16 + (lane & 15);
        offB[i] = rowB * 128 + ((((lane >> 4)) ^ ((rowB >> 1) & 7)) << 4);
    }
    // chunk index = kk*4 + (lane>>4); XOR with the row swizzle commutes with adding kk*4 (bit 2 of the chunk)
    auto load_frags = [&](const char* tile, int kk, bf16x8 (&a)[MI], bf16x8 (&b)[4]) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(tile + (offA[i] ^ (kk << 6)));
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const bf16x8*>(tile + TILE_A + (offB[i] ^ (kk << 6)));
    };
    auto mma = [&](const bf16x8 (&a)[MI], const bf16x8 (&b)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j][i], 0, 0, 0);
    };

    for (int t = 0; t < nk; ++t) {
        const int slot = t & 1;
        const char* cur = smem + slot * STAGE_BYTES;
        if (t + 1 < nk) stage(t + 1, slot ^ 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[MI], b[4];
            load_frags(cur, kk, a, b);
            mma(a, b);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    gemm_epilogue<EPI, MI, (BIG == 1)>(p, acc, m0, n0, batch, wr, wc, lane, lds_ptr(smem + wid * (WROWS * 128)), p.wide_epilogue != 0);
}

// ------------------------------------------------------------------------------------------------------------
// Implicit-GEMM 3 x 3 convolution (round 5): gemm_kernel<EPI, true, 8, 0> (128 x 128 tile, 8 waves of 32 x 64, two LDS stages) with
// the A operand gathered straight from the activation rows -- the im2col matrix (18 bytes written and read per input element,
// 7-8 % of the texture step's kernel time at HBM speed, profiles/r05_texture_stage.md) is never materialised.  A lane stages the
// same tile rows in every k-step, so its output pixel (sample, oy, ox) is decomposed ONCE; a k-step of 64 lies inside one tap
// (Cin % 64 == 0), so per k-step the lane adds the tap's (dy, dx), tests the image border and points the LDS-DMA at the pixel's
// channels or at 128 bytes of zeros.  Same fragment layout, k order and MFMA shape as the GEMM on the im2col matrix: the results
// are bit-identical to it.  Split-K: slice `batch` covers k in [batch K, (batch + 1) K) of the 9 Cin columns (W advances by strideW).
template <int EPI>
__global__ __launch_bounds__(512) void conv_gemm_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 8, WN = 2, BM = 128, BN = 128, MI = 2, WROWS = 32;
    constexpr int TILE_A = BM * BK * 2, TILE_W = BN * BK * 2, STAGE_BYTES = TILE_A + TILE_W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int rg = p.raster_group < 0 ? (tiles_n >= 16 ? 4 : 0) : p.raster_group;
    const int GN = rg > 0 ? rg : tiles_n;
    const int rows_all = tiles_m * p.batch;
    const int per_group = rows_all * GN;
    const int group = wg / per_group;
    const int within = wg - group * per_group;
    const int gn_cur = (tiles_n - group * GN) < GN ? (tiles_n - group * GN) : GN;
    const int rowi = within / gn_cur;
    const int tn = group * GN + (within - rowi * gn_cur);
    const int batch = rowi / tiles_m;
    const int tm = rowi - batch * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const uint16_t* W = p.W + (int64_t)batch * p.strideW;
    const int wr = wid / WN, wc = wid % WN;
    const ConvA& cv = p.conv;

    // the two 8-row pieces this wave stages per k-step: the lane's output pixel, decomposed once
    int iy0[2], ix0[2], pix[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wid * 2 + i) * 8 + (lane >> 3);
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        const int hw_o = cv.Ho * cv.Wo;
        const int n = m / hw_o, rem = m - n * hw_o;
        const int oy = rem / cv.Wo, ox = rem - oy * cv.Wo;
        iy0[i] = oy * cv.stride - cv.pad;
        ix0[i] = ox * cv.stride - cv.pad;
        pix[i] = n * cv.H * cv.W;
    }
    const int kbase = batch * p.K;

    f32x4 acc[4][MI];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    auto stage = [&](int t, int slot) {
        char* base = smem + slot * STAGE_BYTES;
        const int k0 = kbase + t * BK;                 // wave-uniform: the tap and the channel offset of this k-step
        const int tap = k0 / cv.Cin, c0 = k0 - tap * cv.Cin;
        const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;
            const int row = piece * 8 + (lane >> 3);
            const int kc = (lane & 7) ^ ((row >> 1) & 7);
            const int iy = iy0[i] + dy, ix = ix0[i] + dx;
            const bool inside = (unsigned)iy < (unsigned)cv.H && (unsigned)ix < (unsigned)cv.W;
            const uint16_t* g = inside ? cv.x + ((int64_t)(pix[i] + iy * cv.W + ix) * cv.Cin + c0 + kc * 8) : cv.zero + kc * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(base + piece * 1024), 16, 0, 0);
        }
        stage_tile<true, NW, BN>(W, p.ldw, n0, p.N, t * BK, base + TILE_A, wid, lane, tid);
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int offA[MI], offB[4];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int rowA = wr * WROWS + i * 16 + (lane & 15);
        offA[i] = rowA * 128 + ((((lane >> 4)) ^ ((rowA >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rowB = wc * 64 + i * 16 + (lane & 15);
        offB[i] = rowB * 128 + ((((lane >> 4)) ^ ((rowB >> 1) & 7)) << 4);
    }
    for (int t = 0; t < nk; ++t) {
        const int slot = t & 1;
        const char* cur = smem + slot * STAGE_BYTES;
        if (t + 1 < nk) stage(t + 1, slot ^ 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[MI], b[4];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(cur + (offA[i] ^ (kk << 6)));
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const bf16x8*>(cur + TILE_A + (offB[i] ^ (kk << 6)));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j][i], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    gemm_epilogue<EPI, MI, false>(p, acc, m0, n0, batch, wr, wc, lane, lds_ptr(smem + wid * (WROWS * 128)), p.wide_epilogue != 0);
}

// ------------------------------------------------------------------------------------------------------------
// Deep-ring variant: 256x256 tile, BK = 32, 4 LDS stages of 32 KiB (A 16 KiB + W 16 KiB), 8 waves of 128x64.
// Three k-steps of LDS-DMA stay in flight (96 KiB per CU) behind counted s_waitcnt vmcnt + raw s_barrier, and the
// tile carries twice the flops per staged byte of the 128x128 kernel.  Rows are 64 B (4 chunks); the bank swizzle is
// chunk ^ F[(row>>2)&3], F = {0,2,3,1}, which makes every ds_read_b128 lane group hit 16 distinct 16-byte slots.
__device__ __forceinline__ int swz4(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }  // {0,2,3,1}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_deep_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, BKD = 32, NSTG = 4, MI = 8;
    constexpr int TILE = BM * BKD * 2, STAGE = 2 * TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int rg = p.raster_group < 0 ? 4 : p.raster_group;
    const int GN = rg > 0 ? rg : tiles_n;
    const int rows_all = tiles_m * p.batch;
    const int per_group = rows_all * GN;
    const int group = wg / per_group;
    const int within = wg - group * per_group;
    const int gn_cur = (tiles_n - group * GN) < GN ? (tiles_n - group * GN) : GN;
    const int rowi = within / gn_cur;
    const int tn = group * GN + (within - rowi * gn_cur);
    const int batch = rowi / tiles_m;
    const int tm = rowi - batch * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const uint16_t* A = p.A + (int64_t)batch * p.strideA;
    const uint16_t* W = p.W;
    const int wr = wid >> 2, wc = wid & 3;

    f32x4 acc[4][MI];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BKD;
    // per-lane staging coordinates (2 pieces of A and 2 of W per wave and stage)
    auto stage = [&](int t, int slot) {
        char* base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;  // 1 KiB = 16 rows of 64 B
            const int row = piece * 16 + (lane >> 2);
            const int kc = (lane & 3) ^ swz4(row);
            int ga = m0 + row, gw = n0 + row;
            ga = ga < p.M ? ga : p.M - 1;
            gw = gw < p.N ? gw : p.N - 1;
            const uint16_t* sa = A + (int64_t)ga * p.lda + t * BKD + kc * 8;
            const uint16_t* sw = W + (int64_t)gw * p.ldw + t * BKD + kc * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                             (__attribute__((address_space(3))) void*)(base + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sw,
                                             (__attribute__((address_space(3))) void*)(base + TILE + piece * 1024), 16, 0, 0);
        }
    };
    int offA[MI], offB[4];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wr * 128 + i * 16 + (lane & 15);
        offA[i] = row * 64 + (((lane >> 4) ^ swz4(row)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wc * 64 + j * 16 + (lane & 15);
        offB[j] = TILE + row * 64 + (((lane >> 4) ^ swz4(row)) << 4);
    }

    stage(0, 0);
    if (nk > 1) stage(1, 1);
    if (nk > 2) stage(2, 2);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int slot = 0;
    for (int t = 0; t < nk; ++t) {
        const char* cur = smem + slot * STAGE;
        if (t + 3 < nk) stage(t + 3, (slot + 3) & 3);
        bf16x8 a[MI], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(cur + offB[j]);
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(cur + offA[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j][i], 0, 0, 0);
        // k-step t+1 must have landed; the (up to two) later ones that were already issued may stay in flight
        const int later = nk - 2 - t;  // k-steps issued beyond t+1
        if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = (slot + 1) & 3;
    }
    gemm_epilogue<EPI, MI, true>(p, acc, m0, n0, batch, wr, wc, lane);
}

// ------------------------------------------------------------------------------------------------------------
// Phased kernel: 256x256x64 tile, 8 waves (2 along m x 4 along n, 128x64 per wave), 128 KiB of LDS = 2 k-tile buffers
// of 4 half-tiles (16 KiB each).  The k-loop is cut into 4 phases per k-tile (8 per iteration of two k-tiles); a phase
// is {ds_read one register sub-tile, issue the LDS-DMA of ONE half-tile, s_barrier, 16 MFMAs = one accumulator quadrant
// x K=64, s_barrier}.  The two wave rows run one barrier apart, so that on every SIMD one wave is in its MFMA cluster
// while its partner reads LDS and issues loads; s_setprio keeps the matrix pipe fed.  LDS-DMA is waited for with a
// COUNTED vmcnt only (never 0 inside the loop): three half-tiles (48 KiB per CU) stay in flight across the barriers.
//
// Half-tiles are cut so that their LDS regions become free as early as possible:
//   A_h (h = 0,1): for both wave rows, the 64 rows of m-half h        (tile row  = (r>>6)*128 + h*64 + (r&63))
//   W_h          : for all four wave columns, the 32 columns of n-half h (tile col = (r>>5)*64  + h*32 + (r&31))
// Per k-tile in buffer b:   phase 1 reads W_0, A_0 -> quadrant (0,0);   phase 2 reads W_1 -> (0,1);
//                           phase 3 reads A_1 (into A_0's registers) -> (1,1);   phase 4 reads nothing -> (1,0).
// Staging, one half-tile per phase, three phases ahead of its first reader (tiles t, t+1 of this iteration in
// buffers 0, 1):  P1 A_1(t+1)->1 | P2 W_0(t+2)->0 | P3 A_0(t+2)->0 | P4 W_1(t+2)->0, vmcnt(6) |
//                 P5 A_1(t+2)->0 | P6 W_0(t+3)->1 | P7 A_0(t+3)->1 | P8 W_1(t+3)->1, vmcnt(6).
// Hazards.  WAR: a region is restaged two phases after its last read, except W_0 (one phase): its four reads are
// issued first in phase 1 and retired by lgkmcnt(8) before that phase's first barrier.  RAW: the wait that retires a
// staged half-tile sits before the first barrier of phase 4 / 8 of EVERY wave, its readers start in phase 5 / 1.
// Same LDS image as gemm_kernel inside a half-tile (128-byte rows, chunk ^ ((row>>1)&7) on the source and on the reads).
#define R3G_BAR() asm volatile("s_barrier" ::: "memory")
#define R3G_SB() __builtin_amdgcn_sched_barrier(0)

// SPLIT: deterministic split-K over two workgroups per tile (for launches whose 256x256 tiles fill less than half of the
// CUs, i.e. the N = 1024 residual GEMMs of the DiT: 120 tiles).  Workgroup s of a tile runs k-tiles [s nk/2, (s+1) nk/2),
// hands the accumulators of the OTHER wave row (tile rows 128 (1 - s) ...) to its partner through a workspace, waits for
// the partner's accumulators of its own wave row, adds them (two addends: the sum does not depend on the order) and runs
// the epilogue for its 128 rows with its four waves.  Flags carry the launch's epoch (never reset); data and flags move
// with device-scope (sc1) stores and loads, the flag after the wave's data stores have been acknowledged.  Both workgroups of a tile must be resident at once: the
// launcher only takes this path when the whole grid fits the CUs at one workgroup each.
template <int EPI, bool SPLIT = false>
__global__ __launch_bounds__(512) void gemm8_kernel(GemmArgs pa, GemmArgs pb, float* split_ws, unsigned* split_flags,
                                                    unsigned split_epoch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, MI = 8;
    constexpr int HALF = 16384, BUF = 4 * HALF;   // buffer: [A_0][A_1][W_0][W_1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int wg_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    // split: consecutive workgroups (same XCD, same L2) are the two halves of one tile
    const int wg_all = SPLIT ? wg_lin >> 1 : wg_lin;
    const int ksplit = SPLIT ? (wg_lin & 1) : 0;
    const int tiles_a = ((pa.N + BN - 1) / BN) * ((pa.M + BM - 1) / BM) * pa.batch;
    const bool second = wg_all >= tiles_a;
    typedef const char __attribute__((address_space(4))) * kernarg_ptr;
    kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kSecond = (sizeof(GemmArgs) + alignof(GemmArgs) - 1) / alignof(GemmArgs) * alignof(GemmArgs);
    const GemmArgs& p = *(const GemmArgs*)(const GemmArgs __attribute__((address_space(4)))*)(ka + (second ? kSecond : 0));
    (void)pb;
    const int wg = second ? wg_all - tiles_a : wg_all;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int rg = p.raster_group < 0 ? 4 : p.raster_group;
    const int GN = rg > 0 ? rg : tiles_n;
    const int rows_all = tiles_m * p.batch;
    const int per_group = rows_all * GN;
    const int group = wg / per_group;
    const int within = wg - group * per_group;
    const int gn_cur = (tiles_n - group * GN) < GN ? (tiles_n - group * GN) : GN;
    const int rowi = within / gn_cur;
    const int tn = group * GN + (within - rowi * gn_cur);
    const int batch = rowi / tiles_m;
    const int tm = rowi - batch * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wr = wid >> 2, wc = wid & 3;

    // ---- staging sources: this lane's 16-byte chunk of piece (wid*2 + i) of each half-tile, at k = 0
    const uint16_t* srcA[2][2];
    const uint16_t* srcW[2][2];
    {
        const uint16_t* A = p.A + (int64_t)batch * p.strideA;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = (wid * 2 + i) * 8 + (lane >> 3);        // row inside the half-tile
            const int kc = (lane & 7) ^ ((lr >> 1) & 7);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int ga = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
                int gw = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);
                ga = ga < p.M ? ga : p.M - 1;
                gw = gw < p.N ? gw : p.N - 1;
                const int k0 = SPLIT ? ksplit * (p.K / 2) : 0;
                srcA[h][i] = A + (int64_t)ga * p.lda + kc * 8 + k0;
                srcW[h][i] = p.W + (int64_t)gw * p.ldw + kc * 8 + k0;
            }
        }
    }
    char* const dst0 = smem + wid * 2048;   // + buffer * BUF + region * HALF + i * 1024
    auto stage_a = [&](int h, int t, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[h][i] + (int64_t)t * BK),
                                             (__attribute__((address_space(3))) void*)(dst0 + buf * BUF + h * HALF + i * 1024),
                                             16, 0, 0);
    };
    auto stage_w = [&](int h, int t, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[h][i] + (int64_t)t * BK),
                                             (__attribute__((address_space(3))) void*)(dst0 + buf * BUF + (2 + h) * HALF + i * 1024),
                                             16, 0, 0);
    };

    // ---- fragment read addresses (bytes inside a half-tile) for the two 32-wide k-steps
    const int sw = (lane >> 1) & 7;   // == ((row >> 1) & 7): the sub-tile / wave offsets are multiples of 16 rows
    int offA[2], offW[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        offA[kk] = (wr * 64 + (lane & 15)) * 128 + ((((kk << 2) + (lane >> 4)) ^ sw) << 4);
        offW[kk] = (wc * 32 + (lane & 15)) * 128 + ((((kk << 2) + (lane >> 4)) ^ sw) << 4);
    }

    f32x4 acc[4][MI];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], wf0[2][2], wf1[2][2];   // A sub-tile (shared by both m-halves), W_0 and W_1 sub-tiles

    auto read_a = [&](int h, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                af[i][kk] = *reinterpret_cast<const bf16x8*>(smem + buf * BUF + h * HALF + i * 2048 + offA[kk]);
    };
    auto read_w0 = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                wf0[j][kk] = *reinterpret_cast<const bf16x8*>(smem + buf * BUF + 2 * HALF + j * 2048 + offW[kk]);
    };
    auto read_w1 = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                wf1[j][kk] = *reinterpret_cast<const bf16x8*>(smem + buf * BUF + 3 * HALF + j * 2048 + offW[kk]);
    };
    // one accumulator quadrant x K = 64: 16 MFMAs on 8 distinct accumulators per k-step
    auto mma = [&](auto HA, auto HW) {
        constexpr int ha = decltype(HA)::value, hw = decltype(HW)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[hw * 2 + j][ha * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        hw ? wf1[j][kk] : wf0[j][kk], af[i][kk], acc[hw * 2 + j][ha * 4 + i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    const int nk = (SPLIT ? p.K / 2 : p.K) / BK;   // even, >= 2 (checked by the launcher)
    stage_w(0, 0, 0); stage_a(0, 0, 0); stage_w(1, 0, 0); stage_a(1, 0, 0);
    stage_w(0, 1, 1); stage_a(0, 1, 1); stage_w(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    R3G_BAR();
    if (wr == 1) R3G_BAR();   // the second wave row runs one barrier behind the first

    // four phases on the k-tile in buffer `b`; FULL: not the last iteration (every staging slot has a tile to load)
    auto four_phases = [&](auto B, auto FULL, const int t) {
        constexpr int b = decltype(B)::value;
        constexpr bool full = decltype(FULL)::value;
        // tiles staged by these phases: b == 0: A_1(t+1)->1, then W_0, A_0, W_1 of t+2 -> 0
        //                               b == 1: A_1(t+1)->0, then W_0, A_0, W_1 of t+2 -> 1   (t = the k-tile read here)
        // phase 1
        read_w0(b);
        R3G_SB();
        read_a(0, b);
        R3G_SB();
        if (full || b == 0) stage_a(1, t + 1, b ^ 1);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // W_0's four reads have returned: it is restaged next phase
        R3G_BAR();
        R3G_SB();
        mma(I0{}, I0{});
        R3G_SB();
        R3G_BAR();
        // phase 2
        read_w1(b);
        R3G_SB();
        if (full) stage_w(0, t + 2, b);
        R3G_BAR();
        R3G_SB();
        mma(I0{}, I1{});
        R3G_SB();
        R3G_BAR();
        // phase 3
        read_a(1, b);
        R3G_SB();
        if (full) stage_a(0, t + 2, b);
        R3G_BAR();
        R3G_SB();
        mma(I1{}, I1{});
        R3G_SB();
        R3G_BAR();
        // phase 4
        if (full) {
            stage_w(1, t + 2, b);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else if (b == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // last iteration: A_1 of the last k-tile
        }
        R3G_BAR();
        R3G_SB();
        mma(I1{}, I0{});
        R3G_SB();
        R3G_BAR();
    };
    int t = 0;
    for (; t + 2 < nk; t += 2) {
        four_phases(I0{}, std::true_type{}, t);
        four_phases(I1{}, std::true_type{}, t + 1);
    }
    four_phases(I0{}, std::false_type{}, t);
    four_phases(I1{}, std::false_type{}, t + 1);
    if (wr == 0) R3G_BAR();

    if constexpr (SPLIT) {
        // [tile][receiving half][wave column][register group of 4][lane] f32x4: 1 KiB per wave instruction.
        // The exchange uses DEVICE-scope accesses (sc1: stores write through to the memory side, loads do not trust a
        // possibly stale L2 line of another XCD) and no cache-wide fence: an agent-scope release fence writes back the
        // whole L2 (buffer_wbl2) -- measured +100 us per launch with 960 waves doing it.
        const size_t slot = ((size_t)wg_all * 2) * 4 * 32 * 64;
        if (wr != ksplit) {
            f32x4* dst = reinterpret_cast<f32x4*>(split_ws) + slot + ((size_t)(wr * 4 + wc) * 32) * 64 + lane;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + (size_t)(j * MI + i) * 64), "v"(acc[j][i]) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every store of this wave has been performed at device scope
            if (lane == 0) {
                unsigned* flag = &split_flags[(wg_all * 2 + wr) * 4 + wc];
                asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(flag), "v"(split_epoch) : "memory");
            }
            return;
        }
        {
            const unsigned* flag = &split_flags[(wg_all * 2 + wr) * 4 + wc];
            unsigned spins = 0, seen;
            do {
                asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(flag) : "memory");
                seen = __builtin_amdgcn_readfirstlane(seen);
                if (seen == split_epoch) break;
                __builtin_amdgcn_s_sleep(4);
            } while (++spins < (1u << 24));   // never hang the device: a lost partner shows up as a wrong result
            const f32x4* src = reinterpret_cast<const f32x4*>(split_ws) + slot + ((size_t)(wr * 4 + wc) * 32) * 64 + lane;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 t[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[i]) : "v"(src + (size_t)(j * MI + i) * 64) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[j][i] += t[i];
            }
        }
    }
    gemm_epilogue<EPI, MI, true>(p, acc, m0, n0, batch, wr, wc, lane, lds_ptr(smem + wid * (128 * 128)), p.wide_epilogue != 0);
}

// ------------------------------------------------------------------------------------------------------------
// FP8 (OCP e4m3) form of the phased kernel, for BASELINE.json's configs[3] ("config 4"): the same 128-byte LDS rows, LDS-DMA
// staging, phases and hazards -- a k-tile is 128 fp8 instead of 64 bf16, so GemmArgs describes the operands as matrices
// of byte PAIRS (K / 2 columns of 16 bits) and nothing in the staging changes.  A fragment is the 32 bytes of k-block
// (lane >> 4) of a row; v_mfma_scale_f32_16x16x128_f8f6f4 (twice the bf16 MFMA rate) has the C/D layout of the 16x16x32
// bf16 form, so the epilogues are shared.  Operands carry one fp32 scale per row (quantised by quant_fp8_rows_kernel in
// elem.hip); the hardware block scales are set to 1.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int EPI>
__global__ __launch_bounds__(512) void gemm8f_kernel(GemmArgs pa, GemmArgs pb, const float* __restrict__ scale_a,
                                                     const float* __restrict__ scale_w) {
    constexpr bool SPLIT = false;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, MI = 8;
    constexpr int HALF = 16384, BUF = 4 * HALF;   // buffer: [A_0][A_1][W_0][W_1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int wg_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    // split: consecutive workgroups (same XCD, same L2) are the two halves of one tile
    const int wg_all = wg_lin;
    const int tiles_a = ((pa.N + BN - 1) / BN) * ((pa.M + BM - 1) / BM) * pa.batch;
    const bool second = wg_all >= tiles_a;
    typedef const char __attribute__((address_space(4))) * kernarg_ptr;
    kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kSecond = (sizeof(GemmArgs) + alignof(GemmArgs) - 1) / alignof(GemmArgs) * alignof(GemmArgs);
    const GemmArgs& p = *(const GemmArgs*)(const GemmArgs __attribute__((address_space(4)))*)(ka + (second ? kSecond : 0));
    (void)pb;
    const int wg = second ? wg_all - tiles_a : wg_all;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int rg = p.raster_group < 0 ? 4 : p.raster_group;
    const int GN = rg > 0 ? rg : tiles_n;
    const int rows_all = tiles_m * p.batch;
    const int per_group = rows_all * GN;
    const int group = wg / per_group;
    const int within = wg - group * per_group;
    const int gn_cur = (tiles_n - group * GN) < GN ? (tiles_n - group * GN) : GN;
    const int rowi = within / gn_cur;
    const int tn = group * GN + (within - rowi * gn_cur);
    const int batch = rowi / tiles_m;
    const int tm = rowi - batch * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wr = wid >> 2, wc = wid & 3;

    // ---- staging sources: this lane's 16-byte chunk of piece (wid*2 + i) of each half-tile, at k = 0, as 32-bit element
    // offsets from the (wave-uniform) operand bases: the scaled MFMA is not tied to its accumulator registers, so this
    // kernel has no room for eight 64-bit pointers (spilled pointers cost a vmcnt(0) before every LDS-DMA)
    uint32_t offSA[2][2], offSW[2][2];
    const uint16_t* const baseA = p.A + (int64_t)batch * p.strideA;
    const uint16_t* const baseW = p.W;
    {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = (wid * 2 + i) * 8 + (lane >> 3);        // row inside the half-tile
            const int kc = (lane & 7) ^ ((lr >> 1) & 7);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int ga = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
                int gw = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);
                ga = ga < p.M ? ga : p.M - 1;
                gw = gw < p.N ? gw : p.N - 1;
                offSA[h][i] = (uint32_t)((int64_t)ga * p.lda + kc * 8);
                offSW[h][i] = (uint32_t)((int64_t)gw * p.ldw + kc * 8);
            }
        }
    }
    char* const dst0 = smem + wid * 2048;   // + buffer * BUF + region * HALF + i * 1024
    auto stage_a = [&](int h, int t, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseA + (int64_t)t * BK + offSA[h][i]),
                                             (__attribute__((address_space(3))) void*)(dst0 + buf * BUF + h * HALF + i * 1024),
                                             16, 0, 0);
    };
    auto stage_w = [&](int h, int t, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(baseW + (int64_t)t * BK + offSW[h][i]),
                                             (__attribute__((address_space(3))) void*)(dst0 + buf * BUF + (2 + h) * HALF + i * 1024),
                                             16, 0, 0);
    };

    // ---- fragment read addresses (bytes inside a half-tile) for the two 32-wide k-steps
    const int sw = (lane >> 1) & 7;   // == ((row >> 1) & 7): the sub-tile / wave offsets are multiples of 16 rows
    int offA[2], offW[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        offA[kk] = (wr * 64 + (lane & 15)) * 128 + ((((lane >> 4) * 2 + kk) ^ sw) << 4);
        offW[kk] = (wc * 32 + (lane & 15)) * 128 + ((((lane >> 4) * 2 + kk) ^ sw) << 4);
    }

    f32x4 acc[4][MI];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int unit_scale = 0x7F7F7F7F;   // e8m0 127 = 2^0 in every byte
    i32x8 af[4], wf0[2], wf1[2];   // A sub-tile (shared by both m-halves), W_0 and W_1 sub-tiles: 32 fp8 per lane each

    auto frag2 = [&](const char* p0, const char* p1) -> i32x8 {
        const i32x4 lo = *reinterpret_cast<const i32x4*>(p0), hi = *reinterpret_cast<const i32x4*>(p1);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto read_a = [&](int h, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* base = smem + buf * BUF + h * HALF + i * 2048;
            af[i] = frag2(base + offA[0], base + offA[1]);
        }
    };
    auto read_w0 = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const char* base = smem + buf * BUF + 2 * HALF + j * 2048;
            wf0[j] = frag2(base + offW[0], base + offW[1]);
        }
    };
    auto read_w1 = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const char* base = smem + buf * BUF + 3 * HALF + j * 2048;
            wf1[j] = frag2(base + offW[0], base + offW[1]);
        }
    };
    // one accumulator quadrant x K = 128 fp8: 8 block-scaled MFMAs (16x16x128), unit scales (the row scales are applied
    // to the accumulators before the epilogue)
    auto mma = [&](auto HA, auto HW) {
        constexpr int ha = decltype(HA)::value, hw = decltype(HW)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                // inline asm ties the destination to the accumulator: the builtin leaves them untied, hipcc then rotates the
                // 128 accumulator registers through fresh ones and spills the staging addresses (a vmcnt(0) per LDS-DMA)
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                             : "+v"(acc[hw * 2 + j][ha * 4 + i])
                             : "v"(hw ? wf1[j] : wf0[j]), "v"(af[i]), "v"(unit_scale));
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    const int nk = (SPLIT ? p.K / 2 : p.K) / BK;   // even, >= 2 (checked by the launcher)
    stage_w(0, 0, 0); stage_a(0, 0, 0); stage_w(1, 0, 0); stage_a(1, 0, 0);
    stage_w(0, 1, 1); stage_a(0, 1, 1); stage_w(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    R3G_BAR();
    if (wr == 1) R3G_BAR();   // the second wave row runs one barrier behind the first

    // four phases on the k-tile in buffer `b`; FULL: not the last iteration (every staging slot has a tile to load)
    auto four_phases = [&](auto B, auto FULL, const int t) {
        constexpr int b = decltype(B)::value;
        constexpr bool full = decltype(FULL)::value;
        // tiles staged by these phases: b == 0: A_1(t+1)->1, then W_0, A_0, W_1 of t+2 -> 0
        //                               b == 1: A_1(t+1)->0, then W_0, A_0, W_1 of t+2 -> 1   (t = the k-tile read here)
        // phase 1
        read_w0(b);
        R3G_SB();
        read_a(0, b);
        R3G_SB();
        if (full || b == 0) stage_a(1, t + 1, b ^ 1);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // W_0's four reads have returned: it is restaged next phase
        R3G_BAR();
        R3G_SB();
        mma(I0{}, I0{});
        R3G_SB();
        R3G_BAR();
        // phase 2
        read_w1(b);
        R3G_SB();
        if (full) stage_w(0, t + 2, b);
        R3G_BAR();
        R3G_SB();
        mma(I0{}, I1{});
        R3G_SB();
        R3G_BAR();
        // phase 3
        read_a(1, b);
        R3G_SB();
        if (full) stage_a(0, t + 2, b);
        R3G_BAR();
        R3G_SB();
        mma(I1{}, I1{});
        R3G_SB();
        R3G_BAR();
        // phase 4
        if (full) {
            stage_w(1, t + 2, b);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else if (b == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // last iteration: A_1 of the last k-tile
        }
        R3G_BAR();
        R3G_SB();
        mma(I1{}, I0{});
        R3G_SB();
        R3G_BAR();
    };
    int t = 0;
    for (; t + 2 < nk; t += 2) {
        four_phases(I0{}, std::true_type{}, t);
        four_phases(I1{}, std::true_type{}, t + 1);
    }
    four_phases(I0{}, std::false_type{}, t);
    four_phases(I1{}, std::false_type{}, t + 1);
    if (wr == 0) R3G_BAR();

    // fp8 operands carry one fp32 scale per row of A and per row of W: C = (sa[m] sw[n]) * sum_k a8 w8
    {
        f32x4 swv[4];
        float sav[MI];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + ((lane >> 4) << 2);
            swv[j] = *reinterpret_cast<const f32x4*>(scale_w + (n < p.N ? n : p.N - 4));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wr * 128 + i * 16 + (lane & 15);
            sav[i] = scale_a[(int64_t)batch * p.M + (m < p.M ? m : p.M - 1)];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[j][i] *= swv[j] * sav[i];
    }
    gemm_epilogue<EPI, MI, true>(p, acc, m0, n0, batch, wr, wc, lane, lds_ptr(smem + wid * (128 * 128)), p.wide_epilogue != 0);
}


template <int EPI>
hipError_t launch_gemm8_fp8(const GemmArgs& p, const float* scale_a, const float* scale_w, hipStream_t s) {
    const int tiles = ((p.N + 255) / 256) * ((p.M + 255) / 256) * p.batch;
    const size_t lds = 131072;
    auto k = gemm8f_kernel<EPI>;
    static bool done = false;
    if (!done) {
        done = true;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    GemmArgs none{};
    hipLaunchKernelGGL(k, dim3(tiles), dim3(512), lds, s, p, none, scale_a, scale_w);
    return hipGetLastError();
}

#ifdef R3G_GEMM_STAMPS
// Measurement build only (tools/gemm_stamps.py; never in libr3g.so): s_memtime sums per section of the persistent kernel's tile
// loop, per wave of workgroup 8 (the second workgroup of XCD 0).
__device__ unsigned long long g_gemm_stamps[8][8];
#define R3G_STAMP_DECL unsigned long long st_prev = __builtin_amdgcn_s_memtime(), st_acc[6] = {0, 0, 0, 0, 0, 0}
#define R3G_STAMP(k) do { const unsigned long long st_now = __builtin_amdgcn_s_memtime(); st_acc[k] += st_now - st_prev; st_prev = st_now; } while (0)
#define R3G_STAMP_OUT do { if (blockIdx.x == 8 && lane == 0) { for (int k_ = 0; k_ < 6; ++k_) g_gemm_stamps[wid][k_] = st_acc[k_]; } } while (0)
#else
#define R3G_STAMP_DECL
#define R3G_STAMP(k)
#define R3G_STAMP_OUT
#endif
// ------------------------------------------------------------------------------------------------------------
// Persistent form of the phased kernel: the grid is one workgroup per CU (balanced over the dispatch rounds) and every
// workgroup walks tiles w, w + G, w + 2G, ...  Between two tiles it issues the LDS-DMA of the NEXT tile's first k-tile,
// runs the epilogue of the tile it just finished out of a separate 32 KiB of LDS scratch (4 KiB per wave, 32 rows per
// pass), issues the second k-tile and enters the k-loop with the usual counted wait (vmcnt(6): everything older than
// the six newest pieces, so it is exact whatever the epilogue issued).  The prologue latency hides under the epilogue and
// there is no workgroup dispatch between tiles.  s_memtime stamps (profiles/r02_gemm_persistent.md): at K = 1024 a tile
// is k-loop 37 k cycles (93 % of the MFMA issue rate) + epilogue 15-26 k + prologue 6 k in the one-tile-per-workgroup
// kernel; this form removes most of the prologue and part of the epilogue wait (bf16 outputs -2 .. -9 %).  The fp32
// read-modify-write epilogue is bound by HBM (1.3 GB per launch at ~3.5 TB/s) and stays on gemm8_kernel.
// Same k-order and MFMA shape as gemm8_kernel: bit-identical results.
// (Round 4's EARLY variant -- the wait for the next tile's first k-tile inside the epilogue -- measured "no effect" and left the
// library in round 6 together with its option.)
// SLICED (round 6, bf16-output and fused-QKV epilogues): the epilogue's LDS transpose runs in 64-row passes through the wave's own
// staging slices of k-tile buffer 1 (gemm_epilogue's SLICE) instead of 32-row passes through 4 KiB of extra scratch: two passes
// per tile instead of four, V^T rows leave as whole 128-byte lines.  Buffer 1 is idle then: its last reader passed the k-loop's
// closing barriers, the next tile's k-tile 0 goes to buffer 0, and k-tile 1 is staged after the epilogue -- by every wave into
// its OWN slices only, behind its own LDS reads (lgkmcnt is waited for before the stores those reads feed), so there is no
// cross-wave hazard and no extra barrier.
// EPI2 (round 6): the epilogue of the SECOND problem's tiles when two problems with different epilogues share the launch -- a
// single block's linear1 = [fused QKV | MLP-in + GELU] over the same rows is 1 416 + 1 888 tiles = 12.9 rounds of 256 CUs in ONE
// grid where the two launches took 6 + 8 (their last rounds 53 % / 92 % full).  The tile's problem is wave-uniform.
template <int EPI, bool SLICED = false, int EPI2 = EPI>
__global__ __launch_bounds__(512) void gemm8p_kernel(GemmArgs pa, GemmArgs pb, int total_tiles, int rounds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, MI = 8;
    constexpr int HALF = 16384, BUF = 4 * HALF;   // buffer: [A_0][A_1][W_0][W_1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    // Tile walk (round 5): an XCD owns ONE contiguous range of the rasterised tile order -- `rounds` tiles per workgroup it hosts --
    // and its workgroups step through that range side by side (tile = first + local id + step x workgroups of the XCD): the
    // range is rows x the 4 tile columns of a raster group, and consecutive steps of an XCD touch neighbouring A rows; rounds 2-4
    // gave an XCD a different slice of the order at every step (tile = first + step x grid).  Measured on one box (option
    // "gemm_xcd_walk" 0 / 1, tools/bench_gemm.py): geo c_fc 1 109 -> 1 045 us, 30080 x 3072 bf16 175 -> 172, MLP-in equal.  The
    // L2 <-> fabric counters do NOT move (MLP-in FETCH_SIZE 197 -> 201 MB raw: the W panel stayed resident under the old walk
    // too, the bytes are the A rows once per raster group either way) -- the gain is in WHEN the rows are touched, not how often.
    // rounds == 0 (option "gemm_xcd_walk" 0): rounds 2-4's walk, for A/B.
    const int xcd_first_wg = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int xcd_wgs = rounds ? q + (xcd < r ? 1 : 0) : nwg;
    const int xcd_end = rounds ? min(total_tiles, (xcd_first_wg + xcd_wgs) * rounds) : total_tiles;
    const int wg_first = (rounds ? min(total_tiles, xcd_first_wg * rounds) : xcd_first_wg) + (orig >> 3);   // < xcd_end (launch_gemm8p: grid = ceil(tiles / rounds))
    const int tiles_a = ((pa.N + BN - 1) / BN) * ((pa.M + BM - 1) / BM) * pa.batch;
    typedef const char __attribute__((address_space(4))) * kernarg_ptr;
    kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kSecond = (sizeof(GemmArgs) + alignof(GemmArgs) - 1) / alignof(GemmArgs) * alignof(GemmArgs);
    (void)pb;
    const int wr = wid >> 2, wc = wid & 3;

    struct Tile { int second, batch, m0, n0; };
    auto args_of = [&](int second) -> const GemmArgs& {
        return *(const GemmArgs*)(const GemmArgs __attribute__((address_space(4)))*)(ka + (second ? kSecond : 0));
    };
    auto locate = [&](int wg_all) -> Tile {
        Tile t;
        t.second = wg_all >= tiles_a ? 1 : 0;
        const GemmArgs& p = args_of(t.second);
        const int wg = t.second ? wg_all - tiles_a : wg_all;
        const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
        const int rg = p.raster_group < 0 ? 4 : p.raster_group;
        const int GN = rg > 0 ? rg : tiles_n;
        const int rows_all = tiles_m * p.batch;
        const int per_group = rows_all * GN;
        const int group = wg / per_group;
        const int within = wg - group * per_group;
        const int gn_cur = (tiles_n - group * GN) < GN ? (tiles_n - group * GN) : GN;
        const int rowi = within / gn_cur;
        const int tn = group * GN + (within - rowi * gn_cur);
        t.batch = rowi / tiles_m;
        t.m0 = (rowi - t.batch * tiles_m) * BM;
        t.n0 = tn * BN;
        return t;
    };

    const uint16_t* srcA[2][2];
    const uint16_t* srcW[2][2];
    // Round 6: INTERIOR tiles take their sources as four scalar bases + the lane's part of the address (32 bits, rebuilt from an
    // opaque copy of the lane number so that it is not kept alive across the k-loop): 4 multiplies and 2 vector instructions per
    // pointer where the general form (row clamp, 64-bit row x stride product per pointer: 24 quarter-rate multiplies) took ~5,
    // twice per tile.
    auto set_sources = [&](const Tile& tl) {
        const GemmArgs& p = args_of(tl.second);
        const uint16_t* A = p.A + (int64_t)tl.batch * p.strideA;
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        // (not in the ln_3-fold kernel: its epilogue is at the register limit and the second path costs it three spilled bias vectors)
        if (EPI != EPI_BF16_GELU_ERF_LNF && tl.m0 + BM <= p.M && tl.n0 + BN <= p.N) {   // wave-uniform
            const uint32_t lda2 = (uint32_t)p.lda * 2u, ldw2 = (uint32_t)p.ldw * 2u;   // bytes per row (< 2^24: K is an int)
            uint32_t loA[2], loW[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int lr = (wid * 2 + i) * 8 + (lane_s >> 3);
                const int kc = (lane_s & 7) ^ ((lr >> 1) & 7);
                loA[i] = (uint32_t)((lr >> 6) * 128 + (lr & 63)) * lda2 + (uint32_t)kc * 16u;
                loW[i] = (uint32_t)((lr >> 5) * 64 + (lr & 31)) * ldw2 + (uint32_t)kc * 16u;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const char* ba = reinterpret_cast<const char*>(A + (int64_t)(tl.m0 + h * 64) * p.lda);
                const char* bw = reinterpret_cast<const char*>(p.W + (int64_t)(tl.n0 + h * 32) * p.ldw);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    srcA[h][i] = reinterpret_cast<const uint16_t*>(ba + loA[i]);
                    srcW[h][i] = reinterpret_cast<const uint16_t*>(bw + loW[i]);
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = (wid * 2 + i) * 8 + (lane_s >> 3);        // row inside the half-tile
            const int kc = (lane_s & 7) ^ ((lr >> 1) & 7);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int ga = tl.m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
                int gw = tl.n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);
                ga = ga < p.M ? ga : p.M - 1;
                gw = gw < p.N ? gw : p.N - 1;
                srcA[h][i] = A + (int64_t)ga * p.lda + kc * 8;
                srcW[h][i] = p.W + (int64_t)gw * p.ldw + kc * 8;
            }
        }
    };
    char* const dst0 = smem + wid * 2048;   // + buffer * BUF + region * HALF + i * 1024
    auto stage_a = [&](int h, int t, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[h][i] + (int64_t)t * BK),
                                             (__attribute__((address_space(3))) void*)(dst0 + buf * BUF + h * HALF + i * 1024),
                                             16, 0, 0);
    };
    auto stage_w = [&](int h, int t, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[h][i] + (int64_t)t * BK),
                                             (__attribute__((address_space(3))) void*)(dst0 + buf * BUF + (2 + h) * HALF + i * 1024),
                                             16, 0, 0);
    };
    const int sw = (lane >> 1) & 7;
    int offA[2], offW[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        offA[kk] = (wr * 64 + (lane & 15)) * 128 + ((((kk << 2) + (lane >> 4)) ^ sw) << 4);
        offW[kk] = (wc * 32 + (lane & 15)) * 128 + ((((kk << 2) + (lane >> 4)) ^ sw) << 4);
    }
    f32x4 acc[4][MI];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], wf0[2][2], wf1[2][2];
    auto read_a = [&](int h, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                af[i][kk] = *reinterpret_cast<const bf16x8*>(smem + buf * BUF + h * HALF + i * 2048 + offA[kk]);
    };
    auto read_w0 = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                wf0[j][kk] = *reinterpret_cast<const bf16x8*>(smem + buf * BUF + 2 * HALF + j * 2048 + offW[kk]);
    };
    auto read_w1 = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                wf1[j][kk] = *reinterpret_cast<const bf16x8*>(smem + buf * BUF + 3 * HALF + j * 2048 + offW[kk]);
    };
    auto mma = [&](auto HA, auto HW) {
        constexpr int ha = decltype(HA)::value, hw = decltype(HW)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[hw * 2 + j][ha * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        hw ? wf1[j][kk] : wf0[j][kk], af[i][kk], acc[hw * 2 + j][ha * 4 + i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // LAST: the tile's last pair of k-tiles, nothing left to stage
    auto four_phases = [&](auto B, auto LAST, const int t) {
        constexpr int b = decltype(B)::value;
        constexpr bool last = decltype(LAST)::value;
        read_w0(b);
        R3G_SB();
        read_a(0, b);
        R3G_SB();
        if (!last || b == 0) stage_a(1, t + 1, b ^ 1);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        R3G_BAR();
        R3G_SB();
        mma(I0{}, I0{});
        R3G_SB();
        R3G_BAR();
        read_w1(b);
        R3G_SB();
        if (!last) stage_w(0, t + 2, b);
        R3G_BAR();
        R3G_SB();
        mma(I0{}, I1{});
        R3G_SB();
        R3G_BAR();
        read_a(1, b);
        R3G_SB();
        if (!last) stage_a(0, t + 2, b);
        R3G_BAR();
        R3G_SB();
        mma(I1{}, I1{});
        R3G_SB();
        R3G_BAR();
        if (!last) {
            stage_w(1, t + 2, b);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else if (b == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // A_1 of the last k-tile
        }
        R3G_BAR();
        R3G_SB();
        mma(I1{}, I0{});
        R3G_SB();
        R3G_BAR();
    };
    using T = std::true_type;
    using F = std::false_type;

    auto locate_u = [&](int index, Tile& tl, int& nk) {
        tl = locate(index);
        tl.second = __builtin_amdgcn_readfirstlane(tl.second);
        tl.batch = __builtin_amdgcn_readfirstlane(tl.batch);
        tl.m0 = __builtin_amdgcn_readfirstlane(tl.m0);
        tl.n0 = __builtin_amdgcn_readfirstlane(tl.n0);
        nk = args_of(tl.second).K / BK;   // even, >= 4 (checked by the launcher)
    };
    Tile cur;
    int nk;
    int my = wg_first;            // < total_tiles: the grid is never larger than the tile count
    locate_u(my, cur, nk);
    set_sources(cur);
    stage_w(0, 0, 0); stage_a(0, 0, 0); stage_w(1, 0, 0); stage_a(1, 0, 0);
    R3G_STAMP_DECL;
    for (;;) {
        // k-tile 0 is on its way (and, after the first tile, the previous tile's stores); k-tile 1 follows as in gemm8_kernel.
        // vmcnt(6) retires everything older than these six pieces -- exact whatever the epilogue issued.
        R3G_STAMP(5);   // (set_sources again, loop overhead; the first tile: the prologue)
        stage_w(0, 1, 1); stage_a(0, 1, 1); stage_w(1, 1, 1);
        // vmcnt(6) as the BUILTIN (0x0F76 = vmcnt 6, expcnt / lgkmcnt unconstrained), not as inline asm: the compiler's own wait
        // insertion cannot see into inline asm, so it believed the previous tile's bias loads (issued on one path of the
        // epilogue, consumed on another) still pending at the loop header and put an `s_waitcnt vmcnt(0)` in front of the first
        // write to their registers (the clearing of the accumulators) -- AFTER k-tile 1 was issued: every tile of every
        // persistent launch of rounds 2-5 drained its own prefetch there.
        __builtin_amdgcn_s_waitcnt(0x0F76);
        R3G_BAR();
        if (wr == 1) R3G_BAR();   // the second wave row runs one barrier behind the first
        R3G_STAMP(0);   // staging of k-tile 1, wait for k-tile 0 (and the previous tile's stores), opening barriers
        int t = 0;
        for (; t + 2 < nk; t += 2) {
            four_phases(I0{}, F{}, t);
            four_phases(I1{}, F{}, t + 1);
        }
        four_phases(I0{}, T{}, t);
        four_phases(I1{}, T{}, t + 1);
        if (wr == 0) R3G_BAR();
        R3G_STAMP(1);   // k-loop

        const Tile done = cur;
        // Round 5: the lane number the epilogue works with is made opaque once per tile.  Everything the epilogue derives from it
        // (store addresses, row maps, LDS offsets) is invariant across the tiles of a workgroup; left visible, the compiler
        // hoists it all out of the tile loop and spills it around the k-loop (2-3 registers for the bf16 epilogues, 13 for the
        // fp32 residual one, 32 for the fused QKV one -- round 4 read those spills as a cost of the persistent form itself).
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        // the finished tile's bias goes first: a wait for it must not also wait for the LDS-DMA issued below (vmcnt is in
        // order), and its latency hides under the staging (s_memtime: epilogue 10.4 k -> 8.5 k cycles)
        f32x4 bias_pre[4];
        {
            const GemmArgs& pp = args_of(done.second);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = done.n0 + wc * 64 + ((lane_e >> 4) << 2) + j * 16;
                bias_pre[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (pp.bias) bias_pre[j] = *reinterpret_cast<const f32x4*>(pp.bias + (n < pp.N ? n : pp.N - 4));
            }
        }
        my += xcd_wgs;
        const bool more = my < xcd_end;
        if (more) {
            // the next tile's first k-tile goes out before the epilogue and lands under it
            locate_u(my, cur, nk);
            set_sources(cur);
            stage_w(0, 0, 0); stage_a(0, 0, 0); stage_w(1, 0, 0); stage_a(1, 0, 0);
        }
        // (Round 6, measured and NOT kept: hipcc puts an s_waitcnt vmcnt(0) in front of the epilogue's first use of the bias -- it
        // merges "eight newer operations" (a next tile was staged) with "none" (the last tile) -- so every tile waits there for its
        // successor's k-tile 0.  Staging unconditionally + vmcnt(8) as a builtin removes the wait and changes nothing: the k-tile
        // has landed by then (tools/gemm_stamps.py: epilogue 4.4 k ticks either way), while the fused-QKV kernels spill.)
        R3G_STAMP(2);   // bias loads, next tile located, its sources, k-tile 0 issued
        {
            const GemmArgs& pp = args_of(done.second);
            auto run = [&](auto E) __attribute__((always_inline)) {
                constexpr int epi = decltype(E)::value;
                constexpr bool kPre = epi == EPI_BF16 || epi == EPI_BF16_GELU_TANH || epi == EPI_BF16_GELU_ERF || epi == EPI_BF16_GELU_ERF_LNF;
                constexpr bool kSl = SLICED && (kPre || epi == EPI_QKV);
                if constexpr (kSl)
                    // (the launcher takes the SLICED instantiation only for launches with the wide epilogue: a literal `true` lets
                    // the compiler drop the direct-store copy of the fused-QKV epilogue from this kernel)
                    gemm_epilogue<epi, MI, true, 4, false, HALF>(pp, acc, done.m0, done.n0, done.batch, wr, wc, lane_e,
                                                                 lds_ptr(smem + BUF + wid * 2048), true, kPre ? bias_pre : nullptr);
                else
                    gemm_epilogue<epi, MI, true, 2, false>(pp, acc, done.m0, done.n0, done.batch, wr, wc, lane_e,
                                                           lds_ptr(smem + 2 * BUF + wid * 4096), pp.wide_epilogue != 0, kPre ? bias_pre : nullptr);
            };
            if constexpr (EPI2 != EPI) {
                if (done.second) run(std::integral_constant<int, EPI2>{});
                else run(std::integral_constant<int, EPI>{});
            } else {
                run(std::integral_constant<int, EPI>{});
            }
        }
        R3G_STAMP(3);   // epilogue (issue; the stores are waited for under stamp 0 of the next tile)
        if (!more) break;
        // the staging pointers are recomputed rather than kept alive across the epilogue (16 registers it needs); the
        // empty asm keeps the compiler from merging the two computations
        asm volatile("" : "+s"(cur.m0), "+s"(cur.n0), "+s"(cur.batch));
        set_sources(cur);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    R3G_STAMP_OUT;
}

#ifdef R3G_GEMM_STAMPS
}  // anonymous namespace
}  // namespace r3g
extern "C" __attribute__((visibility("default"))) int r3g_debug_gemm_stamps(unsigned long long* out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(r3g::g_gemm_stamps), sizeof(unsigned long long) * 64);
}
namespace r3g {
namespace {
#endif

bool g_gemm_xcd_walk = true;   // persistent kernel: an XCD walks one contiguous range of the tile order (round 5) | 0: rounds 2-4's walk
bool g_gemm_persistent_qkv = true;   // fused QKV launches with more 256x256 tiles than CUs (the double blocks' img + txt pair) on the persistent phased kernel: round 4's 32-row passes cost +17 ms per object, round 6's 64-row passes through the idle k-tile buffer -4 ms (profiles/r06_ab.md)
bool g_gemm_epi_slices = true;   // persistent kernel, bf16 / fused-QKV epilogues: 64-row passes through the wave's slices of k-tile buffer 1 (round 6) | 0: 32-row passes
bool g_gemm_mixed = true;        // a single block's [fused QKV | MLP-in + GELU] as ONE persistent launch (round 6) | 0: two launches

// EPI2 != EPI: p2 is a problem with another epilogue (gemm8p_kernel's EPI2)
template <int EPI, int EPI2 = EPI>
hipError_t launch_gemm8p(const GemmArgs& p, const GemmArgs& p2, int num_cu, hipStream_t s) {
    int tiles = ((p.N + 255) / 256) * ((p.M + 255) / 256) * p.batch;
    if (p2.M > 0) tiles += ((p2.N + 255) / 256) * ((p2.M + 255) / 256) * p2.batch;
    const int rounds = (tiles + num_cu - 1) / num_cu;
    const int grid = (tiles + rounds - 1) / rounds;    // every workgroup gets `rounds` tiles (the last ones one fewer)
    const size_t lds = 163840;
    constexpr bool kSliceable = EPI == EPI_BF16 || EPI == EPI_BF16_GELU_TANH || EPI == EPI_BF16_GELU_ERF || EPI == EPI_QKV || EPI == EPI_BF16_GELU_ERF_LNF;
    auto k = gemm8p_kernel<EPI, false, EPI2>;
    auto ks = gemm8p_kernel<EPI, kSliceable, EPI2>;
    static int state = 0;   // 0 unknown, 1 usable, -1 the device refuses 160 KiB of LDS
    if (state == 0) {
        state = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 1 : -1;
        if (state > 0 && kSliceable)
            state = hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 1 : -1;
    }
    if (state < 0) { (void)hipGetLastError(); return hipErrorNotSupported; }
    const int walk = g_gemm_xcd_walk ? rounds : 0;
    if (kSliceable && g_gemm_epi_slices && p.wide_epilogue) { hipLaunchKernelGGL(ks, dim3(grid), dim3(512), lds, s, p, p2, tiles, walk); }
    else { hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, p, p2, tiles, walk); }
    return hipGetLastError();
}

template <int EPI>
hipError_t launch_gemm8(const GemmArgs& p, const GemmArgs& p2, hipStream_t s) {
    int tiles = ((p.N + 255) / 256) * ((p.M + 255) / 256) * p.batch;
    if (p2.M > 0) tiles += ((p2.N + 255) / 256) * ((p2.M + 255) / 256) * p2.batch;
    const size_t lds = 131072;
    auto k = gemm8_kernel<EPI, false>;
    static bool done = false;
    if (!done) {
        done = true;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(512), lds, s, p, p2, (float*)nullptr, (unsigned*)nullptr, 0u);
    return hipGetLastError();
}

// split-K workspace: one slot of 256 KiB + 8 flags per tile, for at most kSplitMaxTiles tiles, per (device, stream): two
// contexts / streams that use the option at the same time get workspaces (and epochs) of their own
constexpr int kSplitMaxTiles = 128;
struct SplitWs { float* ws = nullptr; unsigned* flags = nullptr; unsigned epoch = 0; };
static std::mutex g_split_mutex;
static std::map<std::pair<int, hipStream_t>, SplitWs> g_split;

template <int EPI>
hipError_t launch_gemm8_split(const GemmArgs& p, const GemmArgs& p2, hipStream_t s) {
    int tiles = ((p.N + 255) / 256) * ((p.M + 255) / 256) * p.batch;
    if (p2.M > 0) tiles += ((p2.N + 255) / 256) * ((p2.M + 255) / 256) * p2.batch;
    if (tiles > kSplitMaxTiles) return hipErrorNotSupported;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    SplitWs w;
    {
        std::lock_guard<std::mutex> lock(g_split_mutex);
        SplitWs& slot = g_split[std::make_pair(dev, s)];
        if (!slot.ws) {
            e = hipMalloc((void**)&slot.ws, (size_t)kSplitMaxTiles * 256 * 256 * 4);
            if (e != hipSuccess) return e;
            e = hipMalloc((void**)&slot.flags, (size_t)kSplitMaxTiles * 8 * 4);
            if (e != hipSuccess) return e;
            e = hipMemset(slot.flags, 0, (size_t)kSplitMaxTiles * 8 * 4);
            if (e != hipSuccess) return e;
        }
        if (++slot.epoch == 0u) slot.epoch = 1u;   // 0 is the flags' initial value
        w = slot;
    }
    const size_t lds = 131072;
    auto k = gemm8_kernel<EPI, true>;
    static bool done = false;
    if (!done) {
        done = true;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(k, dim3(2 * tiles), dim3(512), lds, s, p, p2, w.ws, w.flags, w.epoch);
    return hipGetLastError();
}

template <int EPI>
hipError_t launch_deep(const GemmArgs& p, int batch, hipStream_t s) {
    const int tiles = ((p.N + 255) / 256) * ((p.M + 255) / 256) * batch;
    const size_t lds = 4 * 2 * 256 * 32 * 2;  // 128 KiB
    auto k = gemm_deep_kernel<EPI>;
    static bool done = false;
    if (!done) {
        done = true;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(512), lds, s, p);
    return hipGetLastError();
}

template <int EPI, int NW, int BIG>
hipError_t launch_cfg(const GemmArgs& p, const GemmArgs& p2, bool glds, hipStream_t s) {
    constexpr int BM = BIG ? 256 : 128, BN = BIG == 1 ? 256 : 128;
    int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM) * p.batch;
    if (p2.M > 0) tiles += ((p2.N + BN - 1) / BN) * ((p2.M + BM - 1) / BM) * p2.batch;
    const size_t lds = (size_t)2 * (BM + BN) * BK * 2;
    auto kt = gemm_kernel<EPI, true, NW, BIG>;
    auto kf = gemm_kernel<EPI, false, NW, BIG>;
    if (lds > 64 * 1024) {
        static bool done = false;
        if (!done) {
            done = true;
            (void)hipFuncSetAttribute((const void*)kt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
    }
    if (glds) {
        hipLaunchKernelGGL(kt, dim3(tiles), dim3(NW * 64), lds, s, p, p2);
    } else {
        hipLaunchKernelGGL(kf, dim3(tiles), dim3(NW * 64), lds, s, p, p2);
    }
    return hipGetLastError();
}

// ---- split-K of the 128x128 kernel (round 5): the texture UNets' 3 x 3 convolutions at the coarse levels are GEMMs of a few
// hundred rows over K = 9 Cin up to 23 040 -- 10 to 240 tiles on a chip that holds 512 of these workgroups, each walking 180 to
// 360 k-steps alone (118 / 235 us at 64 ... 384 rows x 1280 columns, profiles/r05_gemm_splitk.md).  S slices of K run as the
// S batches of ONE ordinary EPI_F32 launch (strideA = strideW = K / S along the rows' k axis) into the caller's workspace
// [S][M][N]; this kernel adds the slices in the order s = 0 .. S-1 and applies bias / gate / residual as the fused epilogues do.
// Deterministic: no atomics, the order of the additions is fixed by (M, N, K) alone.
template <bool RESID>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, int64_t MN, int N,
                                                            const float* __restrict__ bias, const float* __restrict__ gate,
                                                            float* __restrict__ C, int64_t ldc) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= MN) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(ws + i);
    int s = 1;
    for (; s + 4 <= S; s += 4) {
        f32x4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = *reinterpret_cast<const f32x4*>(ws + (int64_t)(s + k) * MN + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) v += t[k];
    }
    for (; s < S; ++s) v += *reinterpret_cast<const f32x4*>(ws + (int64_t)s * MN + i);
    const int64_t m = i / N;
    const int n = (int)(i - m * N);
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
    float* dst = C + m * ldc + n;
    if (RESID) {
        if (gate) v = *reinterpret_cast<const f32x4*>(gate + n) * v;
        v = *reinterpret_cast<const f32x4*>(dst) + v;
    }
    *reinterpret_cast<f32x4*>(dst) = v;
}

bool g_gemm_splitk128 = true;
constexpr int kSplit128Slots = 512;    // 128x128 workgroups the chip holds at once (2 per CU)
constexpr int kSplit128MinSteps = 8;   // k-steps of 64 a slice keeps at least

// number of slices: the largest divisor S of K / 64 with S * tiles <= 512 slots and >= 8 k-steps per slice (1: no split)
static int splitk128_factor(int M, int N, int K, int num_cu) {
    if (K < 2048 || K % 64) return 1;
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    const long slots = (long)kSplit128Slots * num_cu / 256;
    if (tiles * 2 > slots) return 1;
    const int nk = K / 64;
    int best = 1;
    for (int S = 2; S <= 64 && S * tiles <= slots && nk / S >= kSplit128MinSteps; ++S)
        if (nk % S == 0) best = S;
    // two slices save half of a short k-loop and pay a pass over the output for it: measured a loss at K = 2560, a gain from 5760
    if (best == 2 && K < 4096) best = 1;
    return best;
}

bool g_conv_implicit = true;

template <int EPI>
hipError_t launch_conv(const GemmArgs& p, hipStream_t s) {
    const int tiles = ((p.N + 127) / 128) * ((p.M + 127) / 128) * p.batch;
    const size_t lds = (size_t)2 * (128 + 128) * BK * 2;
    auto k = conv_gemm_kernel<EPI>;
    static bool done = false;
    if (!done) {
        done = true;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(512), lds, s, p);
    return hipGetLastError();
}

static int g_gemm_waves = 0;  // 0 = automatic tile choice
int g_gemm_raster = -1;
int g_gemm_auto_rule = 2, g_num_cu = 256;
bool g_gemm_splitk = false;      // deterministic split-K for deep-K residual GEMMs on under-filled grids (measured: no gain)
int g_gemm_persistent_resid = 0;   // experiment switch (r3g_set_option "gemm_persistent_resid")
bool g_gemm_persistent = true;   // phased kernel as a persistent grid when there are more 256x256 tiles than CUs
bool g_gemm_phased = true;   // 256x256 tiles run the phased (counted-vmcnt) kernel instead of the two-stage one
bool g_gemm_wide_epilogue = true;
bool g_gemm_gelu_pk = true;   // GELU epilogues in packed fp16 (gemm_common.h)

template <int EPI>
hipError_t launch_epi(const GemmArgs& p, const GemmArgs& p2, bool glds, hipStream_t s) {
    int waves = g_gemm_waves;
    const int batch = p.batch;
    if (waves == 0) {
        // measured on MI355X (profiles/r01_gemm_variants.md): 256x256 tiles (16 waves, half the L2->LDS traffic per
        // flop) win for long K or very many tiles; 128x128 with 8 waves wins slightly for N <= 1024, 4 waves otherwise
        long t256 = (long)((p.N + 255) / 256) * ((p.M + 255) / 256) * batch;
        if (p2.M > 0) t256 += (long)((p2.N + 255) / 256) * ((p2.M + 255) / 256) * p2.batch;
        const long rounds = (t256 + g_num_cu - 1) / g_num_cu;
        const bool fills = t256 * 10 >= rounds * g_num_cu * 9;   // the last dispatch round of 256x256 tiles is >= 90 % full
        if (g_gemm_auto_rule == 0) {
            if (p2.M == 0 && p.N % 256 == 0 && t256 >= 128 && (p.K >= 2048 || t256 >= 2048)) waves = 9;
            else waves = p.N <= 1024 ? 8 : 4;
        } else {
            // A CU keeps ~20 B/clk of operand loads in flight (L1 miss queue x L2 latency, profiles/r01_pmc_gemm_counters.md):
            // the 256x256 tile needs half the bytes per flop of the 128x128 one and wins wherever its coarser grid
            // still fills the machine; otherwise 128x128 with 8 waves (4 per SIMD at two workgroups per CU).
            // rule 2 (round 3): also any launch whose 256x256 grid fills >= 85 % of the CUs in full rounds -- the N = 1024
            // residual GEMMs of the DiT once two or more objects share a launch (236 tiles; one object: 120, stays 128x128)
            const bool wide_enough = p.N >= 4096 || (g_gemm_auto_rule >= 2 && t256 * 100 >= (long)g_num_cu * 85);
            if ((p2.M == 0 || p2.N % 256 == 0) && p.N % 256 == 0 && t256 >= 128 &&
                (p.K >= 2048 || t256 >= 2048 || (wide_enough && fills)))
                waves = g_gemm_phased ? 11 : 9;   // phased 256x256 kernel (falls back to 8 waves when K % 128 != 0)
            else waves = 8;
        }
    }
    if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_RESID_BF16 || EPI == EPI_RESID_F16 || EPI == EPI_BF16) {
        // deterministic split-K over 256x256 tiles (waves == 13 forces it): a deep K on a grid that fills less than half
        // of the CUs with 256x256 tiles and would otherwise run 128x128 tiles at 2.4x the operand traffic
        long tiles = (long)((p.N + 255) / 256) * ((p.M + 255) / 256) * p.batch;
        if (p2.M > 0) tiles += (long)((p2.N + 255) / 256) * ((p2.M + 255) / 256) * p2.batch;
        const bool shape_ok = p.K % 256 == 0 && p.K >= 512 && (p2.M == 0 || p2.K == p.K) && p.N % 256 == 0 &&
                              (p2.M == 0 || p2.N % 256 == 0) && 2 * tiles <= g_num_cu && tiles <= kSplitMaxTiles;
        if (shape_ok && (waves == 13 || (g_gemm_splitk && g_gemm_waves == 0 && EPI == EPI_RESID_F32 && p.K >= 2048 &&
                                         2 * tiles > g_num_cu / 2))) {
            const hipError_t e = launch_gemm8_split<EPI>(p, p2, s);
            if (e != hipErrorNotSupported) return e;
        }
        if (waves == 13) waves = 11;
    }
    if (waves == 13) waves = 11;
    if ((waves == 11 || waves == 12) && p.K % 128 == 0 && (p2.M == 0 || p2.K % 128 == 0)) {   // phased 256x256x64
        {
            // persistent form when a CU gets more than one tile and the output is bf16 (waves == 12 forces it for any
            // epilogue but QKV); it needs 160 KiB of LDS per workgroup
            long tiles = (long)((p.N + 255) / 256) * ((p.M + 255) / 256) * p.batch;
            if (p2.M > 0) tiles += (long)((p2.N + 255) / 256) * ((p2.M + 255) / 256) * p2.batch;
            // read-modify-write epilogues: bit 0 of gemm_persistent_resid admits the fp32 residual form, bit 1 the bf16 one
            // (round 4: the fused QKV epilogue runs in passes of 32 rows through the persistent kernel's 4 KiB of scratch per wave)
            const bool resid_ok = (EPI != EPI_RESID_F32 || (g_gemm_persistent_resid & 1)) &&
                                  ((EPI != EPI_RESID_BF16 && EPI != EPI_RESID_F16) || (g_gemm_persistent_resid & 2)) &&
                                  (EPI != EPI_QKV || (g_gemm_persistent_qkv && p.wide_epilogue));
            if (p.K >= 256 && (p2.M == 0 || p2.K >= 256) &&
                (waves == 12 || (g_gemm_persistent && g_gemm_waves == 0 && tiles > g_num_cu && resid_ok && EPI != EPI_F32))) {
                const hipError_t e = launch_gemm8p<EPI>(p, p2, g_num_cu, s);
                if (e != hipErrorNotSupported) return e;
            }
        }
        return launch_gemm8<EPI>(p, p2, s);
    }
    if (waves == 11 || waves == 12) waves = g_gemm_waves == 0 ? 9 : 8;
    if (waves == 32 && p.K % 32 == 0 && p2.M == 0) return launch_deep<EPI>(p, batch, s);   // deep-ring 256x256x32 kernel
    if (waves == 16) return launch_cfg<EPI, 16, 1>(p, p2, glds, s);
    if (waves == 9) return launch_cfg<EPI, 8, 1>(p, p2, glds, s);   // 256x256 tile, 8 waves of 128x64
    if (waves == 10) return launch_cfg<EPI, 8, 2>(p, p2, glds, s);  // 256x128 tile, 8 waves of 64x64
    if (waves == 8) return launch_cfg<EPI, 8, 0>(p, p2, glds, s);
    return launch_cfg<EPI, 4, 0>(p, p2, glds, s);
}

}  // namespace

static bool g_gemm_glds = true;
void gemm_set_glds(bool on) { g_gemm_glds = on; }
void gemm_set_raster(int group) { g_gemm_raster = group; }
void gemm_set_auto_rule(int rule, int num_cu) {
    if (rule >= 0) g_gemm_auto_rule = rule;
    if (num_cu > 0) g_num_cu = num_cu;
}
void gemm_set_wide_epilogue(bool on) { g_gemm_wide_epilogue = on; }
void gemm_set_gelu_pk(bool on) { g_gemm_gelu_pk = on; }
void gemm_set_phased(bool on) { g_gemm_phased = on; }
void gemm_set_persistent(bool on) { g_gemm_persistent = on; }
void gemm_set_persistent_resid(int mask) { g_gemm_persistent_resid = mask & 3; }
void gemm_set_splitk(bool on) { g_gemm_splitk = on; }
void gemm_set_splitk128(bool on) { g_gemm_splitk128 = on; }
void gemm_set_xcd_walk(bool on) { g_gemm_xcd_walk = on; }
void gemm_set_conv_implicit(bool on) { g_conv_implicit = on; }
bool gemm_conv_implicit() { return g_conv_implicit && g_gemm_glds && g_gemm_waves == 0; }
// the automatic rule of launch_epi (rule 2) for ONE problem: does it take 256 x 256 tiles (the phased kernel, where K % 128 == 0)?
bool gemm_auto_takes_256(int M, int N, int K) {
    if (g_gemm_waves != 0 || g_gemm_auto_rule == 0) return false;
    const long t256 = (long)((N + 255) / 256) * ((M + 255) / 256);
    const long rounds = (t256 + g_num_cu - 1) / g_num_cu;
    const bool fills = t256 * 10 >= rounds * g_num_cu * 9;
    const bool wide_enough = N >= 4096 || (g_gemm_auto_rule >= 2 && t256 * 100 >= (long)g_num_cu * 85);
    return N % 256 == 0 && t256 >= 128 && (K >= 2048 || t256 >= 2048 || (wide_enough && fills));
}
int gemm_splitk128_factor(int M, int N, int K) { return splitk128_factor(M, N, K, g_num_cu); }
void gemm_set_epi_slices(bool on) { g_gemm_epi_slices = on; }
void gemm_set_mixed(bool on) { g_gemm_mixed = on; }
void gemm_set_persistent_qkv(bool on) { g_gemm_persistent_qkv = on; }
void gemm_set_config(int waves) {
    if (waves == 0 || waves == 4 || waves == 8 || waves == 9 || waves == 10 || waves == 11 || waves == 12 || waves == 13 || waves == 16 || waves == 32) g_gemm_waves = waves;
}

static bool gemm_args_ok(const GemmArgs& p) {
    if (p.epi == EPI_QKV && p.qkv.nseg > 3) {
        if (p.qkv.nseg > 64) return false;
        // the device table: sorted, disjoint, at most three segments in any window of 128 rows (a wave's rows)
        const int* t = p.qkv.seg_tab_host;
        if (!t || !p.qkv.seg_tab) return false;
        for (int i = 0; i < p.qkv.nseg; ++i) {
            if (t[4 * i] > t[4 * i + 1]) return false;
            if (i > 0 && t[4 * i] < t[4 * (i - 1) + 1]) return false;
            if (i >= 3 && t[4 * i] - t[4 * (i - 3) + 1] + 1 < 128) return false;
        }
    }
    return !(p.K % 64 != 0 || p.K <= 0 || (p.N & 3) || (p.lda & 7) || (p.ldw & 7) || (p.epi == EPI_QKV && p.N % 64));
}

// One launch for one problem (p2 == nullptr) or for two problems with the same epilogue (and hence kernel).
static int g_last_splitk_slices = 1;     // the K slices the last gemm_launch2 actually ran with (r3g_op_gemm_splitk reports it; test hook)
int gemm_last_splitk_slices() { return g_last_splitk_slices; }

hipError_t gemm_launch2(const GemmArgs& p_in, int batch, const GemmArgs* p2_in, int batch2, hipStream_t s) {
    GemmArgs p = p_in, p2{};
    g_last_splitk_slices = 1;
    p.batch = batch;
    p.raster_group = g_gemm_raster;  // <0: automatic (4 tile-columns per group for 256x256 tiles, row-major otherwise)
    p.wide_epilogue = g_gemm_wide_epilogue ? 1 : 0;
    p.gelu_pk = g_gemm_gelu_pk ? 1 : 0;
    if (p2_in && p2_in->M > 0 && p2_in->N > 0 && batch2 > 0) {
        p2 = *p2_in;
        p2.batch = batch2;
        p2.raster_group = p.raster_group;
        p2.wide_epilogue = p.wide_epilogue;
        p2.gelu_pk = p.gelu_pk;
        if (p2.epi != p.epi || !gemm_args_ok(p2)) return hipErrorInvalidValue;
    }
    if (p.M <= 0 || p.N <= 0 || batch <= 0) {
        if (p2.M > 0) return gemm_launch2(*p2_in, batch2, nullptr, 0, s);
        return hipSuccess;
    }
    if (!gemm_args_ok(p)) return hipErrorInvalidValue;
    // algorithmic bytes of a launch: each operand read once, the result written once (fp32 residual: read + written)
    auto alg_bytes = [](const GemmArgs& g) {
        if (g.M <= 0) return 0.0;
        const double out = g.epi == EPI_RESID_F32 ? 8.0 : (g.epi == EPI_F32 || g.epi == EPI_RESID_BF16 || g.epi == EPI_RESID_F16 ? 4.0 : g.epi == EPI_RESID_BF16_LND ? 2.0 : (g.epi == EPI_FP8_GELU_ERF ? 1.0 : 2.0));
        return (double)g.batch * (2.0 * g.M * g.K + out * (double)g.M * g.N) + 2.0 * (double)g.N * g.K;
    };
    ProfScope ps(PC_GEMM, 2.0 * (double)p.M * p.N * p.K * batch + 2.0 * (double)p2.M * p2.N * p2.K * p2.batch, s,
                 alg_bytes(p) + alg_bytes(p2));
    if (p.conv.x) {
        // implicit-GEMM 3 x 3 convolution on the 128 x 128 kernel (conv_gemm_kernel), split over K by the same rule as below
        const ConvA& cv = p.conv;
        if (p2.M > 0 || batch != 1 || (p.epi != EPI_F32 && p.epi != EPI_RESID_F32) || !g_gemm_glds || cv.Cin < 64 || cv.Cin % 64 ||
            p.K != 9 * cv.Cin || !cv.zero || cv.Ho < 1 || cv.Wo < 1 || p.M % (cv.Ho * cv.Wo) || (p.ldc & 3) ||
            (reinterpret_cast<uintptr_t>(p.C) & 15))
            return hipErrorInvalidValue;
        const int64_t MN = (int64_t)p.M * p.N;
        int S = 1;
        if (p.split_ws && g_gemm_splitk128) {
            S = splitk128_factor(p.M, p.N, p.K, g_num_cu);
            if ((int64_t)S * MN > p.split_ws_elems) S = 1;
        }
        if (S > 1) {
            g_last_splitk_slices = S;
            GemmArgs q = p;
            q.K = p.K / S;
            q.batch = S;
            q.strideW = q.K;
            q.bias = nullptr;
            q.gate = nullptr;
            q.C = p.split_ws;
            q.ldc = p.N;
            q.strideC = MN;
            q.epi = EPI_F32;
            q.split_ws = nullptr;
            const hipError_t e = launch_conv<EPI_F32>(q, s);
            if (e != hipSuccess) return e;
            const unsigned blocks = (unsigned)((MN / 4 + 255) / 256);
            if (p.epi == EPI_RESID_F32)
                hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, (const float*)p.split_ws, S, MN, p.N,
                                   p.bias, p.gate, reinterpret_cast<float*>(p.C), p.ldc);
            else
                hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, (const float*)p.split_ws, S, MN, p.N,
                                   p.bias, p.gate, reinterpret_cast<float*>(p.C), p.ldc);
            return hipGetLastError();
        }
        return p.epi == EPI_F32 ? launch_conv<EPI_F32>(p, s) : launch_conv<EPI_RESID_F32>(p, s);
    }
    if (p.split_ws && g_gemm_splitk128 && g_gemm_waves == 0 && p2.M == 0 && batch == 1 && (p.epi == EPI_F32 || p.epi == EPI_RESID_F32) &&
        (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0) {
        const int S = splitk128_factor(p.M, p.N, p.K, g_num_cu);
        const int64_t MN = (int64_t)p.M * p.N;
        if (S > 1 && (int64_t)S * MN <= p.split_ws_elems) {
            g_last_splitk_slices = S;
            GemmArgs q = p;
            q.K = p.K / S;
            q.batch = S;
            q.strideA = q.K;
            q.strideW = q.K;
            q.bias = nullptr;
            q.gate = nullptr;
            q.C = p.split_ws;
            q.ldc = p.N;
            q.strideC = MN;
            q.epi = EPI_F32;
            q.split_ws = nullptr;
            const hipError_t e = launch_cfg<EPI_F32, 8, 0>(q, GemmArgs{}, g_gemm_glds, s);
            if (e != hipSuccess) return e;
            const unsigned blocks = (unsigned)((MN / 4 + 255) / 256);
            if (p.epi == EPI_RESID_F32)
                hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, (const float*)p.split_ws, S, MN, p.N,
                                   p.bias, p.gate, reinterpret_cast<float*>(p.C), p.ldc);
            else
                hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, (const float*)p.split_ws, S, MN, p.N,
                                   p.bias, p.gate, reinterpret_cast<float*>(p.C), p.ldc);
            return hipGetLastError();
        }
    }
    if (p.epi == EPI_RESID_BF16_ST) {
        const uint16_t* xr = p.resid_src ? p.resid_src : reinterpret_cast<const uint16_t*>(p.C);
        if (p2.M > 0 || batch != 1 || !g_gemm_glds || !p.wide_epilogue || !p.lnd_part || p.K % 128 || p.K < 256 || p.N % 256 || (p.ldc & 7) ||
            (reinterpret_cast<uintptr_t>(p.C) & 15) || (reinterpret_cast<uintptr_t>(xr) & 15) || (reinterpret_cast<uintptr_t>(p.lnd_part) & 15))
            return hipErrorInvalidValue;
        return launch_gemm8<EPI_RESID_BF16_ST>(p, p2, s);
    }
    if (p.epi == EPI_BF16_GELU_ERF_LNF) {
        // one problem on the persistent phased kernel (bf16 output through the sliced LDS transpose)
        const long tiles = (long)(p.N / 256) * ((p.M + 255) / 256);
        if (p2.M > 0 || batch != 1 || !g_gemm_glds || !p.wide_epilogue || !p.lnf_c1 || !p.lnf_stats || !p.bias || p.K % 128 || p.K < 256 ||
            p.N % 256 || (p.ldc & 7) || (reinterpret_cast<uintptr_t>(p.C) & 15) || tiles < 1)
            return hipErrorInvalidValue;
        return launch_gemm8p<EPI_BF16_GELU_ERF_LNF>(p, p2, g_num_cu, s);
    }
    if (p.epi == EPI_RESID_BF16_LND) {
        // one problem on the phased 256 x 256 kernel, wide read-modify-write epilogue (its alignment rules checked here: the
        // epilogue has no other path for this form)
        const uint16_t* xr = p.resid_src ? p.resid_src : reinterpret_cast<const uint16_t*>(p.C);
        if (p2.M > 0 || batch != 1 || !g_gemm_glds || !p.wide_epilogue || !p.lnd_gw || !p.lnd_part || p.gate || p.K % 128 || p.K < 256 ||
            p.N % 256 || (p.ldc & 7) || (reinterpret_cast<uintptr_t>(p.C) & 15) || (reinterpret_cast<uintptr_t>(xr) & 15) ||
            (reinterpret_cast<uintptr_t>(p.lnd_gw) & 15) || (reinterpret_cast<uintptr_t>(p.lnd_part) & 15))
            return hipErrorInvalidValue;
        return launch_gemm8<EPI_RESID_BF16_LND>(p, p2, s);
    }
    switch (p.epi) {
        case EPI_BF16: return launch_epi<EPI_BF16>(p, p2, g_gemm_glds, s);
        case EPI_BF16_GELU_TANH: return launch_epi<EPI_BF16_GELU_TANH>(p, p2, g_gemm_glds, s);
        case EPI_BF16_GELU_ERF: return launch_epi<EPI_BF16_GELU_ERF>(p, p2, g_gemm_glds, s);
        case EPI_RESID_F32: return launch_epi<EPI_RESID_F32>(p, p2, g_gemm_glds, s);
        case EPI_F32: return launch_epi<EPI_F32>(p, p2, g_gemm_glds, s);
        case EPI_QKV: return launch_epi<EPI_QKV>(p, p2, g_gemm_glds, s);
        case EPI_RESID_BF16: return launch_epi<EPI_RESID_BF16>(p, p2, g_gemm_glds, s);
        case EPI_RESID_F16: return launch_epi<EPI_RESID_F16>(p, p2, g_gemm_glds, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t gemm_launch(const GemmArgs& p, int batch, hipStream_t s) { return gemm_launch2(p, batch, nullptr, 0, s); }

// A DiT single block's linear1 (upstream: ONE Linear H -> 3H + mlp_hidden): the fused QKV projection and the MLP-in + GELU(tanh)
// projection read the same rows.  Round 6: both problems as ONE persistent launch of the phased kernel (gemm8p_kernel's EPI2) when
// that grid fills the machine -- 1 416 + 1 888 tiles of 256 x 256 are 12.9 rounds of 256 CUs where the two launches took 6 + 8 --
// otherwise the two launches of rounds 1-5 (MLP-in first).  Same tiles, same k order: bit-identical either way.
hipError_t gemm_launch_qkv_mlp(const GemmArgs& pq_in, const GemmArgs& pm_in, hipStream_t s) {
    GemmArgs pq = pq_in, pm = pm_in;
    auto fallback = [&]() {
        const hipError_t e = gemm_launch(pm_in, 1, s);
        return e != hipSuccess ? e : gemm_launch(pq_in, 1, s);
    };
    if (!g_gemm_mixed || !g_gemm_persistent || !g_gemm_phased || g_gemm_waves != 0 || !g_gemm_glds || !g_gemm_wide_epilogue ||
        pq.epi != EPI_QKV || pm.epi != EPI_BF16_GELU_TANH || pq.conv.x || pm.conv.x || pq.M <= 0 || pm.M <= 0)
        return fallback();
    if (pq.K % 128 || pm.K % 128 || pq.K < 256 || pm.K < 256 || pq.N % 256 || pm.N % 256) return fallback();
    const long tiles = (long)(pq.N / 256) * ((pq.M + 255) / 256) + (long)(pm.N / 256) * ((pm.M + 255) / 256);
    const long rounds = (tiles + g_num_cu - 1) / g_num_cu;
    if (tiles < 2L * g_num_cu || tiles * 10 < rounds * g_num_cu * 9) return fallback();   // the grid must fill >= 90 % of its rounds
    for (GemmArgs* g : {&pq, &pm}) {
        g->batch = 1;
        g->raster_group = g_gemm_raster;
        g->wide_epilogue = 1;
        g->gelu_pk = g_gemm_gelu_pk ? 1 : 0;
        if (!gemm_args_ok(*g)) return hipErrorInvalidValue;
    }
    auto alg_bytes = [](const GemmArgs& g) { return 2.0 * g.M * g.K + 2.0 * (double)g.M * g.N + 2.0 * (double)g.N * g.K; };
    ProfScope ps(PC_GEMM, 2.0 * (double)pq.M * pq.N * pq.K + 2.0 * (double)pm.M * pm.N * pm.K, s, alg_bytes(pq) + alg_bytes(pm));
    const hipError_t e = launch_gemm8p<EPI_QKV, EPI_BF16_GELU_TANH>(pq, pm, g_num_cu, s);
    if (e == hipErrorNotSupported) return fallback();
    return e;
}

// FP8 operands: A8 [M][K] and W8 [N][K] bytes (e4m3), one fp32 scale per row of each.  K % 256 == 0, lda / ldw in bytes.
hipError_t gemm_fp8_launch(const GemmArgs& p_in, const float* scale_a, const float* scale_w, hipStream_t s) {
    GemmArgs p = p_in;
    if (p.K % 256 || p.K < 256 || (p.lda & 15) || (p.ldw & 15) || (p.N & 3) || !scale_a || !scale_w) return hipErrorInvalidValue;
    if ((int64_t)p.M * p.lda >= (1ll << 32) || (int64_t)p.N * p.ldw >= (1ll << 32)) return hipErrorInvalidValue;   // 32-bit staging offsets
    p.batch = 1;
    p.K /= 2; p.lda /= 2; p.ldw /= 2;          // the kernel sees byte pairs (file comment of gemm8f_kernel)
    p.raster_group = g_gemm_raster;
    p.wide_epilogue = g_gemm_wide_epilogue ? 1 : 0;
    ProfScope ps(PC_GEMM, 2.0 * (double)p.M * p.N * (2.0 * p.K), s);
    switch (p.epi) {
        case EPI_BF16: return launch_gemm8_fp8<EPI_BF16>(p, scale_a, scale_w, s);
        case EPI_BF16_GELU_TANH: return launch_gemm8_fp8<EPI_BF16_GELU_TANH>(p, scale_a, scale_w, s);
        case EPI_BF16_GELU_ERF: return launch_gemm8_fp8<EPI_BF16_GELU_ERF>(p, scale_a, scale_w, s);
        case EPI_RESID_F32: return launch_gemm8_fp8<EPI_RESID_F32>(p, scale_a, scale_w, s);
        case EPI_RESID_BF16: return launch_gemm8_fp8<EPI_RESID_BF16>(p, scale_a, scale_w, s);
        case EPI_QKV: return launch_gemm8_fp8<EPI_QKV>(p, scale_a, scale_w, s);
        case EPI_FP8_GELU_ERF: return launch_gemm8_fp8<EPI_FP8_GELU_ERF>(p, scale_a, scale_w, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace r3g
