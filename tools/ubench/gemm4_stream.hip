// NOT PART OF THE LIBRARY (round 5): round 4's 4-wave stream GEMM, kept as the record behind profiles/r04_gemm_stream.md.  It was
// bit-identical to the phased kernels and measured slower (934 against 985 TF/s); the product library no longer builds it.
// gemm4.hip -- the bf16-output GEMMs with a short K (K = 1024: MLP-in + GELU, QKV-shaped projections, the geo decoder's c_fc)
// as ONE stream of k-tiles per compute unit: four waves, one per SIMD, 512 registers each.
//
//   C[m][n] (bf16) = epi( sum_k A[m][k] W[n][k] + bias[n] ),   epi = identity | GELU(tanh) | GELU(erf)
//
// Why another kernel (profiles/r02_gemm_persistent.md, profiles/r03_mfma_util.md): the phased 8-wave kernel of gemm.hip runs
// its k-loop at 93 % of the MFMA issue rate, but a K = 1024 tile is only 16 k-tiles = 37 k cycles of k-loop, and between
// two k-loops a workgroup spends 17-26 k cycles with the matrix pipe idle (epilogue arithmetic on a VALU two waves share,
// the accumulator -> LDS transpose -> store sequence, the restart of the staging pipeline).  Two waves per SIMD leave 256
// registers per wave: 128 accumulators + fragments + addresses fill them, nothing of a finished tile can stay behind.
//
// Here a workgroup is 4 waves (2 x 2), each owning 128 x 128 of the 256 x 256 tile = 256 accumulator registers, and
//   * the k-tiles of the workgroup's tiles (w, w + G, w + 2G, ...) form one stream: the LDS-DMA of the next tile's first
//     k-tiles is issued inside the last k-tiles of the current one -- no drain, no prologue between tiles;
//   * at a tile boundary a wave turns its accumulators into the final bf16 values (bias, GELU, round) and keeps them in 128
//     registers; they leave as 16-byte stores straight from the registers, four per pair of k-tiles, inside the NEXT tile's
//     k-loop.  No LDS transpose: the rows of W are staged in a permuted order (LDS row r of a 32-row group holds column
//     8 (r >> 2 & 3) + 4 (r >> 4) + (r & 3)), so that the two MFMAs of a column pair leave 8 consecutive columns in a lane;
//   * the first MFMAs of a tile take the constant 0 as their C operand (no accumulator clearing);
//   * operands and results go through buffer descriptors: rows past M read as zero and are not written (no clamping, no
//     predication), the k advance and the tile origin live in scalar registers, a lane keeps five 32-bit offsets in all.
// Per k-tile and wave: 128 MFMAs (16x16x32 bf16), 32 ds_read_b128, 16 LDS-DMA pieces of 1 KiB, one s_barrier.
// Same MFMA shape and the same k order per output element as every kernel of gemm.hip, the same epilogue functions
// (gemm_common.h): results are bit-identical to theirs (tools/bench_gemm.py screens it).
//
// LDS (133 120 bytes): two k-tile buffers [A 256 rows x 128 B | W 256 rows x 128 B] with the image of gemm.hip (16-byte
// chunks of a row swizzled by (row >> 1) & 7 on the DMA source and on the fragment reads), then two 1 KiB bias rows.
//
// Synchronisation.  Body t reads buffer b = t & 1.  The only barrier sits in front of the LAST quadrant of a k-tile, behind
// a wave's s_waitcnt vmcnt(0): at that point the wave has every fragment of k-tile t in registers (buffer b is dead for
// it) and its own pieces of k-tile t + 1 have landed in buffer b ^ 1.  Behind the barrier both hold for all four waves:
// the fragments of k-tile t + 1's first quadrant are read (their latency hides under the last quadrant's 16 MFMAs) and the
// pieces of k-tile t + 2 start to go into buffer b -- four in this quadrant, twelve in the first three quadrants of body
// t + 1, the deferred stores in its fourth; everything has most of a k-tile (~2 k cycles) to complete.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gemm_common.h"
#include "kernels.h"

namespace r3g {
namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int G4_BUF = 65536, G4_W = 32768, G4_BIAS = 131072, G4_LDS = 131072 + 2048;

#define G4_BAR() asm volatile("s_barrier" ::: "memory")
#define G4_SB() __builtin_amdgcn_sched_barrier(0)

template <int V>
using IC = std::integral_constant<int, V>;

// a buffer descriptor from wave-uniform inputs (readfirstlane makes the uniformity provable: no waterfall loops)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t g4_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a >> 32));
    const uint32_t nb = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((static_cast<uint64_t>(hi) << 32) | lo), 0,
                                             static_cast<int>(nb), 0x00020000);
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm4_kernel(GemmArgs pa, GemmArgs pb, int total_tiles, int ablate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int wg_first = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tiles_a = ((pa.N + BN - 1) / BN) * ((pa.M + BM - 1) / BM) * pa.batch;
    typedef const char __attribute__((address_space(4))) * kernarg_ptr;
    kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kSecond = (sizeof(GemmArgs) + alignof(GemmArgs) - 1) / alignof(GemmArgs) * alignof(GemmArgs);
    (void)pb;
    auto args_of = [&](int second) -> const GemmArgs& {
        return *(const GemmArgs*)(const GemmArgs __attribute__((address_space(4)))*)(ka + (second ? kSecond : 0));
    };
    struct Tile { int second, batch, m0, n0; };
    // tile index -> (problem, batch, origin): the rasterisation of gemm8p_kernel (groups of 4 tile columns)
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
        return t;     // every input is a kernel argument or the workgroup id: scalar registers
    };

    // ------------------------------------------------------------------------------------------ the load cursor
    // (tile lidx, k-tile lkt): where the NEXT LDS-DMA piece comes from.  Descriptors: A from the tile's first row on (rows
    // past M are out of range: zeros), W whole.  A lane's offsets: the 16-byte chunk (lane & 7), swizzled by the LDS row it
    // lands in, of row (lane >> 3) of the piece; pieces differ by a uniform number of rows.
    __amdgpu_buffer_rsrc_t lA = g4_rsrc(nullptr, 0), lW = g4_rsrc(nullptr, 0), lB = g4_rsrc(nullptr, 0);
    int lit = 0, lkt = 0, lnk = 2;      // the cursor is in the workgroup's lit-th tile (bias row lit & 1), at k-tile lkt of lnk
    uint32_t voffA[2] = {0, 0}, voffW[2] = {0, 0};
    uint32_t lda16 = 0, ldwrow = 0;      // bytes of 8 rows of A; bytes of one row of W
    auto set_load_tile = [&]() {
        lkt = 0;
        const int lidx = wg_first + lit * nwg;
        if (lidx >= total_tiles) {       // nothing left: descriptors of zero bytes, the pieces become zero fills
            lA = g4_rsrc(nullptr, 0);
            lW = g4_rsrc(nullptr, 0);
            lB = g4_rsrc(nullptr, 0);
            return;
        }
        const Tile t = locate(lidx);
        const GemmArgs& p = args_of(t.second);
        lnk = __builtin_amdgcn_readfirstlane(p.K >> 6);
        const uint32_t lda2 = static_cast<uint32_t>(p.lda) * 2u, ldw2 = static_cast<uint32_t>(p.ldw) * 2u;
        lA = g4_rsrc(p.A + (int64_t)t.batch * p.strideA + (int64_t)t.m0 * p.lda, static_cast<uint32_t>(p.M - t.m0) * lda2);
        lW = g4_rsrc(p.W + (int64_t)t.n0 * p.ldw, static_cast<uint32_t>(p.N - t.n0) * ldw2);
        lB = p.bias ? g4_rsrc(p.bias + t.n0, 1024u) : g4_rsrc(nullptr, 0);     // no bias: a row of zeros
        lda16 = __builtin_amdgcn_readfirstlane(lda2 * 8u);
        ldwrow = __builtin_amdgcn_readfirstlane(ldw2);
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            // LDS row of piece i: wid*64 + i*8 + (lane >> 3); (row >> 1) & 7 = (i & 1) * 4 + (lane >> 4)
            const uint32_t kc = static_cast<uint32_t>((lane & 7) ^ (par * 4 + (lane >> 4)));
            voffA[par] = static_cast<uint32_t>(wid * 64 + (lane >> 3)) * lda2 + kc * 16u;
            // W: LDS row r = (i & 3) * 8 + (lane >> 3) of the 32-row group (wid * 2 + (i >> 2)) holds column
            // (i & 1) * 16 + 8 * (lane >> 5) + 4 * ((i >> 1) & 1) + ((lane >> 3) & 3) of that group
            voffW[par] = static_cast<uint32_t>(wid * 64 + 8 * (lane >> 5) + ((lane >> 3) & 3)) * ldw2 + kc * 16u;
        }
    };
    // piece s of the cursor's k-tile into buffer `buf`: s even -> A piece s / 2, s odd -> W piece s / 2 (this wave's 8 + 8)
    const uint32_t lds_wave = static_cast<uint32_t>(wid) * 8192u;      // this wave's eight pieces of a region
    auto piece = [&](auto S, auto BUFC) {
        constexpr int s = decltype(S)::value, i = s >> 1, buf = decltype(BUFC)::value;
        if (ablate & 1) return;
        // the LDS address (-> M0) is formed per piece: hoisted out of the loops, the 64 addresses would occupy scalar registers
        uint32_t wb = lds_wave;
        asm volatile("" : "+s"(wb));
        if constexpr ((s & 1) == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                lA, (__attribute__((address_space(3))) void*)(uintptr_t)(wb + static_cast<uint32_t>(buf * G4_BUF + i * 1024)), 16,
                voffA[i & 1] + static_cast<uint32_t>(i) * lda16, lkt * 128, 0, 0);
        } else {
            constexpr int rowsel = (i >> 2) * 32 + (i & 1) * 16 + ((i >> 1) & 1) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                lW, (__attribute__((address_space(3))) void*)(uintptr_t)(wb + static_cast<uint32_t>(buf * G4_BUF + G4_W + i * 1024)), 16,
                voffW[i & 1] + static_cast<uint32_t>(rowsel) * ldwrow, lkt * 128, 0, 0);
        }
    };
    // the tile's 256 bias values (1 KiB) go with its first k-tile, into the bias row of the tile's parity
    auto bias_piece = [&]() {
        if (wid == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lB, (__attribute__((address_space(3))) void*)(smem + G4_BIAS + (lit & 1) * 1024),
                                                     16, static_cast<uint32_t>(lane) * 16u, 0, 0, 0);
    };
    auto cursor_next = [&]() {      // behind the last piece of a k-tile
        if (++lkt == lnk) {
            ++lit;
            set_load_tile();
        }
    };

    // ------------------------------------------------------------------------------------------ fragments
    const int sw = (lane >> 1) & 7;   // == ((row >> 1) & 7): sub-tile and wave offsets are multiples of 16 rows
    int offA[2], offW[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        offA[kk] = (wr * 128 + (lane & 15)) * 128 + ((((kk << 2) + (lane >> 4)) ^ sw) << 4);
        offW[kk] = G4_W + (wc * 128 + (lane & 15)) * 128 + ((((kk << 2) + (lane >> 4)) ^ sw) << 4);
    }
    bf16x8 af[8], wf[8];
    auto rd_a = [&](auto H, auto BUFC, auto KK) {
        constexpr int h = decltype(H)::value, buf = decltype(BUFC)::value, kk = decltype(KK)::value;
        if (ablate & 16) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            af[h * 4 + i] = *reinterpret_cast<const bf16x8*>(smem + buf * G4_BUF + (h * 4 + i) * 2048 + offA[kk]);
    };
    auto rd_w = [&](auto H, auto BUFC, auto KK) {
        constexpr int h = decltype(H)::value, buf = decltype(BUFC)::value, kk = decltype(KK)::value;
        if (ablate & 16) return;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            wf[h * 4 + j] = *reinterpret_cast<const bf16x8*>(smem + buf * G4_BUF + (h * 4 + j) * 2048 + offW[kk]);
    };

    f32x4 acc[8][8];        // [j: 16-column group][i: 16-row group]
    u32x4 packed[32];       // the finished tile, bf16: [column pair p][i] = 8 consecutive columns of one row
#pragma unroll
    for (int s = 0; s < 32; ++s) packed[s] = (u32x4){0u, 0u, 0u, 0u};
    // 16 MFMAs = one accumulator quadrant x K = 32, in four groups of four with a hook behind each group
    auto quad = [&](auto HA, auto HW, auto ZERO, auto hook) {
        constexpr int ha = decltype(HA)::value, hw = decltype(HW)::value;
        constexpr bool zero = decltype(ZERO)::value != 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int j = hw * 4 + jj, i = ha * 4 + ii;
                // inline asm with the accumulator tied to the accumulation-register file ("a"): the 256 accumulators are
                // exactly that file, everything else lives in the 256 architectural registers.  Left to the builtin, hipcc
                // keeps part of the accumulators in architectural registers and copies them in and out around every MFMA.
                if constexpr (zero)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[j][i]) : "v"(wf[j]), "v"(af[i]));
                else
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(wf[j]), "v"(af[i]));
            }
            G4_SB();
            if (jj == 0) hook(IC<0>{});
            else if (jj == 1) hook(IC<1>{});
            else if (jj == 2) hook(IC<2>{});
            else hook(IC<3>{});
            G4_SB();
        }
    };
    auto no_hook = [](auto) {};

    // ------------------------------------------------------------------------------------------ the finished tile
    __amdgpu_buffer_rsrc_t sC = g4_rsrc(nullptr, 0);   // where the packed tile goes (nothing before the first tile is done)
    uint32_t voffC = 0, ldc32 = 0, scol = 0;           // lane offset; bytes of 16 rows of C; byte offset of the tile's column 0
    auto store_one = [&](auto S) {
        constexpr int s = decltype(S)::value, p = s >> 3, i = s & 7;
        if (ablate & 8) return;
        __builtin_amdgcn_raw_buffer_store_b128(packed[s], sC, voffC + static_cast<uint32_t>(i) * ldc32, scol + p * 64, 0);
    };
    // pair `pr` of k-tiles (0..7) carries stores 4 pr .. 4 pr + 3; u = which of the four
    auto store_slot = [&](int pr, auto U) {
        constexpr int u = decltype(U)::value;
        switch (pr) {
            case 0: store_one(IC<0 + u>{}); break;
            case 1: store_one(IC<4 + u>{}); break;
            case 2: store_one(IC<8 + u>{}); break;
            case 3: store_one(IC<12 + u>{}); break;
            case 4: store_one(IC<16 + u>{}); break;
            case 5: store_one(IC<20 + u>{}); break;
            case 6: store_one(IC<24 + u>{}); break;
            case 7: store_one(IC<28 + u>{}); break;
            default: break;
        }
    };
    // `it`: the workgroup's it-th tile (index wg_first + it * nwg, bias row it & 1).  What a lane needs here is derived from a
    // lane id computed on the spot: values kept from the prologue would sit in registers (or scratch) through every k-loop.
    auto finish_tile = [&](const int it) {
        if (ablate & 4) return;
        const int cidx = wg_first + it * nwg, cpar = it & 1;
        const Tile t = locate(cidx);
        const GemmArgs& p = args_of(t.second);
        int ln;     // (volatile: not hoisted out of the tile loop)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const int q = ln >> 4;
        // The last MFMAs' results are read below, and hipcc pads nothing behind an asm MFMA: without this, the register
        // allocator's copies out of the accumulation file land right behind the MFMA that produces the value (measured: the
        // sixteen accumulator tiles of the last quadrant came out stale).  The empty statements re-define every accumulator
        // BEHIND the wait states, so that no copy can be placed ahead of them.
        asm volatile("s_nop 15" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(acc[j][i]));
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const char* b = smem + G4_BIAS + cpar * 1024 + (wc * 128 + pp * 32 + q * 8) * 4;
            const f32x4 blo = *reinterpret_cast<const f32x4*>(b), bhi = *reinterpret_cast<const f32x4*>(b + 16);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x4 v0 = acc[2 * pp][i] + blo, v1 = acc[2 * pp + 1][i] + bhi;
                if constexpr (EPI == EPI_BF16_GELU_TANH) { v0 = gelu_tanh4(v0); v1 = gelu_tanh4(v1); }
                else if constexpr (EPI == EPI_BF16_GELU_ERF) { v0 = gelu_erf4(v0); v1 = gelu_erf4(v1); }
                packed[pp * 8 + i] = (u32x4){pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]),
                                             pack_bf16(v1[2], v1[3])};
                if ((i & 1) == 1) G4_SB();      // keeps hipcc from reading all 256 accumulators ahead of their use
            }
        }
        const uint32_t ldc2 = static_cast<uint32_t>(p.ldc) * 2u;
        sC = g4_rsrc(reinterpret_cast<const uint16_t*>(p.C) + (int64_t)t.batch * p.strideC + (int64_t)t.m0 * p.ldc,
                     static_cast<uint32_t>(p.M - t.m0) * ldc2);
        voffC = static_cast<uint32_t>(wr * 128 + (ln & 15)) * ldc2 + static_cast<uint32_t>(wc * 128 + q * 8) * 2u;
        ldc32 = __builtin_amdgcn_readfirstlane(ldc2 * 16u);
        scol = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(t.n0) * 2u);
    };

    // ------------------------------------------------------------------------------------------ one k-tile
    // B: the buffer it reads.  FIRST / LAST: of its tile.  On entry the fragments of its first quadrant (A rows 0-63 of the
    // wave, W rows 0-63, k 0-31) are in registers and pieces 0-3 of the next k-tile are on their way.
    auto body = [&](auto BC, auto FIRSTC, auto LASTC, const int pr, const int it) {
        constexpr int b = decltype(BC)::value;
        constexpr bool last = decltype(LASTC)::value != 0;
        using Z = IC<decltype(FIRSTC)::value>;   // the k = 0..31 MFMAs of a tile's first k-tile start from C = 0
        using B0 = IC<b>;
        using B1 = IC<b ^ 1>;
        // ---- k 0-31
        rd_w(IC<1>{}, B0{}, IC<0>{});
        G4_SB();
        quad(IC<0>{}, IC<0>{}, Z{}, [&](auto G) { piece(IC<4 + decltype(G)::value>{}, B1{}); });
        rd_a(IC<1>{}, B0{}, IC<0>{});
        G4_SB();
        quad(IC<0>{}, IC<1>{}, Z{}, [&](auto G) { piece(IC<8 + decltype(G)::value>{}, B1{}); });
        rd_a(IC<0>{}, B0{}, IC<1>{});       // k 32-63 of A rows 0-63 (those registers are free now)
        G4_SB();
        quad(IC<1>{}, IC<1>{}, Z{}, [&](auto G) {
            piece(IC<12 + decltype(G)::value>{}, B1{});
            if constexpr (decltype(G)::value == 3) cursor_next();
        });
        // W rows 0-63, k 32-63: into the registers of W rows 64-127 (k 0-31), which the quadrant above used last
        {
            bf16x8 wn[4];
            if (!(ablate & 16)) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    wn[j] = *reinterpret_cast<const bf16x8*>(smem + b * G4_BUF + j * 2048 + offW[1]);
            }
            G4_SB();
            if constexpr (b == 0) quad(IC<1>{}, IC<0>{}, Z{}, [&](auto G) { store_slot(pr, G); });
            else quad(IC<1>{}, IC<0>{}, Z{}, no_hook);
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = wn[j];
        }
        // ---- k 32-63
        rd_w(IC<1>{}, B0{}, IC<1>{});
        G4_SB();
        quad(IC<0>{}, IC<0>{}, IC<0>{}, no_hook);
        rd_a(IC<1>{}, B0{}, IC<1>{});
        G4_SB();
        quad(IC<0>{}, IC<1>{}, IC<0>{}, no_hook);
        G4_SB();
        quad(IC<1>{}, IC<1>{}, IC<0>{}, no_hook);
        G4_SB();
        // every fragment of this k-tile is in registers; my pieces of the next one have landed
        if (!(ablate & 2)) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            G4_BAR();
        }
        G4_SB();
        {
            // first quadrant of the next k-tile, from the other buffer: its latency hides under the 16 MFMAs below (at the
            // end of a tile the reads wait until the accumulators have been packed: registers)
            bf16x8 an[4], wn[4];
            auto read_next = [&]() {
                if (ablate & 16) return;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    an[i] = *reinterpret_cast<const bf16x8*>(smem + (b ^ 1) * G4_BUF + i * 2048 + offA[0]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    wn[j] = *reinterpret_cast<const bf16x8*>(smem + (b ^ 1) * G4_BUF + j * 2048 + offW[0]);
            };
            if constexpr (!last) read_next();
            G4_SB();
            quad(IC<1>{}, IC<0>{}, IC<0>{}, [&](auto G) {
                if constexpr (decltype(G)::value == 0) { if (lkt == 0) bias_piece(); }
                piece(IC<decltype(G)::value>{}, B0{});
            });
            if constexpr (last) {
                finish_tile(it);
                G4_SB();
                read_next();
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = an[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = wn[j];
        }
        G4_SB();
    };

    // ------------------------------------------------------------------------------------------ prologue
    set_load_tile();
    const int my_tiles = (total_tiles - wg_first + nwg - 1) / nwg;      // >= 1: the grid is never larger than the tile count
    bias_piece();
    piece(IC<0>{}, IC<0>{}); piece(IC<1>{}, IC<0>{}); piece(IC<2>{}, IC<0>{}); piece(IC<3>{}, IC<0>{});
    piece(IC<4>{}, IC<0>{}); piece(IC<5>{}, IC<0>{}); piece(IC<6>{}, IC<0>{}); piece(IC<7>{}, IC<0>{});
    piece(IC<8>{}, IC<0>{}); piece(IC<9>{}, IC<0>{}); piece(IC<10>{}, IC<0>{}); piece(IC<11>{}, IC<0>{});
    piece(IC<12>{}, IC<0>{}); piece(IC<13>{}, IC<0>{}); piece(IC<14>{}, IC<0>{}); piece(IC<15>{}, IC<0>{});
    cursor_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G4_BAR();
    G4_SB();
    rd_a(IC<0>{}, IC<0>{}, IC<0>{});
    rd_w(IC<0>{}, IC<0>{}, IC<0>{});
    piece(IC<0>{}, IC<1>{}); piece(IC<1>{}, IC<1>{}); piece(IC<2>{}, IC<1>{}); piece(IC<3>{}, IC<1>{});
    G4_SB();

    for (int it = 0; it < my_tiles; ++it) {
        const int nk = __builtin_amdgcn_readfirstlane(args_of(locate(wg_first + it * nwg).second).K >> 6);   // even, >= 16 (launcher)
        body(IC<0>{}, IC<1>{}, IC<0>{}, 0, it);
        int pr = 1;
        for (int t = 1; t + 1 < nk; t += 2, ++pr) {
            body(IC<1>{}, IC<0>{}, IC<0>{}, pr, it);
            body(IC<0>{}, IC<0>{}, IC<0>{}, pr, it);
        }
        body(IC<1>{}, IC<0>{}, IC<1>{}, pr, it);
    }
    // the last tile's values: all 32 stores
    store_one(IC<0>{}); store_one(IC<1>{}); store_one(IC<2>{}); store_one(IC<3>{});
    store_one(IC<4>{}); store_one(IC<5>{}); store_one(IC<6>{}); store_one(IC<7>{});
    store_one(IC<8>{}); store_one(IC<9>{}); store_one(IC<10>{}); store_one(IC<11>{});
    store_one(IC<12>{}); store_one(IC<13>{}); store_one(IC<14>{}); store_one(IC<15>{});
    store_one(IC<16>{}); store_one(IC<17>{}); store_one(IC<18>{}); store_one(IC<19>{});
    store_one(IC<20>{}); store_one(IC<21>{}); store_one(IC<22>{}); store_one(IC<23>{});
    store_one(IC<24>{}); store_one(IC<25>{}); store_one(IC<26>{}); store_one(IC<27>{});
    store_one(IC<28>{}); store_one(IC<29>{}); store_one(IC<30>{}); store_one(IC<31>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the zero fills of the cursor past the end still target this LDS
}

int g_gemm4_ablate = 0;

template <int EPI>
hipError_t launch_gemm4_epi(const GemmArgs& p, const GemmArgs& p2, int num_cu, hipStream_t s) {
    int tiles = ((p.N + 255) / 256) * ((p.M + 255) / 256) * p.batch;
    if (p2.M > 0) tiles += ((p2.N + 255) / 256) * ((p2.M + 255) / 256) * p2.batch;
    const int rounds = (tiles + num_cu - 1) / num_cu;
    const int grid = (tiles + rounds - 1) / rounds;    // every workgroup gets `rounds` tiles (the last ones one fewer)
    auto k = gemm4_kernel<EPI>;
    static int state = 0;   // 0 unknown, 1 usable, -1 refused
    if (state == 0)
        state = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS) == hipSuccess ? 1 : -1;
    if (state < 0) { (void)hipGetLastError(); return hipErrorNotSupported; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), G4_LDS, s, p, p2, tiles, g_gemm4_ablate);
    return hipGetLastError();
}

bool g4_problem_ok(const GemmArgs& g) {
    if (g.M <= 0) return true;
    const uint64_t lim = 0xFFFFFFFFull;
    return g.K % 128 == 0 && g.K >= 1024 && g.N % 256 == 0 && (g.lda & 7) == 0 && (g.ldw & 7) == 0 && (g.ldc & 7) == 0 &&
           (g.strideC & 7) == 0 && (g.strideA & 7) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(g.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.W) & 15) == 0 &&
           (g.bias == nullptr || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0) && g.K <= g.lda && g.K <= g.ldw && g.N <= g.ldc &&
           (uint64_t)g.M * (uint64_t)g.lda * 2 <= lim && (uint64_t)g.N * (uint64_t)g.ldw * 2 <= lim &&
           (uint64_t)g.M * (uint64_t)g.ldc * 2 <= lim;
}

}  // namespace

void gemm4_set_ablate(int mask) { g_gemm4_ablate = mask; }   // timing experiments: 1 no LDS-DMA | 2 no barrier | 4 no packing | 8 no stores | 16 no fragment reads

// The stream kernel for a bf16-output launch, or hipErrorNotSupported when the launch is outside what it covers (the caller
// then takes the kernels of gemm.hip).  `force`: also when a compute unit would get fewer than two tiles.
hipError_t launch_gemm4(const GemmArgs& p, const GemmArgs& p2, int num_cu, bool force, hipStream_t s) {
    if (p.epi != EPI_BF16 && p.epi != EPI_BF16_GELU_TANH && p.epi != EPI_BF16_GELU_ERF) return hipErrorNotSupported;
    if (!g4_problem_ok(p) || !g4_problem_ok(p2)) return hipErrorNotSupported;
    long tiles = (long)(p.N / 256) * ((p.M + 255) / 256) * p.batch;
    if (p2.M > 0) tiles += (long)(p2.N / 256) * ((p2.M + 255) / 256) * p2.batch;
    if (!force && tiles < 2L * num_cu) return hipErrorNotSupported;
    switch (p.epi) {
        case EPI_BF16: return launch_gemm4_epi<EPI_BF16>(p, p2, num_cu, s);
        case EPI_BF16_GELU_TANH: return launch_gemm4_epi<EPI_BF16_GELU_TANH>(p, p2, num_cu, s);
        default: return launch_gemm4_epi<EPI_BF16_GELU_ERF>(p, p2, num_cu, s);
    }
}

}  // namespace r3g
