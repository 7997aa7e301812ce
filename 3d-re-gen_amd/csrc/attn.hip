// attn.hip -- fused (flash) attention forward for head dim 64 on gfx950, bf16 in / fp32 softmax / bf16 out.
//
// Replaces F.scaled_dot_product_attention as called by upstream hunyuan3ddit.attention() (joint
// self-attention over cat(cond, latent), L = 4442), attention_blocks.QKVMultiheadAttention (VAE,
// L = 3072), QKVMultiheadCrossAttention (geo decoder: 257^3 query points against 3072 latents) and
// Dinov2 self-attention; the reference reaches them from src/2d_to_3d_models/run.py:77-84.
//
// Formulation (everything about one query lives in lanes q and q+32 of a wave):
//   S^T = K Q^T   : v_mfma_f32_32x32x16_bf16, A = K tile rows (LDS), B = Q^T (registers, loaded once)
//                   -> lane (q = lane&31, h = lane>>5) holds 16 of the 32 keys of a key block
//   softmax       : online, log2 domain, per lane + one cross-half exchange (lane ^ 32) per tile
//   O^T += V^T P^T: A = V^T tile rows (LDS, V is produced pre-transposed by the QKV split kernel),
//                   B = P^T taken straight from the S accumulators: the MFMA k index is mapped to
//                   the keys a lane already holds, so no cross-lane shuffle of P is needed.
// 4 waves x 32 queries per workgroup share the 64-key K / V^T tiles, double-buffered in LDS through
// 16-byte LDS-DMA with the bank swizzle on the source chunk index (same scheme as gemm.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "kernels.h"
#include "prof.h"

namespace r3g {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int KV_TILE = 64;
constexpr int TILE_B = KV_TILE * 64 * 2;  // 8 KiB: K tile [64 keys][64 d] or V^T tile [64 d][64 keys]
constexpr int STAGE_B = 2 * TILE_B;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// v_cvt_pk_bf16_f32 (round to nearest even)
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<const uint32_t*>(&h);
}

template <bool GLDS>
__device__ __forceinline__ void stage_kv(const uint16_t* __restrict__ Kg, const uint16_t* __restrict__ Vtg,
                                         int64_t ldv, int key0, char* lds, int wid, int lane, int tid) {
    if constexpr (GLDS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;  // 8 rows of 128 B
            const int row = piece * 8 + (lane >> 3);
            const int kc = (lane & 7) ^ ((row >> 1) & 7);
            const uint16_t* gk = Kg + (int64_t)(key0 + row) * 64 + kc * 8;
            const uint16_t* gv = Vtg + (int64_t)row * ldv + key0 + kc * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gk,
                                             (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gv,
                                             (__attribute__((address_space(3))) void*)(lds + TILE_B + piece * 1024), 16,
                                             0, 0);
        }
    } else {
        uint4 vk[2], vv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * 256 + tid;
            const int row = c >> 3, kc = c & 7;
            vk[i] = *reinterpret_cast<const uint4*>(Kg + (int64_t)(key0 + row) * 64 + kc * 8);
            vv[i] = *reinterpret_cast<const uint4*>(Vtg + (int64_t)row * ldv + key0 + kc * 8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * 256 + tid;
            const int row = c >> 3, kc = c & 7;
            const int off = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
            *reinterpret_cast<uint4*>(lds + off) = vk[i];
            *reinterpret_cast<uint4*>(lds + TILE_B + off) = vv[i];
        }
    }
}

// Ragged mode: work entry `e` of the launch, read from the kernarg segment at a uniform offset (scalar loads; indexing the
// by-value kernel argument with a run-time index would make the compiler copy it to scratch).  Plain mode: entry e is batch e.
__device__ __forceinline__ int attn_entry_lq(const AttnArgs& p, int e) {
    typedef const char __attribute__((address_space(4)))* kernarg_ptr;
    kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    return *(const int __attribute__((address_space(4)))*)(ka + offsetof(AttnArgs, ent) + (size_t)e * sizeof(AttnEntry) +
                                                            offsetof(AttnEntry, lq));
}
__device__ __forceinline__ AttnEntry attn_entry(const AttnArgs& p, int e) {
    AttnEntry r{};
    if (!p.ragged) {
        r.lq = p.Lq; r.lk = p.Lk; r.bias_key = -1; r.buf = e;
        return r;
    }
    typedef const char __attribute__((address_space(4)))* kernarg_ptr;
    kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    const AttnEntry __attribute__((address_space(4)))* t =
        (const AttnEntry __attribute__((address_space(4)))*)(ka + offsetof(AttnArgs, ent) + (size_t)e * sizeof(AttnEntry));
    r.lq = t->lq; r.lk = t->lk; r.bias_key = t->bias_key; r.bias_log2 = t->bias_log2; r.buf = t->buf;
    return r;
}
// output row of query q of entry e (read at the epilogue only: the three values would otherwise stay live through the key loop)
__device__ __forceinline__ int64_t attn_entry_out_row(int e, int q) {
    typedef const char __attribute__((address_space(4)))* kernarg_ptr;
    kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    const AttnEntry __attribute__((address_space(4)))* t =
        (const AttnEntry __attribute__((address_space(4)))*)(ka + offsetof(AttnArgs, ent) + (size_t)e * sizeof(AttnEntry));
    const int split = t->o_split;
    return q < split ? t->o_row0 + q : t->o_row_split + (q - split);
}
// (head, entry, query tile) of work item `item`: heads outermost, inside a head entry by entry
template <int QT>
__device__ __forceinline__ void attn_work_item(const AttnArgs& p, int item, int& eb, int& hd, int& qt) {
    if (p.ragged) {
        int per_head = 0;
        for (int e = 0; e < p.B; ++e) per_head += (attn_entry_lq(p, e) + QT - 1) / QT;
        hd = item / per_head;
        int rem = item - hd * per_head;
        eb = 0;
        for (;;) {
            const int n = (attn_entry_lq(p, eb) + QT - 1) / QT;
            if (rem < n || eb + 1 >= p.B) break;
            rem -= n;
            ++eb;
        }
        qt = rem;
    } else {
        const int ntq0 = (p.Lq + QT - 1) / QT;
        const int per_head = ntq0 * p.B;
        hd = item / per_head;
        const int rem = item - hd * per_head;
        eb = rem / ntq0;
        qt = rem - eb * ntq0;
    }
}

// ABL: timing-only ablation mask (tools/bench_attn.py; results are garbage for ABL != 0): 1 no exp2, 2 no row max /
// rescale, 4 no PV MFMAs, 8 no QK^T MFMAs, 16 no LDS-DMA in the loop, 32 no per-tile wait + barrier, 64 no V^T reads
template <bool GLDS, int ABL = 0>
__global__ __launch_bounds__(256, 2) void attn_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hd = blockIdx.y;
    const int eb = blockIdx.z;
    const AttnEntry en = attn_entry(p, eb);
    const int b = en.buf;
    const int ql = lane & 31, hh = lane >> 5;
    const int q = blockIdx.x * 128 + wid * 32 + ql;
    const int Lq = en.lq, Lk = en.lk;
    if ((int)blockIdx.x * 128 >= Lq) return;  // ragged: shorter batch entries have fewer query tiles
    const int kvb = p.kv_batch_stride_zero ? 0 : b;
    const uint16_t* Qg = p.Q + (((int64_t)b * p.H + hd) * p.Lq_pad) * 64;
    const uint16_t* Kg = p.K + (((int64_t)kvb * p.H + hd) * p.Lk_pad) * 64;
    const uint16_t* Vtg = p.Vt + (((int64_t)kvb * p.H + hd) * 64) * (int64_t)p.Lk_pad;

    // Q^T fragments: B operand, B[k = 8*hh + j][n = ql] = Q[q][16*ks + 8*hh + j]
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(Qg + (int64_t)q * 64 + ks * 16 + hh * 8);

    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.scale * 1.4426950408889634f;  // scores in log2 units

    // LDS read offsets.  K fragment: row = key (kb*32 + ql), chunk = 2*ks + hh.
    // V^T fragment: row = d (db*32 + ql), first 8-byte piece at chunk 4*kb + 2*ks2, +8*hh bytes; second piece in chunk+1.
    int offK[2], offV[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int row = kb * 32 + ql;
        offK[kb] = row * 128 + ((hh ^ ((row >> 1) & 7)) << 4);  // chunk hh; ks adds (2*ks)<<4 via XOR on bits 5..6
        offV[kb] = row * 128 + ((((row >> 1) & 7)) << 4) + 8 * hh;  // chunk 0 swizzled; chunk c via XOR (c<<4)
    }

    const int ntiles = (Lk + KV_TILE - 1) / KV_TILE;
    const int bias_key = en.bias_key;
    const float bias_raw = p.ragged ? en.bias_log2 / sc : 0.f;  // added to the raw (unscaled) score
    stage_kv<GLDS>(Kg, Vtg, p.Lk_pad, 0, smem, wid, lane, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto tile = [&](auto masked, const int t) {
        const char* cur = smem + (t & 1) * STAGE_B;
        if (!(ABL & 16) && t + 1 < ntiles)
            stage_kv<GLDS>(Kg, Vtg, p.Lk_pad, (t + 1) * KV_TILE, smem + ((t + 1) & 1) * STAGE_B, wid, lane, tid);

        // ---- S^T = K Q^T for the two 32-key blocks
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = (ABL & 8) ? (float)(lane + r) * 0.01f : 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(cur + (offK[kb] ^ (ks << 5)));
                if (ABL & 8) asm volatile("" ::"v"(kf));
                else s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
            }
        }
        // ---- online softmax on the raw scores (scale folded into the exp2 argument);
        //      reg r of block kb <-> key kb*32 + (r&3) + 8*(r>>2) + 4*hh.  Only the last tile can hold padded keys.
        if constexpr (decltype(masked)::value) {
            const int key_base = t * KV_TILE + 4 * hh;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                {
                    const int key = key_base + kb * 32 + (r & 3) + 8 * (r >> 2);
                    if (key >= Lk) s[kb][r] = -INFINITY;
                    else if (key == bias_key) s[kb][r] += bias_raw;
                }
        }
        float mloc = s[0][0];
        if (!(ABL & 2)) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[kb][r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        }
        // rescale only when some query's running max actually grows (wave-uniform branch; exact, not a threshold)
        if (!(ABL & 2) && __any(mloc > m_run)) {
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        const float mb = m_run * sc;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float arg = __builtin_fmaf(s[kb][r], sc, -mb);
                const float pv = (ABL & 1) ? arg : __builtin_amdgcn_exp2f(arg);
                s[kb][r] = pv;
                psum += pv;
            }
        l_run += psum;

        // ---- O^T += V^T P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                union { uint32_t u[4]; bf16x8 v; } pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) pf.u[e] = pack_bf16(s[kb][8 * ks2 + 2 * e], s[kb][8 * ks2 + 2 * e + 1]);
                const int c = 4 * kb + 2 * ks2;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    union { bf16x4 h[2]; bf16x8 v; } vf;
                    if (ABL & 64) {
                        vf.v = qf[(c + db) & 3];
                    } else {   // the lane's 8 keys are one 16-byte chunk of the V^T row (vt_key_pos layout)
                        vf.v = *reinterpret_cast<const bf16x8*>(cur + TILE_B + (offK[db] ^ (c << 4)));
                    }
                    if (ABL & 4) asm volatile("" ::"v"(vf.v), "v"(pf.v));
                    else o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o[db], 0, 0, 0);
                }
            }
        if (!(ABL & 32)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    };
    const bool pad_tail = ntiles * KV_TILE > Lk;
    for (int t = 0; t < ntiles - 1; ++t) tile(std::false_type{}, t);
    if (pad_tail) tile(std::true_type{}, ntiles - 1);
    else tile(std::false_type{}, ntiles - 1);

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    // Output: lane (q, hh) holds dims db*32 + 8g + 4hh + {0..3} of query q.  The two lanes of a query swap halves so
    // that each writes 8 consecutive dims: 8 dwordx4 stores per lane instead of 16 dwordx2 (the store tail of a
    // workgroup is issue bound).  The exchange runs for every lane (a padded query's partner is padded too).
    {
        int64_t orow = (int64_t)b * p.strideO + (int64_t)q * p.ldo;
        if (p.ragged) orow = attn_entry_out_row(eb, q) * p.ldo;
        uint16_t* dst = p.O + orow + hd * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint2 even, odd;   // this lane's packed dims of g = 2gp and g = 2gp + 1
                even.x = pack_bf16(o[db][8 * gp] * inv, o[db][8 * gp + 1] * inv);
                even.y = pack_bf16(o[db][8 * gp + 2] * inv, o[db][8 * gp + 3] * inv);
                odd.x = pack_bf16(o[db][8 * gp + 4] * inv, o[db][8 * gp + 5] * inv);
                odd.y = pack_bf16(o[db][8 * gp + 6] * inv, o[db][8 * gp + 7] * inv);
                // hh = 0 keeps `even` and receives the partner's `even` (dims +4..7); hh = 1 keeps `odd`, receives `odd`
                const uint2 give = hh ? even : odd;
                uint2 got;
                got.x = (uint32_t)__shfl_xor((int)give.x, 32, 64);
                got.y = (uint32_t)__shfl_xor((int)give.y, 32, 64);
                const uint2 mine = hh ? odd : even;
                uint4 out;
                if (hh) { out.x = got.x; out.y = got.y; out.z = mine.x; out.w = mine.y; }
                else { out.x = mine.x; out.y = mine.y; out.z = got.x; out.w = got.y; }
                if (q < Lq) *reinterpret_cast<uint4*>(dst + db * 32 + 16 * gp + 8 * hh) = out;
            }
    }
}


// ------------------------------------------------------------------------------------------------------------
// Software-pipelined variant: inside one wave the QK^T MFMAs of key tile t+1 are issued in the same basic block as
// the exponentials / sums / bf16 packing of tile t, and the row-max reduction of tile t+1 sits beside the PV MFMAs of
// tile t (an in-order wave only overlaps its matrix and vector pipes when the two instruction kinds are interleaved
// in program order).  K tiles therefore live in a 3-deep LDS ring (loaded two tiles ahead), V^T tiles in a 2-deep one.
template <bool GLDS>
__device__ __forceinline__ void stage_half(const uint16_t* __restrict__ g, int64_t ld, int64_t row_elems0, int col0,
                                           char* lds, int wid, int lane, int tid) {
    // one 8 KiB tile = 64 rows of 128 B; source row r starts at g + (row_elems0 + r*ld) + col0
    if constexpr (GLDS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;
            const int row = piece * 8 + (lane >> 3);
            const int kc = (lane & 7) ^ ((row >> 1) & 7);
            const uint16_t* src = g + row_elems0 + (int64_t)row * ld + col0 + kc * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
        }
    } else {
        uint4 v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * 256 + tid;
            const int row = c >> 3, kc = c & 7;
            v[i] = *reinterpret_cast<const uint4*>(g + row_elems0 + (int64_t)row * ld + col0 + kc * 8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * 256 + tid;
            const int row = c >> 3, kc = c & 7;
            *reinterpret_cast<uint4*>(lds + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4)) = v[i];
        }
    }
}

template <bool GLDS>
__global__ __launch_bounds__(256, 2) void attn_kernel_sp(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[5 * TILE_B];  // K ring: 0,1,2 ; V^T ring: 3,4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hd = blockIdx.y;
    const int eb = blockIdx.z;
    const AttnEntry en = attn_entry(p, eb);
    const int b = en.buf;
    const int ql = lane & 31, hh = lane >> 5;
    const int q = blockIdx.x * 128 + wid * 32 + ql;
    const int Lq = en.lq, Lk = en.lk;
    if ((int)blockIdx.x * 128 >= Lq) return;
    const int kvb = p.kv_batch_stride_zero ? 0 : b;
    const uint16_t* Qg = p.Q + (((int64_t)b * p.H + hd) * p.Lq_pad) * 64;
    const uint16_t* Kg = p.K + (((int64_t)kvb * p.H + hd) * p.Lk_pad) * 64;
    const uint16_t* Vtg = p.Vt + (((int64_t)kvb * p.H + hd) * 64) * (int64_t)p.Lk_pad;

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(Qg + (int64_t)q * 64 + ks * 16 + hh * 8);
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.scale * 1.4426950408889634f;
    int offK[2], offV[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int row = kb * 32 + ql;
        offK[kb] = row * 128 + ((hh ^ ((row >> 1) & 7)) << 4);
        offV[kb] = row * 128 + ((((row >> 1) & 7)) << 4) + 8 * hh;
    }
    const int ntiles = (Lk + KV_TILE - 1) / KV_TILE;
    const bool pad_tail = ntiles * KV_TILE > Lk;
    const int bias_key = en.bias_key;
    const float bias_raw = p.ragged ? en.bias_log2 / sc : 0.f;

    auto stage_k = [&](int t, int slot) { stage_half<GLDS>(Kg, 64, (int64_t)t * KV_TILE * 64, 0, smem + slot * TILE_B, wid, lane, tid); };
    auto stage_v = [&](int t, int slot) { stage_half<GLDS>(Vtg, p.Lk_pad, 0, t * KV_TILE, smem + (3 + slot) * TILE_B, wid, lane, tid); };
    auto qk = [&](const char* kt, f32x16 (&sv)[2]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt + (offK[kb] ^ (ks << 5)));
                sv[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sv[kb], 0, 0, 0);
            }
        }
    };

    stage_k(0, 0);
    stage_v(0, 0);
    if (ntiles > 1) stage_k(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 s_cur[2], s_nxt[2];
    qk(smem, s_cur);
    float mloc_nxt = 0.f;  // row max of s_nxt, reduced beside the PV MFMAs

    auto iter = [&](auto masked, auto has_next, const int t, const int kslot_next, const int vslot) {
        if (t + 2 < ntiles) stage_k(t + 2, kslot_next == 2 ? 0 : kslot_next + 1);
        if (t + 1 < ntiles) stage_v(t + 1, vslot ^ 1);
        // ---- row max of tile t and (rarely) the rescale of the running state
        float mloc;
        if constexpr (decltype(masked)::value) {
            const int key_base = t * KV_TILE + 4 * hh;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key_base + kb * 32 + (r & 3) + 8 * (r >> 2);
                    if (key >= Lk) s_cur[kb][r] = -INFINITY;
                    else if (key == bias_key) s_cur[kb][r] += bias_raw;
                }
            mloc = s_cur[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s_cur[kb][r]);
        } else {
            if (t == 0) {
                mloc = s_cur[0][0];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s_cur[kb][r]);
            } else {
                mloc = mloc_nxt;
            }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        if (__any(mloc > m_run)) {
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        const float mb = m_run * sc;
        // ---- region A: QK^T of tile t+1 (matrix pipe)  ||  exp / sum / pack of tile t (vector pipe)
        if constexpr (decltype(has_next)::value) qk(smem + kslot_next * TILE_B, s_nxt);
        float psum = 0.f;
        uint32_t pk[2][8];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[kb][2 * e], sc, -mb));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[kb][2 * e + 1], sc, -mb));
                psum += p0 + p1;
                pk[kb][e] = pack_bf16(p0, p1);
            }
        l_run += psum;
        if constexpr (decltype(has_next)::value) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x402, 14, 0);  // 14 VALU / TRANS
            }
        }
        // ---- region B: PV of tile t (matrix pipe)  ||  row max of tile t+1 (vector pipe)
        const char* vt = smem + (3 + vslot) * TILE_B;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                union { uint32_t u[4]; bf16x8 v; } pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) pf.u[e] = pk[kb][4 * ks2 + e];
                const int c = 4 * kb + 2 * ks2;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vfv = *reinterpret_cast<const bf16x8*>(vt + (offK[db] ^ (c << 4)));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfv, pf.v, o[db], 0, 0, 0);
                }
            }
        if constexpr (decltype(has_next)::value) {
            float mx = s_nxt[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_nxt[kb][r]);
            mloc_nxt = mx;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (decltype(has_next)::value) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) s_cur[kb] = s_nxt[kb];
        }
    };

    int ks_next = 1, vs = 0;  // LDS slots of K(t+1) and V(t)
    for (int t = 0; t < ntiles - 1; ++t) {
        iter(std::false_type{}, std::true_type{}, t, ks_next, vs);
        ks_next = ks_next == 2 ? 0 : ks_next + 1;
        vs ^= 1;
    }
    if (pad_tail) iter(std::true_type{}, std::false_type{}, ntiles - 1, ks_next, vs);
    else iter(std::false_type{}, std::false_type{}, ntiles - 1, ks_next, vs);

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q < Lq) {
        int64_t orow = (int64_t)b * p.strideO + (int64_t)q * p.ldo;
        if (p.ragged) orow = attn_entry_out_row(eb, q) * p.ldo;
        uint16_t* dst = p.O + orow + hd * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk2;
                pk2.x = pack_bf16(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
                pk2.y = pack_bf16(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(dst + db * 32 + 8 * g + 4 * hh) = pk2;
            }
    }
}


// ------------------------------------------------------------------------------------------------------------
// Second-generation kernel (default).  Same decomposition (4 waves x 32 queries share 64-key K / V^T tiles; S^T = K Q^T
// and O^T += V^T P^T on 32x32x16 MFMAs so that one query lives in lanes q and q+32), with the vector work per score
// cut to what the exponential needs:
//  * Q arrives pre-multiplied by scale*log2(e) (the QKV epilogue does it in fp32 before rounding), so a score is
//    already the exp2 argument;
//  * lagging maximum: the score accumulators START at -m (the running stabiliser, kept as a 16-register block that is
//    the C operand of the first QK^T MFMA), so p = exp2(s) with no subtraction per score.  The stabiliser only has to be
//    CLOSE to the maximum: the lane maximum of the 16 shifted scores (8 v_max3) is compared with SCORE_LIMIT and a
//    wave-uniform branch fires when a VALID query's score exceeds it.  Only then is the block's true maximum taken and
//    (O, l, the stabiliser block, the scores) moved by it, in place, before the exponentials -- nothing of the block has
//    entered O or l, an overflowed p is never formed, and no value of the rare path is merged with one of the common path
//    behind the branch (that merge cost 24 register copies per key block while the test sat behind the exponentials);
//    rows past Lq hold stale data and take no part in the decision (a valid query's rounding must not depend on them);
//  * one 32-key score block is live at a time (QK^T -> exp2 -> pack -> PV per block): 16 score registers;
//  * V^T tiles are read with ds_read_b128: the 8 keys a lane feeds into one PV MFMA are contiguous in the V^T layout
//    (vt_key_pos in kernels.h: inside every aligned group of 16 keys the two middle 4-key blocks are swapped);
//  * 1-D grid, work items ordered head-major and cut into 8 contiguous runs, one per XCD (block b runs on XCD b % 8):
//    an XCD walks the query tiles of one head after the other, so its K / V^T stay in its 4 MiB L2.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ float half_max(float x) {   // max over lanes q and q+32
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

constexpr float PSUM_LIMIT = 1024.f;   // (pipelined body) a block whose 16 p of one lane sum to more is re-stabilised
// Round 6 variants of the generation-2 / generation-6 bodies (template VAR, option "attn_variant"; profiles/r06_attention.md):
//  bit 0  after the first key block (whose exact maximum starts the stabiliser) the FAST pass takes no maximum at all: the 8 v_max3,
//         the compare and the branch per key block leave the common path, and a key tile becomes one basic block.  The stabiliser
//         only has to keep exp2 inside the fp32 range, so the test is on the lane's SUM of its 16 exponentials (which the row sum
//         needs anyway) against a generous bound (2^16, or NaN) and only sets a sticky flag; when a VALID query of the workgroup
//         set it, the whole query tile runs again as the SAFE pass = variant 0's body, and nothing of the fast pass is used.
//         Without a re-stabilisation the results are bit-identical to variant 0; a tile that takes the safe pass is bit-identical
//         to it too; where variant 0 would have moved the stabiliser (a score more than 8 above it) and the fast pass does not
//         (sum <= 2^16), P is rounded at another scale: equal within the bf16 rounding of P.
// (A second variant bit -- the row sum on plain v_add_f32 through inline asm, because MI355X_MICROARCH.md prices a v_pk_add_f32 beside
// MFMAs above two plain adds -- measured +2 % on both kernels and was REMOVED: the compiler does not see that an inline-asm add
// reads the result of a v_exp_f32 issued just before it (the transcendental-use hazard needs a wait state it only inserts for
// instructions it knows), and the 64-query kernel produced wrong sums; profiles/r06_attention.md.)
constexpr float PSUM_LIMIT2 = 65536.f;
// sum of 16 values as even / odd partial sums on the packed fp32 adder: (((e0 + e2) + ... + e14) + ((o1 + o3) + ... + o15))
__device__ __forceinline__ float sum16(const float (&pe)[16]) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 ps = (f2){pe[0], pe[1]};
#pragma unroll
    for (int r = 2; r < 16; r += 2) ps += (f2){pe[r], pe[r + 1]};
    return ps[0] + ps[1];
}
constexpr float SCORE_LIMIT = 8.f;     // a block with a score more than 8 (log2 units) above its stabiliser (p > 256) is re-stabilised

// D = A B + C with D in registers DISTINCT from C (hipcc ties vdst to srcC for the builtin and copies the 16 registers
// of the stabiliser block first: 32 v_mov per key tile).  The s_nop covers a VALU write of an operand just before.
__device__ __forceinline__ f32x16 mfma_32x32x16_fresh(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    f32x16 d;
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// PIPE: software-pipelined body -- the QK^T MFMAs of the NEXT 32-key block are issued ahead of the exponentials of the
// current one (an in-order wave overlaps its matrix and vector pipes only when the two kinds alternate in program
// order: ~5 vector instructions fit into the shadow of one 32x32x16 MFMA), two score blocks live, 3-deep K / V^T ring.
// NW: waves per workgroup (4 or 8), 32 queries each.  All of them share the K / V^T tiles: with 8 waves the stream from
// L2 into LDS -- measured at ~16 B/clk/CU with three 4-wave workgroups per CU, the most a CU sustains -- is a third.
// ABL: timing-only ablation mask of the non-pipelined body (1 no exp2, 4 no PV MFMAs, 8 no QK^T MFMAs, 16 no LDS-DMA in
// the loop, 32 no per-tile wait + barrier, 64 no V^T reads, 128 no K reads); results are garbage for ABL != 0.
// (Round 5 also tried this body at FOUR waves per SIMD -- four workgroups per compute unit, registers capped at 128: 22 spilled
// registers inside the key loop, 905 against 995 TFLOP/s on the geo decoder's passes; profiles/r05_attention_phases.md.)
template <bool GLDS, bool PIPE, int NW, int ABL = 0, int VAR = 0>
__global__ __launch_bounds__(NW * 64, (PIPE || NW == 8) ? 2 : 3) void attn2_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[(PIPE ? 3 : 2) * STAGE_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    // ---- work item: XCD-aware remap of the 1-D grid, then (head, batch, query tile)
    int eb, hd, qt;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int qn = nwg >> 3, rn = nwg & 7, xcd = orig & 7;
        const int item = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (orig >> 3);
        constexpr int QT = NW * 32;
        attn_work_item<QT>(p, item, eb, hd, qt);
    }
    const AttnEntry en = attn_entry(p, eb);
    const int b = en.buf;
    const int q = qt * (NW * 32) + wid * 32 + ql;
    const int Lq = en.lq, Lk = en.lk;
    const int kvb = p.kv_batch_stride_zero ? 0 : b;
    const uint16_t* Qg = p.Q + (((int64_t)b * p.H + hd) * p.Lq_pad) * 64;
    const uint16_t* Kg = p.K + (((int64_t)kvb * p.H + hd) * p.Lk_pad) * 64;
    const uint16_t* Vtg = p.Vt + (((int64_t)kvb * p.H + hd) * 64) * (int64_t)p.Lk_pad;

    bf16x8 qf[4];
    {
        const int qrow = q < p.Lq_pad ? q : p.Lq_pad - 1;   // a 256-query tile may reach past the 128-aligned allocation
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[ks] = *reinterpret_cast<const bf16x8*>(Qg + (int64_t)qrow * 64 + ks * 16 + hh * 8);
    }
    if (!p.q_prescaled) {   // test entry point with plain Q: fold scale * log2(e) here (a second bf16 rounding)
        const float sc = p.scale * 1.4426950408889634f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            union { bf16x8 v; uint32_t u[4]; } w;
            w.v = qf[ks];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w.u[e] = pack_bf16(__uint_as_float(w.u[e] << 16) * sc, __uint_as_float(w.u[e] & 0xFFFF0000u) * sc);
            qf[ks] = w.v;
        }
    }

    f32x16 o[2], negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
    float m_run = 0.f, l_run = 0.f;

    // LDS fragment offsets: tile row = lane & 31 (+ 32 for the second row block), 16-byte chunk hh, swizzled by the row;
    // chunk 2*ks + hh of a K row (dims) and chunk 2*(2*kb + ks2) + hh of a V^T row (key positions) follow by XOR
    int off[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int row = rb * 32 + ql;
        off[rb] = row * 128 + ((hh ^ ((row >> 1) & 7)) << 4);
    }

    const int ntiles = (Lk + KV_TILE - 1) / KV_TILE;
    const int bias_key = en.bias_key;
    const float bias_l2 = p.ragged ? en.bias_log2 : 0.f;
    // staging: this lane's two 16-byte chunks of a K tile and of a V^T tile; the per-lane element offsets do not depend on
    // the tile (wave-uniform tile base + constant lane offset: no vector address arithmetic inside the loop)
    constexpr int PPW = 8 / NW;   // 1 KiB pieces (8 tile rows) of each of the two tiles per wave
    // (round 6: BYTE offsets as unsigned 32-bit values -- wave-uniform tile base + zero-extended lane offset is the scalar-base
    // form of the LDS-DMA load, so a key tile costs no 64-bit vector address arithmetic; a V^T row offset stays below 2^32 bytes
    // for any Lk_pad this kernel is launched with: 64 rows x Lk_pad x 2 bytes)
    uint32_t koff[PPW], voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int row = (wid * PPW + i) * 8 + (lane >> 3);
        const int kc = (lane & 7) ^ ((row >> 1) & 7);
        koff[i] = (uint32_t)(row * 64 + kc * 8) * 2u;
        voff[i] = ((uint32_t)row * (uint32_t)p.Lk_pad + (uint32_t)(kc * 8)) * 2u;
    }
    auto stage = [&](int t, char* dst) {
        if constexpr (GLDS) {
            const char* kt = reinterpret_cast<const char*>(Kg + (int64_t)t * (KV_TILE * 64));
            const char* vt = reinterpret_cast<const char*>(Vtg + (int64_t)t * KV_TILE);
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kt + (size_t)koff[i]),
                                                 (__attribute__((address_space(3))) void*)(dst + (wid * PPW + i) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vt + (size_t)voff[i]),
                                                 (__attribute__((address_space(3))) void*)(dst + TILE_B + (wid * PPW + i) * 1024),
                                                 16, 0, 0);
            }
        } else {
            static_assert(GLDS || NW == 4, "register staging is only built for 4-wave workgroups");
            stage_kv<false>(Kg, Vtg, p.Lk_pad, t * KV_TILE, dst, wid, lane, tid);
        }
    };
    const bool pad_tail = ntiles * KV_TILE > Lk;
    if constexpr (PIPE) {
        // scores of block kb of the tile in LDS slot `buf`; C = the stabiliser block (or zero for the very first block)
        auto qk = [&](const char* buf, int kb, bool from_zero) {
            f32x16 sc_;
            const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(buf + off[kb]);
            if (from_zero) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                sc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[0], z, 0, 0, 0);
            } else {
                sc_ = mfma_32x32x16_fresh(k0, qf[0], negm);
            }
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(buf + (off[kb] ^ (ks << 5)));
                sc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sc_, 0, 0, 0);
            }
            return sc_;
        };
        auto mask_block = [&](f32x16& sc_, int t, int kb) {
            const int key_base = t * KV_TILE + kb * 32 + 4 * hh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key_base + (r & 3) + 8 * (r >> 2);
                if (key >= Lk) sc_[r] = -INFINITY;
                else if (key == bias_key) sc_[r] += bias_l2;
            }
        };
        // exponentials, row sum and bf16 packing of one block; `ahead` = the scores already computed (against the same
        // stabiliser) for the following block, corrected when the stabiliser moves
        auto softmax_block = [&](f32x16& sc_, f32x16* ahead, uint32_t (&pk)[8]) {
            float pe[16];
            float ps0, ps1;
            auto expsum = [&]() {
#pragma unroll
                for (int r = 0; r < 16; ++r) pe[r] = __builtin_amdgcn_exp2f(sc_[r]);
                ps0 = 0.f;
                ps1 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r += 2) { ps0 += pe[r]; ps1 += pe[r + 1]; }
                ps0 += ps1;
            };
            expsum();
            if (__any(!(ps0 <= PSUM_LIMIT))) {   // rare: re-stabilise before anything of this block is used
                float mx = sc_[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc_[r]);
                const float grow = fmaxf(half_max(mx), 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-grow);
                m_run += grow;
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; negm[r] = -m_run; sc_[r] -= grow; }
                if (ahead) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) (*ahead)[r] -= grow;
                }
                expsum();
            }
            l_run += ps0;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk[e] = pack_bf16(pe[2 * e], pe[2 * e + 1]);
        };
        auto pv = [&](const char* buf, int kb, const uint32_t (&pk)[8]) {
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                union { uint32_t u[4]; bf16x8 v; } pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) pf.u[e] = pk[4 * ks2 + e];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(buf + TILE_B + (off[db] ^ ((2 * kb + ks2) << 5)));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, o[db], 0, 0, 0);
                }
            }
        };
        stage(0, smem);
        if (ntiles > 1) stage(1, smem + STAGE_B);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // very first block: scores from zero, exact maximum as the initial stabiliser
        f32x16 sA = qk(smem, 0, true), sB;
        if (ntiles == 1 && pad_tail) mask_block(sA, 0, 0);
        {
            float mx = sA[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sA[r]);
            m_run = half_max(mx);
#pragma unroll
            for (int r = 0; r < 16; ++r) { negm[r] = -m_run; sA[r] -= m_run; }
        }
        auto tile = [&](auto masked, auto has_next, const int t, const int slot) {
            const char* cur = smem + slot * STAGE_B;
            const char* nxt = smem + (slot == 2 ? 0 : slot + 1) * STAGE_B;
            if (t + 2 < ntiles) stage(t + 2, smem + (slot == 0 ? 2 : slot - 1) * STAGE_B);
            uint32_t pk[8];
            // block 0 of this tile (scores sA, computed one step ago); block 1's scores are issued first
            sB = qk(cur, 1, false);
            if constexpr (decltype(masked)::value) {
                if (t > 0) mask_block(sA, t, 0);     // (the single-tile case masked its first block in the prologue)
                mask_block(sB, t, 1);
            }
            softmax_block(sA, &sB, pk);
            pv(cur, 0, pk);
            // block 1 (scores sB); the first block of the next tile is issued first
            if constexpr (decltype(has_next)::value) {
                sA = qk(nxt, 0, false);
                softmax_block(sB, &sA, pk);
            } else {
                softmax_block(sB, nullptr, pk);
            }
            pv(cur, 1, pk);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        int slot = 0;
        for (int t = 0; t < ntiles - 1; ++t) {
            tile(std::false_type{}, std::true_type{}, t, slot);
            slot = slot == 2 ? 0 : slot + 1;
        }
        if (pad_tail) tile(std::true_type{}, std::false_type{}, ntiles - 1, slot);
        else tile(std::false_type{}, std::false_type{}, ntiles - 1, slot);
    } else {
    // variant bit 0 (round 6, comment at PSUM_LIMIT2): the FAST pass takes no maximum after the first key block -- a lane whose 16
    // exponentials sum past the bound (or to NaN) only sets a sticky flag -- and when a valid query of the workgroup set it, the
    // whole query tile runs again as the SAFE pass (the body of variant 0, bit for bit); nothing of the fast pass is used then.
    unsigned long long sticky = 0;
    const unsigned long long valid_lanes = __ballot(q < Lq);   // rows past Lq (stale data) set no flag
    auto pass = [&](auto FAST) __attribute__((always_inline)) {
    constexpr bool kFast = decltype(FAST)::value;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
    m_run = 0.f;
    l_run = 0.f;
    stage(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto tile = [&](auto masked, auto first, const int t) {
        const char* cur = smem + (t & 1) * STAGE_B;
        if (!(ABL & 16) && t + 1 < ntiles) stage(t + 1, smem + ((t + 1) & 1) * STAGE_B);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            constexpr bool kFirst = decltype(first)::value;
            const bool first_block = kFirst && kb == 0;
            // ---- scores of 32 keys (log2 units, already minus the stabiliser unless this is the very first block)
            f32x16 s;
            if constexpr ((ABL & 8) != 0) {
                const float tv = __int_as_float(0x3f800000 + (t << 8));   // varies with the tile: not loop invariant
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = negm[r] + tv * (float)(r + 1) * 1e-3f;
                if (!(ABL & 128)) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(cur + (off[kb] ^ (ks << 5)));
                        asm volatile("" ::"v"(kf));
                    }
                }
            } else {
                const bf16x8 k0 = (ABL & 128) ? qf[1] : *reinterpret_cast<const bf16x8*>(cur + off[kb]);
                if (first_block) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[0], z, 0, 0, 0);
                } else {
                    s = mfma_32x32x16_fresh(k0, qf[0], negm);
                }
#pragma unroll
                for (int ks = 1; ks < 4; ++ks) {
                    const bf16x8 kf = (ABL & 128) ? qf[(ks + 1) & 3]
                                                  : *reinterpret_cast<const bf16x8*>(cur + (off[kb] ^ (ks << 5)));
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
                }
            }
            if constexpr (decltype(masked)::value) {   // only the last tile holds padded keys / the weighted key
                const int key_base = t * KV_TILE + kb * 32 + 4 * hh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key_base + (r & 3) + 8 * (r >> 2);
                    if (key >= Lk) s[r] = -INFINITY;
                    else if (key == bias_key) s[r] += bias_l2;
                }
            }
            // lane maximum of the 16 scores (relative to the stabiliser): decides BEFORE the exponentials whether the block
            // has to be re-stabilised, so that the rare path only rewrites values in place (O, l, the stabiliser block and
            // the scores, all by arithmetic on themselves) and nothing computed on it is merged with a value of the common
            // path afterwards -- with the test behind the exponentials (on their sum) the join copied 16 exponentials
            // and the 16-register stabiliser block on the common path of every key block (32 v_mov)
            float mx = 0.f;
            if (first_block || !kFast) {
                mx = fmaxf(s[0], s[1]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
            }
            if (first_block) {   // the stabiliser starts as the exact maximum of the first 32 keys
                m_run = half_max(mx);
#pragma unroll
                for (int r = 0; r < 16; ++r) { negm[r] = -m_run; s[r] -= m_run; }
            } else if (!kFast && __any(q < Lq && !(mx <= SCORE_LIMIT))) {
                // (rows past Lq hold stale data: they must not decide anything, or a valid query's rounding would depend on it)
                // rare: some query's scores outgrew its stabiliser (or are NaN).  s is relative to the old one: the growth
                // is the block maximum itself.  Nothing of this block has entered O or l yet.
                const float grow = fmaxf(half_max(mx), 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-grow);
                m_run += grow;
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; negm[r] -= grow; s[r] -= grow; }
            }
            float pe[16];
            uint32_t pk[8];
#pragma unroll
            for (int r = 0; r < 16; ++r) pe[r] = (ABL & 1) ? s[r] * 0.001f : __builtin_amdgcn_exp2f(s[r]);
#pragma unroll
            for (int e = 0; e < 8; ++e) pk[e] = pack_bf16(pe[2 * e], pe[2 * e + 1]);
            {   // even / odd partial sums: packed fp32 adder, or (variant bit 1) plain adds in the same order
                const float psum = sum16(pe);
                l_run += psum;
                if (kFast && !first_block) sticky |= __ballot(!(psum <= PSUM_LIMIT2)) & valid_lanes;
            }
            // ---- O^T += V^T P^T for these 32 keys: the lane's 8 keys of each 16-key group are one 16-byte chunk of V^T
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                union { uint32_t u[4]; bf16x8 v; } pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) pf.u[e] = pk[4 * ks2 + e];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vf = (ABL & 64) ? qf[(ks2 + db) & 3]
                                                 : *reinterpret_cast<const bf16x8*>(cur + TILE_B + (off[db] ^ ((2 * kb + ks2) << 5)));
                    if (ABL & 4) asm volatile("" ::"v"(vf), "v"(pf.v));
                    else o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, o[db], 0, 0, 0);
                }
            }
        }
        if (!(ABL & 32)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    };
    if (ntiles == 1) {
        if (pad_tail) tile(std::true_type{}, std::true_type{}, 0);
        else tile(std::false_type{}, std::true_type{}, 0);
    } else {
        tile(std::false_type{}, std::true_type{}, 0);
        for (int t = 1; t < ntiles - 1; ++t) tile(std::false_type{}, std::false_type{}, t);
        if (pad_tail) tile(std::true_type{}, std::false_type{}, ntiles - 1);
        else tile(std::false_type{}, std::false_type{}, ntiles - 1);
    }
    };   // pass
    if constexpr ((VAR & 1) != 0 && ABL == 0) {
        pass(std::true_type{});
        if (__syncthreads_or(sticky != 0)) pass(std::false_type{});   // workgroup-uniform: all waves stage the second pass's tiles
    } else {
        pass(std::false_type{});
    }
    }   // !PIPE

    const float inv = 1.0f / half_sum(l_run);
    {
        int64_t orow = (int64_t)b * p.strideO + (int64_t)q * p.ldo;
        if (p.ragged) orow = attn_entry_out_row(eb, q) * p.ldo;
        uint16_t* dst = p.O + orow + hd * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                // lane (q, hh) holds dims db*32 + 8g + 4hh + {0..3}; the two lanes of a query swap halves so that each
                // writes 8 consecutive dims (16-byte stores)
                uint2 even, odd;
                even.x = pack_bf16(o[db][8 * gp] * inv, o[db][8 * gp + 1] * inv);
                even.y = pack_bf16(o[db][8 * gp + 2] * inv, o[db][8 * gp + 3] * inv);
                odd.x = pack_bf16(o[db][8 * gp + 4] * inv, o[db][8 * gp + 5] * inv);
                odd.y = pack_bf16(o[db][8 * gp + 6] * inv, o[db][8 * gp + 7] * inv);
                // permlane32_swap(vdst = even, src = odd): lanes 32-63 of `even` <-> lanes 0-31 of `odd`
                const u32x2 rx = __builtin_amdgcn_permlane32_swap(even.x, odd.x, false, false);
                const u32x2 ry = __builtin_amdgcn_permlane32_swap(even.y, odd.y, false, false);
                // lower lanes: rx[0] = own even, rx[1] = upper's even  -> dims 8g..8g+7   (g = 2gp)
                // upper lanes: rx[0] = lower's odd, rx[1] = own odd    -> dims 8(g+1)..8(g+1)+7
                uint4 out;
                out.x = rx[0]; out.y = ry[0]; out.z = rx[1]; out.w = ry[1];
                if (q < Lq) *reinterpret_cast<uint4*>(dst + db * 32 + 16 * gp + 8 * hh) = out;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Third generation: 64 queries per wave.  The second-generation body with TWO 32-query blocks per wave that share every
// K and V^T fragment read from LDS: a 32-key block costs a wave 8 ds_read_b128 for 16 MFMAs instead of 8 (at the matrix
// pipe's peak rate the 32-query form asks LDS for 128 B/clk/CU, all the LDS has), a 256-query workgroup halves the
// L2 -> LDS stream per query, and the two blocks give every MFMA chain an independent neighbour.  Two workgroups of four
// waves per CU (256 registers per lane).  Per 32-query block the arithmetic, the order of the key blocks and the
// re-stabilise decisions (taken per aligned group of 32 queries) are those of attn2_kernel: the outputs are bit-identical.
template <int NW, int VAR = 0>
__global__ __launch_bounds__(NW * 64, 2) void attn3_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    constexpr int QT = NW * 64;
    int eb, hd, qt;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int qn = nwg >> 3, rn = nwg & 7, xcd = orig & 7;
        const int item = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (orig >> 3);
        attn_work_item<QT>(p, item, eb, hd, qt);
    }
    const AttnEntry en = attn_entry(p, eb);
    const int b = en.buf;
    const int q0 = qt * QT + wid * 64 + ql;   // query of block 0; block 1 holds query q0 + 32
    const int Lq = en.lq, Lk = en.lk;
    const int kvb = p.kv_batch_stride_zero ? 0 : b;
    const uint16_t* Qg = p.Q + (((int64_t)b * p.H + hd) * p.Lq_pad) * 64;
    const uint16_t* Kg = p.K + (((int64_t)kvb * p.H + hd) * p.Lk_pad) * 64;
    const uint16_t* Vtg = p.Vt + (((int64_t)kvb * p.H + hd) * 64) * (int64_t)p.Lk_pad;

    bf16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + qb * 32;
        const int qrow = q < p.Lq_pad ? q : p.Lq_pad - 1;   // a 256-query tile may reach past the 128-aligned allocation
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[qb][ks] = *reinterpret_cast<const bf16x8*>(Qg + (int64_t)qrow * 64 + ks * 16 + hh * 8);
    }
    if (!p.q_prescaled) {
        const float sc = p.scale * 1.4426950408889634f;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                union { bf16x8 v; uint32_t u[4]; } w;
                w.v = qf[qb][ks];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    w.u[e] = pack_bf16(__uint_as_float(w.u[e] << 16) * sc, __uint_as_float(w.u[e] & 0xFFFF0000u) * sc);
                qf[qb][ks] = w.v;
            }
    }

    f32x16 o[2][2], negm[2];
    float m_run[2], l_run[2];
    // variant bit 0 (round 6, comment at PSUM_LIMIT2): the FAST pass takes no maximum after the first key block; a lane whose 16
    // exponentials sum past the bound (or to NaN) only sets a sticky flag.  When a valid query of the workgroup set it, the whole
    // query tile runs again as the SAFE pass -- generation 2's body with its test on the lane maxima, bit-identical to variant 0 --
    // and nothing of the fast pass is used.  (An in-place rare path as in attn2_kernel costs this 256-register kernel a spilled Q
    // fragment inside the key loop, whose reload waits on vmcnt(0) -- i.e. on the LDS-DMA prefetch.)
    unsigned long long sticky = 0;
    const unsigned long long valid_lanes[2] = {__ballot(q0 < Lq), __ballot(q0 + 32 < Lq)};   // rows past Lq (stale data) set no flag

    int off[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int row = rb * 32 + ql;
        off[rb] = row * 128 + ((hh ^ ((row >> 1) & 7)) << 4);
    }

    const int ntiles = (Lk + KV_TILE - 1) / KV_TILE;
    const int bias_key = en.bias_key;
    const float bias_l2 = p.ragged ? en.bias_log2 : 0.f;
    constexpr int PPW = 8 / NW;
    uint32_t koff[PPW], voff[PPW];     // byte offsets, unsigned: scalar tile base + zero-extended lane offset (see attn2_kernel)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int row = (wid * PPW + i) * 8 + (lane >> 3);
        const int kc = (lane & 7) ^ ((row >> 1) & 7);
        koff[i] = (uint32_t)(row * 64 + kc * 8) * 2u;
        voff[i] = ((uint32_t)row * (uint32_t)p.Lk_pad + (uint32_t)(kc * 8)) * 2u;
    }
    auto stage = [&](int t, char* dst) {
        const char* kt = reinterpret_cast<const char*>(Kg + (int64_t)t * (KV_TILE * 64));
        const char* vt = reinterpret_cast<const char*>(Vtg + (int64_t)t * KV_TILE);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kt + (size_t)koff[i]),
                                             (__attribute__((address_space(3))) void*)(dst + (wid * PPW + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vt + (size_t)voff[i]),
                                             (__attribute__((address_space(3))) void*)(dst + TILE_B + (wid * PPW + i) * 1024),
                                             16, 0, 0);
        }
    };
    const bool pad_tail = ntiles * KV_TILE > Lk;
    auto pass = [&](auto FAST) __attribute__((always_inline)) {
    constexpr bool kFast = decltype(FAST)::value;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[qb][0][r] = 0.f; o[qb][1][r] = 0.f; negm[qb][r] = 0.f; }
        m_run[qb] = 0.f;
        l_run[qb] = 0.f;
    }
    stage(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto tile = [&](auto masked, auto first, const int t) {
        const char* cur = smem + (t & 1) * STAGE_B;
        if (t + 1 < ntiles) stage(t + 1, smem + ((t + 1) & 1) * STAGE_B);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            constexpr bool kFirst = decltype(first)::value;
            const bool first_block = kFirst && kb == 0;
            // ---- scores of 32 keys for both query blocks from one set of K fragments
            bf16x8 kf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(cur + (off[kb] ^ (ks << 5)));
            f32x16 s[2];
            if (first_block) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0][0], z, 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[1][0], z, 0, 0, 0);
            } else {
                s[0] = mfma_32x32x16_fresh(kf[0], qf[0][0], negm[0]);
                s[1] = mfma_32x32x16_fresh(kf[0], qf[1][0], negm[1]);
            }
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) {
                s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[0][ks], s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[1][ks], s[1], 0, 0, 0);
            }
            uint32_t pk[2][8];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f32x16& sq = s[qb];
                if constexpr (decltype(masked)::value) {
                    const int key_base = t * KV_TILE + kb * 32 + 4 * hh;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key_base + (r & 3) + 8 * (r >> 2);
                        if (key >= Lk) sq[r] = -INFINITY;
                        else if (key == bias_key) sq[r] += bias_l2;
                    }
                }
                float mx = 0.f;
                if (first_block || !kFast) {
                    mx = fmaxf(sq[0], sq[1]);
#pragma unroll
                    for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sq[r]), sq[r + 1]);
                }
                if (first_block) {
                    m_run[qb] = half_max(mx);
#pragma unroll
                    for (int r = 0; r < 16; ++r) { negm[qb][r] = -m_run[qb]; sq[r] -= m_run[qb]; }
                } else if (!kFast && __any(q0 + qb * 32 < Lq && !(mx <= SCORE_LIMIT))) {   // rows past Lq (stale data) decide nothing
                    const float grow = fmaxf(half_max(mx), 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-grow);
                    m_run[qb] += grow;
                    l_run[qb] *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        o[qb][0][r] *= alpha;
                        o[qb][1][r] *= alpha;
                        negm[qb][r] -= grow;
                        sq[r] -= grow;
                    }
                }
                float pe[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pe[r] = __builtin_amdgcn_exp2f(sq[r]);
#pragma unroll
                for (int e = 0; e < 8; ++e) pk[qb][e] = pack_bf16(pe[2 * e], pe[2 * e + 1]);
                const float psum = sum16(pe);
                l_run[qb] += psum;
                if (kFast && !first_block) sticky |= __ballot(!(psum <= PSUM_LIMIT2)) & valid_lanes[qb];
            }
            // ---- O^T += V^T P^T: one V^T fragment feeds both query blocks
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                union { uint32_t u[4]; bf16x8 v; } pf[2];
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) pf[qb].u[e] = pk[qb][4 * ks2 + e];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(cur + TILE_B + (off[db] ^ ((2 * kb + ks2) << 5)));
                    o[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[0].v, o[0][db], 0, 0, 0);
                    o[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[1].v, o[1][db], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    if (ntiles == 1) {
        if (pad_tail) tile(std::true_type{}, std::true_type{}, 0);
        else tile(std::false_type{}, std::true_type{}, 0);
    } else {
        tile(std::false_type{}, std::true_type{}, 0);
        for (int t = 1; t < ntiles - 1; ++t) tile(std::false_type{}, std::false_type{}, t);
        if (pad_tail) tile(std::true_type{}, std::false_type{}, ntiles - 1);
        else tile(std::false_type{}, std::false_type{}, ntiles - 1);
    }
    };   // pass
    if constexpr ((VAR & 1) != 0) {
        pass(std::true_type{});
        // (the last tile's closing barrier has passed: every wave is done with the LDS tiles; the vote is workgroup-uniform
        // because all four waves stage the tiles of the second pass and meet at its barriers)
        if (__syncthreads_or(sticky != 0)) pass(std::false_type{});
    } else {
        pass(std::false_type{});
    }

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + qb * 32;
        const float inv = 1.0f / half_sum(l_run[qb]);
        int64_t orow = (int64_t)b * p.strideO + (int64_t)q * p.ldo;
        if (p.ragged) orow = attn_entry_out_row(eb, q) * p.ldo;
        uint16_t* dst = p.O + orow + hd * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint2 even, odd;
                even.x = pack_bf16(o[qb][db][8 * gp] * inv, o[qb][db][8 * gp + 1] * inv);
                even.y = pack_bf16(o[qb][db][8 * gp + 2] * inv, o[qb][db][8 * gp + 3] * inv);
                odd.x = pack_bf16(o[qb][db][8 * gp + 4] * inv, o[qb][db][8 * gp + 5] * inv);
                odd.y = pack_bf16(o[qb][db][8 * gp + 6] * inv, o[qb][db][8 * gp + 7] * inv);
                const u32x2 rx = __builtin_amdgcn_permlane32_swap(even.x, odd.x, false, false);
                const u32x2 ry = __builtin_amdgcn_permlane32_swap(even.y, odd.y, false, false);
                uint4 out;
                out.x = rx[0]; out.y = ry[0]; out.z = rx[1]; out.w = ry[1];
                if (q < Lq) *reinterpret_cast<uint4*>(dst + db * 32 + 16 * gp + 8 * hh) = out;
            }
    }
}


// ------------------------------------------------------------------------------------------------------------
// Generation 9 (round 5; opt-in, `attn_generation` 9): EXPLICIT inter-wave phases, the structure VERDICT r4 item 2 asked for.
// Round 4's counters (profiles/r04_attention_bound.md) say the two pipes run together for only 39-53 % of the matrix cycles and
// NEITHER runs ~30 % of the time: waves of independent workgroups meet on a SIMD in whatever phase they happen to be in.  The first
// attempt (two halves of four waves, softmax | matrix phase, one phase apart: "generation 8", removed again) measured 0.6-0.7 of
// generation 2's rate, and tools/ubench/valu_rate.hip says why: ONE wave issues a plain vector instruction every 6 cycles and an
// exponential every 10, whatever the vector pipe could take from its neighbours -- ~440 cycles for the 63 instructions of a block
// against 8 x 16 cycles of the matrix pipe --, so with two waves per SIMD the softmax phase is the whole period.  Here a workgroup is
// 12 waves (three per SIMD, <= 168 registers) = three groups one phase apart, and a block takes a wave three phases:
//     S1(n): [first block of a tile, waves 0-7: LDS-DMA of the tile two ahead]  row maximum test, exponentials 0-7
//     S2(n): LDS reads of the fragments of M(n), exponentials 8-15, bf16 packing, row sum
//     M(n) : the 8 MFMAs (P V of block n interleaved with the scores of block n + 1), s_setprio(1)
// so that on every SIMD one wave is in its matrix phase while the two others share the vector pipe.  Arithmetic and order per
// 32-query block: attn2_kernel's -- bit-identical outputs (tests/test_ops_gpu.py).  MEASURED (profiles/r05_attention_phases.md):
// 852-868 TFLOP/s on the geo decoder's passes against 995-1027 for generations 2 / 6, 640 against 820-834 on the DiT's shape: the
// s_memtime stamps (option attn_stamps) show the two softmax phases at ~1.6x the matrix phase in every setting of s_setprio -- the
// per-wave issue rate again: S1 + S2 is one wave's ~440+ cycles, M is 128, and with <= 3 waves per SIMD no rotation balances that.
// Default stays generation 7 (= 2 / 6); this kernel is kept as the measured answer to the review's question.  LDS ring: 4 stages; global phase of group g: S1(n) = 3n + g, S2(n) = 3n + 1 + g, M(n) = 3n + 2 + g;
// tile t (blocks 2t, 2t + 1) is read from phase 6t - 2 (group 0, K of block 2t in S2(2t - 1)) to phase 6t + 6 (group 2, V^T of block
// 2t + 1 in S2(2t + 1)).  Waves 0-7 stage one K and one V^T piece of tile t + 2 each at the top of their S1(2t) (phase >= 6t: the
// stage's last occupant, tile t - 2, was last read in phase 6t - 6) and wait for THEIR pieces of tile u before the barrier that ends
// phase 6u - 3 -- group 0 at the end of S1(2u - 1), group 1 at the end of M(2u - 2) -- with vmcnt(2): tile u + 1 stays in flight.
// PRIO: 0 no priorities | 1 the matrix phase at s_setprio(1) | 2 the two softmax phases at s_setprio(1)
template <bool STAMP, int PRIO>
__global__ __launch_bounds__(768, 3) void attn6_kernel(AttnArgs p, unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem6[];
    char* const smem = smem6;
    constexpr int R = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;                    // 0, 1, 2: group g runs g phases behind group 0
    const int ql = lane & 31, hh = lane >> 5;
    constexpr int QT = 384;
    int eb, hd, qt;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int qn = nwg >> 3, rn = nwg & 7, xcd = orig & 7;
        const int item = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (orig >> 3);
        attn_work_item<QT>(p, item, eb, hd, qt);
    }
    const AttnEntry en = attn_entry(p, eb);
    const int b = en.buf;
    const int q = qt * QT + wid * 32 + ql;
    const int Lq = en.lq, Lk = en.lk;
    const int kvb = p.kv_batch_stride_zero ? 0 : b;
    const uint16_t* Qg = p.Q + (((int64_t)b * p.H + hd) * p.Lq_pad) * 64;
    const uint16_t* Kg = p.K + (((int64_t)kvb * p.H + hd) * p.Lk_pad) * 64;
    const uint16_t* Vtg = p.Vt + (((int64_t)kvb * p.H + hd) * 64) * (int64_t)p.Lk_pad;

    const int ntiles = (Lk + KV_TILE - 1) / KV_TILE;
    const int nblk = 2 * ntiles;
    const bool stager = wid < 8;
    int koff, voff;
    {
        const int row = (wid & 7) * 8 + (lane >> 3);
        const int kc = (lane & 7) ^ ((row >> 1) & 7);
        koff = row * 64 + kc * 8;
        voff = row * p.Lk_pad + kc * 8;
    }
    auto stage = [&](int t) {
        char* dst = smem + (t % R) * STAGE_B;
        const uint16_t* kt = Kg + (int64_t)t * (KV_TILE * 64);
        const uint16_t* vt = Vtg + (int64_t)t * KV_TILE;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kt + koff),
                                         (__attribute__((address_space(3))) void*)(dst + (wid & 7) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vt + voff),
                                         (__attribute__((address_space(3))) void*)(dst + TILE_B + (wid & 7) * 1024), 16, 0, 0);
    };
    auto wait_tile = [&](int u) {     // this wave's pieces of tile u have landed; tile u + 1 (when it exists) may stay in flight
        if (u + 1 < ntiles) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (stager) {
        stage(0);
        if (ntiles > 1) stage(1);
    }
    bf16x8 qf[4];
    {
        const int qrow = q < p.Lq_pad ? q : p.Lq_pad - 1;   // a 384-query tile may reach past the 128-aligned allocation
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[ks] = *reinterpret_cast<const bf16x8*>(Qg + (int64_t)qrow * 64 + ks * 16 + hh * 8);
    }
    if (!p.q_prescaled) {
        const float sc = p.scale * 1.4426950408889634f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            union { bf16x8 v; uint32_t u[4]; } w;
            w.v = qf[ks];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w.u[e] = pack_bf16(__uint_as_float(w.u[e] << 16) * sc, __uint_as_float(w.u[e] & 0xFFFF0000u) * sc);
            qf[ks] = w.v;
        }
    }
    f32x16 o[2], negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
    float m_run = 0.f, l_run = 0.f;
    int off[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int row = rb * 32 + ql;
        off[rb] = row * 128 + ((hh ^ ((row >> 1) & 7)) << 4);
    }
    const int bias_key = en.bias_key;
    const float bias_l2 = p.ragged ? en.bias_log2 : 0.f;
    const bool pad_tail = ntiles * KV_TILE > Lk;

    // STAMP (timing experiments, option "attn_stamps"): s_memtime ticks of every phase's work (phase start -> barrier arrival) and
    // of its wait at the barrier, summed per wave and added to stamps[group][S1 work, S1 wait, S2 work, S2 wait, M work, M wait, n]
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t0 = 0;
    int st_ph = 0;
#define R3G_PHASE_END() do { __builtin_amdgcn_sched_barrier(0); \
        unsigned long long st_t1 = 0; if (STAMP) { st_t1 = __builtin_amdgcn_s_memtime(); } \
        asm volatile("s_barrier" ::: "memory"); \
        if (STAMP) { const unsigned long long st_t2 = __builtin_amdgcn_s_memtime(); \
            st_acc[2 * st_ph] += st_t1 - st_t0; st_acc[2 * st_ph + 1] += st_t2 - st_t1; st_t0 = st_t2; st_ph = st_ph == 2 ? 0 : st_ph + 1; } \
        __builtin_amdgcn_sched_barrier(0); } while (0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    R3G_PHASE_END();                                  // tiles 0, 1 are in LDS
    for (int g = 0; g < grp; ++g) R3G_PHASE_END();    // group g starts g phases late

    // scores of block 0, from zero (the stabiliser starts as their exact maximum, in S1(0))
    f32x16 sA, sB;
    {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(smem + off[0]);
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[0], z, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(smem + (off[0] ^ (ks << 5)));
            sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sA, 0, 0, 0);
        }
    }
    // One 32-key block: block kb of tile t, scores in `s`; the matrix phase leaves the next block's scores in `s_next` (the two
    // score blocks alternate between two register sets: a loop-carried single set costs 16 register copies per block).
    // The empty asm statements PIN a phase's results in that phase: they are volatile, like the barriers, so nothing that feeds
    // them can sink behind the barrier -- left alone the compiler moved every exponential and conversion of S1 / S2 down to its
    // first use in M(n) (sched_barrier does not bind IR-level sinking), which made M the whole softmax again.
    auto block = [&](auto KB, const int t, f32x16& s, f32x16& s_next) __attribute__((always_inline)) {
        constexpr int kb = decltype(KB)::value;
        const char* cur = smem + (t % R) * STAGE_B;
        // ================= S1 =================
        if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
        if (kb == 0 && stager && t + 2 < ntiles) stage(t + 2);
        if (t == ntiles - 1 && pad_tail) {            // only the last tile holds padded keys / the weighted key
            const int key_base = t * KV_TILE + kb * 32 + 4 * hh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key_base + (r & 3) + 8 * (r >> 2);
                if (key >= Lk) s[r] = -INFINITY;
                else if (key == bias_key) s[r] += bias_l2;
            }
        }
        float mx = fmaxf(s[0], s[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
        if (kb == 0 && t == 0) {
            m_run = half_max(mx);
#pragma unroll
            for (int r = 0; r < 16; ++r) { negm[r] = -m_run; s[r] -= m_run; }
        } else if (__any(q < Lq && !(mx <= SCORE_LIMIT))) {
            const float grow = fmaxf(half_max(mx), 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-grow);
            m_run += grow;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; negm[r] -= grow; s[r] -= grow; }
        }
        float pe[16];
#pragma unroll
        for (int r = 0; r < 8; ++r) pe[r] = __builtin_amdgcn_exp2f(s[r]);
        asm volatile("" : "+v"(pe[0]), "+v"(pe[1]), "+v"(pe[2]), "+v"(pe[3]), "+v"(pe[4]), "+v"(pe[5]), "+v"(pe[6]), "+v"(pe[7]));
        if (kb == 1 && grp == 0) wait_tile(t + 1);
        R3G_PHASE_END();
        // ================= S2 =================
        bf16x8 vf[2][2], kf[4];
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
            for (int db = 0; db < 2; ++db)
                vf[ks2][db] = *reinterpret_cast<const bf16x8*>(cur + TILE_B + (off[db] ^ ((2 * kb + ks2) << 5)));
        {   // K of the next block (behind the very last block: whatever the ring holds -- those scores are never used)
            const char* nxt = kb == 0 ? cur : smem + ((t + 1) % R) * STAGE_B;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(nxt + (off[kb ^ 1] ^ (ks << 5)));
        }
#pragma unroll
        for (int r = 8; r < 16; ++r) pe[r] = __builtin_amdgcn_exp2f(s[r]);
        uint32_t pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[e] = pack_bf16(pe[2 * e], pe[2 * e + 1]);
        {
            f32x2 ps = (f32x2){pe[0], pe[1]};
#pragma unroll
            for (int r = 2; r < 16; r += 2) ps += (f32x2){pe[r], pe[r + 1]};
            l_run += ps[0] + ps[1];
        }
        asm volatile("" : "+v"(pk[0]), "+v"(pk[1]), "+v"(pk[2]), "+v"(pk[3]), "+v"(pk[4]), "+v"(pk[5]), "+v"(pk[6]), "+v"(pk[7]),
                          "+v"(l_run));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        R3G_PHASE_END();
        // ================= M =================
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
        {
            union { uint32_t u[4]; bf16x8 v; } pf0, pf1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pf0.u[e] = pk[e]; pf1.u[e] = pk[4 + e]; }
            // P V of this block (o[db] in the order ks2 = 0, 1: attn2_kernel's) interleaved with the score chain of the next one
            s_next = mfma_32x32x16_fresh(kf[0], qf[0], negm);
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][0], pf0.v, o[0], 0, 0, 0);
            s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[1], s_next, 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][1], pf0.v, o[1], 0, 0, 0);
            s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2], qf[2], s_next, 0, 0, 0);
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][0], pf1.v, o[0], 0, 0, 0);
            s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[3], qf[3], s_next, 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][1], pf1.v, o[1], 0, 0, 0);
        }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (kb == 0 && grp == 1) wait_tile(t + 1);
        R3G_PHASE_END();
    };
    if (STAMP) {
#pragma unroll
        for (int k = 0; k < 6; ++k) st_acc[k] = 0;
        st_ph = 0;
        st_t0 = __builtin_amdgcn_s_memtime();
    }
    for (int t = 0; t < ntiles; ++t) {
        block(std::integral_constant<int, 0>{}, t, sA, sB);
        block(std::integral_constant<int, 1>{}, t, sB, sA);
    }
    if (STAMP && lane == 0 && stamps) {
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(&stamps[grp * 8 + k], st_acc[k]);
        atomicAdd(&stamps[grp * 8 + 6], (unsigned long long)nblk);
    }

    const float inv = 1.0f / half_sum(l_run);
    {
        int64_t orow = (int64_t)b * p.strideO + (int64_t)q * p.ldo;
        if (p.ragged) orow = attn_entry_out_row(eb, q) * p.ldo;
        uint16_t* dst = p.O + orow + hd * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint2 even, odd;
                even.x = pack_bf16(o[db][8 * gp] * inv, o[db][8 * gp + 1] * inv);
                even.y = pack_bf16(o[db][8 * gp + 2] * inv, o[db][8 * gp + 3] * inv);
                odd.x = pack_bf16(o[db][8 * gp + 4] * inv, o[db][8 * gp + 5] * inv);
                odd.y = pack_bf16(o[db][8 * gp + 6] * inv, o[db][8 * gp + 7] * inv);
                const u32x2 rx = __builtin_amdgcn_permlane32_swap(even.x, odd.x, false, false);
                const u32x2 ry = __builtin_amdgcn_permlane32_swap(even.y, odd.y, false, false);
                uint4 out;
                out.x = rx[0]; out.y = ry[0]; out.z = rx[1]; out.w = ry[1];
                if (q < Lq) *reinterpret_cast<uint4*>(dst + db * 32 + 16 * gp + 8 * hh) = out;
            }
    }
    for (int g = grp; g < 2; ++g) R3G_PHASE_END();    // the later groups' last phases end with barriers of their own
#undef R3G_PHASE_END
}

}  // namespace

static bool g_attn_glds = true;
static bool g_attn_pipelined = false;  // measured slower than the plain kernel (2 vs 3 waves/SIMD): profiles/r01_attention_variants.md
static int g_attn_ablate = 0;
void attn_set_ablate(int mask) { g_attn_ablate = mask; }
void attn_set_glds(bool on) { g_attn_glds = on; }
void attn_set_pipelined(bool on) { g_attn_pipelined = on; }

static int g_attn_gen = 7;
static int g_attn_variant = 1;     // option "attn_variant" (round 6): 1 = fast pass without a maximum + sticky sum test (default) | 0 = rounds 2-5 (generations 2 / 6 / 7)
void attn_set_variant(int v) { if (v >= 0 && v <= 1) g_attn_variant = v; }
// generation 7 takes the 64-query-per-wave kernel where its 256-query workgroups make at least four full rounds of the
// 512 slots (the geo decoder's 131072-query passes: +6 %); on the DiT's 4442-query attention the coarser grid costs more
// than the kernel gains (576 workgroups on 512 slots), profiles/r02_attention.md
static int g_wide6_min_items = 2048;
void attn_set_wide_min(int items) { if (items > 0) g_wide6_min_items = items; }
void attn_set_generation(int gen) { if ((gen >= 1 && gen <= 7) || gen == 9) g_attn_gen = gen; }
static int g_attn6_prio = 1;       // option "attn_prio": generation 9's s_setprio use (0 none | 1 matrix phase | 2 softmax phases)
void attn_set_prio(int v) { if (v >= 0 && v <= 2) g_attn6_prio = v; }
static int g_attn_stamps = 0;      // option "attn_stamps": generation 9 prints its per-phase s_memtime sums (timing experiments)
void attn_set_stamps(int on) { g_attn_stamps = on; }

// (the first-generation kernels -- attn_generation 1 and the attn_pipelined option -- scale the scores themselves and
// reject a pre-scaled Q: producers must then leave q plain)
float attn_q_scale(float scale) { return (g_attn_gen >= 2 && !g_attn_pipelined) ? scale * 1.4426950408889634f : 1.0f; }

hipError_t attention_launch(const AttnArgs& p, hipStream_t s) {
    if (p.ragged) {
        if (p.B < 1 || p.B > kAttnMaxEntries) return hipErrorInvalidValue;
        for (int b = 0; b < p.B; ++b) {
            const AttnEntry& en = p.ent[b];
            if (en.lq <= 0 || en.lk <= 0 || en.lq > p.Lq_pad || en.lk > p.Lk_pad || en.buf < 0) return hipErrorInvalidValue;
            const int nt = (en.lk + 63) / 64;
            if (en.bias_key >= 0 && (en.bias_key < (nt - 1) * 64 || nt * 64 == en.lk || en.bias_key >= en.lk))
                return hipErrorInvalidValue;  // the weighted key must sit in the last, padded tile
        }
    }
    if (p.Lq_pad % 128 || p.Lk_pad % 64 || p.Lk <= 0 || p.Lq <= 0 || p.Lk > p.Lk_pad || p.Lq > p.Lq_pad)
        return hipErrorInvalidValue;
    double pairs = (double)p.B * p.Lq * p.Lk;
    if (p.ragged) {
        pairs = 0;
        for (int b = 0; b < p.B; ++b) pairs += (double)p.ent[b].lq * p.ent[b].lk;
    }
    ProfScope ps(PC_ATTN, 4.0 * p.H * pairs * 64, s);
    if (g_attn_gen >= 2 && !g_attn_pipelined) {
        // generation: 2 = 4 waves of 32 queries, 3 = 2 software-pipelined, 4 = 8 waves software-pipelined, 5 = 8 waves,
        // 6 = 4 waves of 64 queries, 7 = 6 where its 256-query workgroups fill the two slots per CU, otherwise 2
        auto count = [&](int qtile) {
            int n = 0;
            if (p.ragged) for (int b = 0; b < p.B; ++b) n += (p.ent[b].lq + qtile - 1) / qtile;
            else n = ((p.Lq + qtile - 1) / qtile) * p.B;
            return n * p.H;
        };
        int gen = g_attn_gen;
        if (gen == 7) gen = (g_attn_glds && count(256) >= g_wide6_min_items) ? 6 : 2;
        const int qtile = (g_attn_glds && gen == 9) ? 384 : (g_attn_glds && gen >= 4) ? 256 : 128;
        const int items = count(qtile);
        if (gen == 9 && g_attn_glds) {     // phased 12-wave kernel (round 5): three groups, S1 | S2 | M
            const int prio = g_attn6_prio;
            auto launch6 = [&](auto ST, unsigned long long* d_st) -> hipError_t {
                constexpr bool kSt = decltype(ST)::value;
                auto k0 = attn6_kernel<kSt, 0>;
                auto k1 = attn6_kernel<kSt, 1>;
                auto k2 = attn6_kernel<kSt, 2>;
                auto k = prio == 0 ? k0 : (prio == 2 ? k2 : k1);
                if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE_B) != hipSuccess) {
                    (void)hipGetLastError();
                    return hipErrorNotSupported;
                }
                hipLaunchKernelGGL(k, dim3(items), dim3(768), 4 * STAGE_B, s, p, d_st);
                return hipGetLastError();
            };
            if (g_attn_stamps) {       // timing experiment: per-phase s_memtime sums of this launch, printed to stderr (synchronous)
                static unsigned long long* d_st = nullptr;
                if (!d_st && hipMalloc((void**)&d_st, 24 * 8) != hipSuccess) return hipErrorOutOfMemory;
                (void)hipMemsetAsync(d_st, 0, 24 * 8, s);
                const hipError_t e = launch6(std::true_type{}, d_st);
                if (e != hipSuccess) return e;
                unsigned long long h[24];
                if (hipMemcpyAsync(h, d_st, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
                    return hipGetLastError();
                for (int g = 0; g < 3; ++g) {
                    const double n = (double)(h[g * 8 + 6] ? h[g * 8 + 6] : 1);
                    fprintf(stderr, "[attn6 stamps] prio %d group %d: ticks per block and wave: S1 %.0f + wait %.0f | S2 %.0f + wait %.0f | M %.0f + wait %.0f\n", prio, g,
                            h[g * 8] / n, h[g * 8 + 1] / n, h[g * 8 + 2] / n, h[g * 8 + 3] / n, h[g * 8 + 4] / n, h[g * 8 + 5] / n);
                }
                return hipSuccess;
            }
            return launch6(std::false_type{}, nullptr);
        }
        if (g_attn_ablate && gen == 2) {
            switch (g_attn_ablate) {
#define R3G_ABL2(m) case m: hipLaunchKernelGGL((attn2_kernel<true, false, 4, m>), dim3(items), dim3(256), 0, s, p); break;
                R3G_ABL2(1) R3G_ABL2(4) R3G_ABL2(8) R3G_ABL2(12) R3G_ABL2(16) R3G_ABL2(48) R3G_ABL2(64) R3G_ABL2(128)
                R3G_ABL2(192) R3G_ABL2(240) R3G_ABL2(241) R3G_ABL2(245) R3G_ABL2(253)
#undef R3G_ABL2
                default: return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
        if (!g_attn_glds) hipLaunchKernelGGL((attn2_kernel<false, false, 4>), dim3(items), dim3(256), 0, s, p);
        else if (gen == 3) hipLaunchKernelGGL((attn2_kernel<true, true, 4>), dim3(items), dim3(256), 0, s, p);
        else if (gen == 4) hipLaunchKernelGGL((attn2_kernel<true, true, 8>), dim3(items), dim3(512), 0, s, p);
        else if (gen == 5) hipLaunchKernelGGL((attn2_kernel<true, false, 8>), dim3(items), dim3(512), 0, s, p);
        else if (gen == 6) {
            if (g_attn_variant == 1) hipLaunchKernelGGL((attn3_kernel<4, 1>), dim3(items), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((attn3_kernel<4, 0>), dim3(items), dim3(256), 0, s, p);
        } else {
            if (g_attn_variant == 1) hipLaunchKernelGGL((attn2_kernel<true, false, 4, 0, 1>), dim3(items), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((attn2_kernel<true, false, 4>), dim3(items), dim3(256), 0, s, p);
        }
        return hipGetLastError();
    }
    if (p.q_prescaled) return hipErrorInvalidValue;   // the first-generation kernels scale the scores themselves
    dim3 grid(p.Lq_pad / 128, p.H, p.B);
    if (g_attn_pipelined) {
        if (g_attn_glds) hipLaunchKernelGGL(attn_kernel_sp<true>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(attn_kernel_sp<false>, grid, dim3(256), 0, s, p);
    } else if (g_attn_glds && g_attn_ablate) {
        switch (g_attn_ablate) {
#define R3G_ABL(m) case m: hipLaunchKernelGGL((attn_kernel<true, m>), grid, dim3(256), 0, s, p); break;
            R3G_ABL(1) R3G_ABL(2) R3G_ABL(3) R3G_ABL(4) R3G_ABL(8) R3G_ABL(12) R3G_ABL(16) R3G_ABL(48) R3G_ABL(64) R3G_ABL(68)
            R3G_ABL(15) R3G_ABL(63) R3G_ABL(127)
#undef R3G_ABL
            default: return hipErrorInvalidValue;
        }
    } else if (g_attn_glds) {
        hipLaunchKernelGGL(attn_kernel<true>, grid, dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL(attn_kernel<false>, grid, dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace r3g
