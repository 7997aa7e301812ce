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

template <bool GLDS>
__global__ __launch_bounds__(256, 2) void attn_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, hd = blockIdx.y;
    const int ql = lane & 31, hh = lane >> 5;
    const int q = blockIdx.x * 128 + wid * 32 + ql;
    const int Lq = p.ragged ? p.lq_b[b] : p.Lq, Lk = p.ragged ? p.lk_b[b] : p.Lk;
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
    const int bias_key = p.ragged ? p.bias_key[b] : -1;
    const float bias_raw = p.ragged ? p.bias_log2[b] / sc : 0.f;  // added to the raw (unscaled) score
    stage_kv<GLDS>(Kg, Vtg, p.Lk_pad, 0, smem, wid, lane, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto tile = [&](auto masked, const int t) {
        const char* cur = smem + (t & 1) * STAGE_B;
        if (t + 1 < ntiles) stage_kv<GLDS>(Kg, Vtg, p.Lk_pad, (t + 1) * KV_TILE, smem + ((t + 1) & 1) * STAGE_B, wid, lane, tid);

        // ---- S^T = K Q^T for the two 32-key blocks
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(cur + (offK[kb] ^ (ks << 5)));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
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
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[kb][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        // rescale only when some query's running max actually grows (wave-uniform branch; exact, not a threshold)
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
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], sc, -mb));
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
                    vf.h[0] = *reinterpret_cast<const bf16x4*>(cur + TILE_B + (offV[db] ^ (c << 4)));
                    vf.h[1] = *reinterpret_cast<const bf16x4*>(cur + TILE_B + (offV[db] ^ ((c + 1) << 4)));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o[db], 0, 0, 0);
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
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
        if (p.ragged) orow = (q < p.o_split[b] ? p.o_row0[b] + q : p.o_row_split[b] + (q - p.o_split[b])) * p.ldo;
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
    const int b = blockIdx.z, hd = blockIdx.y;
    const int ql = lane & 31, hh = lane >> 5;
    const int q = blockIdx.x * 128 + wid * 32 + ql;
    const int Lq = p.ragged ? p.lq_b[b] : p.Lq, Lk = p.ragged ? p.lk_b[b] : p.Lk;
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
    const int bias_key = p.ragged ? p.bias_key[b] : -1;
    const float bias_raw = p.ragged ? p.bias_log2[b] / sc : 0.f;

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
                    union { bf16x4 h[2]; bf16x8 v; } vf;
                    vf.h[0] = *reinterpret_cast<const bf16x4*>(vt + (offV[db] ^ (c << 4)));
                    vf.h[1] = *reinterpret_cast<const bf16x4*>(vt + (offV[db] ^ ((c + 1) << 4)));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o[db], 0, 0, 0);
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
        if (p.ragged) orow = (q < p.o_split[b] ? p.o_row0[b] + q : p.o_row_split[b] + (q - p.o_split[b])) * p.ldo;
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

}  // namespace

static bool g_attn_glds = true;
static bool g_attn_pipelined = false;  // measured slower than the plain kernel (2 vs 3 waves/SIMD): profiles/r01_attention_variants.md
void attn_set_glds(bool on) { g_attn_glds = on; }
void attn_set_pipelined(bool on) { g_attn_pipelined = on; }

hipError_t attention_launch(const AttnArgs& p, hipStream_t s) {
    if (p.ragged) {
        if (p.B > 2) return hipErrorInvalidValue;
        for (int b = 0; b < p.B; ++b) {
            if (p.lq_b[b] <= 0 || p.lk_b[b] <= 0 || p.lq_b[b] > p.Lq_pad || p.lk_b[b] > p.Lk_pad) return hipErrorInvalidValue;
            const int nt = (p.lk_b[b] + 63) / 64;
            if (p.bias_key[b] >= 0 && (p.bias_key[b] < (nt - 1) * 64 || nt * 64 == p.lk_b[b] || p.bias_key[b] >= p.lk_b[b]))
                return hipErrorInvalidValue;  // the weighted key must sit in the last, padded tile
        }
    }
    if (p.Lq_pad % 128 || p.Lk_pad % 64 || p.Lk <= 0 || p.Lq <= 0 || p.Lk > p.Lk_pad || p.Lq > p.Lq_pad)
        return hipErrorInvalidValue;
    double pairs = (double)p.B * p.Lq * p.Lk;
    if (p.ragged) {
        pairs = 0;
        for (int b = 0; b < p.B; ++b) pairs += (double)p.lq_b[b] * p.lk_b[b];
    }
    ProfScope ps(PC_ATTN, 4.0 * p.H * pairs * 64, s);
    dim3 grid(p.Lq_pad / 128, p.H, p.B);
    if (g_attn_pipelined) {
        if (g_attn_glds) hipLaunchKernelGGL(attn_kernel_sp<true>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(attn_kernel_sp<false>, grid, dim3(256), 0, s, p);
    } else if (g_attn_glds) {
        hipLaunchKernelGGL(attn_kernel<true>, grid, dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL(attn_kernel<false>, grid, dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace r3g
