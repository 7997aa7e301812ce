// wave_sum.h -- sum over the 64 lanes of a wave, every lane gets it.  Same pairings in the same order as the butterfly
//     for (d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
// (so the same bits: a + b is commutative), but on the vector pipe: hipcc lowers that loop to six ds_bpermute_b32 -- six dependent round
// trips through the LDS crossbar with an s_waitcnt each, ~700 cycles of latency per sum, two sums per LayerNorm row (round 6: the DiT's
// LayerNorm launches turned out to be bound by this chain, not by HBM: tools/ubench/wave_sum_check.hip checks the equivalence).
//   xor 32 / 16: v_permlane32_swap / v_permlane16_swap (gfx950)      xor 8: DPP row_ror:8
//   xor 4: DPP row_shl:4 on banks 0, 2 + row_shr:4 on banks 1, 3     xor 2 / 1: DPP quad_perm
#ifndef R3G_WAVE_SUM_H
#define R3G_WAVE_SUM_H
#include <hip/hip_runtime.h>
namespace r3g {
// the value lane ^ D holds, D = 1, 2, 4 or 8 (the partner of one butterfly step inside a row of 16 lanes), by DPP
template <int D>
__device__ __forceinline__ float lane_xor(float v) {
    static_assert(D == 1 || D == 2 || D == 4 || D == 8, "inside a row of 16 lanes");
    const int vi = __float_as_int(v);
    if constexpr (D == 1) return __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    if constexpr (D == 2) return __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    if constexpr (D == 8) return __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, 0x128, 0xF, 0xF, false));  // row_ror:8
    int t = __builtin_amdgcn_update_dpp(vi, vi, 0x104, 0xF, 0x5, false);   // row_shl:4, banks 0 and 2: lane + 4
    t = __builtin_amdgcn_update_dpp(t, vi, 0x114, 0xF, 0xA, false);        // row_shr:4, banks 1 and 3: lane - 4
    return __int_as_float(t);
}

__device__ __forceinline__ float wave_sum(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    {
        const u2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    }
    {
        const u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    }
    v += lane_xor<8>(v);
    v += lane_xor<4>(v);
    v += lane_xor<2>(v);
    v += lane_xor<1>(v);
    return v;
}
}  // namespace r3g
#endif
