// Shared helpers for the gfx950 kernels of libmgvae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mgvae.h"

#define MGVAE_CHECK_LAUNCH()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return MGVAE_ELAUNCH;           \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == MGVAE_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MGVAE_ACT_LEAKY) return v > 0.f ? v : v * slope;
    if (act == MGVAE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}
// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
    if (act == MGVAE_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == MGVAE_ACT_LEAKY) return y > 0.f ? 1.f : slope;
    if (act == MGVAE_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

// Sum over the 64 lanes with DPP row shifts / row broadcasts (VALU only -- no LDS round trips like the
// ds_bpermute behind __shfl_xor): six dependent adds.  The total is valid in LANE 63 only.
template <int CTRL, int ROW = 0xf, int BANK = 0xf>
__device__ __forceinline__ float dpp_mov0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW, BANK, false));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v += dpp_mov0<0x111>(v);              // row_shr:1
    v += dpp_mov0<0x112>(v);              // row_shr:2
    v += dpp_mov0<0x114, 0xf, 0xe>(v);    // row_shr:4
    v += dpp_mov0<0x118, 0xf, 0xc>(v);    // row_shr:8   -> lane 15 of every row holds the row total
    v += dpp_mov0<0x142, 0xa>(v);         // row_bcast:15 into rows 1 and 3
    v += dpp_mov0<0x143, 0xc>(v);         // row_bcast:31 into rows 2 and 3
    return v;
}
// full-wave sum, result in every lane (lane 63's total broadcast through an SGPR); call in convergent code only
__device__ __forceinline__ float wave_sum(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_lane63(v)), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
