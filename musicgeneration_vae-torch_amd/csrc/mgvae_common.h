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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
