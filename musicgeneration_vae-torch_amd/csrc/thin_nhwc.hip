// The channels-last ENDS of the island (round 3): the two convolutions whose other side has a single channel, so that the
// encoder trunks start and the decoder ends without a layout change.
//   * mgvae_conv2d_c1_nhwc_{fwd,bwd_weight}: a conv of a ONE-channel map into Cy channels, written channels-last -- the
//     encoder stems' first convs (graph/encodingBlock.py:11-14,42-45: Conv2d(1, 32, (4,1) | (1,4), stride 2 on that axis) +
//     LeakyReLU).  4 taps per output: no matrix product, the pass is bound by writing (reading) the 32-channel map.
//   * mgvae_conv2d_to1_nhwc_{fwd,bwd}: a 1x1 conv of a channels-last map into ONE channel -- the decoder's fit2
//     (graph/decoder.py:186,217: Conv2d(64, 1, 1, bias=False) + Sigmoid).  A dot product over each pixel's channel row.
// HBM-bound: algorithmic bytes = the C-channel tensor once per pass.
#include "mgvae_common.h"

typedef __bf16 tn_b4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ float4 tn_ld4(const T* p);
template <> __device__ __forceinline__ float4 tn_ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 tn_ld4<__bf16>(const __bf16* p) {
    const tn_b4 v = *reinterpret_cast<const tn_b4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <typename T> __device__ __forceinline__ void tn_st4(T* p, float4 v);
template <> __device__ __forceinline__ void tn_st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void tn_st4<__bf16>(__bf16* p, float4 v) {
    tn_b4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;      // round to nearest even
    *reinterpret_cast<tn_b4*>(p) = o;
}

struct C1P {
    int N, H, W, Cy, OH, OW, KH, KW, SH, SW, PH, PW, y_ctot, y_coff, act;
    float slope;
    long rows;      // N * OH * OW
    int PB;         // output pixels per workgroup
};
constexpr int C1_MAXT = 8;      // taps (KH * KW)
// a thread's output pixel as (sample, row, column): two divisions once, then carries (the passes are a few dozen
// instructions per 16 bytes -- the divisions were a third of them)
struct C1Pix {
    int n, oh, ow;
    __device__ __forceinline__ C1Pix(const C1P& p, long r) {
        const int P = p.OH * p.OW;
        n = (int)(r / P);
        const int q = (int)(r - (long)n * P);
        oh = q / p.OW; ow = q - oh * p.OW;
    }
    __device__ __forceinline__ void advance(const C1P& p, int by) {
        ow += by;
        while (ow >= p.OW) { ow -= p.OW; ++oh; }
        while (oh >= p.OH) { oh -= p.OH; ++n; }
    }
};

// LP = Cy / 4 lanes hold one output pixel (a float4 of channels each); 256 / LP pixels per pass
template <int LP, typename T>
__global__ __launch_bounds__(256) void c1_nhwc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                          const C1P p) {
    const int lp = threadIdx.x % LP, slot = threadIdx.x / LP, TT = p.KH * p.KW;
    float4 wr[C1_MAXT];
#pragma unroll
    for (int t = 0; t < C1_MAXT; ++t)
        wr[t] = t < TT ? make_float4(w[(4 * lp) * TT + t], w[(4 * lp + 1) * TT + t], w[(4 * lp + 2) * TT + t], w[(4 * lp + 3) * TT + t])
                       : make_float4(0.f, 0.f, 0.f, 0.f);
    int tkh[C1_MAXT], tkw[C1_MAXT];                  // tap -> (kh, kw), once
#pragma unroll
    for (int t = 0; t < C1_MAXT; ++t) { tkh[t] = t / p.KW; tkw[t] = t - tkh[t] * p.KW; }
    const long r0 = (long)blockIdx.x * p.PB, r1 = r0 + p.PB < p.rows ? r0 + p.PB : p.rows;
    C1Pix px(p, r0 + slot);                          // (n, oh, ow) of the thread's pixel, advanced without divisions
    for (long r = r0 + slot; r < r1; r += 256 / LP, px.advance(p, 256 / LP)) {
        const int oh = px.oh, ow = px.ow;
        const float* xn = x + (size_t)px.n * p.H * p.W;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const int ih0 = oh * p.SH - p.PH, iw0 = ow * p.SW - p.PW;
#pragma unroll
        for (int t = 0; t < C1_MAXT; ++t) {
            if (t < TT) {
                const int ih = ih0 + tkh[t], iw = iw0 + tkw[t];
                const float v = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? xn[ih * p.W + iw] : 0.f;
                a.x += wr[t].x * v; a.y += wr[t].y * v; a.z += wr[t].z * v; a.w += wr[t].w * v;
            }
        }
        a = make_float4(apply_act(a.x, p.act, p.slope), apply_act(a.y, p.act, p.slope), apply_act(a.z, p.act, p.slope),
                        apply_act(a.w, p.act, p.slope));
        tn_st4(y + (size_t)r * p.y_ctot + p.y_coff + 4 * lp, a);
    }
}

// dw[c][t] += sum over output pixels of g[pixel][c] * x[tap t of pixel],  g = dy (* act'(ymask) when a mask is given).
// Same lane layout as the forward; the pixel slots of a workgroup meet in LDS, one atomic per (c, t) and workgroup.
template <int LP, typename T>
__global__ __launch_bounds__(256) void c1_nhwc_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                            const T* __restrict__ ymask, float* __restrict__ dw, const C1P p) {
    constexpr int SLOTS = 256 / LP, U = 4;          // U pixel rows per thread and trip: their loads are in flight together
    __shared__ float red[SLOTS][4 * LP * C1_MAXT + 1];
    const int lp = threadIdx.x % LP, slot = threadIdx.x / LP, TT = p.KH * p.KW;
    float4 acc[C1_MAXT];
#pragma unroll
    for (int t = 0; t < C1_MAXT; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    int tkh[C1_MAXT], tkw[C1_MAXT];
#pragma unroll
    for (int t = 0; t < C1_MAXT; ++t) { tkh[t] = t / p.KW; tkw[t] = t - tkh[t] * p.KW; }
    const long r0 = (long)blockIdx.x * p.PB, r1 = r0 + p.PB < p.rows ? r0 + p.PB : p.rows;
    C1Pix px(p, r0 + slot);
    for (long rb = r0 + slot; rb < r1; rb += (long)U * SLOTS) {
        float4 g[U];
        float xv[U][C1_MAXT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = rb + (long)u * SLOTS;
            const bool ok = r < r1;               // past the chunk: px may point beyond the last sample -- nothing is read
            g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < C1_MAXT; ++t) xv[u][t] = 0.f;
            if (ok) {
                g[u] = tn_ld4(dy + (size_t)r * p.y_ctot + p.y_coff + 4 * lp);
                if (ymask) {
                    const float4 yy = tn_ld4(ymask + (size_t)r * p.y_ctot + p.y_coff + 4 * lp);
                    g[u].x *= act_grad_from_out(yy.x, p.act, p.slope); g[u].y *= act_grad_from_out(yy.y, p.act, p.slope);
                    g[u].z *= act_grad_from_out(yy.z, p.act, p.slope); g[u].w *= act_grad_from_out(yy.w, p.act, p.slope);
                }
                const float* xn = x + (size_t)px.n * p.H * p.W;
                const int ih0 = px.oh * p.SH - p.PH, iw0 = px.ow * p.SW - p.PW;
#pragma unroll
                for (int t = 0; t < C1_MAXT; ++t) {
                    if (t < TT) {
                        const int ih = ih0 + tkh[t], iw = iw0 + tkw[t];
                        if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) xv[u][t] = xn[ih * p.W + iw];
                    }
                }
            }
            px.advance(p, SLOTS);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < C1_MAXT; ++t) {
                acc[t].x += g[u].x * xv[u][t]; acc[t].y += g[u].y * xv[u][t]; acc[t].z += g[u].z * xv[u][t]; acc[t].w += g[u].w * xv[u][t];
            }
    }
#pragma unroll
    for (int t = 0; t < C1_MAXT; ++t) {
        if (t < TT) {
            red[slot][(4 * lp) * TT + t] = acc[t].x; red[slot][(4 * lp + 1) * TT + t] = acc[t].y;
            red[slot][(4 * lp + 2) * TT + t] = acc[t].z; red[slot][(4 * lp + 3) * TT + t] = acc[t].w;
        }
    }
    __syncthreads();
    // one atomic per (channel, tap) and workgroup: the grid is kept at ~512 workgroups so that these do not queue up on the
    // few cache lines of dw (2048 workgroups: 55 us for a 47 MB pass, measured)
    for (int i = threadIdx.x; i < p.Cy * TT; i += 256) {
        float s = 0.f;
#pragma unroll 8
        for (int q = 0; q < SLOTS; ++q) s += red[q][i];
        atomicAdd(&dw[i], s);
    }
}

static int c1_check(const MgvaeConvDesc* d) {
    if (!d || d->Cx != 1 || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->KH <= 0 || d->KW <= 0 || d->SH <= 0 || d->SW <= 0 || d->PH < 0 || d->PW < 0)
        return MGVAE_EINVAL;
    if (d->KH * d->KW > C1_MAXT) return MGVAE_EINVAL;
    if (d->Cy != 16 && d->Cy != 32 && d->Cy != 64) return MGVAE_EINVAL;
    if (d->OH != (d->H + 2 * d->PH - d->KH) / d->SH + 1 || d->OW != (d->W + 2 * d->PW - d->KW) / d->SW + 1 || d->OH <= 0 || d->OW <= 0)
        return MGVAE_EINVAL;
    if (d->y_coff < 0 || d->y_coff + d->Cy > d->y_ctot || (d->y_ctot & 3) || (d->y_coff & 3) || d->x_ctot != 1 || d->x_coff != 0) return MGVAE_EINVAL;
    if ((long)d->N * d->OH * d->OW * d->y_ctot >= (1L << 31)) return MGVAE_EINVAL;
    return MGVAE_OK;
}
static C1P c1_params(const MgvaeConvDesc* d, int groups = 2048) {
    C1P p;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cy = d->Cy; p.OH = d->OH; p.OW = d->OW; p.KH = d->KH; p.KW = d->KW; p.SH = d->SH; p.SW = d->SW;
    p.PH = d->PH; p.PW = d->PW; p.y_ctot = d->y_ctot; p.y_coff = d->y_coff; p.act = d->act; p.slope = d->slope;
    p.rows = (long)d->N * d->OH * d->OW;
    const int pass = 256 / (d->Cy / 4);
    long pb = (p.rows + groups - 1) / groups;       // ~2048 workgroups (forward), ~512 (the reducing weight gradient)
    if (pb < 4 * pass) pb = 4 * pass;
    p.PB = (int)((pb + pass - 1) / pass * pass);
    return p;
}
#define C1_LAUNCH(KERNEL, T, ...)                                                                                              \
    switch (d->Cy) {                                                                                                          \
        case 16: hipLaunchKernelGGL((KERNEL<4, T>), dim3(cdiv(p.rows, p.PB)), dim3(256), 0, s, __VA_ARGS__); break;           \
        case 32: hipLaunchKernelGGL((KERNEL<8, T>), dim3(cdiv(p.rows, p.PB)), dim3(256), 0, s, __VA_ARGS__); break;           \
        default: hipLaunchKernelGGL((KERNEL<16, T>), dim3(cdiv(p.rows, p.PB)), dim3(256), 0, s, __VA_ARGS__); break;          \
    }

extern "C" int mgvae_conv2d_c1_nhwc_fwd(const MgvaeConvDesc* d, const float* x, const float* w, void* y, int storage, void* stream) {
    int rc = c1_check(d);
    if (rc) return rc;
    if (!x || !w || !y || (storage != MGVAE_STORE_F32 && storage != MGVAE_STORE_BF16) || d->act == MGVAE_ACT_SIGMOID) return MGVAE_EINVAL;
    const C1P p = c1_params(d);
    hipStream_t s = as_stream(stream);
    if (storage == MGVAE_STORE_BF16) { C1_LAUNCH(c1_nhwc_fwd_kernel, __bf16, x, w, static_cast<__bf16*>(y), p) }
    else { C1_LAUNCH(c1_nhwc_fwd_kernel, float, x, w, static_cast<float*>(y), p) }
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

extern "C" int mgvae_conv2d_c1_nhwc_bwd_weight(const MgvaeConvDesc* d, const float* x, const void* dy, const void* ymask, float* dw,
                                               int storage, void* stream) {
    int rc = c1_check(d);
    if (rc) return rc;
    if (!x || !dy || !dw || (storage != MGVAE_STORE_F32 && storage != MGVAE_STORE_BF16) || d->act == MGVAE_ACT_SIGMOID) return MGVAE_EINVAL;
    const C1P p = c1_params(d, 512);
    hipStream_t s = as_stream(stream);
    if (storage == MGVAE_STORE_BF16) {
        C1_LAUNCH(c1_nhwc_wgrad_kernel, __bf16, x, static_cast<const __bf16*>(dy), static_cast<const __bf16*>(ymask), dw, p)
    } else {
        C1_LAUNCH(c1_nhwc_wgrad_kernel, float, x, static_cast<const float*>(dy), static_cast<const float*>(ymask), dw, p)
    }
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// ------------------------------------------------------------------------------------------------ C channels -> 1 (1x1)
// y[r] = act(sum_c w[c] x[r, c]); LP = C / 4 lanes per pixel row
template <int LP, typename T>
__global__ __launch_bounds__(256) void to1_nhwc_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                           long rows, int PB, int x_ctot, int x_coff, int act, float slope) {
    const int lp = threadIdx.x % LP, slot = threadIdx.x / LP;
    const float4 wv = make_float4(w[4 * lp], w[4 * lp + 1], w[4 * lp + 2], w[4 * lp + 3]);      // (a slice of a flat parameter buffer: 4-byte aligned)
    const long r0 = (long)blockIdx.x * PB, r1 = r0 + PB < rows ? r0 + PB : rows;
    for (long r = r0 + slot; r < r1; r += 256 / LP) {
        const float4 v = tn_ld4(x + (size_t)r * x_ctot + x_coff + 4 * lp);
        float s = (v.x * wv.x + v.y * wv.y) + (v.z * wv.z + v.w * wv.w);
#pragma unroll
        for (int o = LP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, LP);
        if (lp == 0) y[r] = apply_act(s, act, slope);
    }
}
// g = dy * act'(y);  dx[r, c] = g w[c];  dw[c] += sum_r g x[r, c]
template <int LP, typename T>
__global__ __launch_bounds__(256) void to1_nhwc_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ y,
                                                           const float* __restrict__ dy, T* __restrict__ dx, float* __restrict__ dw,
                                                           long rows, int PB, int x_ctot, int x_coff, int act, float slope) {
    constexpr int SLOTS = 256 / LP, U = 4;
    __shared__ float red[SLOTS][4 * LP + 1];
    const int lp = threadIdx.x % LP, slot = threadIdx.x / LP;
    const float4 wv = make_float4(w[4 * lp], w[4 * lp + 1], w[4 * lp + 2], w[4 * lp + 3]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long r0 = (long)blockIdx.x * PB, r1 = r0 + PB < rows ? r0 + PB : rows;
    for (long rb = r0 + slot; rb < r1; rb += (long)U * SLOTS) {
        float g[U];
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = rb + (long)u * SLOTS;
            const bool ok = r < r1;
            const long rr = ok ? r : r0;
            g[u] = ok ? dy[rr] * act_grad_from_out(y[rr], act, slope) : 0.f;
            v[u] = dw ? tn_ld4(x + (size_t)rr * x_ctot + x_coff + 4 * lp) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = rb + (long)u * SLOTS;
            acc.x += g[u] * v[u].x; acc.y += g[u] * v[u].y; acc.z += g[u] * v[u].z; acc.w += g[u] * v[u].w;
            if (dx && r < r1) tn_st4(dx + (size_t)r * x_ctot + x_coff + 4 * lp, make_float4(g[u] * wv.x, g[u] * wv.y, g[u] * wv.z, g[u] * wv.w));
        }
    }
    if (!dw) return;
    red[slot][4 * lp] = acc.x; red[slot][4 * lp + 1] = acc.y; red[slot][4 * lp + 2] = acc.z; red[slot][4 * lp + 3] = acc.w;
    __syncthreads();
    if (threadIdx.x < 4 * LP) {
        float s = 0.f;
#pragma unroll 8
        for (int q = 0; q < SLOTS; ++q) s += red[q][threadIdx.x];
        atomicAdd(&dw[threadIdx.x], s);
    }
}

static int to1_check(long rows, int C, int x_ctot, int x_coff, int storage) {
    if (rows <= 0 || (C != 16 && C != 32 && C != 64 && C != 128 && C != 256)) return MGVAE_EINVAL;
    if (x_coff < 0 || x_coff + C > x_ctot || (x_ctot & 3) || (x_coff & 3)) return MGVAE_EINVAL;
    if (storage != MGVAE_STORE_F32 && storage != MGVAE_STORE_BF16) return MGVAE_EINVAL;
    if (rows * x_ctot >= (1L << 40)) return MGVAE_EINVAL;
    return MGVAE_OK;
}
static int to1_pb(long rows, int C, int groups = 2048) {
    const int pass = 256 / (C / 4);
    long pb = (rows + groups - 1) / groups;
    if (pb < 4 * pass) pb = 4 * pass;
    return (int)((pb + pass - 1) / pass * pass);
}
#define TO1_LAUNCH(KERNEL, T, ...)                                                                                             \
    switch (C) {                                                                                                              \
        case 16: hipLaunchKernelGGL((KERNEL<4, T>), dim3(cdiv(rows, PB)), dim3(256), 0, s, __VA_ARGS__); break;               \
        case 32: hipLaunchKernelGGL((KERNEL<8, T>), dim3(cdiv(rows, PB)), dim3(256), 0, s, __VA_ARGS__); break;               \
        case 64: hipLaunchKernelGGL((KERNEL<16, T>), dim3(cdiv(rows, PB)), dim3(256), 0, s, __VA_ARGS__); break;              \
        case 128: hipLaunchKernelGGL((KERNEL<32, T>), dim3(cdiv(rows, PB)), dim3(256), 0, s, __VA_ARGS__); break;             \
        default: hipLaunchKernelGGL((KERNEL<64, T>), dim3(cdiv(rows, PB)), dim3(256), 0, s, __VA_ARGS__); break;              \
    }

extern "C" int mgvae_conv2d_to1_nhwc_fwd(const void* x, const float* w, float* y, long rows, int C, int x_ctot, int x_coff, int act,
                                         float slope, int storage, void* stream) {
    int rc = to1_check(rows, C, x_ctot, x_coff, storage);
    if (rc) return rc;
    if (!x || !w || !y) return MGVAE_EINVAL;
    const int PB = to1_pb(rows, C);
    hipStream_t s = as_stream(stream);
    if (storage == MGVAE_STORE_BF16) { TO1_LAUNCH(to1_nhwc_fwd_kernel, __bf16, static_cast<const __bf16*>(x), w, y, rows, PB, x_ctot, x_coff, act, slope) }
    else { TO1_LAUNCH(to1_nhwc_fwd_kernel, float, static_cast<const float*>(x), w, y, rows, PB, x_ctot, x_coff, act, slope) }
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

extern "C" int mgvae_conv2d_to1_nhwc_bwd(const void* x, const float* w, const float* y, const float* dy, void* dx, float* dw, long rows,
                                         int C, int x_ctot, int x_coff, int act, float slope, int storage, void* stream) {
    int rc = to1_check(rows, C, x_ctot, x_coff, storage);
    if (rc) return rc;
    if (!x || !w || !y || !dy || (!dx && !dw)) return MGVAE_EINVAL;
    const int PB = to1_pb(rows, C, dw ? 512 : 2048);         // the weight gradient ends in atomics on C addresses: fewer, fatter workgroups
    hipStream_t s = as_stream(stream);
    if (storage == MGVAE_STORE_BF16) {
        TO1_LAUNCH(to1_nhwc_bwd_kernel, __bf16, static_cast<const __bf16*>(x), w, y, dy, static_cast<__bf16*>(dx), dw, rows, PB, x_ctot, x_coff,
                   act, slope)
    } else {
        TO1_LAUNCH(to1_nhwc_bwd_kernel, float, static_cast<const float*>(x), w, y, dy, static_cast<float*>(dx), dw, rows, PB, x_ctot, x_coff, act,
                   slope)
    }
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// ------------------------------------------------------------------------------------------------ storage-type change
// a dense tensor fp32 <-> bf16 (the fp32 stems' concat entering a bf16 island, its gradient coming back); n % 4 == 0
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_storage_kernel(const TS* __restrict__ src, TD* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) tn_st4(dst + 4 * i, tn_ld4(src + 4 * i));
}
extern "C" int mgvae_cast_storage(const void* src, int src_storage, void* dst, int dst_storage, size_t n, void* stream) {
    if (!src || !dst || n == 0 || (n & 3) || src_storage == dst_storage) return MGVAE_EINVAL;
    if ((src_storage != MGVAE_STORE_F32 && src_storage != MGVAE_STORE_BF16) || (dst_storage != MGVAE_STORE_F32 && dst_storage != MGVAE_STORE_BF16))
        return MGVAE_EINVAL;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    if (src_storage == MGVAE_STORE_F32)
        hipLaunchKernelGGL((cast_storage_kernel<float, __bf16>), dim3(blocks), dim3(256), 0, as_stream(stream), static_cast<const float*>(src),
                           static_cast<__bf16*>(dst), n4);
    else
        hipLaunchKernelGGL((cast_storage_kernel<__bf16, float>), dim3(blocks), dim3(256), 0, as_stream(stream), static_cast<const __bf16*>(src),
                           static_cast<float*>(dst), n4);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}
