// CBAM (channel attention then spatial attention) fused with the residual + activation
// that follows it in every block of the reference (graph/cbam.py, graph/encodingBlock.py,
// graph/decoder.py).  HBM-bound: the feature map u is swept twice in forward (channel
// pool; spatial pool) plus one apply pass, reductions over H*W use wavefront shuffles
// (one wave per (n, c) plane) and reductions over C keep pixels on the lanes so every
// load is coalesced in NCHW.
//
//   cg[n,c]  = sigmoid(W2 relu(W1 avg_hw(u)) + W2 relu(W1 max_hw(u)))
//   v        = u * cg
//   sg[n,p]  = sigmoid(conv3x3([mean_c v, max_c v]))
//   o        = v * sg
//   y        = o                      (mode 0)
//            = act(u   + o)           (mode 1)
//            = act(res + o)           (mode 2)
#include "mgvae_common.h"

struct CbamSave {   // carve-up of the `save` workspace (floats)
    float *cg, *avg, *mx, *hid, *s_in, *sg;
    int *amax_hw, *amax_c;
};
static __host__ __device__ inline size_t cbam_save_floats(int N, int C, int P) {
    return (size_t)4 * N * C + (((size_t)2 * N * (C / 16) + 3) & ~(size_t)3) + (size_t)4 * N * P;   // hid padded: 16-B aligned maps
}
static inline CbamSave carve(float* s, int N, int C, int P) {
    CbamSave r;
    const size_t nc = (size_t)N * C, np = (size_t)N * P;
    r.cg = s; r.avg = s + nc; r.mx = s + 2 * nc; r.amax_hw = reinterpret_cast<int*>(s + 3 * nc);
    r.hid = s + 4 * nc;
    float* q = r.hid + (((size_t)2 * N * (C / 16) + 3) & ~(size_t)3);
    r.s_in = q; r.amax_c = reinterpret_cast<int*>(q + 2 * np); r.sg = q + 3 * np;
    return r;
}

// ---------------------------------------------------------------- F1: channel pooling
__global__ __launch_bounds__(256) void cbam_chan_pool_kernel(const float* __restrict__ u, float* __restrict__ avg,
                                                             float* __restrict__ mx, int* __restrict__ amax,
                                                             int NC, int P) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nc = blockIdx.x * 4 + wave;
    if (nc >= NC) return;
    const float* up = u + (size_t)nc * P;
    float s = 0.f, m = -INFINITY;
    int mi = 0x7fffffff;
    for (int i = lane; i < P; i += 64) {
        const float v = up[i];
        s += v;
        if (v > m) { m = v; mi = i; }   // strict >: first occurrence wins inside a lane
    }
    s = wave_sum(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if (lane == 0) { avg[nc] = s / (float)P; mx[nc] = m; amax[nc] = mi; }
}

// ---------------------------------------------------------------- F2: shared MLP + sigmoid
// F2a: hidden[n, {avg,max}, j] = relu(W1[j,:] . pool[n,:]) -- one wave per (n, j), so the launch has
// N*Cr waves instead of N latency-bound workgroups.  F2b: gate[n,c], one thread per (n, c).
__global__ __launch_bounds__(256) void cbam_chan_hidden_kernel(const float* __restrict__ avg, const float* __restrict__ mx,
                                                               const float* __restrict__ w1, float* __restrict__ hid,
                                                               int N, int C) {
    const int Cr = C / 16;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int id = blockIdx.x * 4 + wave;
    if (id >= N * Cr) return;
    const int n = id / Cr, j = id - n * Cr;
    const float* wr = w1 + (size_t)j * C;
    const float* ap = avg + (size_t)n * C;
    const float* mp = mx + (size_t)n * C;
    float a = 0.f, m = 0.f;
    for (int c = lane; c < C; c += 64) { const float w = wr[c]; a += w * ap[c]; m += w * mp[c]; }
    a = wave_sum(a); m = wave_sum(m);
    if (lane == 0) {
        hid[(size_t)n * 2 * Cr + j] = a > 0.f ? a : 0.f;
        hid[(size_t)n * 2 * Cr + Cr + j] = m > 0.f ? m : 0.f;
    }
}

__global__ __launch_bounds__(256) void cbam_chan_gate_kernel(const float* __restrict__ hid, const float* __restrict__ w2,
                                                             float* __restrict__ cg, int N, int C) {
    const int Cr = C / 16;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= N * C) return;
    const int n = id / C, c = id - n * C;
    const float* ha = hid + (size_t)n * 2 * Cr;
    const float* hm = ha + Cr;
    const float* wr = w2 + (size_t)c * Cr;
    float a = 0.f, m = 0.f;
    for (int j = 0; j < Cr; ++j) { const float w = wr[j]; a += w * ha[j]; m += w * hm[j]; }
    cg[id] = 1.f / (1.f + expf(-(a + m)));
}

// ---------------------------------------------------------------- F3: spatial pooling
// PXB pixels x (256/PXB) channel groups per workgroup; pixels run along the lanes (coalesced in
// NCHW).  PXB shrinks for small maps so the launch still fills the chip and the per-thread
// channel loop stays short.
template <int PXB>
__global__ __launch_bounds__(256) void cbam_spatial_pool_kernel(const float* __restrict__ u, const float* __restrict__ cg,
                                                                float* __restrict__ s_in, int* __restrict__ amax_c,
                                                                int N, int C, int P) {
    constexpr int CGB = 256 / PXB;
    __shared__ float ssum[CGB][PXB], smax[CGB][PXB];
    __shared__ int sidx[CGB][PXB];
    const int px = threadIdx.x % PXB, grp = threadIdx.x / PXB;
    const long gp = (long)blockIdx.x * PXB + px;
    const bool ok = gp < (long)N * P;
    const int n = ok ? (int)(gp / P) : 0, pp = ok ? (int)(gp - (long)n * P) : 0;
    float s = 0.f, m = -INFINITY;
    int mi = 0x7fffffff;
    if (ok) {
        const float* up = u + (size_t)n * C * P + pp;
        const float* gp_ = cg + (size_t)n * C;
        for (int c = grp; c < C; c += CGB) {
            const float v = up[(size_t)c * P] * gp_[c];
            s += v;
            if (v > m) { m = v; mi = c; }
        }
    }
    ssum[grp][px] = s; smax[grp][px] = m; sidx[grp][px] = mi;
    __syncthreads();
    if (grp == 0 && ok) {
        float ts = 0.f, tm = -INFINITY; int ti = 0x7fffffff;
#pragma unroll 4
        for (int g = 0; g < CGB; ++g) {
            ts += ssum[g][px];
            const float om = smax[g][px]; const int oi = sidx[g][px];
            if (om > tm || (om == tm && oi < ti)) { tm = om; ti = oi; }
        }
        s_in[(size_t)n * 2 * P + pp] = ts / (float)C;
        s_in[(size_t)n * 2 * P + P + pp] = tm;
        amax_c[(size_t)n * P + pp] = ti;
    }
}

static inline int pick_pxb(long total_px) {
    if (total_px >= 64 * 512) return 64;
    if (total_px >= 32 * 512) return 32;
    if (total_px >= 16 * 256) return 16;
    return 8;
}

// ---------------------------------------------------------------- F4: 3x3 conv (2->1) + sigmoid
__global__ __launch_bounds__(256) void cbam_spatial_gate_kernel(const float* __restrict__ s_in, const float* __restrict__ wsp,
                                                                float* __restrict__ sg, int N, int H, int W) {
    const int P = H * W;
    const long gp = (long)blockIdx.x * 256 + threadIdx.x;
    if (gp >= (long)N * P) return;
    const int n = (int)(gp / P), pp = (int)(gp - (long)n * P);
    const int h = pp / W, w = pp - h * W;
    float acc = 0.f;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const float* sp = s_in + ((size_t)n * 2 + ch) * P;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = h + kh - 1;
            if ((unsigned)hh >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ww = w + kw - 1;
                if ((unsigned)ww >= (unsigned)W) continue;
                acc += sp[hh * W + ww] * wsp[ch * 9 + kh * 3 + kw];
            }
        }
    }
    sg[gp] = 1.f / (1.f + expf(-acc));
}

// ---------------------------------------------------------------- F5: apply + residual + act
__global__ __launch_bounds__(256) void cbam_apply_kernel(const float* __restrict__ u, const float* __restrict__ res,
                                                         const float* __restrict__ cg, const float* __restrict__ sg,
                                                         float* __restrict__ y, int N, int C, int P, int y_ctot,
                                                         int y_coff, int mode, int act, float slope) {
    const long total = (long)N * C * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int nc = (int)(i / P), pp = (int)(i - (long)nc * P);
        const int n = nc / C, c = nc - n * C;
        const float uu = u[i];
        const float o = (uu * cg[nc]) * sg[(size_t)n * P + pp];
        float v;
        if (mode == 0) v = o;
        else if (mode == 1) v = apply_act(uu + o, act, slope);
        else v = apply_act(res[i] + o, act, slope);
        y[((size_t)n * y_ctot + y_coff + c) * P + pp] = v;
    }
}

// float4 variant (P % 4 == 0: four pixels of one plane per thread)
__global__ __launch_bounds__(256) void cbam_apply_vec_kernel(const float* __restrict__ u, const float* __restrict__ res,
                                                             const float* __restrict__ cg, const float* __restrict__ sg,
                                                             float* __restrict__ y, int N, int C, int P, int y_ctot,
                                                             int y_coff, int mode, int act, float slope) {
    const long total4 = (long)N * C * P / 4;
    for (long i4 = (long)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * 256) {
        const long i = i4 * 4;
        const int nc = (int)(i / P), pp = (int)(i - (long)nc * P);
        const int n = nc / C, c = nc - n * C;
        const float4 uu = *reinterpret_cast<const float4*>(u + i);
        const float4 sv = *reinterpret_cast<const float4*>(sg + (size_t)n * P + pp);
        const float g = cg[nc];
        float4 o = make_float4((uu.x * g) * sv.x, (uu.y * g) * sv.y, (uu.z * g) * sv.z, (uu.w * g) * sv.w);
        if (mode != 0) {
            float4 b = uu;
            if (mode == 2) b = *reinterpret_cast<const float4*>(res + i);
            o.x = apply_act(b.x + o.x, act, slope); o.y = apply_act(b.y + o.y, act, slope);
            o.z = apply_act(b.z + o.z, act, slope); o.w = apply_act(b.w + o.w, act, slope);
        }
        *reinterpret_cast<float4*>(y + ((size_t)n * y_ctot + y_coff + c) * P + pp) = o;
    }
}

// ================================================================ backward
// B1: dt[n,p] = (sum_c g * u * cg) * sg * (1 - sg),  g = dy * act'(y)
template <int PXB>
__global__ __launch_bounds__(256) void cbam_bwd_spatial_kernel(const float* __restrict__ u, const float* __restrict__ y,
                                                               const float* __restrict__ dy, const float* __restrict__ cg,
                                                               const float* __restrict__ sg, float* __restrict__ dt,
                                                               int N, int C, int P, int y_ctot, int y_coff, int mode,
                                                               int act, float slope) {
    constexpr int CGB = 256 / PXB;
    __shared__ float ssum[CGB][PXB];
    const int px = threadIdx.x % PXB, grp = threadIdx.x / PXB;
    const long gp = (long)blockIdx.x * PXB + px;
    const bool ok = gp < (long)N * P;
    const int n = ok ? (int)(gp / P) : 0, pp = ok ? (int)(gp - (long)n * P) : 0;
    float s = 0.f;
    if (ok) {
        const float* up = u + (size_t)n * C * P + pp;
        const float* yp = y + ((size_t)n * y_ctot + y_coff) * P + pp;
        const float* dyp = dy + ((size_t)n * y_ctot + y_coff) * P + pp;
        const float* gp_ = cg + (size_t)n * C;
        for (int c = grp; c < C; c += CGB) {
            float g = dyp[(size_t)c * P];
            if (mode != 0) g *= act_grad_from_out(yp[(size_t)c * P], act, slope);
            s += g * (up[(size_t)c * P] * gp_[c]);
        }
    }
    ssum[grp][px] = s;
    __syncthreads();
    if (grp == 0 && ok) {
        float t = 0.f;
#pragma unroll 4
        for (int g = 0; g < CGB; ++g) t += ssum[g][px];
        const float sgg = sg[gp];
        dt[gp] = t * sgg * (1.f - sgg);
    }
}

// B2: ds_in = conv_transpose3x3(dt, wsp); dwsp += corr(s_in, dt)
__global__ __launch_bounds__(256) void cbam_bwd_sgate_kernel(const float* __restrict__ dt, const float* __restrict__ s_in,
                                                             const float* __restrict__ wsp, float* __restrict__ ds_in,
                                                             float* __restrict__ dwsp, int N, int H, int W) {
    __shared__ float red[4][18];
    const int P = H * W;
    const long gp = (long)blockIdx.x * 256 + threadIdx.x;
    const bool ok = gp < (long)N * P;
    float dw[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) dw[i] = 0.f;
    if (ok) {
        const int n = (int)(gp / P), pp = (int)(gp - (long)n * P);
        const int h = pp / W, w = pp - h * W;
        const float* dtp = dt + (size_t)n * P;
        const float mydt = dtp[pp];
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                // gradient wrt s_in[h][w]: outputs at (h+1-kh, w+1-kw) used tap (kh,kw) on it
                const int oh = h + 1 - kh, ow = w + 1 - kw;
                if ((unsigned)oh < (unsigned)H && (unsigned)ow < (unsigned)W) {
                    const float t = dtp[oh * W + ow];
                    d0 += t * wsp[kh * 3 + kw];
                    d1 += t * wsp[9 + kh * 3 + kw];
                }
                // weight gradient: output (h,w) read s_in at (h+kh-1, w+kw-1)
                const int ih = h + kh - 1, iw = w + kw - 1;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
                    dw[kh * 3 + kw] += mydt * s_in[((size_t)n * 2) * P + ih * W + iw];
                    dw[9 + kh * 3 + kw] += mydt * s_in[((size_t)n * 2 + 1) * P + ih * W + iw];
                }
            }
        }
        ds_in[((size_t)n * 2) * P + pp] = d0;
        ds_in[((size_t)n * 2 + 1) * P + pp] = d1;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const float s = wave_sum(dw[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 18 && dwsp)
        atomicAdd(&dwsp[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// B3: per (n,c) plane: dv, dcg, du (without the channel-pool terms), dres
__global__ __launch_bounds__(256) void cbam_bwd_channel_kernel(
    const float* __restrict__ u, const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ cg,
    const float* __restrict__ sg, const float* __restrict__ ds_in, const int* __restrict__ amax_c,
    float* __restrict__ du, float* __restrict__ dres, float* __restrict__ dcg, int N, int C, int P, int y_ctot,
    int y_coff, int mode, int act, float slope) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nc = blockIdx.x * 4 + wave;
    if (nc >= N * C) return;
    const int n = nc / C, c = nc - n * C;
    const float* up = u + (size_t)nc * P;
    const float* yp = y + ((size_t)n * y_ctot + y_coff + c) * P;
    const float* dyp = dy + ((size_t)n * y_ctot + y_coff + c) * P;
    const float* sgp = sg + (size_t)n * P;
    const float* dmean = ds_in + (size_t)n * 2 * P;
    const float* dmax = dmean + P;
    const int* amp = amax_c + (size_t)n * P;
    const float g_c = cg[nc], invC = 1.f / (float)C;
    float acc = 0.f;
    for (int i = lane; i < P; i += 64) {
        float g = dyp[i];
        if (mode != 0) g *= act_grad_from_out(yp[i], act, slope);
        float dv = g * sgp[i] + dmean[i] * invC;
        if (amp[i] == c) dv += dmax[i];
        acc += dv * up[i];
        du[(size_t)nc * P + i] = dv * g_c + (mode == 1 ? g : 0.f);
        if (mode == 2) dres[(size_t)nc * P + i] = g;
    }
    acc = wave_sum(acc);
    if (lane == 0) dcg[nc] = acc * g_c * (1.f - g_c);      // = d(pre-sigmoid): the gate MLP backward starts from it
}

// float4 variant of B3 (P % 4 == 0)
__global__ __launch_bounds__(256) void cbam_bwd_channel_vec_kernel(
    const float* __restrict__ u, const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ cg,
    const float* __restrict__ sg, const float* __restrict__ ds_in, const int* __restrict__ amax_c,
    float* __restrict__ du, float* __restrict__ dres, float* __restrict__ dcg, int N, int C, int P, int y_ctot,
    int y_coff, int mode, int act, float slope) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nc = blockIdx.x * 4 + wave;
    if (nc >= N * C) return;
    const int n = nc / C, c = nc - n * C;
    const float4* up = reinterpret_cast<const float4*>(u + (size_t)nc * P);
    const float4* yp = reinterpret_cast<const float4*>(y + ((size_t)n * y_ctot + y_coff + c) * P);
    const float4* dyp = reinterpret_cast<const float4*>(dy + ((size_t)n * y_ctot + y_coff + c) * P);
    const float4* sgp = reinterpret_cast<const float4*>(sg + (size_t)n * P);
    const float4* dmean = reinterpret_cast<const float4*>(ds_in + (size_t)n * 2 * P);
    const float4* dmax = reinterpret_cast<const float4*>(ds_in + (size_t)n * 2 * P + P);
    const int4* amp = reinterpret_cast<const int4*>(amax_c + (size_t)n * P);
    float4* dup = reinterpret_cast<float4*>(du + (size_t)nc * P);
    float4* drp = mode == 2 ? reinterpret_cast<float4*>(dres + (size_t)nc * P) : nullptr;
    const float g_c = cg[nc], invC = 1.f / (float)C;
    float acc = 0.f;
    for (int i = lane; i < (P >> 2); i += 64) {
        float4 g = dyp[i];
        if (mode != 0) {
            const float4 yy = yp[i];
            g.x *= act_grad_from_out(yy.x, act, slope); g.y *= act_grad_from_out(yy.y, act, slope);
            g.z *= act_grad_from_out(yy.z, act, slope); g.w *= act_grad_from_out(yy.w, act, slope);
        }
        const float4 s4 = sgp[i], dm = dmean[i], dx = dmax[i], uu = up[i];
        const int4 am = amp[i];
        float4 dv = make_float4(g.x * s4.x + dm.x * invC, g.y * s4.y + dm.y * invC, g.z * s4.z + dm.z * invC,
                                g.w * s4.w + dm.w * invC);
        if (am.x == c) dv.x += dx.x;
        if (am.y == c) dv.y += dx.y;
        if (am.z == c) dv.z += dx.z;
        if (am.w == c) dv.w += dx.w;
        acc += (dv.x * uu.x + dv.y * uu.y) + (dv.z * uu.z + dv.w * uu.w);
        const float k = mode == 1 ? 1.f : 0.f;
        dup[i] = make_float4(dv.x * g_c + k * g.x, dv.y * g_c + k * g.y, dv.z * g_c + k * g.z, dv.w * g_c + k * g.w);
        if (mode == 2) drp[i] = g;
    }
    acc = wave_sum(acc);
    if (lane == 0) dcg[nc] = acc * g_c * (1.f - g_c);      // = d(pre-sigmoid): the gate MLP backward starts from it
}

// small planes (P <= G*E): G lanes per (n, c) plane, several planes per wave (see norm.hip, small planes)
template <int G, int E>
__global__ __launch_bounds__(256) void cbam_bwd_channel_mini_kernel(
    const float* __restrict__ u, const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ cg,
    const float* __restrict__ sg, const float* __restrict__ ds_in, const int* __restrict__ amax_c,
    float* __restrict__ du, float* __restrict__ dres, float* __restrict__ dcg, int N, int C, int P, int y_ctot,
    int y_coff, int mode, int act, float slope) {
    const int l = threadIdx.x % G;
    const int plane = (blockIdx.x * 256 + threadIdx.x) / G;
    const bool live = plane < N * C;
    const int nc = live ? plane : N * C - 1;
    const int n = nc / C, c = nc - n * C;
    const float* up = u + (size_t)nc * P;
    const float* yp = y + ((size_t)n * y_ctot + y_coff + c) * P;
    const float* dyp = dy + ((size_t)n * y_ctot + y_coff + c) * P;
    const float* sgp = sg + (size_t)n * P;
    const float* dmean = ds_in + (size_t)n * 2 * P;
    const float* dmax = dmean + P;
    const int* amp = amax_c + (size_t)n * P;
    const float g_c = cg[nc], invC = 1.f / (float)C;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int i = l + G * k;
        if (i < P) {
            float g = dyp[i];
            if (mode != 0) g *= act_grad_from_out(yp[i], act, slope);
            float dv = g * sgp[i] + dmean[i] * invC;
            if (amp[i] == c) dv += dmax[i];
            acc += dv * up[i];
            if (live) {
                du[(size_t)nc * P + i] = dv * g_c + (mode == 1 ? g : 0.f);
                if (mode == 2) dres[(size_t)nc * P + i] = g;
            }
        }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
    if (l == 0 && live) dcg[nc] = acc * g_c * (1.f - g_c);
}

// B4a: MLP backward, fully parallel: dpre[n,c] = dcg * cg * (1 - cg) is already what B3 stored; (1) one wave per
// (n, j): dh = W2[:,j] . dpre[n,:], masked by the two ReLUs; (2) one thread per (n, c): davg / dmaxp.
__global__ __launch_bounds__(256) void cbam_bwd_dh_kernel(const float* __restrict__ dpre, const float* __restrict__ hid,
                                                          const float* __restrict__ w2, float* __restrict__ dh, int N, int C) {
    const int Cr = C / 16;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int id = blockIdx.x * 4 + wave;
    if (id >= N * Cr) return;
    const int n = id / Cr, j = id - n * Cr;
    const float* dp = dpre + (size_t)n * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += w2[(size_t)c * Cr + j] * dp[c];
    s = wave_sum(s);
    if (lane == 0) {
        dh[(size_t)n * 2 * Cr + j] = hid[(size_t)n * 2 * Cr + j] > 0.f ? s : 0.f;
        dh[(size_t)n * 2 * Cr + Cr + j] = hid[(size_t)n * 2 * Cr + Cr + j] > 0.f ? s : 0.f;
    }
}

__global__ __launch_bounds__(256) void cbam_bwd_dpool_kernel(const float* __restrict__ dh, const float* __restrict__ w1,
                                                             float* __restrict__ davg, float* __restrict__ dmaxp, int N, int C) {
    const int Cr = C / 16;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= N * C) return;
    const int n = id / C, c = id - n * C;
    const float* da = dh + (size_t)n * 2 * Cr;
    const float* dm = da + Cr;
    float a = 0.f, m = 0.f;
    for (int j = 0; j < Cr; ++j) { const float w = w1[(size_t)j * C + c]; a += w * da[j]; m += w * dm[j]; }
    davg[id] = a; dmaxp[id] = m;
}

// B4b: weight gradients as a batched reduction over the samples; one thread owns one (c, j)
// pair of BOTH matrices for a chunk of samples (blockIdx.y).  Small C has few (c, j) pairs, so the
// sample axis supplies the parallelism there (chunks > 1 -> atomic accumulation); the loop is
// unrolled by four with independent loads so the L2 latencies overlap instead of chaining.
//   dw2[c][j] += sum_n dpre[n,c] * (ha[n,j] + hm[n,j])
//   dw1[j][c] += sum_n dha[n,j] * avg[n,c] + dhm[n,j] * max[n,c]
__global__ __launch_bounds__(256) void cbam_bwd_mlp_wgrad_kernel(const float* __restrict__ dpre, const float* __restrict__ hid,
                                                                 const float* __restrict__ dh, const float* __restrict__ avg,
                                                                 const float* __restrict__ mx, float* __restrict__ dw1,
                                                                 float* __restrict__ dw2, int N, int C, int per) {
    const int Cr = C / 16;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= C * Cr) return;
    const int c = f / Cr, j = f - c * Cr;
    const int n0 = blockIdx.y * per, n1 = min(N, n0 + per);
    float a2 = 0.f, a1 = 0.f;
    int n = n0;
    for (; n + 4 <= n1; n += 4) {
        float d[4], ha[4], hm[4], da[4], dm[4], av[4], mv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* hn = hid + (size_t)(n + q) * 2 * Cr;
            const float* dn = dh + (size_t)(n + q) * 2 * Cr;
            d[q] = dpre[(size_t)(n + q) * C + c]; ha[q] = hn[j]; hm[q] = hn[Cr + j];
            da[q] = dn[j]; dm[q] = dn[Cr + j];
            av[q] = avg[(size_t)(n + q) * C + c]; mv[q] = mx[(size_t)(n + q) * C + c];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { a2 += d[q] * (ha[q] + hm[q]); a1 += da[q] * av[q] + dm[q] * mv[q]; }
    }
    for (; n < n1; ++n) {
        const float* hn = hid + (size_t)n * 2 * Cr;
        const float* dn = dh + (size_t)n * 2 * Cr;
        a2 += dpre[(size_t)n * C + c] * (hn[j] + hn[Cr + j]);
        a1 += dn[j] * avg[(size_t)n * C + c] + dn[Cr + j] * mx[(size_t)n * C + c];
    }
    // always atomic: the same CBAM may be running its backward on another stream (two halves of a batch, see
    // graph/model.py::Model.encode_phrase), and the sample chunks of one launch share the element anyway
    if (dw2) atomicAdd(&dw2[(size_t)c * Cr + j], a2);
    if (dw1) atomicAdd(&dw1[(size_t)j * C + c], a1);
}

// B5: du += davg/P + [p == argmax_hw] dmaxp
__global__ __launch_bounds__(256) void cbam_bwd_finish_kernel(float* __restrict__ du, const float* __restrict__ davg,
                                                              const float* __restrict__ dmaxp, const int* __restrict__ amax_hw,
                                                              long total, int P) {
    const float invP = 1.f / (float)P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int nc = (int)(i / P), pp = (int)(i - (long)nc * P);
        float v = du[i] + davg[nc] * invP;
        if (amax_hw[nc] == pp) v += dmaxp[nc];
        du[i] = v;
    }
}

__global__ __launch_bounds__(256) void cbam_fill_kernel(float* __restrict__ t, float v, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) t[i] = v;
}
static void cbam_fill(float* t, float v, long n, hipStream_t s) {
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(cbam_fill_kernel, dim3(blocks), dim3(256), 0, s, t, v, n);
}

// ================================================================ host entry points
extern "C" size_t mgvae_cbam_save_floats(int N, int C, int H, int W) { return cbam_save_floats(N, C, H * W); }
extern "C" size_t mgvae_cbam_bwd_scratch_floats(int N, int C, int H, int W) {
    return (size_t)3 * N * H * W + (size_t)3 * N * C + (size_t)2 * N * (C / 16);
}

static int cbam_check(int N, int C, int H, int W, int ctot, int coff, int mode, int act) {
    if (N <= 0 || C < 16 || (C % 16) != 0 || C > 4096 || H <= 0 || W <= 0) return MGVAE_EINVAL;
    if (coff < 0 || coff + C > ctot || mode < 0 || mode > 2) return MGVAE_EINVAL;
    if (act == MGVAE_ACT_SIGMOID) return MGVAE_EINVAL;
    if ((long)N * C * H * W >= (1L << 31)) return MGVAE_EINVAL;
    return MGVAE_OK;
}

extern "C" int mgvae_cbam_fwd(const float* u, const float* res, const float* w1, const float* w2, const float* wsp,
                              float* y, float* save, int N, int C, int H, int W, int y_ctot, int y_coff, int mode,
                              int act, float slope, int parts, void* stream) {
    int rc = cbam_check(N, C, H, W, y_ctot, y_coff, mode, act);
    if (rc) return rc;
    if ((parts & 3) == 0 || parts > 7) return MGVAE_EINVAL;
    const bool chan = parts & 1, spat = parts & 2, pooled = parts & 4;
    if (!u || !y || !save || (mode == 2 && !res) || (chan && (!w1 || !w2)) || (spat && !wsp)) return MGVAE_EINVAL;
    hipStream_t s = as_stream(stream);
    const int P = H * W, NC = N * C, Cr = C / 16;
    CbamSave sv = carve(save, N, C, P);
    if (!chan) cbam_fill(sv.cg, 1.f, NC, s);
    if (!spat) cbam_fill(sv.sg, 1.f, (long)N * P, s);
    if (chan) {
    if (!pooled)   // else mgvae_instance_norm_fwd already wrote avg / max / argmax into `save`
    hipLaunchKernelGGL(cbam_chan_pool_kernel, dim3(cdiv(NC, 4)), dim3(256), 0, s, u, sv.avg, sv.mx, sv.amax_hw, NC, P);
    hipLaunchKernelGGL(cbam_chan_hidden_kernel, dim3(cdiv((long)N * Cr, 4)), dim3(256), 0, s, sv.avg, sv.mx, w1, sv.hid, N, C);
    hipLaunchKernelGGL(cbam_chan_gate_kernel, dim3(cdiv(NC, 256)), dim3(256), 0, s, sv.hid, w2, sv.cg, N, C);
    }
    if (spat) {
    switch (pick_pxb((long)N * P)) {
        case 64: hipLaunchKernelGGL(cbam_spatial_pool_kernel<64>, dim3(cdiv((long)N * P, 64)), dim3(256), 0, s, u, sv.cg, sv.s_in, sv.amax_c, N, C, P); break;
        case 32: hipLaunchKernelGGL(cbam_spatial_pool_kernel<32>, dim3(cdiv((long)N * P, 32)), dim3(256), 0, s, u, sv.cg, sv.s_in, sv.amax_c, N, C, P); break;
        case 16: hipLaunchKernelGGL(cbam_spatial_pool_kernel<16>, dim3(cdiv((long)N * P, 16)), dim3(256), 0, s, u, sv.cg, sv.s_in, sv.amax_c, N, C, P); break;
        default: hipLaunchKernelGGL(cbam_spatial_pool_kernel<8>, dim3(cdiv((long)N * P, 8)), dim3(256), 0, s, u, sv.cg, sv.s_in, sv.amax_c, N, C, P); break;
    }
    hipLaunchKernelGGL(cbam_spatial_gate_kernel, dim3(cdiv((long)N * P, 256)), dim3(256), 0, s, sv.s_in, wsp, sv.sg, N, H, W);
    }
    const long total = (long)NC * P;
    const int blocks = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
    if ((P & 3) == 0)
        hipLaunchKernelGGL(cbam_apply_vec_kernel, dim3(blocks), dim3(256), 0, s, u, res, sv.cg, sv.sg, y, N, C, P, y_ctot, y_coff,
                           mode, act, slope);
    else
    hipLaunchKernelGGL(cbam_apply_kernel, dim3(blocks), dim3(256), 0, s, u, res, sv.cg, sv.sg, y, N, C, P, y_ctot,
                       y_coff, mode, act, slope);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

extern "C" int mgvae_cbam_bwd(const float* u, const float* y, const float* dy, const float* w1, const float* w2,
                              const float* wsp, const float* save, float* du, float* dres, float* dw1, float* dw2,
                              float* dwsp, float* scratch, int N, int C, int H, int W, int y_ctot, int y_coff,
                              int mode, int act, float slope, int parts, void* stream) {
    int rc = cbam_check(N, C, H, W, y_ctot, y_coff, mode, act);
    if (rc) return rc;
    if ((parts & 3) == 0 || parts > 7) return MGVAE_EINVAL;
    const bool chan = parts & 1, spat = parts & 2, defer = parts & 4;
    if (!u || !y || !dy || !save || !du || !scratch || (mode == 2 && !dres) || (chan && (!w1 || !w2)) || (spat && !wsp))
        return MGVAE_EINVAL;
    hipStream_t s = as_stream(stream);
    const int P = H * W, NC = N * C, Cr = C / 16;
    CbamSave sv = carve(const_cast<float*>(save), N, C, P);
    float* dt = scratch;                         // [N,P]
    float* ds_in = scratch + (size_t)N * P;      // [N,2,P]
    float* dcg = scratch + (size_t)3 * N * P;    // [N,C]
    float* davg = dcg + NC;
    float* dmaxp = davg + NC;
    float* dh = dmaxp + NC;                      // [N,2,Cr]
    if (!spat) {
        cbam_fill(ds_in, 0.f, (long)2 * N * P, s);          // no spatial gate: its pooled inputs get no gradient
        cbam_fill(reinterpret_cast<float*>(sv.amax_c), 0.f, (long)N * P, s);
    } else {
    switch (pick_pxb((long)N * P)) {
        case 64: hipLaunchKernelGGL(cbam_bwd_spatial_kernel<64>, dim3(cdiv((long)N * P, 64)), dim3(256), 0, s, u, y, dy, sv.cg, sv.sg, dt, N, C, P, y_ctot, y_coff, mode, act, slope); break;
        case 32: hipLaunchKernelGGL(cbam_bwd_spatial_kernel<32>, dim3(cdiv((long)N * P, 32)), dim3(256), 0, s, u, y, dy, sv.cg, sv.sg, dt, N, C, P, y_ctot, y_coff, mode, act, slope); break;
        case 16: hipLaunchKernelGGL(cbam_bwd_spatial_kernel<16>, dim3(cdiv((long)N * P, 16)), dim3(256), 0, s, u, y, dy, sv.cg, sv.sg, dt, N, C, P, y_ctot, y_coff, mode, act, slope); break;
        default: hipLaunchKernelGGL(cbam_bwd_spatial_kernel<8>, dim3(cdiv((long)N * P, 8)), dim3(256), 0, s, u, y, dy, sv.cg, sv.sg, dt, N, C, P, y_ctot, y_coff, mode, act, slope); break;
    }
    hipLaunchKernelGGL(cbam_bwd_sgate_kernel, dim3(cdiv((long)N * P, 256)), dim3(256), 0, s, dt, sv.s_in, wsp, ds_in,
                       dwsp, N, H, W);
    }
    if (P <= 24)
        hipLaunchKernelGGL((cbam_bwd_channel_mini_kernel<8, 3>), dim3(cdiv((long)NC * 8, 256)), dim3(256), 0, s, u, y, dy, sv.cg,
                           sv.sg, ds_in, sv.amax_c, du, dres, dcg, N, C, P, y_ctot, y_coff, mode, act, slope);
    else if (P <= 96)
        hipLaunchKernelGGL((cbam_bwd_channel_mini_kernel<16, 6>), dim3(cdiv((long)NC * 16, 256)), dim3(256), 0, s, u, y, dy, sv.cg,
                           sv.sg, ds_in, sv.amax_c, du, dres, dcg, N, C, P, y_ctot, y_coff, mode, act, slope);
    else if ((P & 3) == 0)
        hipLaunchKernelGGL(cbam_bwd_channel_vec_kernel, dim3(cdiv(NC, 4)), dim3(256), 0, s, u, y, dy, sv.cg, sv.sg, ds_in,
                           sv.amax_c, du, dres, dcg, N, C, P, y_ctot, y_coff, mode, act, slope);
    else
        hipLaunchKernelGGL(cbam_bwd_channel_kernel, dim3(cdiv(NC, 4)), dim3(256), 0, s, u, y, dy, sv.cg, sv.sg, ds_in,
                           sv.amax_c, du, dres, dcg, N, C, P, y_ctot, y_coff, mode, act, slope);
    if (!chan) { MGVAE_CHECK_LAUNCH(); return MGVAE_OK; }   // no channel gate: du is complete
    hipLaunchKernelGGL(cbam_bwd_dh_kernel, dim3(cdiv((long)N * Cr, 4)), dim3(256), 0, s, dcg, sv.hid, w2, dh, N, C);
    hipLaunchKernelGGL(cbam_bwd_dpool_kernel, dim3(cdiv(NC, 256)), dim3(256), 0, s, dh, w1, davg, dmaxp, N, C);
    if (dw1 || dw2) {
        long chunks = 32768 / ((long)C * Cr);            // aim for >= 32k threads in flight
        chunks = chunks < 1 ? 1 : (chunks > 16 ? 16 : chunks);
        const int per = (int)cdiv(N, chunks) < 4 ? 4 : (int)cdiv(N, chunks);
        hipLaunchKernelGGL(cbam_bwd_mlp_wgrad_kernel, dim3(cdiv((long)C * Cr, 256), cdiv(N, per)), dim3(256), 0, s, dcg, sv.hid,
                           dh, sv.avg, sv.mx, dw1, dw2, N, C, per);
    }
    if (defer) { MGVAE_CHECK_LAUNCH(); return MGVAE_OK; }   // the caller's InstanceNorm backward adds the tail
    const long total = (long)NC * P;
    const int blocks = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
    hipLaunchKernelGGL(cbam_bwd_finish_kernel, dim3(blocks), dim3(256), 0, s, du, davg, dmaxp, sv.amax_hw, total, P);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

#include "norm_cbam_nhwc.inc"
