// InstanceNorm2d (affine, biased variance, no running statistics) with a fused
// activation, its backward, whole-map average pooling and the bias-gradient reduction.
// All are HBM-bound: one 64-lane wave owns one (sample, channel) plane, reads are
// lane-contiguous, reductions are wavefront shuffles (no LDS round trip).
#include "mgvae_common.h"

// optional fused CBAM channel pooling of the normalised output: (avg, max, first argmax) per plane
__device__ __forceinline__ void pool_finish(float ps, float pm, int pi, int P, int lane, float* avg, float* mx, int* idx) {
    ps = wave_sum(ps);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(pm, o, 64);
        const int oi = __shfl_xor(pi, o, 64);
        if (om > pm || (om == pm && oi < pi)) { pm = om; pi = oi; }
    }
    if (lane == 0) { *avg = ps / (float)P; *mx = pm; *idx = pi; }
}

// ------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void instance_norm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ y, float* __restrict__ stats, int NC, int C, int P, int y_ctot, int y_coff,
    float eps, int act, float slope, float* __restrict__ pavg, float* __restrict__ pmax, int* __restrict__ pidx) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nc = blockIdx.x * 4 + wave;
    if (nc >= NC) return;
    const int n = nc / C, c = nc - n * C;
    const float* xp = x + (size_t)nc * P;
    float s = 0.f;
    for (int i = lane; i < P; i += 64) s += xp[i];
    const float mean = wave_sum(s) / (float)P;
    float v = 0.f;
    for (int i = lane; i < P; i += 64) { const float d = xp[i] - mean; v += d * d; }
    const float var = wave_sum(v) / (float)P;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) { stats[2 * nc] = mean; stats[2 * nc + 1] = rstd; }
    const float g = gamma[c], b = beta[c];
    float* yp = y + ((size_t)n * y_ctot + y_coff + c) * P;
    float ps = 0.f, pm = -INFINITY; int pi = 0x7fffffff;
    for (int i = lane; i < P; i += 64) {
        const float o = apply_act((xp[i] - mean) * rstd * g + b, act, slope);
        yp[i] = o;
        ps += o;
        if (o > pm) { pm = o; pi = i; }
    }
    if (pavg) pool_finish(ps, pm, pi, P, lane, pavg + nc, pmax + nc, pidx + nc);
}

// ------------------------------------------------------------------------ backward
__global__ __launch_bounds__(256) void instance_norm_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ stats, const float* __restrict__ dy, float* __restrict__ dx,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int NC, int C, int P, int dy_ctot, int dy_coff,
    int act, float slope, const float* __restrict__ addc, const float* __restrict__ addp, const int* __restrict__ addi) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nc = blockIdx.x * 4 + wave;
    if (nc >= NC) return;
    const int n = nc / C, c = nc - n * C;
    const float* xp = x + (size_t)nc * P;
    const float* dyp = dy + ((size_t)n * dy_ctot + dy_coff + c) * P;
    float* dxp = dx + (size_t)nc * P;
    const float mean = stats[2 * nc], rstd = stats[2 * nc + 1];
    const float g = gamma[c], b = beta[c];
    // deferred tail of the CBAM backward: + davg/P everywhere, + dmax at the arg-max pixel
    const float ac = addc ? addc[nc] / (float)P : 0.f, ap = addc ? addp[nc] : 0.f;
    const int ai = addc ? addi[nc] : -1;
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < P; i += 64) {
        const float xh = (xp[i] - mean) * rstd;
        float gr = dyp[i] + ac + (i == ai ? ap : 0.f);
        if (act != MGVAE_ACT_NONE) {
            const float u = xh * g + b;
            gr *= (u > 0.f) ? 1.f : (act == MGVAE_ACT_LEAKY ? slope : 0.f);
        }
        s1 += gr; s2 += gr * xh;
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) {
        if (dgamma) atomicAdd(&dgamma[c], s2);
        if (dbeta) atomicAdd(&dbeta[c], s1);
    }
    const float m1 = s1 / (float)P, m2 = s2 / (float)P, k = g * rstd;
    for (int i = lane; i < P; i += 64) {
        const float xh = (xp[i] - mean) * rstd;
        float gr = dyp[i] + ac + (i == ai ? ap : 0.f);
        if (act != MGVAE_ACT_NONE) {
            const float u = xh * g + b;
            gr *= (u > 0.f) ? 1.f : (act == MGVAE_ACT_LEAKY ? slope : 0.f);
        }
        dxp[i] = k * (gr - m1 - xh * m2);
    }
}

// ------------------------------------------------------------------------ small planes
// P <= G*E (the 3x2 ... 12x8 maps of the deep encoder / first decoder blocks): a whole wave per plane would leave
// most lanes idle, so G = 8 or 16 lanes share a plane (8 or 4 planes per wave), the plane lives in E registers per
// lane and the reductions are xor-shuffles inside the lane group.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
}

template <int G, int E>
__global__ __launch_bounds__(256) void instance_norm_fwd_mini_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ y, float* __restrict__ stats, int NC, int C, int P, int y_ctot, int y_coff,
    float eps, int act, float slope, float* __restrict__ pavg, float* __restrict__ pmax, int* __restrict__ pidx) {
    const int l = threadIdx.x % G;
    const int plane = (blockIdx.x * 256 + threadIdx.x) / G;
    const bool live = plane < NC;
    const int nc = live ? plane : NC - 1;
    const int n = nc / C, c = nc - n * C;
    const float* xp = x + (size_t)nc * P;
    float v[E];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) { const int i = l + G * k; v[k] = i < P ? xp[i] : 0.f; s += v[k]; }
    const float mean = group_sum<G>(s) / (float)P;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) { const float d = (l + G * k) < P ? v[k] - mean : 0.f; q += d * d; }
    const float rstd = 1.0f / sqrtf(group_sum<G>(q) / (float)P + eps);
    if (l == 0 && live) { stats[2 * nc] = mean; stats[2 * nc + 1] = rstd; }
    const float g = gamma[c], b = beta[c];
    float* yp = y + ((size_t)n * y_ctot + y_coff + c) * P;
    float ps = 0.f, pm = -INFINITY; int pi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int i = l + G * k;
        if (i < P) {
            const float o = apply_act((v[k] - mean) * rstd * g + b, act, slope);
            if (live) yp[i] = o;
            ps += o;
            if (o > pm) { pm = o; pi = i; }
        }
    }
    if (pavg) {
        ps = group_sum<G>(ps);
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) {
            const float om = __shfl_xor(pm, o, G);
            const int oi = __shfl_xor(pi, o, G);
            if (om > pm || (om == pm && oi < pi)) { pm = om; pi = oi; }
        }
        if (l == 0 && live) { pavg[nc] = ps / (float)P; pmax[nc] = pm; pidx[nc] = pi; }
    }
}

template <int G, int E>
__global__ __launch_bounds__(256) void instance_norm_bwd_mini_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ stats, const float* __restrict__ dy, float* __restrict__ dx,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int NC, int C, int P, int dy_ctot, int dy_coff,
    int act, float slope, const float* __restrict__ addc, const float* __restrict__ addp, const int* __restrict__ addi) {
    const int l = threadIdx.x % G;
    const int plane = (blockIdx.x * 256 + threadIdx.x) / G;
    const bool live = plane < NC;
    const int nc = live ? plane : NC - 1;
    const int n = nc / C, c = nc - n * C;
    const float* xp = x + (size_t)nc * P;
    const float* dyp = dy + ((size_t)n * dy_ctot + dy_coff + c) * P;
    const float mean = stats[2 * nc], rstd = stats[2 * nc + 1];
    const float g = gamma[c], b = beta[c];
    const float ac = addc ? addc[nc] / (float)P : 0.f, ap = addc ? addp[nc] : 0.f;
    const int ai = addc ? addi[nc] : -1;
    float xh[E], gr[E];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int i = l + G * k;
        const bool in = i < P;
        xh[k] = in ? (xp[i] - mean) * rstd : 0.f;
        float t = in ? dyp[i] + ac + (i == ai ? ap : 0.f) : 0.f;
        if (act != MGVAE_ACT_NONE) {
            const float u = xh[k] * g + b;
            t *= (u > 0.f) ? 1.f : (act == MGVAE_ACT_LEAKY ? slope : 0.f);
        }
        gr[k] = t;
        s1 += t; s2 += t * xh[k];
    }
    s1 = group_sum<G>(s1); s2 = group_sum<G>(s2);
    if (l == 0 && live) {
        if (dgamma) atomicAdd(&dgamma[c], s2);
        if (dbeta) atomicAdd(&dbeta[c], s1);
    }
    const float m1 = s1 / (float)P, m2 = s2 / (float)P, kk = g * rstd;
    float* dxp = dx + (size_t)nc * P;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int i = l + G * k;
        if (i < P && live) dxp[i] = kk * (gr[k] - m1 - xh[k] * m2);
    }
}

// ------------------------------------------------------------------------ register-resident variants
// WPP waves per plane: 1 (a wave owns a plane) or 4 (the whole workgroup owns one big plane: a quarter of the
// registers per lane, so four times the waves in flight -- 184 VGPRs held the P = 5760 planes to 2 waves per SIMD)
template <int WPP>
__device__ __forceinline__ float plane_sum(float v, float* sh, int wave, int lane) {
    v = wave_sum(v);
    if constexpr (WPP == 1) return v;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    const float r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;
}

// P % 4 == 0 and P <= 256*NV: the whole plane lives in NV float4 registers per lane, so x (and dy)
// are read from HBM exactly once, as 16-byte lane-contiguous loads.
template <int NV, int WPP>
__global__ __launch_bounds__(256) void instance_norm_fwd_vec_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ y, float* __restrict__ stats, int NC, int C, int P, int y_ctot, int y_coff,
    float eps, int act, float slope, float* __restrict__ pavg, float* __restrict__ pmax, int* __restrict__ pidx) {
    __shared__ float sh[4];
    __shared__ int shi[4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nc = WPP == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    if (nc >= NC) return;
    const int L = WPP == 1 ? lane : (int)threadIdx.x;      // this thread's float4 slot within the plane
    constexpr int STR = 64 * WPP;
    const int n = nc / C, c = nc - n * C;
    const int P4 = P >> 2;
    const float4* xp = reinterpret_cast<const float4*>(x + (size_t)nc * P);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = L + STR * k;
        v[k] = i < P4 ? xp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mean = plane_sum<WPP>(s, sh, wave, lane) / (float)P;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (L + STR * k < P4) {
            const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, d = v[k].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float var = plane_sum<WPP>(q, sh, wave, lane) / (float)P;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (L == 0) { stats[2 * nc] = mean; stats[2 * nc + 1] = rstd; }
    const float g = gamma[c], b = beta[c];
    float4* yp = reinterpret_cast<float4*>(y + ((size_t)n * y_ctot + y_coff + c) * P);
    float ps = 0.f, pm = -INFINITY; int pi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = L + STR * k;
        if (i < P4) {
            float4 o;
            o.x = apply_act((v[k].x - mean) * rstd * g + b, act, slope);
            o.y = apply_act((v[k].y - mean) * rstd * g + b, act, slope);
            o.z = apply_act((v[k].z - mean) * rstd * g + b, act, slope);
            o.w = apply_act((v[k].w - mean) * rstd * g + b, act, slope);
            yp[i] = o;
            ps += (o.x + o.y) + (o.z + o.w);
            if (o.x > pm) { pm = o.x; pi = 4 * i; }
            if (o.y > pm) { pm = o.y; pi = 4 * i + 1; }
            if (o.z > pm) { pm = o.z; pi = 4 * i + 2; }
            if (o.w > pm) { pm = o.w; pi = 4 * i + 3; }
        }
    }
    if (pavg) {
        if constexpr (WPP == 1) {
            pool_finish(ps, pm, pi, P, lane, pavg + nc, pmax + nc, pidx + nc);
        } else {                                   // wave-level (sum, max, first argmax), then across the four waves
            ps = wave_sum(ps);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float om = __shfl_xor(pm, o, 64);
                const int oi = __shfl_xor(pi, o, 64);
                if (om > pm || (om == pm && oi < pi)) { pm = om; pi = oi; }
            }
            __shared__ float shs[4], shm[4];
            if (lane == 0) { shs[wave] = ps; shm[wave] = pm; shi[wave] = pi; }
            __syncthreads();
            if (threadIdx.x == 0) {
                float ts = 0.f, tm = -INFINITY; int ti = 0x7fffffff;
                for (int w = 0; w < 4; ++w) {
                    ts += shs[w];
                    if (shm[w] > tm || (shm[w] == tm && shi[w] < ti)) { tm = shm[w]; ti = shi[w]; }
                }
                pavg[nc] = ts / (float)P; pmax[nc] = tm; pidx[nc] = ti;
            }
        }
    }
}

template <int NV, int WPP>
__global__ __launch_bounds__(256) void instance_norm_bwd_vec_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ stats, const float* __restrict__ dy, float* __restrict__ dx,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int NC, int C, int P, int dy_ctot, int dy_coff,
    int act, float slope, const float* __restrict__ addc, const float* __restrict__ addp, const int* __restrict__ addi) {
    __shared__ float sh[4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nc = WPP == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    if (nc >= NC) return;
    const int L = WPP == 1 ? lane : (int)threadIdx.x;
    constexpr int STR = 64 * WPP;
    const int n = nc / C, c = nc - n * C;
    const int P4 = P >> 2;
    const float ac = addc ? addc[nc] / (float)P : 0.f, ap = addc ? addp[nc] : 0.f;
    const int ai = addc ? addi[nc] : -1;
    const float4* xp = reinterpret_cast<const float4*>(x + (size_t)nc * P);
    const float4* dyp = reinterpret_cast<const float4*>(dy + ((size_t)n * dy_ctot + dy_coff + c) * P);
    float4* dxp = reinterpret_cast<float4*>(dx + (size_t)nc * P);
    const float mean = stats[2 * nc], rstd = stats[2 * nc + 1];
    const float g = gamma[c], b = beta[c];
    const float neg = act == MGVAE_ACT_LEAKY ? slope : 0.f;
    float4 xh[NV], gr[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = L + STR * k;
        if (i < P4) {
            const float4 xv = xp[i];
            float4 gv = dyp[i];
            gv.x += ac; gv.y += ac; gv.z += ac; gv.w += ac;
            if ((ai >> 2) == i) { const int q = ai & 3; if (q == 0) gv.x += ap; else if (q == 1) gv.y += ap; else if (q == 2) gv.z += ap; else gv.w += ap; }
            float4 h;
            h.x = (xv.x - mean) * rstd; h.y = (xv.y - mean) * rstd; h.z = (xv.z - mean) * rstd; h.w = (xv.w - mean) * rstd;
            if (act != MGVAE_ACT_NONE) {
                gv.x *= (h.x * g + b > 0.f) ? 1.f : neg; gv.y *= (h.y * g + b > 0.f) ? 1.f : neg;
                gv.z *= (h.z * g + b > 0.f) ? 1.f : neg; gv.w *= (h.w * g + b > 0.f) ? 1.f : neg;
            }
            xh[k] = h; gr[k] = gv;
            s1 += (gv.x + gv.y) + (gv.z + gv.w);
            s2 += (gv.x * h.x + gv.y * h.y) + (gv.z * h.z + gv.w * h.w);
        } else {
            xh[k] = make_float4(0.f, 0.f, 0.f, 0.f); gr[k] = xh[k];
        }
    }
    s1 = plane_sum<WPP>(s1, sh, wave, lane); s2 = plane_sum<WPP>(s2, sh, wave, lane);
    if (L == 0) {
        if (dgamma) atomicAdd(&dgamma[c], s2);
        if (dbeta) atomicAdd(&dbeta[c], s1);
    }
    const float m1 = s1 / (float)P, m2 = s2 / (float)P, kq = g * rstd;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = L + STR * k;
        if (i < P4) {
            float4 o;
            o.x = kq * (gr[k].x - m1 - xh[k].x * m2); o.y = kq * (gr[k].y - m1 - xh[k].y * m2);
            o.z = kq * (gr[k].z - m1 - xh[k].z * m2); o.w = kq * (gr[k].w - m1 - xh[k].w * m2);
            dxp[i] = o;
        }
    }
}

static inline int pick_nv(int P) {          // float4 slots per lane, or 0 for the generic kernel
    if (P & 3) return 0;
    const int need = (P / 4 + 63) / 64;
    return need <= 1 ? 1 : need <= 2 ? 2 : need <= 6 ? 6 : need <= 24 ? 24 : 0;   // 24: one workgroup per plane, 6 slots per lane
}

// ------------------------------------------------------------------------ row mean
__global__ __launch_bounds__(256) void rowmean_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          int rows, int L) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* xp = x + (size_t)r * L;
    float s = 0.f;
    for (int i = 0; i < L; ++i) s += xp[i];
    out[r] = s / (float)L;
}

__global__ __launch_bounds__(256) void rowmean_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx,
                                                          long total, int L) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    dx[i] = dout[i / L] / (float)L;
}

// ------------------------------------------------------------------------ bias gradient
// db[c] += sum over (n, p) of t[n, coff + c, p]; blockIdx.x = channel, blockIdx.y = slice of the samples
// (enough workgroups to fill the chip even for 32..64 channels), one atomic per workgroup.
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ t, int N, int C, int P,
                                                          int ctot, int coff, float* __restrict__ db) {
    __shared__ float part[4];
    const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    for (int n = blockIdx.y * 4 + wave; n < N; n += 4 * gridDim.y) {   // one wave walks whole (n, c) planes
        const float* tp = t + ((size_t)n * ctot + coff + c) * P;
        if ((P & 3) == 0) {
            const float4* t4 = reinterpret_cast<const float4*>(tp);
            for (int i = lane; i < (P >> 2); i += 64) { const float4 v = t4[i]; s += (v.x + v.y) + (v.z + v.w); }
        } else {
            for (int i = lane; i < P; i += 64) s += tp[i];
        }
    }
    s = wave_sum(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&db[c], part[0] + part[1] + part[2] + part[3]);
}

// ------------------------------------------------------------------------ BatchNorm2d
// (BarDiscriminator / Refiner only: C <= 64 channels, launch-bound).  One workgroup per channel
// reduces over (N, P); training mode also updates the running statistics like torch does
// (momentum, unbiased variance) and saves (mean, rstd) for backward.
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void batch_norm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ rmean,
                                                             float* __restrict__ rvar, float* __restrict__ y,
                                                             float* __restrict__ stats, int N, int C, int P, int training,
                                                             float momentum, float eps, int act, float slope) {
    __shared__ float sh[4];
    const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float cnt = (float)N * (float)P;
    float mean, var;
    if (training) {
        float s = 0.f;
        for (int n = wave; n < N; n += 4) {
            const float* xp = x + ((size_t)n * C + c) * P;
            for (int i = lane; i < P; i += 64) s += xp[i];
        }
        mean = block_sum(s, sh) / cnt;
        float v = 0.f;
        for (int n = wave; n < N; n += 4) {
            const float* xp = x + ((size_t)n * C + c) * P;
            for (int i = lane; i < P; i += 64) { const float d = xp[i] - mean; v += d * d; }
        }
        var = block_sum(v, sh) / cnt;
        if (threadIdx.x == 0) {
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (cnt > 1.f ? var * cnt / (cnt - 1.f) : var);
        }
    } else {
        mean = rmean[c]; var = rvar[c];
    }
    const float rstd = 1.0f / sqrtf(var + eps);
    if (threadIdx.x == 0) { stats[2 * c] = mean; stats[2 * c + 1] = rstd; }
    const float g = gamma[c], b = beta[c];
    for (int n = wave; n < N; n += 4) {
        const float* xp = x + ((size_t)n * C + c) * P;
        float* yp = y + ((size_t)n * C + c) * P;
        for (int i = lane; i < P; i += 64) yp[i] = apply_act((xp[i] - mean) * rstd * g + b, act, slope);
    }
}

__global__ __launch_bounds__(256) void batch_norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ stats,
                                                             const float* __restrict__ dy, float* __restrict__ dx,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int N,
                                                             int C, int P, int training, int act, float slope) {
    __shared__ float sh[4];
    const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float cnt = (float)N * (float)P;
    const float mean = stats[2 * c], rstd = stats[2 * c + 1], g = gamma[c], b = beta[c];
    float s1 = 0.f, s2 = 0.f;
    for (int n = wave; n < N; n += 4) {
        const float* xp = x + ((size_t)n * C + c) * P;
        const float* dp = dy + ((size_t)n * C + c) * P;
        for (int i = lane; i < P; i += 64) {
            const float xh = (xp[i] - mean) * rstd;
            float gr = dp[i];
            if (act != MGVAE_ACT_NONE) gr *= (xh * g + b > 0.f) ? 1.f : (act == MGVAE_ACT_LEAKY ? slope : 0.f);
            s1 += gr; s2 += gr * xh;
        }
    }
    s1 = block_sum(s1, sh); s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) {
        if (dgamma) dgamma[c] += s2;
        if (dbeta) dbeta[c] += s1;
    }
    const float m1 = training ? s1 / cnt : 0.f, m2 = training ? s2 / cnt : 0.f, k = g * rstd;
    for (int n = wave; n < N; n += 4) {
        const float* xp = x + ((size_t)n * C + c) * P;
        const float* dp = dy + ((size_t)n * C + c) * P;
        float* dxp = dx + ((size_t)n * C + c) * P;
        for (int i = lane; i < P; i += 64) {
            const float xh = (xp[i] - mean) * rstd;
            float gr = dp[i];
            if (act != MGVAE_ACT_NONE) gr *= (xh * g + b > 0.f) ? 1.f : (act == MGVAE_ACT_LEAKY ? slope : 0.f);
            dxp[i] = k * (gr - m1 - xh * m2);
        }
    }
}

extern "C" int mgvae_batch_norm_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float* y, float* stats, int N, int C, int P, int training,
                                    float momentum, float eps, int act, float slope, void* stream) {
    if (!x || !gamma || !beta || !running_mean || !running_var || !y || !stats || N <= 0 || C <= 0 || P <= 0) return MGVAE_EINVAL;
    if (act == MGVAE_ACT_SIGMOID) return MGVAE_EINVAL;
    hipLaunchKernelGGL(batch_norm_fwd_kernel, dim3(C), dim3(256), 0, as_stream(stream), x, gamma, beta, running_mean,
                       running_var, y, stats, N, C, P, training, momentum, eps, act, slope);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

extern "C" int mgvae_batch_norm_bwd(const float* x, const float* gamma, const float* beta, const float* stats,
                                    const float* dy, float* dx, float* dgamma, float* dbeta, int N, int C, int P,
                                    int training, int act, float slope, void* stream) {
    if (!x || !gamma || !beta || !stats || !dy || !dx || N <= 0 || C <= 0 || P <= 0) return MGVAE_EINVAL;
    if (act == MGVAE_ACT_SIGMOID) return MGVAE_EINVAL;
    hipLaunchKernelGGL(batch_norm_bwd_kernel, dim3(C), dim3(256), 0, as_stream(stream), x, gamma, beta, stats, dy, dx,
                       dgamma, dbeta, N, C, P, training, act, slope);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

extern "C" int mgvae_instance_norm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                                       float* stats, int N, int C, int P, int y_ctot, int y_coff, float eps,
                                       int act, float slope, float* pool_avg, float* pool_max, int* pool_idx,
                                       void* stream) {
    if ((pool_avg != nullptr) != (pool_max != nullptr) || (pool_avg != nullptr) != (pool_idx != nullptr)) return MGVAE_EINVAL;
    if (!x || !gamma || !beta || !y || !stats || N <= 0 || C <= 0 || P <= 0) return MGVAE_EINVAL;
    if (y_coff < 0 || y_coff + C > y_ctot || act == MGVAE_ACT_SIGMOID) return MGVAE_EINVAL;
    const int NC = N * C;
    const bool al = ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
    void* prof_tok = nullptr;      // measurement hook (bench.py roofline.hbm): algorithmic bytes = read x + write y
    mgvae_prof_record_begin(MGVAE_PROF_INORM_FWD, 0, 2.0 * 4.0 * (double)NC * P, stream, &prof_tok);
#define MGVAE_INF(NV) hipLaunchKernelGGL((instance_norm_fwd_vec_kernel<NV, 1>), dim3(cdiv(NC, 4)), dim3(256), 0, as_stream(stream), x, \
                                         gamma, beta, y, stats, NC, C, P, y_ctot, y_coff, eps, act, slope, pool_avg, pool_max, pool_idx)
    if (P <= 24) {
        hipLaunchKernelGGL((instance_norm_fwd_mini_kernel<8, 3>), dim3(cdiv((long)NC * 8, 256)), dim3(256), 0, as_stream(stream), x,
                           gamma, beta, y, stats, NC, C, P, y_ctot, y_coff, eps, act, slope, pool_avg, pool_max, pool_idx);
    } else if (P <= 96) {
        hipLaunchKernelGGL((instance_norm_fwd_mini_kernel<16, 6>), dim3(cdiv((long)NC * 16, 256)), dim3(256), 0, as_stream(stream), x,
                           gamma, beta, y, stats, NC, C, P, y_ctot, y_coff, eps, act, slope, pool_avg, pool_max, pool_idx);
    } else
    switch (al ? pick_nv(P) : 0) {
        case 1: MGVAE_INF(1); break;
        case 2: MGVAE_INF(2); break;
        case 6: MGVAE_INF(6); break;
        case 24:
            hipLaunchKernelGGL((instance_norm_fwd_vec_kernel<6, 4>), dim3(NC), dim3(256), 0, as_stream(stream), x, gamma, beta, y, stats,
                               NC, C, P, y_ctot, y_coff, eps, act, slope, pool_avg, pool_max, pool_idx);
            break;
        default:
            hipLaunchKernelGGL(instance_norm_fwd_kernel, dim3(cdiv(NC, 4)), dim3(256), 0, as_stream(stream), x, gamma,
                               beta, y, stats, NC, C, P, y_ctot, y_coff, eps, act, slope, pool_avg, pool_max, pool_idx);
    }
#undef MGVAE_INF
    mgvae_prof_record_end(prof_tok, stream);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

extern "C" int mgvae_instance_norm_bwd(const float* x, const float* gamma, const float* beta, const float* stats,
                                       const float* dy, float* dx, float* dgamma, float* dbeta, int N, int C,
                                       int P, int dy_ctot, int dy_coff, int act, float slope, const float* add_const,
                                       const float* add_point, const int* add_index, void* stream) {
    if ((add_const != nullptr) != (add_point != nullptr) || (add_const != nullptr) != (add_index != nullptr)) return MGVAE_EINVAL;
    if (!x || !gamma || !beta || !stats || !dy || !dx || N <= 0 || C <= 0 || P <= 0) return MGVAE_EINVAL;
    if (dy_coff < 0 || dy_coff + C > dy_ctot || act == MGVAE_ACT_SIGMOID) return MGVAE_EINVAL;
    const int NC = N * C;
    const bool al = ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0;
    void* prof_tok = nullptr;      // algorithmic bytes = read x + read dy + write dx
    mgvae_prof_record_begin(MGVAE_PROF_INORM_BWD, 0, 3.0 * 4.0 * (double)NC * P, stream, &prof_tok);
#define MGVAE_INB(NV) hipLaunchKernelGGL((instance_norm_bwd_vec_kernel<NV, 1>), dim3(cdiv(NC, 4)), dim3(256), 0, as_stream(stream), x, \
                                         gamma, beta, stats, dy, dx, dgamma, dbeta, NC, C, P, dy_ctot, dy_coff, act, slope, add_const, \
                                         add_point, add_index)
    if (P <= 24) {
        hipLaunchKernelGGL((instance_norm_bwd_mini_kernel<8, 3>), dim3(cdiv((long)NC * 8, 256)), dim3(256), 0, as_stream(stream), x,
                           gamma, beta, stats, dy, dx, dgamma, dbeta, NC, C, P, dy_ctot, dy_coff, act, slope, add_const,
                           add_point, add_index);
    } else if (P <= 96) {
        hipLaunchKernelGGL((instance_norm_bwd_mini_kernel<16, 6>), dim3(cdiv((long)NC * 16, 256)), dim3(256), 0, as_stream(stream), x,
                           gamma, beta, stats, dy, dx, dgamma, dbeta, NC, C, P, dy_ctot, dy_coff, act, slope, add_const,
                           add_point, add_index);
    } else
    switch (al ? pick_nv(P) : 0) {
        case 1: MGVAE_INB(1); break;
        case 2: MGVAE_INB(2); break;
        case 6: MGVAE_INB(6); break;
        case 24:
            hipLaunchKernelGGL((instance_norm_bwd_vec_kernel<6, 4>), dim3(NC), dim3(256), 0, as_stream(stream), x, gamma, beta, stats, dy,
                               dx, dgamma, dbeta, NC, C, P, dy_ctot, dy_coff, act, slope, add_const, add_point, add_index);
            break;
        default:
            hipLaunchKernelGGL(instance_norm_bwd_kernel, dim3(cdiv(NC, 4)), dim3(256), 0, as_stream(stream), x, gamma,
                               beta, stats, dy, dx, dgamma, dbeta, NC, C, P, dy_ctot, dy_coff, act, slope, add_const,
                               add_point, add_index);
    }
#undef MGVAE_INB
    mgvae_prof_record_end(prof_tok, stream);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

extern "C" int mgvae_rowmean_fwd(const float* x, float* out, int rows, int L, void* stream) {
    if (!x || !out || rows <= 0 || L <= 0) return MGVAE_EINVAL;
    hipLaunchKernelGGL(rowmean_fwd_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, as_stream(stream), x, out, rows, L);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

extern "C" int mgvae_rowmean_bwd(const float* dout, float* dx, int rows, int L, void* stream) {
    if (!dout || !dx || rows <= 0 || L <= 0) return MGVAE_EINVAL;
    const long total = (long)rows * L;
    hipLaunchKernelGGL(rowmean_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), dout, dx, total, L);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

extern "C" int mgvae_channel_sum_accum(const float* t, int N, int C, int P, int ctot, int coff, float* db,
                                       void* stream) {
    if (!t || !db || N <= 0 || C <= 0 || P <= 0 || coff < 0 || coff + C > ctot) return MGVAE_EINVAL;
    int ny = cdiv(2048, C);                       // >= 2048 workgroups when the batch allows it
    ny = ny > cdiv(N, 4) ? cdiv(N, 4) : ny;
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C, ny), dim3(256), 0, as_stream(stream), t, N, C, P, ctot, coff, db);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}
