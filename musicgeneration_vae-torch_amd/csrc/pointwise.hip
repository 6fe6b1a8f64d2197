// Small HBM-bound ops of the training step: activation backward, channel-slice copies,
// embedding gather / scatter-add, Philox dropout + Gaussian prior noise, the BCE losses
// (with the reference's label smoothing and missed-note count), the reparameterisation
// sampler + KL term, and the fused flat Adam update.
#include "mgvae_common.h"

static inline int grid_for(size_t n, int cap = 8192) {
    size_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b < (size_t)cap ? b : (size_t)cap);
}

// ------------------------------------------------------------------ activation backward
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                      float* __restrict__ dx, int N, int C, int P, int y_ctot, int y_coff,
                                                      int dy_ctot, int dy_coff, int dx_ctot, int dx_coff, int act,
                                                      float slope) {
    const long total = (long)N * C * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int nc = (int)(i / P), pp = (int)(i - (long)nc * P);
        const int n = nc / C, c = nc - n * C;
        const float yy = y[((size_t)n * y_ctot + y_coff + c) * P + pp];
        const float g = dy[((size_t)n * dy_ctot + dy_coff + c) * P + pp];
        dx[((size_t)n * dx_ctot + dx_coff + c) * P + pp] = g * act_grad_from_out(yy, act, slope);
    }
}

extern "C" int mgvae_act_bwd(const float* y, const float* dy, float* dx, int N, int C, int P, int y_ctot, int y_coff,
                             int dy_ctot, int dy_coff, int dx_ctot, int dx_coff, int act, float slope, void* stream) {
    if (!y || !dy || !dx || N <= 0 || C <= 0 || P <= 0) return MGVAE_EINVAL;
    if (y_coff < 0 || y_coff + C > y_ctot || dy_coff < 0 || dy_coff + C > dy_ctot || dx_coff < 0 || dx_coff + C > dx_ctot)
        return MGVAE_EINVAL;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for((size_t)N * C * P)), dim3(256), 0, as_stream(stream), y, dy, dx, N,
                       C, P, y_ctot, y_coff, dy_ctot, dy_coff, dx_ctot, dx_coff, act, slope);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// ------------------------------------------------------------------ copies / adds
__global__ __launch_bounds__(256) void copy2d_kernel(float* __restrict__ dst, size_t dpitch, const float* __restrict__ src,
                                                     size_t spitch, size_t width, size_t rows) {
    const size_t total = width * rows;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / width, c = i - r * width;
        dst[r * dpitch + c] = src[r * spitch + c];
    }
}

extern "C" int mgvae_copy2d(float* dst, size_t dpitch, const float* src, size_t spitch, size_t width, size_t rows,
                            void* stream) {
    if (!dst || !src || width == 0 || rows == 0 || dpitch < width || spitch < width) return MGVAE_EINVAL;
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(width * rows)), dim3(256), 0, as_stream(stream), dst, dpitch, src,
                       spitch, width, rows);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] += src[i];
}

extern "C" int mgvae_add_inplace(float* dst, const float* src, size_t n, void* stream) {
    if (!dst || !src || n == 0) return MGVAE_EINVAL;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), dst, src, n);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

__global__ __launch_bounds__(256) void add_inplace_bf16_kernel(__bf16* __restrict__ dst, const __bf16* __restrict__ src, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        uint2 a = reinterpret_cast<const uint2*>(dst)[i];
        const uint2 b = reinterpret_cast<const uint2*>(src)[i];
        auto add2 = [](unsigned x, unsigned y) {            // two packed bf16: add in fp32, round to nearest even
            const float lo = __uint_as_float(x << 16) + __uint_as_float(y << 16);
            const float hi = __uint_as_float(x & 0xffff0000u) + __uint_as_float(y & 0xffff0000u);
            auto rne = [](float f) { unsigned u = __float_as_uint(f); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
            return rne(lo) | (rne(hi) << 16);
        };
        a.x = add2(a.x, b.x); a.y = add2(a.y, b.y);
        reinterpret_cast<uint2*>(dst)[i] = a;
    }
}
__global__ __launch_bounds__(256) void add_inplace_f4_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 a = dst[i]; const float4 b = src[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        dst[i] = a;
    }
}
extern "C" int mgvae_add_inplace_typed(void* dst, const void* src, size_t n, int storage, void* stream) {
    if (!dst || !src || n == 0 || (n & 3) || (storage != 0 && storage != 1)) return MGVAE_EINVAL;
    if (storage == 1)
        hipLaunchKernelGGL(add_inplace_bf16_kernel, dim3(grid_for(n / 4, 4096)), dim3(256), 0, as_stream(stream),
                           static_cast<__bf16*>(dst), static_cast<const __bf16*>(src), n / 4);
    else
        hipLaunchKernelGGL(add_inplace_f4_kernel, dim3(grid_for(n / 4, 4096)), dim3(256), 0, as_stream(stream),
                           static_cast<float4*>(dst), static_cast<const float4*>(src), n / 4);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = a[i] * b[i];
}

extern "C" int mgvae_mul(const float* a, const float* b, float* out, size_t n, void* stream) {
    if (!a || !b || !out || n == 0) return MGVAE_EINVAL;
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), a, b, out, n);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// ------------------------------------------------------------------ grouped sums over the pitch axis
// out[r, g] = sum_{i < gsize} x[r, g*gsize + i]  (BarDiscriminator feature front-ends:
// graph/bar_discriminator.py:32-34 chord folding 60 -> 12 x 5, :86-87 on/off sum over 60 pitches)
__global__ __launch_bounds__(256) void group_sum_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                            long rows, int groups, int gsize) {
    const long total = rows * groups;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float* xp = x + i * gsize;      // (r*groups + g) * gsize == r*W + g*gsize
        float s = 0.f;
        for (int k = 0; k < gsize; ++k) s += xp[k];
        out[i] = s;
    }
}
__global__ __launch_bounds__(256) void group_sum_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx,
                                                            long total, int gsize) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) dx[i] = dout[i / gsize];
}

extern "C" int mgvae_group_sum_fwd(const float* x, float* out, size_t rows, int groups, int gsize, void* stream) {
    if (!x || !out || rows == 0 || groups <= 0 || gsize <= 0) return MGVAE_EINVAL;
    hipLaunchKernelGGL(group_sum_fwd_kernel, dim3(grid_for(rows * groups)), dim3(256), 0, as_stream(stream), x, out,
                       (long)rows, groups, gsize);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}
extern "C" int mgvae_group_sum_bwd(const float* dout, float* dx, size_t rows, int groups, int gsize, void* stream) {
    if (!dout || !dx || rows == 0 || groups <= 0 || gsize <= 0) return MGVAE_EINVAL;
    hipLaunchKernelGGL(group_sum_bwd_kernel, dim3(grid_for(rows * groups * gsize)), dim3(256), 0, as_stream(stream), dout, dx,
                       (long)(rows * groups * gsize), gsize);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// ------------------------------------------------------------------ MaxPool2d(2) / activation / axpby (Refiner, graph/refiner.py)
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int* __restrict__ idx,
                                                           long planes, int H, int W, int OH, int OW) {
    const long total = planes * OH * OW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pl = i / (OH * OW); const int r = (int)(i - pl * OH * OW);
        const int oh = r / OW, ow = r - oh * OW;
        const float* xp = x + pl * H * W + (2 * oh) * W + 2 * ow;
        float m = xp[0]; int mi = 0;
        if (xp[1] > m) { m = xp[1]; mi = 1; }
        if (xp[W] > m) { m = xp[W]; mi = 2; }
        if (xp[W + 1] > m) { m = xp[W + 1]; mi = 3; }
        y[i] = m; idx[i] = mi;
    }
}
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                           float* __restrict__ dx, long planes, int H, int W, int OH, int OW) {
    const long total = planes * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pl = i / (H * W); const int r = (int)(i - pl * H * W);
        const int h = r / W, w = r - h * W;
        const int oh = h >> 1, ow = w >> 1;
        float v = 0.f;
        if (oh < OH && ow < OW) {
            const long o = pl * OH * OW + oh * OW + ow;
            if (idx[o] == ((h & 1) << 1 | (w & 1))) v = dy[o];
        }
        dx[i] = v;
    }
}
extern "C" int mgvae_maxpool2_fwd(const float* x, float* y, int* idx, size_t planes, int H, int W, void* stream) {
    if (!x || !y || !idx || planes == 0 || H < 2 || W < 2) return MGVAE_EINVAL;
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for(planes * (H / 2) * (W / 2))), dim3(256), 0, as_stream(stream), x, y, idx,
                       (long)planes, H, W, H / 2, W / 2);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}
extern "C" int mgvae_maxpool2_bwd(const float* dy, const int* idx, float* dx, size_t planes, int H, int W, void* stream) {
    if (!dy || !dx || !idx || planes == 0 || H < 2 || W < 2) return MGVAE_EINVAL;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(planes * H * W)), dim3(256), 0, as_stream(stream), dy, idx, dx,
                       (long)planes, H, W, H / 2, W / 2);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act, float slope) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = apply_act(x[i], act, slope);
}
extern "C" int mgvae_act_fwd(const float* x, float* y, size_t n, int act, float slope, void* stream) {
    if (!x || !y || n == 0) return MGVAE_EINVAL;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, y, n, act, slope);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                                                    float a, float b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = a * x[i] + b * y[i];
}
extern "C" int mgvae_axpby(const float* x, const float* y, float* out, float a, float b, size_t n, void* stream) {
    if (!x || !y || !out || n == 0) return MGVAE_EINVAL;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, y, out, a, b, n);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// ------------------------------------------------------------------ embedding
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* __restrict__ idx, const float* __restrict__ table,
                                                            float* __restrict__ out, int B, int D, int rows, size_t pitch) {
    const long total = (long)B * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / D), d = (int)(i - (long)b * D);
        const int64_t r = idx[b];
        out[(size_t)b * pitch + d] = (r >= 0 && r < rows) ? table[(size_t)r * D + d] : __int_as_float(0x7fc00000);
    }
}

__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ idx, const float* __restrict__ dout,
                                                            float* __restrict__ dtable, int B, int D, int rows, size_t pitch) {
    const long total = (long)B * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / D), d = (int)(i - (long)b * D);
        const int64_t r = idx[b];
        if (r >= 0 && r < rows) atomicAdd(&dtable[(size_t)r * D + d], dout[(size_t)b * pitch + d]);
    }
}

extern "C" int mgvae_embedding_fwd(const int64_t* idx, const float* table, float* out, int B, int D, int rows,
                                   size_t out_pitch, void* stream) {
    if (!idx || !table || !out || B <= 0 || D <= 0 || rows <= 0 || out_pitch < (size_t)D) return MGVAE_EINVAL;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for((size_t)B * D)), dim3(256), 0, as_stream(stream), idx, table,
                       out, B, D, rows, out_pitch);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

extern "C" int mgvae_embedding_bwd(const int64_t* idx, const float* dout, float* dtable, int B, int D, int rows,
                                   size_t dout_pitch, void* stream) {
    if (!idx || !dout || !dtable || B <= 0 || D <= 0 || rows <= 0 || dout_pitch < (size_t)D) return MGVAE_EINVAL;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid_for((size_t)B * D)), dim3(256), 0, as_stream(stream), idx, dout,
                       dtable, B, D, rows, dout_pitch);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// ------------------------------------------------------------------ Philox4x32-10
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32_10(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }   // [0,1)

__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          float* __restrict__ mask, size_t n, float p, uint64_t seed,
                                                          uint64_t offset) {
    const float scale = 1.f / (1.f - p);
    const size_t groups = (n + 3) / 4;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (size_t)gridDim.x * 256) {
        const U4 r = philox4x32_10(g, offset, seed);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = g * 4 + k;
            if (i < n) {
                const float m = u01(rr[k]) >= p ? scale : 0.f;
                mask[i] = m; y[i] = x[i] * m;
            }
        }
    }
}

extern "C" int mgvae_dropout_fwd(const float* x, float* y, float* mask, size_t n, float p, uint64_t seed,
                                 uint64_t offset, void* stream) {
    if (!x || !y || !mask || n == 0 || !(p >= 0.f && p < 1.f)) return MGVAE_EINVAL;
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, as_stream(stream), x, y, mask, n, p,
                       seed, offset);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, size_t n, float sigma, uint64_t seed,
                                                    uint64_t offset) {
    const size_t groups = (n + 3) / 4;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (size_t)gridDim.x * 256) {
        const U4 r = philox4x32_10(g, offset, seed);
        // Box-Muller on (0,1] x [0,1)
        const float u1 = 1.0f - u01(r.x), u2 = u01(r.y), u3 = 1.0f - u01(r.z), u4 = u01(r.w);
        const float ra = sqrtf(-2.f * logf(u1)), rb = sqrtf(-2.f * logf(u3));
        const float v[4] = {ra * cosf(6.28318530718f * u2), ra * sinf(6.28318530718f * u2),
                            rb * cosf(6.28318530718f * u4), rb * sinf(6.28318530718f * u4)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = g * 4 + k;
            if (i < n) out[i] = v[k] * sigma;
        }
    }
}

extern "C" int mgvae_randn(float* out, size_t n, float sigma, uint64_t seed, uint64_t offset, void* stream) {
    if (!out || n == 0) return MGVAE_EINVAL;
    hipLaunchKernelGGL(randn_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, as_stream(stream), out, n, sigma, seed,
                       offset);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// ------------------------------------------------------------------ BCE losses
#define BCE_BLOCKS 1024
__device__ __forceinline__ float bce_target(const float* __restrict__ targets, const float* __restrict__ prior,
                                            float tconst, size_t i, int mode) {
    if (mode == 2) return tconst;
    const float t = targets[i];
    if (mode == 1) return ((t * 0.82f) + (float)(0.1 / 60)) + prior[i % 60];
    return t;
}

__global__ __launch_bounds__(256) void bce_partial_kernel(const float* __restrict__ x, const float* __restrict__ targets,
                                                          const float* __restrict__ prior, float tconst, size_t n,
                                                          int mode, int count_term, float* __restrict__ partial) {
    __shared__ float sa[4], sb[4];
    float acc = 0.f, cnt = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float xx = x[i];
        const float t = bce_target(targets, prior, tconst, i, mode);
        const float lx = fmaxf(logf(xx), -100.f), l1x = fmaxf(logf(1.f - xx), -100.f);
        acc -= t * lx + (1.f - t) * l1x;
        if (count_term) {
            const float lab = targets[i];
            const float o = xx > 0.3f ? 1.f : 0.f;
            cnt += (lab - o > 0.0001f) ? 1.f : 0.f;
        }
    }
    acc = wave_sum(acc); cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = acc; sb[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sa[0] + sa[1] + sa[2] + sa[3];
        partial[2 * blockIdx.x + 1] = sb[0] + sb[1] + sb[2] + sb[3];
    }
}

__global__ __launch_bounds__(256) void bce_final_kernel(const float* __restrict__ partial, int nblocks, size_t n,
                                                        float* __restrict__ loss_out) {
    __shared__ double sa[4], sb[4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += partial[2 * i]; b += partial[2 * i + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double bce = (sa[0] + sa[1] + sa[2] + sa[3]) / (double)n;
        const double cnt = sb[0] + sb[1] + sb[2] + sb[3];
        loss_out[0] = (float)bce + (float)cnt * 0.005f;
    }
}

extern "C" size_t mgvae_bce_partial_floats(void) { return 2 * BCE_BLOCKS; }

extern "C" int mgvae_bce_fwd(const float* x, const float* targets, const float* prior, float tconst, size_t n,
                             int target_mode, int count_term, float* partial, float* loss_out, void* stream) {
    if (!x || !partial || !loss_out || n == 0 || target_mode < 0 || target_mode > 2) return MGVAE_EINVAL;
    if ((target_mode != 2 || count_term) && !targets) return MGVAE_EINVAL;
    if (target_mode == 1 && !prior) return MGVAE_EINVAL;
    const int blocks = grid_for(n, BCE_BLOCKS);
    hipLaunchKernelGGL(bce_partial_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, targets, prior, tconst, n,
                       target_mode, count_term, partial);
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, blocks, n, loss_out);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ targets,
                                                      const float* __restrict__ prior, float tconst, size_t n, int mode,
                                                      const float* __restrict__ gscale, float* __restrict__ dx) {
    const float gs = gscale[0] / (float)n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float xx = x[i];
        const float t = bce_target(targets, prior, tconst, i, mode);
        dx[i] = gs * (xx - t) / fmaxf((1.f - xx) * xx, 1e-12f);
    }
}

extern "C" int mgvae_bce_bwd(const float* x, const float* targets, const float* prior, float tconst, size_t n,
                             int target_mode, const float* gscale, float* dx, void* stream) {
    if (!x || !gscale || !dx || n == 0 || target_mode < 0 || target_mode > 2) return MGVAE_EINVAL;
    if (target_mode != 2 && !targets) return MGVAE_EINVAL;
    if (target_mode == 1 && !prior) return MGVAE_EINVAL;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, targets, prior, tconst, n,
                       target_mode, gscale, dx);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// ------------------------------------------------------------------ reparameterise + KL
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ logvar,
                                                             const float* __restrict__ eps, float* __restrict__ z,
                                                             float* __restrict__ partial, size_t n) {
    __shared__ float sa[4];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float m = mean[i], lv = logvar[i];
        z[i] = m + eps[i] * expf(0.5f * lv);
        acc += 1.f + lv - m * m - expf(lv);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sa[0] + sa[1] + sa[2] + sa[3];
}

__global__ __launch_bounds__(256) void kl_final_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out) {
    __shared__ double sa[4];
    double a = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) a += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)(-0.5 * (sa[0] + sa[1] + sa[2] + sa[3]));
}

extern "C" int mgvae_reparam_kl_fwd(const float* mean, const float* logvar, const float* eps, float* z, float* partial,
                                    float* kl_out, size_t n, void* stream) {
    if (!mean || !logvar || !eps || !z || !partial || !kl_out || n == 0) return MGVAE_EINVAL;
    const int blocks = grid_for(n, BCE_BLOCKS);
    hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), mean, logvar, eps, z, partial, n);
    hipLaunchKernelGGL(kl_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, blocks, kl_out);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ logvar,
                                                             const float* __restrict__ eps, const float* __restrict__ dz,
                                                             const float* __restrict__ gkl, float* __restrict__ dmean,
                                                             float* __restrict__ dlogvar, size_t n) {
    const float gk = gkl[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float m = mean[i], lv = logvar[i], g = dz[i];
        dmean[i] = g + gk * m;
        dlogvar[i] = g * eps[i] * 0.5f * expf(0.5f * lv) + gk * 0.5f * (expf(lv) - 1.f);
    }
}

extern "C" int mgvae_reparam_kl_bwd(const float* mean, const float* logvar, const float* eps, const float* dz,
                                    const float* gkl, float* dmean, float* dlogvar, size_t n, void* stream) {
    if (!mean || !logvar || !eps || !dz || !gkl || !dmean || !dlogvar || n == 0) return MGVAE_EINVAL;
    hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), mean, logvar, eps, dz,
                       gkl, dmean, dlogvar, n);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// ------------------------------------------------------------------ bit-packed piano rolls -> fp32
// A bar is 96x60 = 5760 {0,1} cells: 720 bytes packed (LSB first, numpy packbits bitorder='little') against 23 KB as
// fp32.  The host ships the packed rows; this kernel expands them on the device: one thread per output byte-group
// of 8 cells (two float4 stores).
__global__ __launch_bounds__(256) void unpack_bits_kernel(const unsigned char* __restrict__ packed, float* __restrict__ out,
                                                          size_t nbytes, size_t nbits) {
    for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < nbytes; b += (size_t)gridDim.x * 256) {
        const unsigned v = packed[b];
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = (float)((v >> k) & 1u);
        const size_t o = b * 8;
        if (o + 8 <= nbits) {
            float4* o4 = reinterpret_cast<float4*>(out + o);
            o4[0] = make_float4(f[0], f[1], f[2], f[3]);
            o4[1] = make_float4(f[4], f[5], f[6], f[7]);
        } else {
            for (int k = 0; k < 8 && o + k < nbits; ++k) out[o + k] = f[k];
        }
    }
}

extern "C" int mgvae_unpack_bits(const unsigned char* packed, float* out, size_t nbits, void* stream) {
    if (!packed || !out || nbits == 0 || ((uintptr_t)out & 15)) return MGVAE_EINVAL;
    const size_t nbytes = (nbits + 7) / 8;
    hipLaunchKernelGGL(unpack_bits_kernel, dim3(grid_for(nbytes, 4096)), dim3(256), 0, as_stream(stream), packed, out, nbytes, nbits);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// ------------------------------------------------------------------ fused flat Adam
// 7 streams of n floats (read p,g,m,v; write p,m,v): pure HBM traffic, float4 per lane.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, const float* __restrict__ hyper,
                                                   float eps, float gscale) {
    const float step = hyper[0], sbc2 = hyper[1], b1 = hyper[2], b2 = hyper[3];
    const size_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = ga[k] * gscale;
            ma[k] = b1 * ma[k] + (1.f - b1) * gr;
            va[k] = b2 * va[k] + (1.f - b2) * gr * gr;
            pa[k] -= step * ma[k] / (sqrtf(va[k]) / sbc2 + eps);
        }
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0) {
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const float gr = g[i] * gscale;
            const float mm = b1 * m[i] + (1.f - b1) * gr;
            const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
            m[i] = mm; v[i] = vv;
            p[i] -= step * mm / (sqrtf(vv) / sbc2 + eps);
        }
    }
}

extern "C" int mgvae_adam_step(float* p, const float* g, float* m, float* v, size_t n, const float* hyper, float eps,
                               float grad_scale, void* stream) {
    if (!p || !g || !m || !v || !hyper || n == 0) return MGVAE_EINVAL;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return MGVAE_EINVAL;
    void* prof_tok = nullptr;      // measurement hook (bench.py roofline.hbm): reads p, g, m, v and writes p, m, v
    mgvae_prof_record_begin(MGVAE_PROF_ADAM, 0, 7.0 * 4.0 * (double)n, stream, &prof_tok);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1, 4096)), dim3(256), 0, as_stream(stream), p, g, m, v, n, hyper,
                       eps, grad_scale);
    mgvae_prof_record_end(prof_tok, stream);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// ---- bf16 gradient transport of the data-parallel exchange (hipops/dist.py, transport "bf16") ----------------------
// The flat fp32 gradient travels as bf16 and is summed in fp32 where it arrives; HBM-bound streaming kernels, 8 values
// per lane and instruction on the fp32 side.
typedef __bf16 bf16x4_pw __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        bf16x4_pw o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;     // round to nearest even
        reinterpret_cast<bf16x4_pw*>(dst)[i] = o;
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) dst[i] = (__bf16)src[i];
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const bf16x4_pw v = reinterpret_cast<const bf16x4_pw*>(src)[i];
        reinterpret_cast<float4*>(dst)[i] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) dst[i] = (float)src[i];
}
// dst[j] = bf16( sum_r rows[r][j] ), the sum in fp32 in rank order (deterministic); n a multiple of 4
__global__ __launch_bounds__(256) void bf16_rows_sum_kernel(const __bf16* __restrict__ rows, __bf16* __restrict__ dst, int R,
                                                            size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int r = 0; r < R; ++r) {
            const bf16x4_pw v = reinterpret_cast<const bf16x4_pw*>(rows + (size_t)r * n)[i];
            a0 += (float)v[0]; a1 += (float)v[1]; a2 += (float)v[2]; a3 += (float)v[3];
        }
        bf16x4_pw o;
        o[0] = (__bf16)a0; o[1] = (__bf16)a1; o[2] = (__bf16)a2; o[3] = (__bf16)a3;
        reinterpret_cast<bf16x4_pw*>(dst)[i] = o;
    }
}

extern "C" int mgvae_f32_to_bf16(const float* src, void* dst, size_t n, void* stream) {
    if (!src || !dst || n == 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return MGVAE_EINVAL;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n / 4 + 1, 4096)), dim3(256), 0, as_stream(stream), src,
                       static_cast<__bf16*>(dst), n);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}
extern "C" int mgvae_bf16_to_f32(const void* src, float* dst, size_t n, void* stream) {
    if (!src || !dst || n == 0 || ((uintptr_t)dst & 15) || ((uintptr_t)src & 7)) return MGVAE_EINVAL;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n / 4 + 1, 4096)), dim3(256), 0, as_stream(stream),
                       static_cast<const __bf16*>(src), dst, n);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}
extern "C" int mgvae_bf16_rows_sum(const void* rows, void* dst, int R, size_t n, void* stream) {
    if (!rows || !dst || R <= 0 || n == 0 || (n & 3) || ((uintptr_t)rows & 7) || ((uintptr_t)dst & 7)) return MGVAE_EINVAL;
    hipLaunchKernelGGL(bf16_rows_sum_kernel, dim3(grid_for(n / 4 + 1, 4096)), dim3(256), 0, as_stream(stream),
                       static_cast<const __bf16*>(rows), static_cast<__bf16*>(dst), R, n);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}
