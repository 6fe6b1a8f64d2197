// Direct (halo-tile) convolution on the fp32 matrix pipe for gfx950.
//
// Forward conv and the data-gradient / transposed-conv forward share one kernel: both are a
// "gather convolution" of a SOURCE tensor whose result has pixels on the MFMA lanes.
//   FWD      : source = X, taps = the KHxKW window, tile space = output pixels, source stride = S
//   BWD_DATA : one launch per stride phase; source = Y, taps = the taps that exist in that
//              phase, tile space = the phase's sub-grid of X, source stride = 1
// Per workgroup: (64*TI) output channels x (64*TJ) pixels.  The pixel tile is RTI whole rows of
// one image (or G whole small images), so for a chunk of CC source channels the source rows
// WITH HALO are staged once into LDS ([CC][G][SRI][SWp], zero-filled outside the tensor) and
// every tap of the window is a shifted LDS read -- a 3x3 conv fetches each input element once
// per chunk instead of 9 times, with no per-element bounds test in the inner loop.
// Weights come pre-packed ([chunk][i][t*CC + c], zero padded; mgvae_conv_pack) so a chunk's
// A tile is one contiguous float4 stream.  Inner loop per k-pair (2 channels of one tap):
// TI + TJ ds_read_b32 (both conflict-free: odd row pitch / lane-linear pixels) feed TI*TJ
// v_mfma_f32_32x32x2_f32.
#include "mgvae_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct DcP {
    const float* src; const float* wp; const float* bias; float* out;
    int N, Cs, SRH, SRW, s_ctot;      // source tensor [N, s_ctot, SRH, SRW], Cs channels used
    int Itot, o_ctot, OHt, OWt;       // output tensor [N, o_ctot, OHt, OWt], Itot channels written
    int PR, PC;                       // tile space: rows / cols per image
    int str_h, str_w, org_h, org_w;   // source coord = org + tile coord * str + tap
    int T;                            // taps
    int dh[16], dw[16];               // tap source offsets
    int dhmin, dwmin;
    int o_r0, o_sr, o_c0, o_sc;       // output coord = o_r0 + o_sr * r, o_c0 + o_sc * c
    int RTI, G, SRI, SWp, plane;      // tile geometry (plane = G*SRI*SWp)
    int tiles_per_img;                // G == 1: row tiles per image
    int CC, nchunk, Kp, Ipad, lda;    // channel chunking; Kp = padded CC*T; lda = Kp + 1
    int act; float slope;
    unsigned src_bytes;               // extent of the source tensor (buffer descriptor: masked elements read as 0)
};
#define DC_MAXX 8                     // halo-tile elements per thread and chunk that the register prefetch can hold

template <int TI, int TJ>
__global__ __launch_bounds__(256) void dconv_kernel(const DcP p) {
    constexpr int IT = 64 * TI, JT = 64 * TJ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                              // [IT][lda]
    float* Xs = As + IT * p.lda;                   // [CC][plane]
    int* goff = reinterpret_cast<int*>(Xs + p.CC * p.plane);   // [plane] source offset (no channel) or -1
    __shared__ int toff[16];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const int i0 = blockIdx.y * IT;
    const int HWs = p.SRH * p.SRW;

    // ---- which pixels does this workgroup own --------------------------------------------
    int img0, r0;                                  // first image, first tile-space row
    if (p.G == 1) { img0 = blockIdx.x / p.tiles_per_img; r0 = (blockIdx.x - img0 * p.tiles_per_img) * p.RTI; }
    else { img0 = blockIdx.x * p.G; r0 = 0; }
    const int rows_here = min(p.RTI, p.PR - r0);   // valid tile rows in each image of this tile
    const int px_per_img = p.RTI * p.PC;

    // ---- one-time tables -------------------------------------------------------------------
    if (tid < p.T) toff[tid] = (p.dh[tid] - p.dhmin) * p.SWp + (p.dw[tid] - p.dwmin);
    for (int q = tid; q < p.plane; q += 256) {
        const int g = q / (p.SRI * p.SWp), rem = q - g * (p.SRI * p.SWp);
        const int sr = rem / p.SWp, sc = rem - sr * p.SWp;
        const int n = img0 + g;
        const int row = p.org_h + r0 * p.str_h + p.dhmin + sr;
        const int col = p.org_w + p.dwmin + sc;
        const bool ok = n < p.N && (unsigned)row < (unsigned)p.SRH && (unsigned)col < (unsigned)p.SRW;
        goff[q] = ok ? (n * p.s_ctot) * HWs + row * p.SRW + col : -1;
    }

    // ---- per-lane operand addresses ----------------------------------------------------------
    int boff[TJ];                                  // LDS offset of this lane's pixel (tap 0, channel h)
    bool jok[TJ]; int obase[TJ];
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
        const int j = wj * 32 * TJ + tj * 32 + l31;
        const int g = j / px_per_img, rem = j - g * px_per_img;
        const int lr = rem / p.PC, c = rem - lr * p.PC;
        const bool ok = g < p.G && lr < rows_here && (img0 + g) < p.N;
        jok[tj] = ok;
        boff[tj] = ok ? g * (p.SRI * p.SWp) + (lr * p.str_h) * p.SWp + c * p.str_w + h * p.plane : h * p.plane;
        obase[tj] = ((img0 + g) * p.o_ctot) * (p.OHt * p.OWt) + (p.o_r0 + p.o_sr * (r0 + lr)) * p.OWt + p.o_c0 + p.o_sc * c;
    }
    int aoff[TI];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) aoff[ti] = (wi * 32 * TI + ti * 32 + l31) * p.lda + h;

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

    const int KQ = p.Kp >> 2;                      // float4 per packed row
    const int nA4 = IT * KQ;
    const int nx = p.CC * p.plane;
    // Register prefetch: the weight tile and the halo tile of chunk ch+1 are loaded from global memory while the
    // MFMAs of chunk ch run; one LDS buffer is enough (a chunk is committed to LDS after the barrier that ends the
    // previous chunk's MFMAs).  Per-thread counts are bounded by the planner (Kp <= 72, CC * plane <= MAXX * 256).
    constexpr int MAXA4 = (IT * 18 + 255) / 256;
    constexpr int MAXX = DC_MAXX;
    float4 ra[MAXA4];
    float rx[MAXX];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
    __syncthreads();                               // goff / toff are in place
    // the halo-tile elements this thread stages are the same in every chunk: decode them once
    int xoff[MAXX], xc[MAXX];                      // source offset without the chunk's channel base (-1: padding), channel in chunk
#pragma unroll
    for (int m = 0; m < MAXX; ++m) {
        const int e = tid + 256 * m;
        const int ec = e < nx ? e : 0;
        const int c = ec / p.plane, q = ec - c * p.plane;
        const int o = goff[q];
        xc[m] = c;
        xoff[m] = (e < nx && o >= 0) ? c * HWs + o : -1;
    }
    auto prefetch = [&](int ch) {
        const float4* wsrc = reinterpret_cast<const float4*>(p.wp + ((size_t)ch * p.Ipad + i0) * p.Kp);
#pragma unroll
        for (int q = 0; q < MAXA4; ++q) {
            const int e = tid + 256 * q;
            ra[q] = wsrc[e < nA4 ? e : 0];             // contiguous float4 stream; the tail is loaded, not stored
        }
        const int c0 = ch * p.CC;
#pragma unroll
        for (int m = 0; m < MAXX; ++m) {
            const bool ok = (xoff[m] >= 0) & ((c0 + xc[m]) < p.Cs);
            rx[m] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ok ? (c0 * HWs + xoff[m]) * 4 : -1, 0, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < MAXA4; ++q) {
            const int e = tid + 256 * q;
            if (e < nA4) {
                const int row = e / KQ, col = (e - row * KQ) << 2;
                float* d = As + row * p.lda + col;
                d[0] = ra[q].x; d[1] = ra[q].y; d[2] = ra[q].z; d[3] = ra[q].w;
            }
        }
#pragma unroll
        for (int m = 0; m < MAXX; ++m) {
            const int e = tid + 256 * m;
            if (e < nx) Xs[e] = rx[m];
        }
    };
    prefetch(0);

    for (int ch = 0; ch < p.nchunk; ++ch) {
        commit();
        __syncthreads();
        if (ch + 1 < p.nchunk) prefetch(ch + 1);
        // ---- MFMA over (tap, channel pair) ----
        for (int t = 0; t < p.T; ++t) {
            const int to = toff[t];
            const float* ap = As + t * p.CC;
            const float* bp = Xs + to;
#pragma unroll 4
            for (int cc = 0; cc < p.CC; cc += 2) {
                float a[TI], b[TJ];
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) a[ti] = ap[aoff[ti] + cc];
#pragma unroll
                for (int tj = 0; tj < TJ; ++tj) b[tj] = bp[boff[tj] + cc * p.plane];
#pragma unroll
                for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TJ; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: D row (i) = (reg&3) + 8*(reg>>2) + 4*h, D col (pixel) = lane&31 ----
    const int ostr = p.OHt * p.OWt;
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
        if (!jok[tj]) continue;
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gi = i0 + wi * 32 * TI + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (gi < p.Itot) {
                    float v = acc[ti][tj][r];
                    if (p.bias) v += p.bias[gi];
                    p.out[obase[tj] + gi * ostr] = apply_act(v, p.act, p.slope);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight packing: wp[chunk][i][t*CC + c] = W(i, chunk*CC + c, tapw[t]), zero padded
struct PackP {
    const float* w; float* wp;
    int mode;            // 0: FWD  W[i][cs][tap];  1: BWD_DATA  W[cs][i][tap]
    int Itot, Cs, KK;    // KK = KH*KW
    int T; int tapw[16];
    int CC, nchunk, Kp, Ipad;
};

__global__ __launch_bounds__(256) void dconv_pack_kernel(const PackP p) {
    const long total = (long)p.nchunk * p.Ipad * p.Kp;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int k = (int)(e % p.Kp);
        const long r = e / p.Kp;
        const int i = (int)(r % p.Ipad), ch = (int)(r / p.Ipad);
        float v = 0.f;
        if (k < p.CC * p.T && i < p.Itot) {
            const int t = k / p.CC, c = k - t * p.CC;
            const int cs = ch * p.CC + c;
            if (cs < p.Cs)
                v = p.mode == 0 ? p.w[((size_t)i * p.Cs + cs) * p.KK + p.tapw[t]]
                                : p.w[((size_t)cs * p.Itot + i) * p.KK + p.tapw[t]];
        }
        p.wp[e] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// host side: planning
struct Plan {
    DcP k; PackP pk;
    int tile;            // 0: 128x128, 1: 64(i)x128(j), 2: 128(i)x64(j), 3: 64x64
    dim3 grid; size_t lds; size_t wp_floats;
    bool empty;          // phase without taps: output = act(bias)
    double flops;
};

extern int g_mgvae_cus;
#define g_dc_cus g_mgvae_cus

static int pick_cc(int T, int Cs) {
    int cc = T >= 16 ? 4 : T >= 9 ? 8 : T >= 6 ? 12 : T >= 4 ? 16 : T == 3 ? 24 : T == 2 ? 32 : 64;
    int cs_even = (Cs + 1) & ~1;
    if (cc > cs_even) cc = cs_even;
    return cc < 2 ? 2 : cc;
}

// mode 0: FWD.  mode 1: BWD_DATA phase `ph` (0 .. SH*SW-1).  Returns false if unsupported.
static bool make_plan(const MgvaeConvDesc* d, int mode, int ph, Plan& P) {
    DcP& k = P.k; PackP& pk = P.pk;
    k = DcP{}; pk = PackP{};
    P.empty = false;
    int tapw[16];
    if (mode == 0) {
        k.N = d->N; k.Cs = d->Cx; k.SRH = d->H; k.SRW = d->W; k.s_ctot = d->x_ctot;
        k.Itot = d->Cy; k.o_ctot = d->y_ctot; k.OHt = d->OH; k.OWt = d->OW;
        k.PR = d->OH; k.PC = d->OW; k.str_h = d->SH; k.str_w = d->SW; k.org_h = -d->PH; k.org_w = -d->PW;
        k.T = d->KH * d->KW;
        for (int t = 0; t < k.T; ++t) { k.dh[t] = t / d->KW; k.dw[t] = t % d->KW; tapw[t] = t; }
        k.o_r0 = 0; k.o_sr = 1; k.o_c0 = 0; k.o_sc = 1;
    } else {
        const int rh = ph / d->SW, rw = ph % d->SW;
        const int kh0 = (rh + d->PH) % d->SH, kw0 = (rw + d->PW) % d->SW;
        const int nkh = kh0 < d->KH ? (d->KH - kh0 + d->SH - 1) / d->SH : 0;
        const int nkw = kw0 < d->KW ? (d->KW - kw0 + d->SW - 1) / d->SW : 0;
        const int qh = (rh + d->PH - kh0) / d->SH, qw = (rw + d->PW - kw0) / d->SW;
        const int Ha = rh < d->H ? (d->H - rh + d->SH - 1) / d->SH : 0;
        const int Wb = rw < d->W ? (d->W - rw + d->SW - 1) / d->SW : 0;
        k.N = d->N; k.Cs = d->Cy; k.SRH = d->OH; k.SRW = d->OW; k.s_ctot = d->y_ctot;
        k.Itot = d->Cx; k.o_ctot = d->x_ctot; k.OHt = d->H; k.OWt = d->W;
        k.PR = Ha; k.PC = Wb; k.str_h = 1; k.str_w = 1; k.org_h = 0; k.org_w = 0;
        k.T = nkh * nkw;
        for (int t = 0; t < k.T; ++t) {
            const int jh = t / nkw, jw = t % nkw;
            k.dh[t] = qh - jh; k.dw[t] = qw - jw;
            tapw[t] = (kh0 + d->SH * jh) * d->KW + kw0 + d->SW * jw;
        }
        k.o_r0 = rh; k.o_sr = d->SH; k.o_c0 = rw; k.o_sc = d->SW;
        if (Ha == 0 || Wb == 0) { P.empty = true; P.wp_floats = 0; P.grid = dim3(0, 0, 0); return true; }
    }
    if (k.T > 16) return false;
    k.act = d->act; k.slope = d->slope;
    if (k.T == 0) {                      // a phase no tap reaches: result is act(bias); handled by T=1 with zero weights
        k.T = 1; k.dh[0] = 0; k.dw[0] = 0; tapw[0] = -1;
    }
    k.dhmin = k.dh[0]; k.dwmin = k.dw[0];
    int dhmax = k.dh[0], dwmax = k.dw[0];
    for (int t = 1; t < k.T; ++t) {
        k.dhmin = k.dh[t] < k.dhmin ? k.dh[t] : k.dhmin; k.dwmin = k.dw[t] < k.dwmin ? k.dw[t] : k.dwmin;
        dhmax = k.dh[t] > dhmax ? k.dh[t] : dhmax; dwmax = k.dw[t] > dwmax ? k.dw[t] : dwmax;
    }
    if (k.PC > 128) return false;
    // ---- tile shape: prefer 128 pixels, drop to 64 when that leaves the chip under-filled ----
    const long px_img = (long)k.PR * k.PC, px_all = px_img * k.N;
    int ti = k.Itot > 64 ? 2 : 1;
    auto geometry = [&](int JT) {
        if (px_img >= JT) { k.G = 1; k.RTI = JT / k.PC; if (k.RTI < 1) k.RTI = 1; k.tiles_per_img = cdiv(k.PR, k.RTI); }
        else { k.G = (int)(JT / px_img); k.RTI = k.PR; k.tiles_per_img = 1; if (k.G > k.N) k.G = k.N; }
        return k.G == 1 ? (long)k.N * k.tiles_per_img : (long)cdiv(k.N, k.G);
    };
    int tj = px_all > 64 ? 2 : 1;
    long jt_tiles = geometry(64 * tj);
    const long want = (long)g_dc_cus * 3 / 2;
    if (jt_tiles * cdiv(k.Itot, 64 * ti) < want && tj == 2) { tj = 1; jt_tiles = geometry(64); }
    if (jt_tiles * cdiv(k.Itot, 64 * ti) < want && ti == 2) ti = 1;
    if (k.RTI * k.PC > 64 * tj) return false;      // a single row wider than the tile
    k.SRI = (k.RTI - 1) * k.str_h + (dhmax - k.dhmin) + 1;
    k.SWp = (k.PC - 1) * k.str_w + (dwmax - k.dwmin) + 1;
    k.plane = k.G * k.SRI * k.SWp;
    // ---- channel chunk under an LDS budget of 64 KB ----
    const int IT = 64 * ti;
    int cc = pick_cc(k.T, k.Cs);
    auto lds_bytes = [&](int c) {
        const int kp = ((c * k.T + 3) / 4) * 4;
        return (size_t)4 * ((size_t)IT * (kp + 1) + (size_t)c * k.plane + k.plane);
    };
    while (cc > 2 && (lds_bytes(cc) > 60 * 1024 || (long)cc * k.plane > DC_MAXX * 256)) cc -= 2;
    if (lds_bytes(cc) > 60 * 1024 || (long)cc * k.plane > DC_MAXX * 256 || ((cc * k.T + 3) / 4) * 4 > 72) return false;
    {
        const size_t sb = (size_t)k.N * k.s_ctot * k.SRH * k.SRW * 4;
        if (sb >= ((size_t)1 << 32)) return false;
        k.src_bytes = (unsigned)sb;
    }
    k.CC = cc; k.nchunk = cdiv(k.Cs, cc); k.Kp = ((cc * k.T + 3) / 4) * 4; k.lda = k.Kp + 1;
    k.Ipad = cdiv(k.Itot, IT) * IT;
    P.lds = lds_bytes(cc);
    P.tile = (ti == 2 ? 0 : 1) + (tj == 2 ? 0 : 2);
    P.grid = dim3((unsigned)jt_tiles, (unsigned)cdiv(k.Itot, IT), 1);
    P.wp_floats = (size_t)k.nchunk * k.Ipad * k.Kp;
    pk.mode = mode; pk.Itot = k.Itot; pk.Cs = k.Cs; pk.KK = d->KH * d->KW; pk.T = k.T;
    for (int t = 0; t < k.T; ++t) pk.tapw[t] = tapw[t] < 0 ? 0 : tapw[t];
    if (tapw[0] < 0) pk.Cs = 0;                    // tap-less phase: pack zeros
    pk.CC = cc; pk.nchunk = k.nchunk; pk.Kp = k.Kp; pk.Ipad = k.Ipad;
    P.flops = 2.0 * k.Itot * (double)px_all * k.Cs * (tapw[0] < 0 ? 0 : k.T);
    return true;
}

static int check_desc(const MgvaeConvDesc* d) {
    if (!d) return MGVAE_EINVAL;
    if (d->N <= 0 || d->Cx <= 0 || d->Cy <= 0 || d->H <= 0 || d->W <= 0 || d->OH <= 0 || d->OW <= 0) return MGVAE_EINVAL;
    if (d->KH <= 0 || d->KW <= 0 || d->SH <= 0 || d->SW <= 0 || d->PH < 0 || d->PW < 0) return MGVAE_EINVAL;
    if (d->KH * d->KW > 16 || d->SH * d->SW > 8) return MGVAE_EINVAL;
    if ((d->H + 2 * d->PH - d->KH) / d->SH + 1 != d->OH) return MGVAE_EINVAL;
    if ((d->W + 2 * d->PW - d->KW) / d->SW + 1 != d->OW) return MGVAE_EINVAL;
    if (d->x_coff != 0 || d->y_coff != 0) return MGVAE_EINVAL;     // packed path: pass sliced pointers
    if (d->Cx > d->x_ctot || d->Cy > d->y_ctot) return MGVAE_EINVAL;
    const long xe = (long)d->N * d->x_ctot * d->H * d->W, ye = (long)d->N * d->y_ctot * d->OH * d->OW;
    if (xe >= (1L << 31) || ye >= (1L << 31)) return MGVAE_EINVAL;
    return MGVAE_OK;
}

// number of floats of the packed-weight workspace for this conv in the given mode (0 = unsupported)
extern "C" size_t mgvae_conv_pack_floats(const MgvaeConvDesc* d, int mode) {
    if (check_desc(d) || mode < 0 || mode > 1) return 0;
    const int nph = mode == 0 ? 1 : d->SH * d->SW;
    size_t tot = 0;
    for (int ph = 0; ph < nph; ++ph) {
        Plan P;
        if (!make_plan(d, mode, ph, P)) return 0;
        tot += (P.wp_floats + 63) / 64 * 64;
    }
    return tot;
}

extern "C" int mgvae_conv_pack(const MgvaeConvDesc* d, int mode, const float* w, float* packed, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!w || !packed || mode < 0 || mode > 1) return MGVAE_EINVAL;
    const int nph = mode == 0 ? 1 : d->SH * d->SW;
    size_t off = 0;
    for (int ph = 0; ph < nph; ++ph) {
        Plan P;
        if (!make_plan(d, mode, ph, P)) return MGVAE_EINVAL;
        if (P.empty || P.wp_floats == 0) continue;
        P.pk.w = w; P.pk.wp = packed + off;
        const long total = (long)P.wp_floats;
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(dconv_pack_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), P.pk);
        off += (P.wp_floats + 63) / 64 * 64;
    }
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

static int launch_plan(const Plan& P, hipStream_t s) {
    switch (P.tile) {
        case 0: hipLaunchKernelGGL((dconv_kernel<2, 2>), P.grid, dim3(256), P.lds, s, P.k); break;
        case 1: hipLaunchKernelGGL((dconv_kernel<1, 2>), P.grid, dim3(256), P.lds, s, P.k); break;
        case 2: hipLaunchKernelGGL((dconv_kernel<2, 1>), P.grid, dim3(256), P.lds, s, P.k); break;
        default: hipLaunchKernelGGL((dconv_kernel<1, 1>), P.grid, dim3(256), P.lds, s, P.k); break;
    }
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

extern thread_local int g_prof_note[16];
// shared with conv_igemm.hip's profiler
extern "C" int mgvae_prof_record_begin(int kind, int tile, double flops, void* stream, void** token);
extern "C" int mgvae_prof_record_end(void* token, void* stream);

static int run_mode(const MgvaeConvDesc* d, int mode, const float* src, const float* packed, const float* bias,
                    float* out, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!src || !packed || !out) return MGVAE_EINVAL;
    const int nph = mode == 0 ? 1 : d->SH * d->SW;
    size_t off = 0;
    for (int ph = 0; ph < nph; ++ph) {
        Plan P;
        if (!make_plan(d, mode, ph, P)) return MGVAE_EINVAL;
        if (P.empty) continue;
        P.k.src = src; P.k.wp = packed + off; P.k.bias = bias; P.k.out = out;
        void* tok = nullptr;
        {   // geometry note for the per-launch profile: N,Cs,SRH,SRW,Itot,PR,PC,T,CC,nchunk,G*1000+RTI,grid
            const DcP& k = P.k;
            int v[14] = {k.N, k.Cs, k.SRH, k.SRW, k.Itot, k.PR, k.PC, k.T, k.CC, k.nchunk, k.G * 1000 + k.RTI,
                         (int)P.grid.x, (int)P.grid.y, (int)(P.lds / 1024)};
            for (int i = 0; i < 14; ++i) g_prof_note[i] = v[i];
        }
        mgvae_prof_record_begin(3 + mode, P.tile, P.flops, stream, &tok);
        rc = launch_plan(P, as_stream(stream));
        mgvae_prof_record_end(tok, stream);
        if (rc) return rc;
        off += (P.wp_floats + 63) / 64 * 64;
    }
    return MGVAE_OK;
}

extern "C" int mgvae_conv2d_fwd_packed(const MgvaeConvDesc* d, const float* x, const float* packed, const float* bias,
                                       float* y, void* stream) {
    return run_mode(d, 0, x, packed, bias, y, stream);
}

extern "C" int mgvae_conv2d_bwd_data_packed(const MgvaeConvDesc* d, const float* y, const float* packed,
                                            const float* bias, float* x, void* stream) {
    return run_mode(d, 1, y, packed, bias, x, stream);
}

