// Implicit-GEMM convolution family for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32): Conv2d forward, Conv2d data-gradient == ConvTranspose2d
// forward (stride-phase decomposed so no zero taps are multiplied), and the weight
// gradient (split-K over pixels, fp32 atomics into the caller-zeroed gradient).
//
// One workgroup = 256 threads = 4 waves arranged 2x2; wave (wi, wj) owns TI x TJ MFMA
// tiles of 32x32, so the workgroup tile is (64*TI) x (64*TJ).  The MFMA "column" (lane)
// index always runs along the memory-contiguous axis of the written tensor (pixels for
// activations, (cx,kh,kw) for weight gradients) so every accumulator register stores as
// two 128-byte segments.  Operand tiles are staged global -> registers -> LDS with a
// two-deep LDS ring; the gather loaders keep the K index wave-uniform so tap decode is
// done once per wave-instruction, and LDS images are laid out so both the staging
// writes and the MFMA fragment reads (ds_read_b32, lanes 0-31 / 32-63) are
// conflict-free.
#include "mgvae_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 16
#define LDP (BK + 1)   // padded row length of the "row-major, k fastest" LDS images

struct IgemmP {
    const float* X;     // image-side tensor  [N, x_ctot, H, W]
    const float* Y;     // feature-side tensor [N, y_ctot, OH, OW]
    const float* Wt;    // [Cy, Cx, KH, KW]
    const float* bias;  // fwd: [Cy]; bwd_data: [Cx]; may be null
    float* out;         // fwd: Y; bwd_data: X; bwd_weight: dWt
    int N, Cx, H, W, Cy, OH, OW, KH, KW, SH, SW, PH, PW;
    int x_ctot, x_coff, y_ctot, y_coff;
    int act;
    float slope;
    int kchunk;         // bwd_weight: pixels per split
    int ksplit;         // fwd / bwd_data: K splits (blockIdx.z = phase * ksplit + split); >1 => atomic epilogue
    const int2* ktab;   // fwd / bwd_data: per-K gather table {src offset, dh | dw << 16}; bwd_data: [phase][stride]
    const int* wtab;    // bwd_data: per-K weight offset (cy * Cx * KH*KW + tap), [phase][stride]
    int ktab_stride;    // entries per phase (padded by 16 so a wave can always read its whole row group)
    int w_transposed;   // bwd_data: Wt is [Cy][KH*KW][Cx] (mgvae_weight_transpose) -> lane-contiguous A loads
    unsigned x_bytes, y_bytes, w_bytes;   // extents of X / Y / Wt for the buffer descriptors (< 4 GiB each)
    int xcd_remap;      // 1: workgroup ids are permuted so that each XCD (own L2) walks a contiguous run of tiles
    // optional epilogue factor act'(m) read from a tensor shaped like the output (fwd / bwd_data): the activation
    // gradient of the layer that PRODUCED this conv's input, folded into the conv's own data gradient
    const float* mask;
    int mask_ctot, mask_coff, mask_act;
    float mask_slope;
    int bf16;           // 1: igemm_bf16_kernel with bf16 operands; 3: with three-term bf16 splits (fp32 accuracy)
};

// Operand loads go through buffer descriptors: a masked element gets the offset 0xFFFFFFFF, which the
// hardware range-checks and answers with 0.0f.  No select on the loaded VALUE means nothing consumes a
// load before the LDS store that follows the MFMA block, so the whole K tile stays in flight under it.
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const float* ptr, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptr), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, int elem, bool ok) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, ok ? elem * 4 : -1, 0, 0));
}

__device__ __forceinline__ float buf_load_b(rsrc_t r, int byte_off) {     // byte_off >= 2^31 (e.g. -1): reads as 0
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

#ifndef MGVAE_ABL
#define MGVAE_ABL 0   // timing-only ablations of the main loop (WRONG results): 1 no global loads, 2 no LDS stores, 4 no barrier, 8 no LDS fragment reads
#endif
#ifndef MGVAE_FRAG_AHEAD
#define MGVAE_FRAG_AHEAD 1
#endif
enum { MODE_FWD = 0, MODE_BWD_DATA = 1, MODE_BWD_WEIGHT = 2 };
enum { KT_PAD_TAP = 31, KT_MAX_TAPS = 31 };
__device__ __forceinline__ int kt_dh(int y) { return (int)(signed char)(y & 0xff); }
__device__ __forceinline__ int kt_dw(int y) { return (int)(signed char)((y >> 8) & 0xff); }
__device__ __forceinline__ int kt_tap(int y) { return (int)((unsigned)y >> 16); }

// ---------------------------------------------------------------------------------------
// MFMA over one staged K-tile.  A image: A_IK ? As[i][LDP] : As[k][IT];
//                               B image: B_KJ ? Bs[k][JT]  : Bs[j][LDP].
template <int TI, int TJ, bool A_IK, bool B_KJ, int BKc>
__device__ __forceinline__ void mma_tile(const float* __restrict__ As, const float* __restrict__ Bs,
                                         f32x16 (&acc)[TI][TJ], int a_off, int b_off) {
    constexpr int IT = 64 * TI, JT = 64 * TJ, LDPc = BKc + 1;
    // a_off / b_off: this lane's element offset at (k-step 0, MFMA tile 0), computed once per kernel (mma_offsets);
    // every fragment read below is then base + compile-time constant = one ds_read with an immediate offset
    const float* __restrict__ ap = As + a_off;
    const float* __restrict__ bp = Bs + b_off;
    // register ring: the LDS reads of k-step kk+AHEAD are issued before the MFMAs of k-step kk
    constexpr int AHEAD = MGVAE_FRAG_AHEAD, RING = AHEAD + 1, KS = BKc / 2;
    float a[RING][TI], b[RING][TJ];
    auto fetch = [&](int kk, int s) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) a[s][ti] = A_IK ? ap[ti * 32 * LDPc + 2 * kk] : ap[2 * kk * IT + ti * 32];
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) b[s][tj] = B_KJ ? bp[2 * kk * JT + tj * 32] : bp[tj * 32 * LDPc + 2 * kk];
    };
#pragma unroll
    for (int kk = 0; kk < AHEAD && kk < KS; ++kk) fetch(kk, kk % RING);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        if (!(MGVAE_ABL & 8) && kk + AHEAD < KS) fetch(kk + AHEAD, (kk + AHEAD) % RING);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs (the scheduler sinks them otherwise)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj)
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(MGVAE_ABL & 8) ? 0 : kk % RING][ti], b[(MGVAE_ABL & 8) ? 0 : kk % RING][tj], acc[ti][tj], 0, 0, 0);
    }
}
// lane offsets of mma_tile: A image A_IK ? As[i][LDP] : As[k][IT]; B image B_KJ ? Bs[k][JT] : Bs[j][LDP];
// i = wi*32*TI + ti*32 + l31, j = wj*32*TJ + tj*32 + l31, k = 2*kk + h
template <int TI, int TJ, bool A_IK, bool B_KJ, int BKc>
__device__ __forceinline__ void mma_offsets(int wi, int wj, int l31, int h, int* a_off, int* b_off) {
    constexpr int IT = 64 * TI, JT = 64 * TJ, LDPc = BKc + 1;
    *a_off = A_IK ? (wi * 32 * TI + l31) * LDPc + h : h * IT + wi * 32 * TI + l31;
    *b_off = B_KJ ? h * JT + wj * 32 * TJ + l31 : (wj * 32 * TJ + l31) * LDPc + h;
}

// Running decode of a K index that enumerates (channel, tap): advance by `step`.
struct ChanTap {
    int c, t;
    __device__ __forceinline__ void init(int k, int T) { c = k / T; t = k - c * T; }
    __device__ __forceinline__ void advance(int step, int T) {
        t += step;
        while (t >= T) { t -= T; ++c; }
    }
};

// ---------------------------------------------------------------------------------------
template <int MODE, int TI, int TJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void igemm_kernel(const IgemmP p) {
    constexpr int IT = 64 * TI, JT = 64 * TJ;
    constexpr bool A_IK = (MODE != MODE_BWD_DATA);   // weights [cy][k] / dY [cy][pix]: k contiguous
    constexpr bool B_KJ = (MODE != MODE_BWD_WEIGHT); // gathers with pixels along lanes
    // K tile: 16 (17-34 KB of LDS per workgroup -> 4+ workgroups per CU to hide the loaders' latency; measured
    // 27.2 -> 26.4 ms/step against 32), except the narrower weight-gradient tiles, which prefer 32
    constexpr int BKc = (TI * TJ >= 4 || MODE != MODE_BWD_WEIGHT) ? 16 : 32;
    constexpr int LDPc = BKc + 1;          // padded row length of the "row-major, k fastest" LDS images
    constexpr int RP = 256 / BKc;          // rows per pass of the "lanes along k" loaders
    constexpr int A_ELEMS = A_IK ? IT * LDPc : BKc * IT;
    constexpr int B_ELEMS = B_KJ ? BKc * JT : JT * LDPc;
    constexpr int NA = IT * BKc / 256, NB = JT * BKc / 256;

    __shared__ float lds[2 * (A_ELEMS + B_ELEMS)];
    float* As0 = lds;
    float* Bs0 = lds + 2 * A_ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    // Hardware deals consecutive workgroup ids round-robin over the 8 XCDs.  Permute the ids so that XCD x
    // executes the x-th contiguous eighth of the (z, y, x) tile order: the tiles sharing an A row block /
    // a K split then share one L2 instead of being replicated into all eight.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_remap) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned lin = bx + gridDim.x * (by + gridDim.y * bz);
        const unsigned xcd = lin & 7, slot = lin >> 3, per = total >> 3, rem = total & 7;
        const unsigned l2 = xcd < rem ? xcd * (per + 1) + slot : rem * (per + 1) + (xcd - rem) * per + slot;
        bx = l2 % gridDim.x;
        const unsigned t2 = l2 / gridDim.x;
        by = t2 % gridDim.y; bz = t2 / gridDim.y;
    }
    const int i0 = by * IT, j0 = bx * JT;

    const int HW = p.H * p.W, P = p.OH * p.OW;

    // ---------------- per-mode problem shape + tap tables ------------------------------
    int Itot, Jtot, kbeg, kend, T;
    // bwd_data phase constants
    int rh = 0, rw = 0, Wb = 1, Pp = 1;
    int Ktot = 0, split = 0;
    if constexpr (MODE == MODE_FWD) {
        Itot = p.Cy; Jtot = p.N * P; T = p.KH * p.KW; Ktot = p.Cx * T;
        split = (int)bz;
        const int kc = ((Ktot + p.ksplit - 1) / p.ksplit + BKc - 1) / BKc * BKc;
        kbeg = split * kc; kend = min(Ktot, kbeg + kc);
    } else if constexpr (MODE == MODE_BWD_DATA) {
        const int ph = (int)bz / p.ksplit;
        split = (int)bz - ph * p.ksplit;
        rh = ph / p.SW; rw = ph - rh * p.SW;
        const int kh0 = (rh + p.PH) % p.SH, kw0 = (rw + p.PW) % p.SW;
        const int nkh = kh0 < p.KH ? (p.KH - kh0 + p.SH - 1) / p.SH : 0;
        const int nkw = kw0 < p.KW ? (p.KW - kw0 + p.SW - 1) / p.SW : 0;
        const int qh = (rh + p.PH - kh0) / p.SH, qw = (rw + p.PW - kw0) / p.SW;
        const int Ha = rh < p.H ? (p.H - rh + p.SH - 1) / p.SH : 0;
        Wb = rw < p.W ? (p.W - rw + p.SW - 1) / p.SW : 0;
        Pp = Ha * Wb;
        Itot = p.Cx; Jtot = p.N * Pp; T = nkh * nkw; Ktot = p.Cy * T;
        {
            const int kc = ((Ktot + p.ksplit - 1) / p.ksplit + BKc - 1) / BKc * BKc;
            kbeg = split * kc; kend = min(Ktot, kbeg + kc);
            if (split > 0 && kbeg >= Ktot) return;   // nothing left for this split (split 0 still writes bias)
        }
        if (Jtot == 0 || j0 >= Jtot) return;   // uniform across the workgroup
        (void)qh; (void)qw;
        if (T == 0) { T = 1; kbeg = 0; kend = 0; }   // no taps: loop never runs
    } else {
        Itot = p.Cy; Jtot = p.Cx * p.KH * p.KW; T = 1;
        kbeg = (int)bz * p.kchunk;
        kend = min(p.N * P, kbeg + p.kchunk);
    }
    const int2* __restrict__ ktab = p.ktab;
    const int* __restrict__ wtab = p.wtab;
    if constexpr (MODE == MODE_BWD_DATA) {
        ktab += (size_t)((int)bz / p.ksplit) * p.ktab_stride;
        wtab += (size_t)((int)bz / p.ksplit) * p.ktab_stride;
    }

    // ---------------- loader state ------------------------------------------------------
    // "lanes along j/i" mapping (KJ / KI images): each wave-instruction covers 64 columns
    // of one K row; the K row is wave-uniform.
    // Each wave owns a CONTIGUOUS group of K rows, so its gather constants are one scalar load.
    constexpr int JC = JT / 64, IC = IT / 64;
    const int jc = wave % JC, jkr0 = (wave / JC) * NB;
    const int ic = wave % IC, ikr0 = (wave / IC) * NA;
    // "lanes along k" mapping (IK / JK images): 16 lanes per row, 16 rows per pass.
    const int kl = tid % BKc, rr = tid / BKc;

    // B gather state (FWD: from X; BWD_DATA: from Y)
    bool bj_valid = false; int b_pix = 0, b_r0 = 0, b_c0 = 0, b_RH = 1, b_RW = 1;
    // A gather state (BWD_DATA weights)
    bool ai_valid = false; int a_i = 0;
    // BWD_WEIGHT state
    int w_n = 0, w_p = 0;                      // running pixel of this thread's k lane
    int bj_off[NB], bj_tap[NB];                // per-row (cx,kh,kw) constants: source offset, tap index
    int a_row4[NA];                            // per-row byte offset into dY (out of range past the last row)
    int2 w_pe = make_int2(0, -1);              // running pixel's {offset of its window origin in X, invalid-tap bits}
    (void)bj_off; (void)bj_tap; (void)a_row4; (void)w_pe;

    if constexpr (MODE == MODE_FWD) {
        const int j = j0 + jc * 64 + lane;
        bj_valid = j < Jtot;
        const int jj = bj_valid ? j : 0;
        const int n = jj / P, pp = jj - n * P;
        const int oh = pp / p.OW, ow = pp - oh * p.OW;
        b_r0 = oh * p.SH - p.PH; b_c0 = ow * p.SW - p.PW;
        b_pix = (n * p.x_ctot + p.x_coff) * HW + b_r0 * p.W + b_c0;
        b_RH = p.H; b_RW = p.W;
    } else if constexpr (MODE == MODE_BWD_DATA) {
        const int j = j0 + jc * 64 + lane;
        bj_valid = j < Jtot;
        const int jj = bj_valid ? j : 0;
        const int n = jj / Pp, pp = jj - n * Pp;
        const int a = pp / Wb, b = pp - a * Wb;
        b_r0 = a; b_c0 = b;
        b_pix = (n * p.y_ctot + p.y_coff) * P + a * p.OW + b;
        b_RH = p.OH; b_RW = p.OW;
        a_i = i0 + ic * 64 + lane;
        ai_valid = a_i < Itot;
    } else {
        // ktab here is the per-PIXEL table of the geometry: entry p = {(oh*SH)*W + ow*SW, one bit per tap that falls
        // outside the image for that pixel}.  The window origin may lie in the padding (negative offsets are fine:
        // every tap that is used lies inside).
        const int kp = kbeg + kl;
        w_n = kp / P; w_p = kp - w_n * P;
        w_pe = ktab[w_p];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const int gj = j0 + rr + RP * r;
            if (gj < Jtot) {
                const int KK = p.KH * p.KW;
                const int cx = gj / KK, t = gj - cx * KK;
                const int kh = t / p.KW, kw = t - kh * p.KW;
                bj_tap[r] = t;
                bj_off[r] = cx * HW + (kh - p.PH) * p.W + (kw - p.PW);
            } else {
                bj_tap[r] = KT_PAD_TAP; bj_off[r] = 0;                // never in range
            }
        }
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            const int gi = i0 + rr + RP * r;
            a_row4[r] = gi < Itot ? gi * P * 4 : (int)0x80000000;
        }
    }

    // Validity of a gathered element = (this thread's pixel, the K row's tap): one bit per tap, computed once.  A set
    // bit of `nmask` means "outside the image" (bit KT_PAD_TAP: the padding rows past the end of K; all bits: no pixel).
    // In the loop an element then costs three vector instructions (bit -> 0 / -1, offset, or) instead of two
    // compares, four adds and a select plus eight scalar ones.
    unsigned nmask = 0xffffffffu;
    if constexpr (MODE != MODE_BWD_WEIGHT) {
        unsigned m = 1u << KT_PAD_TAP;
        for (int t = 0; t < T; ++t) {
            const int y = ktab[t].y;                  // channel 0 lists the taps in order
            const bool in = ((unsigned)(b_r0 + kt_dh(y)) < (unsigned)b_RH) & ((unsigned)(b_c0 + kt_dw(y)) < (unsigned)b_RW);
            m |= in ? 0u : (1u << t);
        }
        if (bj_valid) nmask = m;
    }
    const int b_pix4 = b_pix * 4;
    // weight rows past the end: an offset that stays out of range whatever (small) K offset is added to it
    int a_base4[NA];
    (void)a_base4; (void)b_pix4;
    if constexpr (MODE == MODE_FWD) {
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            const int gi = i0 + rr + RP * r;
            a_base4[r] = gi < Itot ? (gi * Ktot + kl) * 4 : (int)0x80000000;
        }
    }
    const int KKw = p.KH * p.KW;
    const int a_off_bd = ai_valid ? (p.w_transposed ? a_i : a_i * KKw) : 0x20000000;   // elements; << 2 in the loop
    (void)a_off_bd;

    float ra[NA], rb[NB];
    const rsrc_t rX = make_rsrc(p.X, p.x_bytes), rY = make_rsrc(p.Y, p.y_bytes), rW = make_rsrc(p.Wt, p.w_bytes);
    (void)rX; (void)rY; (void)rW;

    // All loads are UNCONDITIONAL buffer loads (masked elements are sent out of range and read as 0),
    // so the compiler emits one straight-line block: every load of a K tile is in flight together (a
    // predicated load would be a branch + wait per element).  The per-K gather constants come from
    // the ktab table through scalar loads (the K row is wave-uniform).
    // The per-K gather constants come from scalar loads; with 16-deep K tiles an SMEM round trip per tile would sit
    // in front of every tile's global loads, so the entries of tile t+1 are requested while tile t's loads are issued.
    int2 e_pre[NB];
    int wo_pre[NA];
    (void)e_pre; (void)wo_pre;
    auto prefetch_tables = [&](int k0) {
        if constexpr (MODE != MODE_BWD_WEIGHT) {
            const int2* __restrict__ kt = ktab + (k0 + jkr0);
#pragma unroll
            for (int r = 0; r < NB; ++r) e_pre[r] = kt[r];           // one scalar load (rows are contiguous)
        }
        if constexpr (MODE == MODE_BWD_DATA) {
            const int* __restrict__ wt = wtab + (k0 + ikr0);
#pragma unroll
            for (int r = 0; r < NA; ++r) wo_pre[r] = wt[r];
        }
    };
    auto load_tile = [&](int k0) {
        if constexpr ((MGVAE_ABL & 64) != 0) {          // timing only: the same number of loads, trivial addresses
#pragma unroll
            for (int r = 0; r < NA; ++r) ra[r] = buf_load(MODE == MODE_BWD_WEIGHT ? rY : rW, (int)threadIdx.x + 256 * r + (k0 & 1023), true);
#pragma unroll
            for (int r = 0; r < NB; ++r) rb[r] = buf_load(MODE == MODE_FWD ? rX : (MODE == MODE_BWD_DATA ? rY : rX), (int)threadIdx.x + 256 * r + (k0 & 1023), true);
            return;
        }
        // ------------------------------ A operand ------------------------------
        if constexpr (MODE == MODE_FWD) {
            // K rows past kend multiply B rows that read as 0 (padding taps), so only the row bound is checked
            const int k04 = k0 * 4;
#pragma unroll
            for (int r = 0; r < NA; ++r) ra[r] = buf_load_b(rW, a_base4[r] + k04);
        } else if constexpr (MODE == MODE_BWD_DATA) {
            int wo[NA];
#pragma unroll
            for (int r = 0; r < NA; ++r) wo[r] = wo_pre[r];
#pragma unroll
            for (int r = 0; r < NA; ++r) ra[r] = buf_load_b(rW, (wo[r] + a_off_bd) << 2);
        } else {
            // past kend: an offset that is out of range with or without a row offset added (buffers are < 2 GB)
            const bool kok = (k0 + kl) < kend;
            const int abase4 = kok ? ((w_n * p.y_ctot + p.y_coff) * P + w_p) * 4 : 0x7ffffffc;
#pragma unroll
            for (int r = 0; r < NA; ++r) ra[r] = buf_load_b(rY, abase4 + a_row4[r]);
        }
        // ------------------------------ B operand ------------------------------
        if constexpr (MODE != MODE_BWD_WEIGHT) {
            const rsrc_t src = (MODE == MODE_FWD) ? rX : rY;
            int2 e[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) e[r] = e_pre[r];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const int inval = __builtin_amdgcn_sbfe((int)nmask, (unsigned)kt_tap(e[r].y), 1u);   // 0 or -1
                rb[r] = buf_load_b(src, (b_pix4 + e[r].x * 4) | inval);
            }
        } else {
            const bool kok = (k0 + kl) < kend;
            const int ninv = kok ? w_pe.y : -1;
            const int base = (w_n * p.x_ctot + p.x_coff) * HW + w_pe.x;
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const int inval = __builtin_amdgcn_sbfe(ninv, (unsigned)bj_tap[r], 1u);   // 0 or -1
                rb[r] = buf_load_b(rX, ((base + bj_off[r]) << 2) | inval);
            }
            // advance this thread's pixel by one K tile; its table entry is back before the next tile's loads
            w_p += BKc;
            while (w_p >= P) { w_p -= P; ++w_n; }
            w_pe = ktab[w_p];
        }
    };

    auto store_tile = [&](int buf) {
        float* As = As0 + buf * A_ELEMS;
        float* Bs = Bs0 + buf * B_ELEMS;
        if constexpr (A_IK) {
#pragma unroll
            for (int r = 0; r < NA; ++r) As[(rr + RP * r) * LDPc + kl] = ra[r];
        } else {
#pragma unroll
            for (int r = 0; r < NA; ++r) As[(ikr0 + r) * IT + ic * 64 + lane] = ra[r];
        }
        if constexpr (B_KJ) {
#pragma unroll
            for (int r = 0; r < NB; ++r) Bs[(jkr0 + r) * JT + jc * 64 + lane] = rb[r];
        } else {
#pragma unroll
            for (int r = 0; r < NB; ++r) Bs[(rr + RP * r) * LDPc + kl] = rb[r];
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

    int mma_a_off, mma_b_off;
    mma_offsets<TI, TJ, A_IK, B_KJ, BKc>(wi, wj, l31, h, &mma_a_off, &mma_b_off);
    const int nt = kend > kbeg ? (kend - kbeg + BKc - 1) / BKc : 0;
    if (nt > 0) {
        prefetch_tables(kbeg);
        load_tile(kbeg);
        prefetch_tables(kbeg + BKc);                   // (the tables are padded past their end)
        store_tile(0);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            if (!(MGVAE_ABL & 1) && t + 1 < nt) load_tile(kbeg + (t + 1) * BKc);
            mma_tile<TI, TJ, A_IK, B_KJ, BKc>(As0 + buf * A_ELEMS, Bs0 + buf * B_ELEMS, acc, mma_a_off, mma_b_off);
            // entries of tile t+2: requested after the last LDS fragment read (SMEM and LDS share a counter, and an
            // outstanding scalar load would turn every fragment wait into a full drain), in flight under the LDS
            // store, the barrier and the start of the next trip
            if (!(MGVAE_ABL & (1 | 64))) prefetch_tables(kbeg + (t + 2) * BKc);
            if (!(MGVAE_ABL & 2) && (!(MGVAE_ABL & 16) || p.N < 0) && t + 1 < nt) store_tile(buf ^ 1);
            if (!(MGVAE_ABL & 4)) __syncthreads();
        }
    }

    // ---------------- epilogue -----------------------------------------------------------
    // D row (i) = (reg&3) + 8*(reg>>2) + 4*h ; D col (j) = lane&31
    if constexpr (MODE == MODE_BWD_WEIGHT) {
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
            const int gj = j0 + wj * 32 * TJ + tj * 32 + l31;
            if (gj >= Jtot) continue;
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gi = i0 + wi * 32 * TI + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (gi < Itot) atomicAdd(&p.out[(size_t)gi * Jtot + gj], acc[ti][tj][r]);
                }
            }
        }
    } else {
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
            const int gj = j0 + wj * 32 * TJ + tj * 32 + l31;
            if (gj >= Jtot) continue;
            int obase, cstride, mbase = 0;
            if constexpr (MODE == MODE_FWD) {
                const int n = gj / P, pp = gj - n * P;
                obase = (n * p.y_ctot + p.y_coff) * P + pp; cstride = P;
                mbase = (n * p.mask_ctot + p.mask_coff) * P + pp;
            } else {
                const int n = gj / Pp, pp = gj - n * Pp;
                const int a = pp / Wb, b = pp - a * Wb;
                const int pix = (rh + p.SH * a) * p.W + rw + p.SW * b;
                obase = (n * p.x_ctot + p.x_coff) * HW + pix;
                mbase = (n * p.mask_ctot + p.mask_coff) * HW + pix;
                cstride = HW;
            }
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gi = i0 + wi * 32 * TI + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (gi < Itot) {
                        float v = acc[ti][tj][r];
                        // act'(m) is a per-element factor, so it distributes over the K splits
                        const float mk = p.mask ? act_grad_from_out(p.mask[mbase + gi * cstride], p.mask_act, p.mask_slope) : 1.f;
                        if (p.ksplit > 1) {       // split-K: caller zeroed the output, activation runs afterwards
                            if (p.bias && split == 0) v += p.bias[gi];
                            atomicAdd(&p.out[obase + gi * cstride], v * mk);
                        } else {
                            if (p.bias) v += p.bias[gi];
                            p.out[obase + gi * cstride] = apply_act(v, p.act, p.slope) * mk;
                        }
                    }
                }
            }
        }
    }
}

// =======================================================================================
// bf16-compute variant (BASELINE.json configs 3-4: "bf16 compute / fp32 master weights"): tensors stay fp32 in HBM,
// the loaders round to bf16 (RNE, v_cvt_pk_bf16_f32) while staging into LDS and the product runs on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- 1/16 of the matrix-pipe time of the fp32 kernel, half the LDS
// traffic.  Every LDS image is [row][k] (k fastest, 40 bf16 = 80 B per row: 16-byte aligned rows whose ds_read_b128
// fragment reads are bank-conflict-free): an MFMA operand is 8 consecutive k of one row = ONE 128-bit LDS read.
//   * column-mapped loaders (a thread holds NB consecutive K rows of one column -- gathers, transposed weights):
//     pack the NB values and store them as 128-bit rows;
//   * k-mapped loaders (weights [cy][k], dY [cy][pix], the weight-gradient gather): a thread owns a PAIR of
//     consecutive k = one packed dword.
// Geometry tables, split-K, bias / activation / mask epilogue are those of the fp32 kernel.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16_t;
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    bf16x2_t v;
    v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// NS = 1: plain bf16 operands.  NS = 3: every fp32 operand is split EXACTLY into three bf16 terms x = hi + mid + lo
// (8 + 8 + 8 mantissa bits) kept as three LDS planes, and a product is the six bf16 MFMAs whose terms are >= 2^-16
// relative: hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid.  bf16 x bf16 is exact in the fp32 accumulator, the
// dropped terms are <= 2^-24 relative, so the result has fp32 accuracy at 6/16 of the fp32 matrix-pipe time.
// Truncating split (3 VALU ops + 2 subtractions, all exact): hi = top 8 mantissa bits, mid = the next 8, lo = the rest;
// every term is a bf16 value held in a float whose low 16 bits are zero.
__device__ __forceinline__ void split3(float x, float& hi, float& mid, float& lo) {
    hi = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    const float r1 = x - hi;              // exact, <= 16 significant bits
    mid = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    lo = r1 - mid;                        // exact, <= 8 significant bits: a bf16 value
}
// two floats that ARE bf16 values -> one packed dword (low half = first): a byte permute, no rounding needed
__device__ __forceinline__ unsigned pack_bf16_exact(float lo_elem, float hi_elem) {
    return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}

template <int MODE, int TI, int TJ, int NS>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(const IgemmP p) {
    constexpr int IT = 64 * TI, JT = 64 * TJ;
    constexpr int BKc = 32, LDB = BKc + 8;
    constexpr bool A_K = (MODE != MODE_BWD_DATA);      // A loaded with lanes along k
    constexpr bool B_K = (MODE == MODE_BWD_WEIGHT);    // B loaded with lanes along k
    constexpr int A_ELEMS = IT * LDB, B_ELEMS = JT * LDB;
    constexpr int NA = IT * BKc / 256, NB = JT * BKc / 256;   // values per thread and K tile
    constexpr int NA2 = NA / 2, NB2 = NB / 2;                  // k pairs per thread (k-mapped loaders)

    __shared__ __attribute__((aligned(16))) u16_t lds[2 * NS * (A_ELEMS + B_ELEMS)];   // [buffer][plane] per operand
    u16_t* As0 = lds;
    u16_t* Bs0 = lds + 2 * NS * A_ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const int i0 = blockIdx.y * IT, j0 = blockIdx.x * JT;
    const int bz = blockIdx.z;
    const int HW = p.H * p.W, P = p.OH * p.OW;

    int Itot, Jtot, kbeg, kend, T;
    int rh = 0, rw = 0, Wb = 1, Pp = 1;
    int Ktot = 0, split = 0;
    if constexpr (MODE == MODE_FWD) {
        Itot = p.Cy; Jtot = p.N * P; T = p.KH * p.KW; Ktot = p.Cx * T;
        split = bz;
        const int kc = ((Ktot + p.ksplit - 1) / p.ksplit + BKc - 1) / BKc * BKc;
        kbeg = split * kc; kend = min(Ktot, kbeg + kc);
    } else if constexpr (MODE == MODE_BWD_DATA) {
        const int ph = bz / p.ksplit;
        split = bz - ph * p.ksplit;
        rh = ph / p.SW; rw = ph - rh * p.SW;
        const int kh0 = (rh + p.PH) % p.SH, kw0 = (rw + p.PW) % p.SW;
        const int nkh = kh0 < p.KH ? (p.KH - kh0 + p.SH - 1) / p.SH : 0;
        const int nkw = kw0 < p.KW ? (p.KW - kw0 + p.SW - 1) / p.SW : 0;
        const int Ha = rh < p.H ? (p.H - rh + p.SH - 1) / p.SH : 0;
        Wb = rw < p.W ? (p.W - rw + p.SW - 1) / p.SW : 0;
        Pp = Ha * Wb;
        Itot = p.Cx; Jtot = p.N * Pp; T = nkh * nkw; Ktot = p.Cy * T;
        const int kc = ((Ktot + p.ksplit - 1) / p.ksplit + BKc - 1) / BKc * BKc;
        kbeg = split * kc; kend = min(Ktot, kbeg + kc);
        if (split > 0 && kbeg >= Ktot) return;
        if (Jtot == 0 || j0 >= Jtot) return;
        if (T == 0) { T = 1; kbeg = 0; kend = 0; }
    } else {
        Itot = p.Cy; Jtot = p.Cx * p.KH * p.KW; T = 1;
        kbeg = bz * p.kchunk;
        kend = min(p.N * P, kbeg + p.kchunk);
    }
    const int2* __restrict__ ktab = p.ktab;
    const int* __restrict__ wtab = p.wtab;
    if constexpr (MODE == MODE_BWD_DATA) {
        ktab += (size_t)(bz / p.ksplit) * p.ktab_stride;
        wtab += (size_t)(bz / p.ksplit) * p.ktab_stride;
    }

    // column mapping: a wave-instruction covers 64 columns of one (wave-uniform) K row
    constexpr int JC = JT / 64, IC = IT / 64;
    const int jc = wave % JC, jkr0 = (wave / JC) * NB;
    const int ic = wave % IC, ikr0 = (wave / IC) * NA;
    // k mapping: 16 lanes x 2 consecutive k per row, 16 rows per pass
    const int kl2 = (tid & 15) * 2, rr = tid >> 4;

    bool bj_valid = false; int b_pix = 0, b_r0 = 0, b_c0 = 0, b_RH = 1, b_RW = 1;
    bool ai_valid = false; int a_i = 0;
    int w_n = 0, w_p = 0;                          // (sample, pixel) of this thread's FIRST k of the pair
    int bj_off[NB2 > 0 ? NB2 : 1], bj_tap[NB2 > 0 ? NB2 : 1];
    int a_row4[NA2 > 0 ? NA2 : 1];                 // weight gradient: per-row byte offset into dY
    int2 w_pe0 = make_int2(0, -1), w_pe1 = make_int2(0, -1);   // per-pixel table entries of the two k of the pair
    (void)bj_off; (void)bj_tap; (void)a_row4; (void)w_pe0; (void)w_pe1;

    if constexpr (MODE == MODE_FWD) {
        const int j = j0 + jc * 64 + lane;
        bj_valid = j < Jtot;
        const int jj = bj_valid ? j : 0;
        const int n = jj / P, pp = jj - n * P;
        const int oh = pp / p.OW, ow = pp - oh * p.OW;
        b_r0 = oh * p.SH - p.PH; b_c0 = ow * p.SW - p.PW;
        b_pix = (n * p.x_ctot + p.x_coff) * HW + b_r0 * p.W + b_c0;
        b_RH = p.H; b_RW = p.W;
    } else if constexpr (MODE == MODE_BWD_DATA) {
        const int j = j0 + jc * 64 + lane;
        bj_valid = j < Jtot;
        const int jj = bj_valid ? j : 0;
        const int n = jj / Pp, pp = jj - n * Pp;
        const int a = pp / Wb, b = pp - a * Wb;
        b_r0 = a; b_c0 = b;
        b_pix = (n * p.y_ctot + p.y_coff) * P + a * p.OW + b;
        b_RH = p.OH; b_RW = p.OW;
        a_i = i0 + ic * 64 + lane;
        ai_valid = a_i < Itot;
    } else {
        // per-pixel table (see igemm_kernel): {window origin offset, invalid-tap bits}
        const int kp = kbeg + kl2;
        w_n = kp / P; w_p = kp - w_n * P;
        w_pe0 = ktab[w_p];
        w_pe1 = ktab[w_p + 1 >= P ? w_p + 1 - P : w_p + 1];
#pragma unroll
        for (int r = 0; r < NB2; ++r) {
            const int gj = j0 + rr + 16 * r;
            if (gj < Jtot) {
                const int KK = p.KH * p.KW;
                const int cx = gj / KK, t = gj - cx * KK;
                const int kh = t / p.KW, kw = t - kh * p.KW;
                bj_tap[r] = t;
                bj_off[r] = cx * HW + (kh - p.PH) * p.W + (kw - p.PW);
            } else {
                bj_tap[r] = KT_PAD_TAP; bj_off[r] = 0;
            }
        }
#pragma unroll
        for (int r = 0; r < NA2; ++r) {
            const int gi = i0 + rr + 16 * r;
            a_row4[r] = gi < Itot ? gi * P * 4 : (int)0x80000000;
        }
    }

    // one validity bit per tap, as in igemm_kernel
    unsigned nmask = 0xffffffffu;
    if constexpr (MODE != MODE_BWD_WEIGHT) {
        unsigned m = 1u << KT_PAD_TAP;
        for (int t = 0; t < T; ++t) {
            const int y = ktab[t].y;
            const bool in = ((unsigned)(b_r0 + kt_dh(y)) < (unsigned)b_RH) & ((unsigned)(b_c0 + kt_dw(y)) < (unsigned)b_RW);
            m |= in ? 0u : (1u << t);
        }
        if (bj_valid) nmask = m;
    }
    const int b_pix4 = b_pix * 4;
    int a_base4[NA2 > 0 ? NA2 : 1];
    (void)a_base4; (void)b_pix4;
    if constexpr (MODE == MODE_FWD) {
#pragma unroll
        for (int r = 0; r < NA2; ++r) {
            const int gi = i0 + rr + 16 * r;
            a_base4[r] = gi < Itot ? (gi * Ktot + kl2) * 4 : (int)0x80000000;
        }
    }
    const int a_off_bd = ai_valid ? (p.w_transposed ? a_i : a_i * (p.KH * p.KW)) : 0x20000000;
    (void)a_off_bd;

    float ra[NA], rb[NB];
    const rsrc_t rX = make_rsrc(p.X, p.x_bytes), rY = make_rsrc(p.Y, p.y_bytes), rW = make_rsrc(p.Wt, p.w_bytes);
    (void)rX; (void)rY; (void)rW;

    auto load_tile = [&](int k0) {
        // (sample, pixel) of the two k of this thread's pair (weight gradient only)
        int n1 = w_n, p1 = w_p + 1;
        if constexpr (MODE == MODE_BWD_WEIGHT) { if (p1 >= P) { p1 -= P; ++n1; } }
        // ------------------------------ A operand ------------------------------
        if constexpr (MODE == MODE_FWD) {
            const int k04 = k0 * 4;
#pragma unroll
            for (int r = 0; r < NA2; ++r) {
                ra[2 * r] = buf_load_b(rW, a_base4[r] + k04);
                ra[2 * r + 1] = buf_load_b(rW, a_base4[r] + k04 + 4);
            }
        } else if constexpr (MODE == MODE_BWD_DATA) {
            const int* __restrict__ wt = wtab + (k0 + ikr0);
            int wo[NA];
#pragma unroll
            for (int r = 0; r < NA; ++r) wo[r] = wt[r];
#pragma unroll
            for (int r = 0; r < NA; ++r) ra[r] = buf_load_b(rW, (wo[r] + a_off_bd) << 2);
        } else {
            const bool k0ok = (k0 + kl2) < kend, k1ok = (k0 + kl2 + 1) < kend;
            const int abase0 = k0ok ? ((w_n * p.y_ctot + p.y_coff) * P + w_p) * 4 : 0x7ffffffc;
            const int abase1 = k1ok ? ((n1 * p.y_ctot + p.y_coff) * P + p1) * 4 : 0x7ffffffc;
#pragma unroll
            for (int r = 0; r < NA2; ++r) {
                ra[2 * r] = buf_load_b(rY, abase0 + a_row4[r]);
                ra[2 * r + 1] = buf_load_b(rY, abase1 + a_row4[r]);
            }
        }
        // ------------------------------ B operand ------------------------------
        if constexpr (MODE != MODE_BWD_WEIGHT) {
            const rsrc_t src = (MODE == MODE_FWD) ? rX : rY;
            const int2* __restrict__ kt = ktab + (k0 + jkr0);
            int2 e[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) e[r] = kt[r];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const int inval = __builtin_amdgcn_sbfe((int)nmask, (unsigned)kt_tap(e[r].y), 1u);
                rb[r] = buf_load_b(src, (b_pix4 + e[r].x * 4) | inval);
            }
        } else {
            const bool k0ok = (k0 + kl2) < kend, k1ok = (k0 + kl2 + 1) < kend;
            const int ninv0 = k0ok ? w_pe0.y : -1, ninv1 = k1ok ? w_pe1.y : -1;
            const int base0 = (w_n * p.x_ctot + p.x_coff) * HW + w_pe0.x;
            const int base1 = (n1 * p.x_ctot + p.x_coff) * HW + w_pe1.x;
#pragma unroll
            for (int r = 0; r < NB2; ++r) {
                rb[2 * r] = buf_load_b(rX, ((base0 + bj_off[r]) << 2) | __builtin_amdgcn_sbfe(ninv0, (unsigned)bj_tap[r], 1u));
                rb[2 * r + 1] = buf_load_b(rX, ((base1 + bj_off[r]) << 2) | __builtin_amdgcn_sbfe(ninv1, (unsigned)bj_tap[r], 1u));
            }
            w_p += BKc;
            while (w_p >= P) { w_p -= P; ++w_n; }
            w_pe0 = ktab[w_p];
            w_pe1 = ktab[w_p + 1 >= P ? w_p + 1 - P : w_p + 1];
        }
    };

    // plane s of a register array: s = 0 the value rounded to bf16 (NS = 1) or its hi / mid / lo term (NS = 3)
    auto term = [&](float x, int s) -> float {
        if constexpr (NS == 1) { return x; }
        else { float hi, mid, lo; split3(x, hi, mid, lo); return s == 0 ? hi : (s == 1 ? mid : lo); }
    };
    auto pk = [&](float a0, float a1) -> unsigned {
        if constexpr (NS == 1) return pack_bf16(a0, a1); else return pack_bf16_exact(a0, a1);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            u16_t* As = As0 + (buf * NS + s) * A_ELEMS;
            u16_t* Bs = Bs0 + (buf * NS + s) * B_ELEMS;
            if constexpr (A_K) {
#pragma unroll
                for (int r = 0; r < NA2; ++r)
                    *reinterpret_cast<unsigned*>(As + (rr + 16 * r) * LDB + kl2) = pk(term(ra[2 * r], s), term(ra[2 * r + 1], s));
            } else {
                u16_t* row = As + (ic * 64 + lane) * LDB + ikr0;
#pragma unroll
                for (int q = 0; q < NA / 8; ++q)
                    *reinterpret_cast<uint4*>(row + 8 * q) =
                        make_uint4(pk(term(ra[8 * q], s), term(ra[8 * q + 1], s)), pk(term(ra[8 * q + 2], s), term(ra[8 * q + 3], s)),
                                   pk(term(ra[8 * q + 4], s), term(ra[8 * q + 5], s)), pk(term(ra[8 * q + 6], s), term(ra[8 * q + 7], s)));
            }
            if constexpr (B_K) {
#pragma unroll
                for (int r = 0; r < NB2; ++r)
                    *reinterpret_cast<unsigned*>(Bs + (rr + 16 * r) * LDB + kl2) = pk(term(rb[2 * r], s), term(rb[2 * r + 1], s));
            } else {
                u16_t* row = Bs + (jc * 64 + lane) * LDB + jkr0;
#pragma unroll
                for (int q = 0; q < NB / 8; ++q)
                    *reinterpret_cast<uint4*>(row + 8 * q) =
                        make_uint4(pk(term(rb[8 * q], s), term(rb[8 * q + 1], s)), pk(term(rb[8 * q + 2], s), term(rb[8 * q + 3], s)),
                                   pk(term(rb[8 * q + 4], s), term(rb[8 * q + 5], s)), pk(term(rb[8 * q + 6], s), term(rb[8 * q + 7], s)));
            }
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

    const int nt = kend > kbeg ? (kend - kbeg + BKc - 1) / BKc : 0;
    if (nt > 0) {
        load_tile(kbeg);
        store_tile(0);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            if (t + 1 < nt) load_tile(kbeg + (t + 1) * BKc);
            const u16_t* As = As0 + buf * NS * A_ELEMS;
            const u16_t* Bs = Bs0 + buf * NS * B_ELEMS;
#pragma unroll
            for (int ks = 0; ks < BKc / 16; ++ks) {
                bf16x8_t a[NS][TI], b[NS][TJ];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
#pragma unroll
                    for (int ti = 0; ti < TI; ++ti)
                        a[s][ti] = *reinterpret_cast<const bf16x8_t*>(As + s * A_ELEMS + (wi * 32 * TI + ti * 32 + l31) * LDB + ks * 16 + h * 8);
#pragma unroll
                    for (int tj = 0; tj < TJ; ++tj)
                        b[s][tj] = *reinterpret_cast<const bf16x8_t*>(Bs + s * B_ELEMS + (wj * 32 * TJ + tj * 32 + l31) * LDB + ks * 16 + h * 8);
                }
#pragma unroll
                for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TJ; ++tj) {
                        if constexpr (NS == 3) {      // small terms first
                            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][ti], b[1][tj], acc[ti][tj], 0, 0, 0);
                            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][ti], b[2][tj], acc[ti][tj], 0, 0, 0);
                            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][ti], b[0][tj], acc[ti][tj], 0, 0, 0);
                            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][ti], b[1][tj], acc[ti][tj], 0, 0, 0);
                            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][ti], b[0][tj], acc[ti][tj], 0, 0, 0);
                        }
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][ti], b[0][tj], acc[ti][tj], 0, 0, 0);
                    }
            }
            if (t + 1 < nt) store_tile(buf ^ 1);
            __syncthreads();
        }
    }

    // ---------------- epilogue (as in the fp32 kernel) -----------------------------------
    if constexpr (MODE == MODE_BWD_WEIGHT) {
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
            const int gj = j0 + wj * 32 * TJ + tj * 32 + l31;
            if (gj >= Jtot) continue;
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gi = i0 + wi * 32 * TI + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (gi < Itot) atomicAdd(&p.out[(size_t)gi * Jtot + gj], acc[ti][tj][r]);
                }
            }
        }
    } else {
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
            const int gj = j0 + wj * 32 * TJ + tj * 32 + l31;
            if (gj >= Jtot) continue;
            int obase, cstride, mbase = 0;
            if constexpr (MODE == MODE_FWD) {
                const int n = gj / P, pp = gj - n * P;
                obase = (n * p.y_ctot + p.y_coff) * P + pp; cstride = P;
                mbase = (n * p.mask_ctot + p.mask_coff) * P + pp;
            } else {
                const int n = gj / Pp, pp = gj - n * Pp;
                const int a = pp / Wb, b = pp - a * Wb;
                const int pix = (rh + p.SH * a) * p.W + rw + p.SW * b;
                obase = (n * p.x_ctot + p.x_coff) * HW + pix;
                mbase = (n * p.mask_ctot + p.mask_coff) * HW + pix;
                cstride = HW;
            }
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gi = i0 + wi * 32 * TI + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (gi < Itot) {
                        float v = acc[ti][tj][r];
                        const float mk = p.mask ? act_grad_from_out(p.mask[mbase + gi * cstride], p.mask_act, p.mask_slope) : 1.f;
                        if (p.ksplit > 1) {
                            if (p.bias && split == 0) v += p.bias[gi];
                            atomicAdd(&p.out[obase + gi * cstride], v * mk);
                        } else {
                            if (p.bias) v += p.bias[gi];
                            p.out[obase + gi * cstride] = apply_act(v, p.act, p.slope) * mk;
                        }
                    }
                }
            }
        }
    }
}

// Forward of a "thin" convolution (Cx*KH*KW <= 16: the C=1 stems of graph/encodingBlock.py:12-15 and the first
// layers of graph/bar_discriminator.py): K <= 16 would idle the MFMA and the layer is bound by writing Y.  One
// thread per output pixel gathers its <= 16 taps once and produces every output channel from them; the weights
// (and bias) sit in LDS and are read as broadcasts, the stores of one channel are contiguous along the lanes.
constexpr int THIN_W_MAX = 4096;          // Cy * J floats that fit the LDS weight stage
template <int JMAX, int ACT>
__global__ __launch_bounds__(256) void thin_fwd_kernel(const IgemmP p, int J) {
    __shared__ float wsh[THIN_W_MAX];
    __shared__ float bsh[THIN_W_MAX / 4];
    const int P = p.OH * p.OW, HW = p.H * p.W, KK = p.KH * p.KW;
    for (int t = threadIdx.x; t < p.Cy * J; t += 256) wsh[t] = p.Wt[t];
    for (int t = threadIdx.x; t < p.Cy; t += 256) bsh[t] = p.bias ? p.bias[t] : 0.f;
    __syncthreads();
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int oh = i / p.OW, ow = i - oh * p.OW;
    const int r0 = oh * p.SH - p.PH, c0 = ow * p.SW - p.PW;
    const float* xb = p.X + ((size_t)n * p.x_ctot + p.x_coff) * HW;
    float xv[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int jj = j < J ? j : 0;
        const int cx = jj / KK, t = jj - cx * KK;
        const int kh = t / p.KW, kw = t - kh * p.KW;
        const int r = r0 + kh, c = c0 + kw;
        const bool ok = (j < J) & ((unsigned)r < (unsigned)p.H) & ((unsigned)c < (unsigned)p.W);
        const float v = xb[ok ? cx * HW + r * p.W + c : 0];
        xv[j] = ok ? v : 0.f;
    }
    float* ob = p.out + ((size_t)n * p.y_ctot + p.y_coff) * P + i;
#pragma unroll 8
    for (int cy = 0; cy < p.Cy; ++cy) {
        const float* wr = wsh + cy * J;
        float v = bsh[cy];
#pragma unroll
        for (int j = 0; j < JMAX; ++j) v += wr[j < J ? j : 0] * xv[j];   // xv[j] = 0 beyond J
        ob[(size_t)cy * P] = apply_act(v, ACT, p.slope);
    }
}

// =======================================================================================
// Skinny GEMMs: the Linear layers (H = W = 1, 1x1 kernel) at batch <= 128 -- z-discriminators, the
// encoder / decoder fully connected layers, the flattened decoder stems.  The tiled kernel above is
// latency-bound there (20 us for 33 MFLOP: one 64-wide pixel tile, K walked serially through LDS).
// Here the batch IS the MFMA column block and nothing goes through LDS on the way in: with a K chunk
// of 8 laid out as k = kb + 4*h + s (h = lane >> 5, s = MFMA step 0..3), both operands of
// v_mfma_f32_32x32x2_f32 are one float4 global load per lane (row = lane & 31, 16 contiguous bytes).
// A workgroup owns 32 output features; its four waves interleave the K chunks and combine through
// LDS, then write rows of 32 contiguous features.  grid.y splits K further (atomics into a
// zeroed output, activation by a follow-up pass) only when grid.x alone cannot fill the chip.
struct SkinnyP {
    const float* A;      // weight: [I][K] (A_KI = false) or [K][I] (A_KI = true), leading dimension a_ld
    const float* B;      // activations [N rows][K], row stride b_ld (channel-sliced buffers)
    const float* bias;   // [I] or null
    float* out;          // [N rows][I], row stride o_ld
    int I, N, K, a_ld, b_ld, o_ld, act, kchunk;
    float slope;
};

template <int NT, bool A_KI>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const SkinnyP p) {
    // 4 waves share one block of 32 output features: wave w takes the K chunks (of 8) w, w+4, ...; the problem is
    // latency-bound (few waves, every load a miss), so a wave first issues the loads of CH chunks, then multiplies.
    // The four partial blocks meet in LDS (one slab per wave -- LDS float atomics serialise per lane), two column
    // tiles at a time.
    constexpr int CH = NT <= 2 ? 4 : 3;
    constexpr int STRIDE = 4 * 8;
    constexpr int TP = NT < 2 ? NT : 2;                 // column tiles per reduction pass
    __shared__ float red[4][TP * 32][33];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * 32;
    const int kbeg = blockIdx.y * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    // rows past the edge are clamped, computed and never written: no predication on the loads
    const int ia = min(i0 + l31, p.I - 1);
    const float* ap = A_KI ? p.A + ia : p.A + (size_t)ia * p.a_ld;
    const float* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bp[t] = p.B + (size_t)min(32 * t + l31, p.N - 1) * p.b_ld;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int kb0 = kbeg + wave * 8; kb0 < kend; kb0 += STRIDE * CH) {
        float4 a[CH], b[CH][NT];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int k = min(kb0 + c * STRIDE, kend - 8) + 4 * h;      // clamped: a chunk past the end is loaded, not used
            if constexpr (A_KI) {
                a[c].x = ap[(size_t)k * p.a_ld]; a[c].y = ap[(size_t)(k + 1) * p.a_ld];
                a[c].z = ap[(size_t)(k + 2) * p.a_ld]; a[c].w = ap[(size_t)(k + 3) * p.a_ld];
            } else {
                a[c] = *reinterpret_cast<const float4*>(ap + k);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) b[c][t] = *reinterpret_cast<const float4*>(bp[t] + k);
        }
        __builtin_amdgcn_sched_barrier(0);      // all CH chunks' loads are in flight before the first MFMA waits
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            // a chunk past the end multiplies zeros (no branch: the compiler would sink the loads into it)
            const float m = (kb0 + c * STRIDE < kend) ? 1.f : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].x * m, b[c][t].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].y * m, b[c][t].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].z * m, b[c][t].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].w * m, b[c][t].w, acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += TP) {
        if (t0 > 0) __syncthreads();                   // the previous pass has been read
#pragma unroll
        for (int t = 0; t < TP; ++t)
            if (t0 + t < NT) {
#pragma unroll
                for (int r = 0; r < 16; ++r)           // D row (i) = (r&3) + 8*(r>>2) + 4*h ; D col (j) = lane & 31
                    red[wave][32 * t + l31][(r & 3) + 8 * (r >> 2) + 4 * h] = acc[t0 + t][r];
            }
        __syncthreads();
        for (int e = tid; e < TP * 32 * 32; e += 256) {
            const int il = e & 31, jl = e >> 5;
            const int i = i0 + il, j = 32 * t0 + jl;
            if (i < p.I && j < p.N) {
                float v = (red[0][jl][il] + red[1][jl][il]) + (red[2][jl][il] + red[3][jl][il]);
                float* o = p.out + (size_t)j * p.o_ld + i;
                if (gridDim.y > 1) {
                    if (p.bias && blockIdx.y == 0) v += p.bias[i];
                    atomicAdd(o, v);
                } else {
                    if (p.bias) v += p.bias[i];
                    *o = apply_act(v, p.act, p.slope);
                }
            }
        }
    }
}

// dW[i][j] += sum_n dy[n][i] * x[n][j]: the reduction is only the batch, the output is the whole weight -- bound
// by the read-modify-write of dW.  One wave owns a 32 x (32*TJ) block for the full batch (no split, so plain +=,
// no atomics); both operands are lane-contiguous 128-byte rows; eight k-steps of loads are issued before their MFMAs.
template <int TJ>
__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const float* __restrict__ dy, int dy_ld, const float* __restrict__ x,
                                                           int x_ld, float* __restrict__ dw, int I, int J, int N) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i0 = (blockIdx.y * 4 + wave) * 32, j0 = blockIdx.x * 32 * TJ;
    if (i0 >= I) return;
    const float* ap = dy + min(i0 + l31, I - 1);
    const float* bp[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t) bp[t] = x + min(j0 + 32 * t + l31, J - 1);
    f32x16 acc[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int k = 0; k < N; k += 16) {
        float a[8], b[8][TJ];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = k + 2 * u + h;
            const int kc = min(kk, N - 1);             // clamped; a row past the batch is zeroed below
            a[u] = ap[(size_t)kc * dy_ld];
#pragma unroll
            for (int t = 0; t < TJ; ++t) b[u][t] = bp[t][(size_t)kc * x_ld];
        }
        __builtin_amdgcn_sched_barrier(0);      // eight k-steps of loads in flight before the first MFMA waits
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float av = (k + 2 * u + h) < N ? a[u] : 0.f;
#pragma unroll
            for (int t = 0; t < TJ; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][t], acc[t], 0, 0, 0);
        }
    }
    // read-modify-write of the owned block: all 16 reads of a column tile in flight, then the adds and stores
    // (rows / columns past the edge are clamped for the read and skipped for the write)
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
        const int j = j0 + 32 * t + l31;
        const int jc = min(j, J - 1);
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = dw[(size_t)min(i0 + (r & 3) + 8 * (r >> 2) + 4 * h, I - 1) * J + jc];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (i < I && j < J) dw[(size_t)i * J + j] = old[r] + acc[t][r];
        }
    }
}

// zero / activate a channel-sliced tensor [N, C, P] living in a buffer with ctot channels
__global__ __launch_bounds__(256) void slice_zero_kernel(float* __restrict__ t, long row, long pitch, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / row, r = i - n * row;
        t[n * pitch + r] = 0.f;
    }
}
__global__ __launch_bounds__(256) void slice_act_kernel(float* __restrict__ t, long row, long pitch, long total, int act,
                                                        float slope) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / row, r = i - n * row;
        t[n * pitch + r] = apply_act(t[n * pitch + r], act, slope);
    }
}

// ---------------------------------------------------------------------------------------
// host side
int g_mgvae_cus = 256;
#define g_cus g_mgvae_cus
static bool g_prof = false;
struct ProfEntry { int kind, tile; double flops; hipEvent_t e0, e1; IgemmP p; unsigned gx, gy, gz; };
static std::vector<ProfEntry> g_prof_entries;
static std::mutex g_prof_mu;

static int validate(const MgvaeConvDesc* d) {
    if (!d) return MGVAE_EINVAL;
    if (d->N <= 0 || d->Cx <= 0 || d->Cy <= 0 || d->H <= 0 || d->W <= 0 || d->OH <= 0 || d->OW <= 0) return MGVAE_EINVAL;
    if (d->KH <= 0 || d->KW <= 0 || d->SH <= 0 || d->SW <= 0 || d->PH < 0 || d->PW < 0) return MGVAE_EINVAL;
    if (d->KH * d->KW > 16) return MGVAE_EINVAL;
    if ((d->H + 2 * d->PH - d->KH) / d->SH + 1 != d->OH) return MGVAE_EINVAL;
    if ((d->W + 2 * d->PW - d->KW) / d->SW + 1 != d->OW) return MGVAE_EINVAL;
    if (d->x_coff < 0 || d->x_coff + d->Cx > d->x_ctot) return MGVAE_EINVAL;
    if (d->y_coff < 0 || d->y_coff + d->Cy > d->y_ctot) return MGVAE_EINVAL;
    const long xe = (long)d->N * d->x_ctot * d->H * d->W, ye = (long)d->N * d->y_ctot * d->OH * d->OW;
    // tensors are addressed through 32-bit byte offsets of buffer descriptors: < 2^30 elements each
    if (xe >= (1L << 30) || ye >= (1L << 30) || (long)d->Cx * d->Cy * d->KH * d->KW >= (1L << 30)) return MGVAE_EINVAL;
    return MGVAE_OK;
}

// matrix-operand precision of the tiled conv kernels (process-wide; the thin / skinny paths always run fp32)
static std::atomic<int> g_compute_bf16{0};
extern "C" int mgvae_set_compute_dtype(int dtype) {
    if (dtype != MGVAE_COMPUTE_F32 && dtype != MGVAE_COMPUTE_BF16 && dtype != MGVAE_COMPUTE_F32_BF16X3) return MGVAE_EINVAL;
    g_compute_bf16.store(dtype == MGVAE_COMPUTE_BF16 ? 1 : (dtype == MGVAE_COMPUTE_F32_BF16X3 ? 3 : 0));
    return MGVAE_OK;
}
extern "C" int mgvae_get_compute_dtype(void) {
    const int v = g_compute_bf16.load();
    return v == 1 ? MGVAE_COMPUTE_BF16 : (v == 3 ? MGVAE_COMPUTE_F32_BF16X3 : MGVAE_COMPUTE_F32);
}

static IgemmP make_params(const MgvaeConvDesc* d) {
    IgemmP p{};
    p.N = d->N; p.Cx = d->Cx; p.H = d->H; p.W = d->W; p.Cy = d->Cy; p.OH = d->OH; p.OW = d->OW;
    p.KH = d->KH; p.KW = d->KW; p.SH = d->SH; p.SW = d->SW; p.PH = d->PH; p.PW = d->PW;
    p.x_ctot = d->x_ctot; p.x_coff = d->x_coff; p.y_ctot = d->y_ctot; p.y_coff = d->y_coff;
    p.act = d->act; p.slope = d->slope; p.kchunk = 0; p.ksplit = 1; p.ktab = nullptr; p.wtab = nullptr; p.ktab_stride = 0; p.w_transposed = 0;
    p.mask = nullptr; p.mask_ctot = 0; p.mask_coff = 0; p.mask_act = MGVAE_ACT_NONE; p.mask_slope = 0.f;
    p.bf16 = g_compute_bf16.load(std::memory_order_relaxed);
    static const int xcd = getenv("MGVAE_XCD") ? atoi(getenv("MGVAE_XCD")) : 0;
    p.xcd_remap = xcd;
    p.x_bytes = (unsigned)((size_t)d->N * d->x_ctot * d->H * d->W * 4);
    p.y_bytes = (unsigned)((size_t)d->N * d->y_ctot * d->OH * d->OW * 4);
    p.w_bytes = (unsigned)((size_t)d->Cx * d->Cy * d->KH * d->KW * 4);
    return p;
}

// ---- per-geometry gather tables (device-resident, built once per geometry and cached) -----------
// One int4 per K index: {source offset (channel + tap), tap row offset, tap col offset, weight
// offset}.  The kernels read them with scalar loads.  This is the only memory the library owns:
// a few hundred KB in total, allocated on the first call of a geometry (so run one warm-up step
// before capturing a hipGraph).
struct KtabKey {
    int v[14];
    bool operator<(const KtabKey& o) const { return memcmp(v, o.v, sizeof(v)) < 0; }
};
struct KtabVal { int2* dev; int* wdev; int stride; };
static std::map<KtabKey, KtabVal> g_ktab;
static std::mutex g_ktab_mu;

// entry.y = tap row offset (int8) | tap col offset (int8) << 8 | tap index << 16.  The tap index selects one bit of the
// loader's per-pixel validity mask (<= 31 taps); KT_PAD_TAP marks the padding rows past the end of K (never valid).
static inline int2 kt_entry(int off, int dh, int dw, int tap) { return make_int2(off, (dh & 0xff) | ((dw & 0xff) << 8) | (tap << 16)); }

static int get_ktab(const MgvaeConvDesc* d, int mode, IgemmP& p, int wtrans = 0) {
    KtabKey key{{mode + 16 * wtrans, d->Cx, d->H, d->W, d->Cy, d->OH, d->OW, d->KH, d->KW, d->SH, d->SW, d->PH, d->PW, 0}};
    std::lock_guard<std::mutex> lk(g_ktab_mu);
    auto it = g_ktab.find(key);
    if (it == g_ktab.end()) {
        std::vector<int2> host;
        std::vector<int> whost;
        int str = 0;
        const int KK = d->KH * d->KW;
        if (KK > KT_MAX_TAPS || d->KH > 127 || d->KW > 127) return MGVAE_EINVAL;   // one validity bit per tap, int8 tap offsets
        if (mode == MODE_BWD_WEIGHT) {
            const int P = d->OH * d->OW;
            str = P + 64;
            host.assign(str, make_int2(0, -1));
            for (int oh = 0; oh < d->OH; ++oh)
                for (int ow = 0; ow < d->OW; ++ow) {
                    const int r0 = oh * d->SH, c0 = ow * d->SW;
                    unsigned m = 1u << KT_PAD_TAP;
                    for (int t = 0; t < KK; ++t) {
                        const int ih = r0 + t / d->KW - d->PH, iw = c0 + t % d->KW - d->PW;
                        if (ih < 0 || ih >= d->H || iw < 0 || iw >= d->W) m |= 1u << t;
                    }
                    host[(size_t)oh * d->OW + ow] = make_int2(r0 * d->W + c0, (int)m);
                }
        } else if (mode == MODE_FWD) {
            str = d->Cx * KK + 96;   // padding: the loaders prefetch the table two K tiles ahead
            host.assign(str, kt_entry(0, 0, 0, KT_PAD_TAP));
            for (int c = 0; c < d->Cx; ++c)
                for (int t = 0; t < KK; ++t) {
                    const int kh = t / d->KW, kw = t % d->KW;
                    host[(size_t)c * KK + t] = kt_entry(c * d->H * d->W + kh * d->W + kw, kh, kw, t);
                }
        } else {
            const int Z = d->SH * d->SW;
            int maxT = 1;
            for (int ph = 0; ph < Z; ++ph) {
                const int rh = ph / d->SW, rw = ph % d->SW;
                const int kh0 = (rh + d->PH) % d->SH, kw0 = (rw + d->PW) % d->SW;
                const int nkh = kh0 < d->KH ? (d->KH - kh0 + d->SH - 1) / d->SH : 0;
                const int nkw = kw0 < d->KW ? (d->KW - kw0 + d->SW - 1) / d->SW : 0;
                if (nkh * nkw > maxT) maxT = nkh * nkw;
            }
            str = d->Cy * maxT + 96;
            host.assign((size_t)Z * str, kt_entry(0, 0, 0, KT_PAD_TAP));
            whost.assign((size_t)Z * str, 0);
            for (int ph = 0; ph < Z; ++ph) {
                const int rh = ph / d->SW, rw = ph % d->SW;
                const int kh0 = (rh + d->PH) % d->SH, kw0 = (rw + d->PW) % d->SW;
                const int nkh = kh0 < d->KH ? (d->KH - kh0 + d->SH - 1) / d->SH : 0;
                const int nkw = kw0 < d->KW ? (d->KW - kw0 + d->SW - 1) / d->SW : 0;
                const int qh = (rh + d->PH - kh0) / d->SH, qw = (rw + d->PW - kw0) / d->SW;
                const int T = nkh * nkw;
                for (int c = 0; c < d->Cy; ++c)
                    for (int t = 0; t < T; ++t) {
                        const int jh = t / nkw, jw = t % nkw;
                        const size_t idx = (size_t)ph * str + (size_t)c * T + t;
                        host[idx] = kt_entry(c * d->OH * d->OW + (qh - jh) * d->OW + (qw - jw), qh - jh, qw - jw, t);
                        const int tapw = (kh0 + d->SH * jh) * d->KW + kw0 + d->SW * jw;
                        whost[idx] = wtrans ? (c * KK + tapw) * d->Cx : c * d->Cx * KK + tapw;
                    }
            }
        }
        KtabVal v{nullptr, nullptr, str};
        if (hipMalloc(&v.dev, host.size() * sizeof(int2)) != hipSuccess) return MGVAE_ELAUNCH;
        if (hipMemcpy(v.dev, host.data(), host.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) return MGVAE_ELAUNCH;
        if (!whost.empty()) {
            if (hipMalloc(&v.wdev, whost.size() * sizeof(int)) != hipSuccess) return MGVAE_ELAUNCH;
            if (hipMemcpy(v.wdev, whost.data(), whost.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return MGVAE_ELAUNCH;
        }
        it = g_ktab.emplace(key, v).first;
    }
    p.ktab = it->second.dev; p.wtab = it->second.wdev; p.ktab_stride = it->second.stride;
    return MGVAE_OK;
}

// tile ids: 0 = 128x128, 1 = 64(i)x128(j), 2 = 128(i)x64(j), 3 = 64x64.
// Under-filled launches first split K (keeps the efficient big tiles), then shrink the tile.
static int pick_tile(long Itot, long Jtot, int Z, long Kmin, int* ksplit, bool allow_split = true) {
    int ti = Itot > 64 ? 2 : 1, tj = Jtot > 64 ? 2 : 1;
    auto wgs = [&](int a, int b) { return (long)cdiv(Itot, 64 * a) * cdiv(Jtot, 64 * b) * Z; };
    const long want = (long)g_cus * 2;
    static const int env_nosplit = getenv("MGVAE_NOSPLIT") ? atoi(getenv("MGVAE_NOSPLIT")) : 0;
    if (env_nosplit == 1) allow_split = false;
    const long maxsplit = (allow_split && Kmin / (BK * 8) > 1) ? Kmin / (BK * 8) : 1;
    if (wgs(ti, tj) * maxsplit < want && tj == 2) tj = 1;
    if (wgs(ti, tj) * maxsplit < want && ti == 2) ti = 1;
    long sp = cdiv(want, wgs(ti, tj));
    if (sp > maxsplit) sp = maxsplit;
    if (sp > 32) sp = 32;
    *ksplit = sp < 1 ? 1 : (int)sp;
    return (ti == 2 ? 0 : 1) + (tj == 2 ? 0 : 2);
}
static inline int tile_it(int tile) { return (tile & 1) ? 64 : 128; }
static inline int tile_jt(int tile) { return (tile & 2) ? 64 : 128; }

static void zero_slice(float* t, int N, int C, long P, int ctot, hipStream_t s) {
    const long row = (long)C * P, total = row * N;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(slice_zero_kernel, dim3(blocks), dim3(256), 0, s, t, row, (long)ctot * P, total);
}
static void act_slice(float* t, int N, int C, long P, int ctot, int act, float slope, hipStream_t s) {
    const long row = (long)C * P, total = row * N;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(slice_act_kernel, dim3(blocks), dim3(256), 0, s, t, row, (long)ctot * P, total, act, slope);
}

template <int MODE>
static int launch(int tile, dim3 grid, const IgemmP& p, hipStream_t s) {
    if (p.bf16 == 1) {
        switch (tile) {
            case 0: hipLaunchKernelGGL((igemm_bf16_kernel<MODE, 2, 2, 1>), grid, dim3(256), 0, s, p); break;
            case 1: hipLaunchKernelGGL((igemm_bf16_kernel<MODE, 1, 2, 1>), grid, dim3(256), 0, s, p); break;
            case 2: hipLaunchKernelGGL((igemm_bf16_kernel<MODE, 2, 1, 1>), grid, dim3(256), 0, s, p); break;
            default: hipLaunchKernelGGL((igemm_bf16_kernel<MODE, 1, 1, 1>), grid, dim3(256), 0, s, p); break;
        }
        MGVAE_CHECK_LAUNCH();
        return MGVAE_OK;
    }
    if (p.bf16 == 3) {      // three-term split: the 128x128 tile would need 120 KB of LDS -> exec_* map it to 128x64
        switch (tile) {
            case 2: hipLaunchKernelGGL((igemm_bf16_kernel<MODE, 2, 1, 3>), grid, dim3(256), 0, s, p); break;
            case 1: hipLaunchKernelGGL((igemm_bf16_kernel<MODE, 1, 2, 3>), grid, dim3(256), 0, s, p); break;
            default: hipLaunchKernelGGL((igemm_bf16_kernel<MODE, 1, 1, 3>), grid, dim3(256), 0, s, p); break;
        }
        MGVAE_CHECK_LAUNCH();
        return MGVAE_OK;
    }
    switch (tile) {
        case 0: hipLaunchKernelGGL((igemm_kernel<MODE, 2, 2>), grid, dim3(256), 0, s, p); break;
        case 1: hipLaunchKernelGGL((igemm_kernel<MODE, 1, 2>), grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((igemm_kernel<MODE, 2, 1>), grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL((igemm_kernel<MODE, 1, 1>), grid, dim3(256), 0, s, p); break;
    }
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

static thread_local bool t_no_prof = false;   // set while autotuning so trial launches are not recorded

template <int MODE>
static int run(int tile, dim3 grid, const IgemmP& p, hipStream_t s, double flops) {
    if (!g_prof || t_no_prof) return launch<MODE>(tile, grid, p, s);
    ProfEntry pe{MODE, tile, flops, nullptr, nullptr, p, grid.x, grid.y, grid.z};
    if (hipEventCreate(&pe.e0) != hipSuccess || hipEventCreate(&pe.e1) != hipSuccess) return MGVAE_ELAUNCH;
    hipEventRecord(pe.e0, s);
    int rc = launch<MODE>(tile, grid, p, s);
    hipEventRecord(pe.e1, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_entries.push_back(pe);
    return rc;
}

thread_local int g_prof_note[16] = {0};
// profiler hooks shared with conv_direct.hip (kinds 3 = direct fwd, 4 = direct bwd_data)
extern "C" int mgvae_prof_record_begin(int kind, int tile, double flops, void* stream, void** token) {
    *token = nullptr;
    if (!g_prof) return MGVAE_OK;
    ProfEntry* pe = new ProfEntry{kind, tile, flops, nullptr, nullptr, IgemmP{}, 0, 0, 0};
    IgemmP& q = pe->p;
    q.N = g_prof_note[0]; q.Cx = g_prof_note[1]; q.H = g_prof_note[2]; q.W = g_prof_note[3]; q.Cy = g_prof_note[4];
    q.OH = g_prof_note[5]; q.OW = g_prof_note[6]; q.KH = g_prof_note[7]; q.KW = g_prof_note[8]; q.SH = g_prof_note[9];
    q.SW = g_prof_note[10]; pe->gx = g_prof_note[11]; pe->gy = g_prof_note[12]; pe->gz = g_prof_note[13];
    if (hipEventCreate(&pe->e0) != hipSuccess || hipEventCreate(&pe->e1) != hipSuccess) { delete pe; return MGVAE_ELAUNCH; }
    hipEventRecord(pe->e0, as_stream(stream));
    *token = pe;
    return MGVAE_OK;
}
extern "C" int mgvae_prof_record_end(void* token, void* stream) {
    if (!token) return MGVAE_OK;
    ProfEntry* pe = static_cast<ProfEntry*>(token);
    hipEventRecord(pe->e1, as_stream(stream));
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_entries.push_back(*pe);
    delete pe;
    return MGVAE_OK;
}

// ---- launch helpers: one function per mode that runs a given (tile, split) configuration -----------

static int exec_fwd(const MgvaeConvDesc* d, IgemmP p, int tile, int ksplit, hipStream_t s) {
    if (p.bf16 == 3 && tile == 0) tile = 2;        // no 128x128 tile in the three-term mode (LDS)
    const long I = d->Cy, J = (long)d->N * d->OH * d->OW;
    const long P = (long)d->OH * d->OW;
    float* y = p.out + (size_t)d->y_coff * P;          // first channel of the written slice
    p.ksplit = ksplit;
    if (ksplit > 1) { p.act = MGVAE_ACT_NONE; zero_slice(y, d->N, d->Cy, P, d->y_ctot, s); }
    dim3 grid(cdiv(J, tile_jt(tile)), cdiv(I, tile_it(tile)), ksplit);
    int rc = run<MODE_FWD>(tile, grid, p, s, 2.0 * I * J * d->Cx * d->KH * d->KW);
    if (rc == MGVAE_OK && ksplit > 1 && d->act != MGVAE_ACT_NONE) act_slice(y, d->N, d->Cy, P, d->y_ctot, d->act, d->slope, s);
    return rc;
}

static int exec_bwd_data(const MgvaeConvDesc* d, IgemmP p, int tile, int ksplit, hipStream_t s) {
    if (p.bf16 == 3 && tile == 0) tile = 2;
    const int Z = d->SH * d->SW;
    const long I = d->Cx, J = (long)d->N * cdiv(d->H, d->SH) * cdiv(d->W, d->SW);   // largest phase
    const long HW = (long)d->H * d->W;
    float* x = p.out + (size_t)d->x_coff * HW;         // first channel of the written slice
    p.ksplit = ksplit;
    if (ksplit > 1) { p.act = MGVAE_ACT_NONE; zero_slice(x, d->N, d->Cx, HW, d->x_ctot, s); }
    dim3 grid(cdiv(J, tile_jt(tile)), cdiv(I, tile_it(tile)), Z * ksplit);
    // algorithmic flops: every (output pixel, tap) pair that exists = same as the forward conv
    const double flops = 2.0 * d->Cy * d->Cx * d->KH * d->KW * (double)d->N * d->OH * d->OW;
    int rc = run<MODE_BWD_DATA>(tile, grid, p, s, flops);
    if (rc == MGVAE_OK && ksplit > 1 && d->act != MGVAE_ACT_NONE) act_slice(x, d->N, d->Cx, HW, d->x_ctot, d->act, d->slope, s);
    return rc;
}

static int exec_bwd_weight(const MgvaeConvDesc* d, IgemmP p, int tile, long splits, hipStream_t s) {
    if (p.bf16 == 3 && tile == 0) tile = 2;
    const long I = d->Cy, J = (long)d->Cx * d->KH * d->KW, M = (long)d->N * d->OH * d->OW;
    if (splits < 1) splits = 1;
    long kchunk = cdiv(M, splits);
    kchunk = (kchunk + 31) / 32 * 32;
    splits = cdiv(M, kchunk);
    p.kchunk = (int)kchunk;
    dim3 grid(cdiv(J, tile_jt(tile)), cdiv(I, tile_it(tile)), (unsigned)splits);
    return run<MODE_BWD_WEIGHT>(tile, grid, p, s, 2.0 * I * J * M);
}

// ---- autotuner ("cudnn.benchmark" of the reference, agent/barGen2.py:27): the first time a
// (mode, geometry, batch) is seen outside stream capture, every sensible (tile, split-K) pair is
// timed with hipEvents on the caller's data and the fastest is cached for the life of the process.
struct Choice { int tile; int split; };
static std::map<KtabKey, Choice> g_choice;
static std::mutex g_choice_mu;

// MGVAE_AUTOTUNE_FILE=<path>: decisions are appended to / preloaded from a text file (one line per key), so a
// restarted job -- or the other ranks of a node -- skips the trial launches.
static const char* choice_file() {
    static const char* f = getenv("MGVAE_AUTOTUNE_FILE");
    return (f && *f) ? f : nullptr;
}
static void choice_load_locked() {
    static bool loaded = false;
    if (loaded) return;
    loaded = true;
    const char* f = choice_file();
    if (!f) return;
    FILE* fp = fopen(f, "r");
    if (!fp) return;
    KtabKey k; Choice c;
    for (;;) {
        int n = 0;
        for (int i = 0; i < 14; ++i) n += fscanf(fp, "%d", &k.v[i]);
        n += fscanf(fp, "%d %d", &c.tile, &c.split);
        if (n != 16) break;
        if (c.tile >= 0 && c.tile < 16 && c.split >= 1 && c.split <= 4096) g_choice[k] = c;     // tile / 4: loop form of the x3 family (3 = halo)
    }
    fclose(fp);
}
static bool choice_lookup(const KtabKey& key, Choice* c) {
    std::lock_guard<std::mutex> lk(g_choice_mu);
    choice_load_locked();
    auto it = g_choice.find(key);
    if (it == g_choice.end()) return false;
    *c = it->second;
    return true;
}
static void choice_store(const KtabKey& key, const Choice& c) {
    std::lock_guard<std::mutex> lk(g_choice_mu);
    g_choice[key] = c;
    if (const char* f = choice_file()) {
        if (FILE* fp = fopen(f, "a")) {
            char line[256]; int o = 0;
            for (int i = 0; i < 14; ++i) o += snprintf(line + o, sizeof(line) - o, "%d ", key.v[i]);
            snprintf(line + o, sizeof(line) - o, "%d %d\n", c.tile, c.split);
            fputs(line, fp);     // one write per line: concurrent ranks interleave whole lines only
            fclose(fp);
        }
    }
}

static bool autotune_on() {
    static const int v = getenv("MGVAE_AUTOTUNE") ? atoi(getenv("MGVAE_AUTOTUNE")) : 1;
    return v != 0;
}
static bool is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}
static KtabKey choice_key(const MgvaeConvDesc* d, int mode, int wtrans) {
    return KtabKey{{100 + mode + 16 * wtrans + 1000 * g_compute_bf16.load(std::memory_order_relaxed), d->Cx, d->H, d->W, d->Cy, d->OH, d->OW, d->KH, d->KW, d->SH, d->SW, d->PH, d->PW, d->N}};
}

template <class Exec>
static Choice tune(const std::vector<Choice>& cands, Exec&& exec, hipStream_t s) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    Choice best = cands[0];
    float best_ms = 1e30f;
    t_no_prof = true;
    // candidates are timed alone on the chip: work queued on other streams (side-stream weight gradients, the other
    // encoder trunk) would otherwise be charged to whichever candidate happens to run beside it
    hipDeviceSynchronize();
    for (const Choice& c : cands) {
        if (exec(c) != MGVAE_OK) continue;          // warm-up (also uploads nothing new: tables exist)
        for (int rep = 0; rep < 2; ++rep) {         // best of two single launches
            hipEventRecord(e0, s);
            exec(c);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best_ms) { best_ms = ms; best = c; }
        }
    }
    t_no_prof = false;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}

static std::vector<Choice> gemm_candidates(long I, long J, int Z, long Kmin) {
    std::vector<Choice> v;
    const long maxsplit = Kmin / (BK * 8) > 1 ? Kmin / (BK * 8) : 1;
    for (int tile = 0; tile < 4; ++tile) {
        if (!(tile & 1) && I <= 64) continue;       // 128-row tiles need more than 64 rows
        if (!(tile & 2) && J <= 64) continue;
        const long wg = (long)cdiv(I, tile_it(tile)) * cdiv(J, tile_jt(tile)) * Z;
        for (int sp = 1; sp <= 32; sp *= 2) {
            if (sp > maxsplit) break;
            if (sp > 1 && wg * sp > (long)g_cus * 12) break;        // already far more workgroups than needed
            if (wg * sp * 8 < g_cus && sp * 2 <= maxsplit) continue;  // hopelessly under-filled
            v.push_back(Choice{tile, sp});
        }
    }
    if (v.empty()) v.push_back(Choice{3, 1});
    return v;
}

// ---- skinny (Linear, batch <= 128) dispatch --------------------------------------------------------------
static bool skinny_enabled() {
    static const int v = getenv("MGVAE_SKINNY") ? atoi(getenv("MGVAE_SKINNY")) : 1;
    return v != 0;
}
static bool is_linear(const MgvaeConvDesc* d) {
    return d->H == 1 && d->W == 1 && d->OH == 1 && d->OW == 1 && d->KH == 1 && d->KW == 1 && d->N <= 128;
}
static void skinny_note(const MgvaeConvDesc* d, unsigned gx, unsigned gy) {
    g_prof_note[0] = d->N; g_prof_note[1] = d->Cx; g_prof_note[2] = 1; g_prof_note[3] = 1; g_prof_note[4] = d->Cy;
    g_prof_note[5] = 1; g_prof_note[6] = 1; g_prof_note[7] = 1; g_prof_note[8] = 1; g_prof_note[9] = 1; g_prof_note[10] = 1;
    g_prof_note[11] = gx; g_prof_note[12] = gy; g_prof_note[13] = 1;
}
// out[n][i] = act(bias[i] + sum_k A(i,k) * B[n][k]).  Returns false when the shape / alignment does not qualify.
static bool skinny_gemm(const MgvaeConvDesc* d, int mode, const float* A, bool a_ki, int I, int K, const float* B, int b_ld,
                        const float* bias, float* out, int o_ld, hipStream_t s, void* stream) {
    if (!skinny_enabled() || (K & 7) || (b_ld & 3) || ((uintptr_t)B & 15) || (!a_ki && ((uintptr_t)A & 15))) return false;
    SkinnyP p{A, B, bias, out, I, d->N, K, a_ki ? I : K, b_ld, o_ld, d->act, 0, d->slope};
    const int gx = cdiv(I, 32);
    int ks = 1;                                            // split K only while the chip is under-filled and K is deep
    while (gx * ks < g_cus && K / (ks * 2) >= 512) ks *= 2;  // a wave then walks >= 128 of K: >= 4 rounds of loads
    p.kchunk = cdiv(cdiv(K, ks), 32) * 32;
    ks = cdiv(K, p.kchunk);
    const dim3 grid(gx, ks);
    const int nt = cdiv(d->N, 32);
    if (ks > 1) zero_slice(out, d->N, I, 1, o_ld, s);
    void* tok = nullptr;
    skinny_note(d, grid.x, grid.y);
    mgvae_prof_record_begin(mode, 5, 2.0 * I * (double)K * d->N, stream, &tok);
#define MGVAE_SK(NT)                                                                                         \
    if (a_ki) hipLaunchKernelGGL((skinny_gemm_kernel<NT, true>), grid, dim3(256), 0, s, p);                 \
    else hipLaunchKernelGGL((skinny_gemm_kernel<NT, false>), grid, dim3(256), 0, s, p)
    switch (nt) {
        case 1: MGVAE_SK(1); break;
        case 2: MGVAE_SK(2); break;
        case 3: MGVAE_SK(3); break;
        default: MGVAE_SK(4); break;
    }
#undef MGVAE_SK
    mgvae_prof_record_end(tok, stream);
    if (ks > 1 && d->act != MGVAE_ACT_NONE) act_slice(out, d->N, I, 1, o_ld, d->act, d->slope, s);
    return true;
}

static int check_mask(const MgvaeActMask* m, int C) {
    if (!m) return MGVAE_OK;
    if (!m->src || m->coff < 0 || m->coff + C > m->ctot || m->act < MGVAE_ACT_NONE || m->act > MGVAE_ACT_SIGMOID) return MGVAE_EINVAL;
    return MGVAE_OK;
}
static void set_mask(IgemmP& p, const MgvaeActMask* m) {
    if (!m) return;
    p.mask = m->src; p.mask_ctot = m->ctot; p.mask_coff = m->coff; p.mask_act = m->act; p.mask_slope = m->slope;
}

static int fwd_impl(const MgvaeConvDesc* d, const float* x, const float* w, const float* bias, float* y, void* stream,
                    const MgvaeActMask* m) {
    int rc = validate(d);
    if (rc) return rc;
    if (!x || !w || !y || check_mask(m, d ? d->Cy : 0)) return MGVAE_EINVAL;
    IgemmP p = make_params(d);
    p.X = x; p.Wt = w; p.bias = bias; p.out = y; p.Y = nullptr;     // kernels add y_coff themselves
    set_mask(p, m);
    hipStream_t s = as_stream(stream);
    const long I = d->Cy, J = (long)d->N * d->OH * d->OW, K = (long)d->Cx * d->KH * d->KW;
    if (!m && K <= 16 && d->N <= 65535 && I * K <= THIN_W_MAX && I <= THIN_W_MAX / 4) {
        const dim3 grid(cdiv((long)d->OH * d->OW, 256), d->N);
        void* tok = nullptr;
        g_prof_note[0] = d->N; g_prof_note[1] = d->Cx; g_prof_note[2] = d->H; g_prof_note[3] = d->W; g_prof_note[4] = d->Cy;
        g_prof_note[5] = d->OH; g_prof_note[6] = d->OW; g_prof_note[7] = d->KH; g_prof_note[8] = d->KW; g_prof_note[9] = d->SH;
        g_prof_note[10] = d->SW; g_prof_note[11] = grid.x; g_prof_note[12] = grid.y; g_prof_note[13] = 1;
        mgvae_prof_record_begin(MODE_FWD, 6, 2.0 * I * J * K, stream, &tok);
#define MGVAE_THIN_FWD(JM)                                                                                              \
    switch (d->act) {                                                                                                   \
        case MGVAE_ACT_RELU: hipLaunchKernelGGL((thin_fwd_kernel<JM, MGVAE_ACT_RELU>), grid, dim3(256), 0, s, p, (int)K); break;       \
        case MGVAE_ACT_LEAKY: hipLaunchKernelGGL((thin_fwd_kernel<JM, MGVAE_ACT_LEAKY>), grid, dim3(256), 0, s, p, (int)K); break;     \
        case MGVAE_ACT_SIGMOID: hipLaunchKernelGGL((thin_fwd_kernel<JM, MGVAE_ACT_SIGMOID>), grid, dim3(256), 0, s, p, (int)K); break; \
        default: hipLaunchKernelGGL((thin_fwd_kernel<JM, MGVAE_ACT_NONE>), grid, dim3(256), 0, s, p, (int)K); break;                   \
    }
        if (K <= 4) { MGVAE_THIN_FWD(4) } else { MGVAE_THIN_FWD(16) }
#undef MGVAE_THIN_FWD
        mgvae_prof_record_end(tok, stream);
        MGVAE_CHECK_LAUNCH();
        return MGVAE_OK;
    }
    if (!m && is_linear(d) && skinny_gemm(d, MODE_FWD, w, false, d->Cy, d->Cx, x + d->x_coff, d->x_ctot, bias, y + d->y_coff,
                                    d->y_ctot, s, stream)) {
        MGVAE_CHECK_LAUNCH();
        return MGVAE_OK;
    }
    rc = get_ktab(d, MODE_FWD, p);
    if (rc) return rc;
    if (autotune_on()) {
        const KtabKey key = choice_key(d, MODE_FWD, 0);
        Choice c;
        bool have = choice_lookup(key, &c);
        if (!have && !is_capturing(s)) {       // no trial launches under stream capture; a cached decision is still used
            c = tune(gemm_candidates(I, J, 1, K), [&](const Choice& q) { return exec_fwd(d, p, q.tile, q.split, s); }, s);
            choice_store(key, c);
            have = true;
        }
        if (have) return exec_fwd(d, p, c.tile, c.split, s);
    }
    int ksplit = 1;
    const int tile = pick_tile(I, J, 1, K, &ksplit);
    return exec_fwd(d, p, tile, ksplit, s);
}

extern "C" int mgvae_conv2d_fwd(const MgvaeConvDesc* d, const float* x, const float* w, const float* bias,
                                float* y, void* stream) {
    return fwd_impl(d, x, w, bias, y, stream, nullptr);
}
extern "C" int mgvae_conv2d_fwd_masked(const MgvaeConvDesc* d, const float* x, const float* w, const float* bias,
                                       float* y, const MgvaeActMask* mask, void* stream) {
    return fwd_impl(d, x, w, bias, y, stream, mask);
}

static int bwd_data_impl(const MgvaeConvDesc* d, const float* y, const float* w, const float* bias, float* x,
                         void* stream, int wtrans, const MgvaeActMask* m = nullptr) {
    int rc = validate(d);
    if (rc) return rc;
    if (!x || !w || !y || check_mask(m, d ? d->Cx : 0)) return MGVAE_EINVAL;
    if (!m && is_linear(d) && skinny_gemm(d, MODE_BWD_DATA, w, true, d->Cx, d->Cy, y + d->y_coff, d->y_ctot, bias, x + d->x_coff,
                                    d->x_ctot, as_stream(stream), stream)) {      // KH*KW == 1: w_t has w's layout
        MGVAE_CHECK_LAUNCH();
        return MGVAE_OK;
    }
    IgemmP p = make_params(d);
    p.Y = y; p.Wt = w; p.bias = bias; p.out = x; p.X = nullptr;     // kernels add x_coff themselves
    set_mask(p, m);
    rc = get_ktab(d, MODE_BWD_DATA, p, wtrans);
    p.w_transposed = wtrans;
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    const int Z = d->SH * d->SW;
    const long I = d->Cx;
    const long J = (long)d->N * cdiv(d->H, d->SH) * cdiv(d->W, d->SW);   // largest phase
    // smallest per-phase K: Cy * (fewest taps a phase has, at least 1)
    const long tmin = (long)(d->KH / d->SH > 0 ? d->KH / d->SH : 1) * (d->KW / d->SW > 0 ? d->KW / d->SW : 1);
    if (autotune_on()) {
        const KtabKey key = choice_key(d, MODE_BWD_DATA, wtrans);
        Choice c;
        bool have = choice_lookup(key, &c);
        if (!have && !is_capturing(s)) {
            c = tune(gemm_candidates(I, J, Z, (long)d->Cy * tmin), [&](const Choice& q) { return exec_bwd_data(d, p, q.tile, q.split, s); }, s);
            choice_store(key, c);
            have = true;
        }
        if (have) return exec_bwd_data(d, p, c.tile, c.split, s);
    }
    int ksplit = 1;
    const int tile = pick_tile(I, J, Z, (long)d->Cy * tmin, &ksplit);
    return exec_bwd_data(d, p, tile, ksplit, s);
}

extern "C" int mgvae_conv2d_bwd_data(const MgvaeConvDesc* d, const float* y, const float* w, const float* bias,
                                     float* x, void* stream) {
    return bwd_data_impl(d, y, w, bias, x, stream, 0);
}

// X = conv_transpose(Y, Wt) * act'(mask): w_transposed selects the weight layout of the two entry points above/below
extern "C" int mgvae_conv2d_bwd_data_masked(const MgvaeConvDesc* d, const float* y, const float* w, int w_transposed,
                                            const float* bias, float* x, const MgvaeActMask* mask, void* stream) {
    return bwd_data_impl(d, y, w, bias, x, stream, w_transposed ? 1 : 0, mask);
}

// same with w_t = mgvae_weight_transpose(w): [Cy][KH*KW][Cx], so the weight operand loads are
// contiguous along the lanes (the [Cy][Cx][KH][KW] layout strides them by KH*KW floats)
extern "C" int mgvae_conv2d_bwd_data_tw(const MgvaeConvDesc* d, const float* y, const float* w_t, const float* bias,
                                        float* x, void* stream) {
    return bwd_data_impl(d, y, w_t, bias, x, stream, 1);
}

__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                               int Cy, int Cx, int KK) {
    // out[cy][kk][cx] = in[cy][cx][kk]; a workgroup moves a [64 cx][KK] slab of one cy through LDS
    extern __shared__ float tile[];
    const int cy = blockIdx.y, cx0 = blockIdx.x * 64;
    const int n = min(64, Cx - cx0);
    const float* src = w + ((size_t)cy * Cx + cx0) * KK;
    for (int e = threadIdx.x; e < n * KK; e += 256) tile[e] = src[e];      // contiguous read
    __syncthreads();
    for (int e = threadIdx.x; e < n * KK; e += 256) {
        const int kk = e / n, c = e - kk * n;
        wt[((size_t)cy * KK + kk) * Cx + cx0 + c] = tile[c * KK + kk];     // contiguous write per kk row
    }
}

extern "C" int mgvae_weight_transpose(const float* w, float* w_t, int Cy, int Cx, int KK, void* stream) {
    if (!w || !w_t || Cy <= 0 || Cx <= 0 || KK <= 0 || KK > 64) return MGVAE_EINVAL;
    hipLaunchKernelGGL(weight_transpose_kernel, dim3(cdiv(Cx, 64), Cy), dim3(256), 64 * KK * sizeof(float), as_stream(stream),
                       w, w_t, Cy, Cx, KK);
    MGVAE_CHECK_LAUNCH();
    return MGVAE_OK;
}

// Weight gradient of a "thin" convolution (Cx*KH*KW <= 16: the C=1 stems, K1 of SURVEY 2.2): a GEMM
// with <= 16 columns would waste the MFMA, and it is purely HBM-bound (dY is read once).  Pixels sit
// on the lanes (coalesced dY reads); a wave takes 64-pixel chunks of one sample, gathers the <= 16
// input taps of its pixel ONCE and reuses them for a block of CYB output channels, keeping the
// CYB x JMAX partial sums in registers ACROSS chunks (persistent waves: every workgroup ends with
// CYB*JMAX atomics onto the same few cache lines, so the grid is capped at about two workgroups per
// CU).  DPP-reduce over the lanes, combine the four waves through LDS.
template <int CYB, int JMAX>
__global__ __launch_bounds__(256) void thin_bwd_weight_kernel(const IgemmP p, int J) {
    __shared__ float red[4][CYB * JMAX];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int P = p.OH * p.OW, HW = p.H * p.W, KK = p.KH * p.KW;
    const int cy0 = blockIdx.y * CYB;
    const int cps = (P + 63) >> 6;                     // chunks per sample
    const int total = p.N * cps;
    int joff[JMAX], jdh[JMAX], jdw[JMAX];             // wave-uniform tap constants
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int jj = j < J ? j : 0;
        const int cx = jj / KK, t = jj - cx * KK;
        const int kh = t / p.KW, kw = t - kh * p.KW;
        jdh[j] = j < J ? kh : -(1 << 28); jdw[j] = kw; joff[j] = cx * HW + kh * p.W + kw;
    }
    float acc[CYB][JMAX];
#pragma unroll
    for (int c = 0; c < CYB; ++c)
#pragma unroll
        for (int j = 0; j < JMAX; ++j) acc[c][j] = 0.f;
    for (int ch = blockIdx.x * 4 + wave; ch < total; ch += gridDim.x * 4) {
        const int n = ch / cps;                        // wave-uniform
        const int i = (ch - n * cps) * 64 + lane;
        const bool pin = i < P;
        const int ii = pin ? i : 0;
        const int oh = ii / p.OW, ow = ii - oh * p.OW;
        const int r0 = oh * p.SH - p.PH, c0 = ow * p.SW - p.PW;
        const int xo = r0 * p.W + c0;
        const float* xb = p.X + ((size_t)n * p.x_ctot + p.x_coff) * HW;
        const float* yb = p.Y + ((size_t)n * p.y_ctot + p.y_coff + cy0) * P;
        float xv[JMAX];
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const bool ok = pin & ((unsigned)(r0 + jdh[j]) < (unsigned)p.H) & ((unsigned)(c0 + jdw[j]) < (unsigned)p.W);
            const float v = xb[ok ? xo + joff[j] : 0];
            xv[j] = ok ? v : 0.f;
        }
        // no branch on the channel bound: a clamped (valid) address keeps all CYB loads in flight together; the
        // sums of channels >= Cy are garbage that is never written
        float g[CYB];
#pragma unroll
        for (int c = 0; c < CYB; ++c) g[c] = yb[(size_t)(cy0 + c < p.Cy ? c : 0) * P + ii];
#pragma unroll
        for (int c = 0; c < CYB; ++c)
#pragma unroll
            for (int j = 0; j < JMAX; ++j) acc[c][j] += g[c] * xv[j];   // xv is 0 outside the image / sample
    }
#pragma unroll
    for (int c = 0; c < CYB; ++c)
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const float sum = wave_sum_lane63(acc[c][j]);
            if (lane == 63) red[wave][c * JMAX + j] = sum;
        }
    __syncthreads();
    for (int t = threadIdx.x; t < CYB * JMAX; t += 256) {
        const int c = t / JMAX, j = t - c * JMAX;
        if (j < J && cy0 + c < p.Cy)
            atomicAdd(&p.out[(size_t)(cy0 + c) * J + j], (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]));
    }
}

extern "C" int mgvae_conv2d_bwd_weight(const MgvaeConvDesc* d, const float* x, const float* y, float* dw,
                                       void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    if (!x || !dw || !y) return MGVAE_EINVAL;
    if ((long)d->Cx * d->KH * d->KW <= 16) {
        IgemmP q = make_params(d);
        q.X = x; q.Y = y; q.out = dw;
        const int J = d->Cx * d->KH * d->KW;
        void* tok = nullptr;
        const double fl = 2.0 * d->Cy * J * (double)d->N * d->OH * d->OW;
        g_prof_note[0] = d->N; g_prof_note[1] = d->Cx; g_prof_note[2] = d->H; g_prof_note[3] = d->W; g_prof_note[4] = d->Cy;
        g_prof_note[5] = d->OH; g_prof_note[6] = d->OW; g_prof_note[7] = d->KH; g_prof_note[8] = d->KW; g_prof_note[9] = d->SH;
        const int P = d->OH * d->OW;
        const long chunks = (long)d->N * cdiv(P, 64);
        const int cyb = J <= 4 ? 32 : 8;
        const int ycount = cdiv(d->Cy, cyb);
        long gx = cdiv(chunks, 4);
        const long cap = 2 * g_cus / ycount > 32 ? 2 * g_cus / ycount : 32;   // ~two workgroups per CU (see kernel)
        if (gx > cap) gx = cap;
        const dim3 grid((unsigned)gx, ycount, 1);
        g_prof_note[10] = d->SW; g_prof_note[11] = grid.x; g_prof_note[12] = grid.y; g_prof_note[13] = grid.z;
        mgvae_prof_record_begin(MODE_BWD_WEIGHT, 6, fl, stream, &tok);
        if (J <= 4) hipLaunchKernelGGL((thin_bwd_weight_kernel<32, 4>), grid, dim3(256), 0, as_stream(stream), q, J);
        else hipLaunchKernelGGL((thin_bwd_weight_kernel<8, 16>), grid, dim3(256), 0, as_stream(stream), q, J);
        mgvae_prof_record_end(tok, stream);
        MGVAE_CHECK_LAUNCH();
        return MGVAE_OK;
    }
    if (is_linear(d) && skinny_enabled()) {
        const dim3 grid(cdiv(d->Cx, 128), cdiv(d->Cy, 128));
        void* tok = nullptr;
        skinny_note(d, grid.x, grid.y);
        mgvae_prof_record_begin(MODE_BWD_WEIGHT, 5, 2.0 * d->Cy * (double)d->Cx * d->N, stream, &tok);
        hipLaunchKernelGGL(skinny_wgrad_kernel<4>, grid, dim3(256), 0, as_stream(stream), y + d->y_coff, d->y_ctot, x + d->x_coff,
                           d->x_ctot, dw, d->Cy, d->Cx, d->N);
        mgvae_prof_record_end(tok, stream);
        MGVAE_CHECK_LAUNCH();
        return MGVAE_OK;
    }
    IgemmP p = make_params(d);
    p.X = x; p.Y = y; p.out = dw; p.Wt = nullptr; p.bias = nullptr;
    { const int rc = get_ktab(d, MODE_BWD_WEIGHT, p); if (rc != MGVAE_OK) return rc; }
    hipStream_t s = as_stream(stream);
    const long I = d->Cy, J = (long)d->Cx * d->KH * d->KW, M = (long)d->N * d->OH * d->OW;
    const long max_splits = cdiv(M, BK * 8);
    if (autotune_on()) {
        const KtabKey key = choice_key(d, MODE_BWD_WEIGHT, 0);
        Choice c;
        bool have = choice_lookup(key, &c);
        if (!have && !is_capturing(s)) {
            // trial launches accumulate, so they write a scratch gradient, never the caller's
            float* scratch = nullptr;
            if (hipMalloc(&scratch, (size_t)I * J * sizeof(float)) == hipSuccess) {
                hipMemsetAsync(scratch, 0, (size_t)I * J * sizeof(float), s);
                std::vector<Choice> cands;
                for (int tile = 0; tile < 4; ++tile) {
                    if (!(tile & 1) && I <= 64) continue;
                    if (!(tile & 2) && J <= 64) continue;
                    const long tiles = (long)cdiv(I, tile_it(tile)) * cdiv(J, tile_jt(tile));
                    const long q = cdiv((long)g_cus * 2, tiles);
                    long last = -1;
                    for (long sp : {q / 2, q, q * 2, q * 4}) {
                        if (sp < 1) sp = 1;
                        if (sp > max_splits) sp = max_splits;
                        if (sp == last) continue;
                        last = sp;
                        cands.push_back(Choice{tile, (int)sp});
                    }
                }
                if (cands.empty()) cands.push_back(Choice{3, 1});
                IgemmP ps = p; ps.out = scratch;
                c = tune(cands, [&](const Choice& q) { return exec_bwd_weight(d, ps, q.tile, q.split, s); }, s);
                hipStreamSynchronize(s);
                hipFree(scratch);
                choice_store(key, c);
                have = true;
            }
        }
        if (have) return exec_bwd_weight(d, p, c.tile, c.split, s);
    }
    // heuristic: big tiles always (they run ~1.4x the MFMA rate of 64x64); the pixel reduction supplies
    // the parallelism through split-K, shrinking the tile only when even that cannot fill the chip
    int ti = I > 64 ? 2 : 1, tj = J > 64 ? 2 : 1;
    auto ntiles = [&]() { return (long)cdiv(I, 64 * ti) * cdiv(J, 64 * tj); };
    if (ntiles() * max_splits < g_cus && tj == 2) tj = 1;
    if (ntiles() * max_splits < g_cus && ti == 2) ti = 1;
    const int tile = (ti == 2 ? 0 : 1) + (tj == 2 ? 0 : 2);
    long splits = cdiv((long)g_cus * 3, ntiles());
    if (splits > max_splits) splits = max_splits;
    return exec_bwd_weight(d, p, tile, splits, s);
}

extern "C" int mgvae_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof = on != 0;
    return MGVAE_OK;
}

static FILE* g_prof_detail = nullptr;
extern "C" int mgvae_prof_detail(const char* path) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_detail) { fclose(g_prof_detail); g_prof_detail = nullptr; }
    if (path && path[0]) {
        g_prof_detail = fopen(path, "w");
        if (!g_prof_detail) return MGVAE_EINVAL;
        fprintf(g_prof_detail, "kind,tile,N,Cx,H,W,Cy,OH,OW,KH,KW,SH,SW,gx,gy,gz,us,gflop,tflops\n");
    }
    return MGVAE_OK;
}

extern "C" int mgvae_prof_collect(MgvaeProfRec* out, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    constexpr int SLOTS = 16;          // variant ids per kind (the x3 family: 3 loop forms + the halo form, x 4 tile shapes)
    MgvaeProfRec recs[MGVAE_PROF_KINDS * SLOTS];
    for (int k = 0; k < MGVAE_PROF_KINDS; ++k)
        for (int t = 0; t < SLOTS; ++t) recs[k * SLOTS + t] = MgvaeProfRec{k, t, 0, 0.0, 0.0};
    for (auto& pe : g_prof_entries) {
        float ms = 0.f;
        hipEventSynchronize(pe.e1);
        hipEventElapsedTime(&ms, pe.e0, pe.e1);
        MgvaeProfRec& r = recs[pe.kind * SLOTS + (pe.tile < SLOTS ? pe.tile : 0)];
        r.launches += 1; r.ms += ms; r.flops += pe.flops;
        if (g_prof_detail && (pe.kind < 5 || pe.kind >= MGVAE_PROF_NHWC_FWD)) {
            const IgemmP& q = pe.p;
            fprintf(g_prof_detail, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%u,%u,%u,%.1f,%.3f,%.2f\n", pe.kind, pe.tile, q.N, q.Cx,
                    q.H, q.W, q.Cy, q.OH, q.OW, q.KH, q.KW, q.SH, q.SW, pe.gx, pe.gy, pe.gz, ms * 1e3, pe.flops * 1e-9,
                    pe.flops / (ms * 1e-3) * 1e-12);
        }
        hipEventDestroy(pe.e0); hipEventDestroy(pe.e1);
    }
    g_prof_entries.clear();
    if (g_prof_detail) fflush(g_prof_detail);
    int n = 0;
    for (int i = 0; i < MGVAE_PROF_KINDS * SLOTS && n < cap; ++i)
        if (recs[i].launches > 0) out[n++] = recs[i];
    return n;
}

extern "C" const char* mgvae_kernel_name(int kind, int tile) {
    static const char* tiles[5] = {"2, 2", "1, 2", "2, 1", "1, 1", "4, 2"};
    static char buf[5][7][48];
    static char nbuf[9][16][48];
    if (kind >= MGVAE_PROF_NHWC_FWD && kind < MGVAE_PROF_NHWC_FWD + 9 && tile >= 0 && tile < (kind >= MGVAE_PROF_NHWC_X3_FWD ? 16 : 4)) {
        const int m = kind - MGVAE_PROF_NHWC_FWD;
        static const char* fam[6] = {"nhwc_igemm_kernel<%d, %s>", "nhwc_igemm_bf16_kernel<%d, %s>", "nhwc_igemm_x3_kernel<%d, %s>",
                                     "nhwc_igemm_x3s_kernel<%d, %s>", "nhwc_igemm_x3w_kernel<%d, %s>", "nhwc_halo_x3_kernel<%d, %s>"};
        if (tile >= 12) snprintf(nbuf[m][tile], 48, fam[5], m % 3, tile == 15 ? "4, 1" : ((tile & 2) ? "2, 1" : "2, 2"));   // halo form: <MODE, TI, TJ>
        else snprintf(nbuf[m][tile], 48, fam[m / 3 + tile / 4], m % 3, tiles[tile & 3]);
        return nbuf[m][tile];
    }
    if (kind == MGVAE_PROF_ADAM) return "adam_kernel";
    if (kind == MGVAE_PROF_INORM_FWD) return "instance_norm_fwd_*";
    if (kind == MGVAE_PROF_INORM_BWD) return "instance_norm_bwd_*";
    if (kind < 0 || kind > 4 || tile < 0 || tile > 6) return "?";
    if (tile == 5) snprintf(buf[kind][tile], 48, kind == 2 ? "skinny_wgrad_kernel" : "skinny_gemm_kernel (%s)", kind == 0 ? "fwd" : "bwd_data");
    else if (tile == 6) snprintf(buf[kind][tile], 48, kind == 2 ? "thin_bwd_weight_kernel" : "thin_fwd_kernel");
    else if (kind < 3) snprintf(buf[kind][tile], 48, "igemm_kernel<%d, %s>", kind, tiles[tile]);
    else snprintf(buf[kind][tile], 48, "dconv_kernel<%s> (%s)", tiles[tile], kind == 3 ? "fwd" : "bwd_data");
    return buf[kind][tile];
}

extern "C" int mgvae_device_info(char* arch, size_t arch_len, int* cu_count) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MGVAE_ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return MGVAE_ENODEV;
    if (arch && arch_len) {
        size_t i = 0;
        for (; i + 1 < arch_len && prop.gcnArchName[i]; ++i) arch[i] = prop.gcnArchName[i];
        arch[i] = 0;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (prop.multiProcessorCount > 0) g_cus = prop.multiProcessorCount;
    return MGVAE_OK;
}

extern "C" const char* mgvae_strerror(int code) {
    switch (code) {
        case MGVAE_OK: return "ok";
        case MGVAE_EINVAL: return "invalid argument or unsupported geometry";
        case MGVAE_ELAUNCH: return "kernel launch failed";
        case MGVAE_ENODEV: return "no usable gfx950 device";
        default: return "unknown error";
    }
}

#include "conv_nhwc.inc"
#include "conv_nhwc_bf16.inc"
#include "conv_nhwc_x3.inc"
