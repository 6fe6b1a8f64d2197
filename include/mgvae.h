/*
 * mgvae.h -- C ABI of libmgvae_hip.so: the MI355X (gfx950) kernels behind the bar-VAE
 * training hot path of KMU-AELAB-MusicProject/MusicGeneration_VAE-torch.
 *
 * The reference has no FFI: its hot path is reached through torch.nn modules
 * (graph/*.py) that dispatch to ATen/cuDNN.  Each entry point below replaces the
 * library kernel(s) one reference module call issues; the citation on every
 * prototype names that call site (paths relative to the reference root).
 *
 * Conventions
 *   - plain pointers + sizes; all tensors are fp32, NCHW, device memory owned by the
 *     caller (PyTorch's allocator); the library allocates nothing per call -- its only own
 *     memory is a small cache of per-geometry gather tables (a few hundred KB in total)
 *     filled on the first call of each conv geometry (run one warm-up step before
 *     capturing a hipGraph);
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*);
 *     no implicit synchronisation, safe under hipGraph stream capture.  One exception,
 *     outside capture only: the FIRST call of a (mode, conv geometry, batch) uploads that
 *     table synchronously and, unless MGVAE_AUTOTUNE=0, times the candidate (tile, split-K)
 *     launches with events on `stream` (weight-gradient trials accumulate into a
 *     hipMalloc'd scratch buffer, never into the caller's gradient); the decision is cached
 *     for the process (MGVAE_AUTOTUNE_FILE persists it);
 *   - return 0 on success, a negative MGVAE_E* code otherwise (mgvae_strerror());
 *   - "accumulate" outputs (weight / bias / affine gradients, embedding gradient) are
 *     ADDED into the destination, which the caller zeroes once per optimizer step;
 *   - channel-sliced tensors: a tensor argument described by (ctot, coff) is the
 *     channel range [coff, coff+C) of a buffer with ctot channels, so concatenations
 *     (torch.cat(dim=1) in the reference) are written in place with no copy.
 */
#ifndef MGVAE_H
#define MGVAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGVAE_OK 0
#define MGVAE_EINVAL (-1)   /* bad descriptor / null pointer / unsupported geometry */
#define MGVAE_ELAUNCH (-2)  /* hipLaunchKernel reported an error */
#define MGVAE_ENODEV (-3)   /* no gfx950 device / wrong architecture */

enum { MGVAE_ACT_NONE = 0, MGVAE_ACT_RELU = 1, MGVAE_ACT_LEAKY = 2, MGVAE_ACT_SIGMOID = 3 };

/* Geometry of one convolution, always in CONVOLUTION sense:
 *   X [N, x_ctot, H,  W ]  channels [x_coff, x_coff+Cx)   (the larger, "image" side)
 *   Y [N, y_ctot, OH, OW]  channels [y_coff, y_coff+Cy)   (the "feature" side)
 *   Wt[Cy, Cx, KH, KW]     (torch Conv2d layout; a ConvTranspose2d weight
 *                           [Cin_T, Cout_T, kh, kw] is the same memory with
 *                           Cy = Cin_T, Cx = Cout_T)
 *   OH = (H + 2*PH - KH) / SH + 1, likewise OW (output_padding of a transposed
 *   convolution is absorbed in H/W).
 * act/slope: activation fused on the tensor the call WRITES (fwd: Y, bwd_data /
 * transposed fwd: X).                                                              */
typedef struct MgvaeConvDesc {
    int32_t N, Cx, H, W, Cy, OH, OW;
    int32_t KH, KW, SH, SW, PH, PW;
    int32_t x_ctot, x_coff, y_ctot, y_coff;
    int32_t act;
    float slope;
} MgvaeConvDesc;

const char* mgvae_strerror(int code);
/* library / device probe: returns 0 and fills arch string ("gfx950...") + CU count */
int mgvae_device_info(char* arch, size_t arch_len, int* cu_count);

/* ---- convolution family (implicit GEMM on v_mfma_f32_32x32x2_f32) -----------------
 * nn.Conv2d forward:  graph/encodingBlock.py:12-15,43-46,74-77,107-108;
 * graph/decoder.py:79,122,172,175; nn.Linear (H=W=1): graph/encoder.py:22,38,
 * graph/decoder.py:166-167, graph/z_discriminator.py:13-24                            */
int mgvae_conv2d_fwd(const MgvaeConvDesc* d, const float* x, const float* w, const float* bias,
                     float* y, void* stream);
/* dX of nn.Conv2d == forward of nn.ConvTranspose2d (graph/decoder.py:12-15,43-46,
 * 73-77,116-120): X = conv_transpose(Y, Wt) (+ bias[Cx]), stride-phase decomposed    */
int mgvae_conv2d_bwd_data(const MgvaeConvDesc* d, const float* y, const float* w, const float* bias,
                          float* x, void* stream);
/* Same result from w_t = mgvae_weight_transpose(w) ([Cy][KH*KW][Cx]): the weight operand then
 * loads contiguously along the lanes.  The transpose is one small HBM pass per call.            */
int mgvae_weight_transpose(const float* w, float* w_t, int Cy, int Cx, int KK, void* stream);
int mgvae_conv2d_bwd_data_tw(const MgvaeConvDesc* d, const float* y, const float* w_t, const float* bias,
                             float* x, void* stream);
/* dWt += corr(X, Y): weight gradient of either layer type (split-K, fp32 atomics)    */
/* Matrix-operand precision of the tiled conv kernels (process-wide; default fp32).  BF16 = BASELINE.json configs
 * 3-4 ("bf16 compute / fp32 master weights"): tensors stay fp32 in HBM, operands are rounded to bf16 (RNE) while
 * being staged, products accumulate in fp32 on v_mfma_f32_32x32x16_bf16.  Thin (K<=16) and skinny (Linear) paths
 * stay fp32. */
/* F32_BF16X3 (opt-in): fp32-ACCURATE products on the bf16 matrix pipe -- every fp32 operand is split exactly into three
 * bf16 terms (8+8+8 mantissa bits) and a product is the six bf16 MFMAs whose terms are >= 2^-16 relative; bf16*bf16
 * is exact in the fp32 accumulator and the dropped terms are <= 2^-24 relative.  6/16 of the fp32 matrix-pipe time. */
enum { MGVAE_COMPUTE_F32 = 0, MGVAE_COMPUTE_BF16 = 1, MGVAE_COMPUTE_F32_BF16X3 = 2 };
int mgvae_set_compute_dtype(int dtype);
int mgvae_get_compute_dtype(void);

/* Activation-gradient mask for the *_masked entry points: out[n,c,p] *= act'(src[n, coff+c, p]) in the epilogue.
 * `src` is shaped like the written tensor ([N, ctot, H, W] at channel offset coff).  Use: the layer that produced a
 * conv's input fused an activation into its own forward (reference graph/encodingBlock.py:89-91: conv1 -> ReLU ->
 * conv2); instead of a separate dy*act'(y) pass in that layer's backward, conv2's data gradient -- whose output
 * has exactly that shape and whose saved input IS the activated tensor -- applies the factor while storing.
 * act' is expressed through the activation's OUTPUT (relu: src>0; leaky: src>0 ? 1 : slope; sigmoid: src(1-src)). */
typedef struct MgvaeActMask {
    const float* src;
    int32_t ctot, coff, act;
    float slope;
} MgvaeActMask;
int mgvae_conv2d_fwd_masked(const MgvaeConvDesc* d, const float* x, const float* w, const float* bias, float* y,
                            const MgvaeActMask* mask, void* stream);   /* mask over Y: ConvTranspose2d d/dx */
int mgvae_conv2d_bwd_data_masked(const MgvaeConvDesc* d, const float* y, const float* w, int w_transposed,
                                 const float* bias, float* x, const MgvaeActMask* mask, void* stream);   /* mask over X */

/* ---- channels-last (NHWC) family: the same three products on tensors stored [N, H, W, C] (x_ctot / y_ctot are then the
 * channel pitch of a pixel row) and weights stored [Cy, KH, KW, Cx] -- torch.channels_last for activations AND weights,
 * so logical shapes and state_dict entries are unchanged.  K = (tap, channel) is contiguous in memory for both forward
 * operands: 16-byte loads, no per-element gather arithmetic, no per-step weight repacking (csrc/conv_nhwc.inc).
 * Replaces the same reference calls as the NCHW entry points above (graph/encodingBlock.py:87-126 for the encoder
 * trunks).  Channel counts and slice offsets must be multiples of 16 / 4.  `mask` (nullable): epilogue factor act'(mask)
 * as in the *_masked entry points.                                                                                      */
int mgvae_conv2d_nhwc_fwd(const MgvaeConvDesc* d, const float* x, const float* w, const float* bias, float* y,
                          const MgvaeActMask* mask, void* stream);
int mgvae_conv2d_nhwc_bwd_data(const MgvaeConvDesc* d, const float* y, const float* w, const float* bias, float* x,
                               const MgvaeActMask* mask, void* stream);
int mgvae_conv2d_nhwc_bwd_weight(const MgvaeConvDesc* d, const float* x, const float* y, float* dw, void* stream);
/* bf16-STORAGE forms (BASELINE.json configs 3-4): x / y are bf16 channels-last tensors (passed as void*; a mask's `src`
 * then points at bf16 too), the two weight operands are bf16 copies of the fp32 channels-last master made once per
 * optimizer step by mgvae_pack_conv_weights_bf16 -- wk [Cy, KH*KW, Cx] for the forward, wt [Cx, KH*KW, Cy] for the data
 * gradient -- bias fp32, accumulation fp32 (v_mfma_f32_32x32x16_bf16), the weight gradient is ADDED to the fp32 master
 * gradient dw [Cy, KH*KW, Cx].  Channel counts multiples of 64, slice offsets multiples of 8 (csrc/conv_nhwc_bf16.inc).   */
int mgvae_pack_conv_weights_bf16(const float* w, void* wk, void* wt, int Cy, int T, int Cx, void* stream);
/* `ws, ws_bytes`: CALLER-OWNED split-K workspace (the library allocates nothing).  mgvae_conv2d_nhwc_bf16_workspace(d, mode)
 * -- mode 0 forward, 1 data gradient / transposed-conv forward, 2 weight gradient (always 0) -- returns the bytes the largest
 * deterministic split the tuner may pick for this geometry needs; 0 when the launch fills the chip unsplit.  A null or short
 * workspace is legal: the split shrinks to what fits (same values up to summation order of the K chunks).                 */
size_t mgvae_conv2d_nhwc_bf16_workspace(const MgvaeConvDesc* d, int mode);
int mgvae_conv2d_nhwc_bf16_fwd(const MgvaeConvDesc* d, const void* x, const void* wk, const float* bias, void* y,
                               const MgvaeActMask* mask, void* ws, size_t ws_bytes, void* stream);
int mgvae_conv2d_nhwc_bf16_bwd_data(const MgvaeConvDesc* d, const void* y, const void* wt, const float* bias, void* x,
                                    const MgvaeActMask* mask, void* ws, size_t ws_bytes, void* stream);
int mgvae_conv2d_nhwc_bf16_bwd_weight(const MgvaeConvDesc* d, const void* x, const void* y, float* dw, void* stream);
/* fp32-STORAGE forms on the bf16 matrix pipe ("x3", csrc/conv_nhwc_x3.inc): same tensors and results as the fp32 entry
 * points above (x / y / dw fp32, fp32 accumulation), but every fp32 operand value enters the matrix pipe as the exact sum
 * of three bf16 values and each product as six bf16 MFMAs (relative error < 2^-22 per product: fp32 grade) -- 16 / 6 of
 * the fp32 MFMA rate.  Activations are split while staged; weights once per optimizer step by mgvae_pack_conv_weights_x3
 * into wk3 (forward; rows = Cy, k = Cx) and wt3 (data gradient / transposed-conv forward; rows = Cx, k = Cy), bf16, 3 * Cy *
 * KH*KW * Cx elements each.  The buffers are OPAQUE to the caller (only these kernels read them); their layout is
 * [plane][row][tap][k], or -- when both channel counts are multiples of 32 -- block-major [tap][k / 32][row][plane][32], where
 * the 24 KB a 128-row K tile of 32 needs are one contiguous run of whole cache lines (csrc/conv_nhwc_x3.inc::x3w_blocked).
 * Channel counts multiples of 16, slice offsets multiples of 4.                                                         */
int mgvae_pack_conv_weights_x3(const float* w, void* wk3, void* wt3, int Cy, int T, int Cx, void* stream);
/* every conv weight of a network in one launch per optimizer step: `items` is a DEVICE array of n records
 * {const float* w; void* wk; void* wt; int32 Cy, T, Cx, block0} (40 bytes; block0 = running sum of
 * T * ceil(Cy / 32) * ceil(Cx / 32)), total_blocks that sum over all records; planes = 3: the x3 split (layouts of
 * mgvae_pack_conv_weights_x3), planes = 1: the bf16 copies of mgvae_pack_conv_weights_bf16.                           */
int mgvae_pack_conv_weights_grouped(const void* items, int n, int total_blocks, int planes, void* stream);
/* workspace: as for the bf16 family above (caller-owned, size from mgvae_conv2d_nhwc_x3_workspace, null / short is legal) */
size_t mgvae_conv2d_nhwc_x3_workspace(const MgvaeConvDesc* d, int mode);
int mgvae_conv2d_nhwc_x3_fwd(const MgvaeConvDesc* d, const float* x, const void* wk3, const float* bias, float* y,
                             const MgvaeActMask* mask, void* ws, size_t ws_bytes, void* stream);
int mgvae_conv2d_nhwc_x3_bwd_data(const MgvaeConvDesc* d, const float* y, const void* wt3, const float* bias, float* x,
                                  const MgvaeActMask* mask, void* ws, size_t ws_bytes, void* stream);
int mgvae_conv2d_nhwc_x3_bwd_weight(const MgvaeConvDesc* d, const float* x, const float* y, float* dw, void* stream);

int mgvae_conv2d_bwd_weight(const MgvaeConvDesc* d, const float* x, const float* y, float* dw,
                            void* stream);
/* Faster path of the two calls above: direct (halo-tile) convolution with pre-packed weights.
 * mode 0 packs for mgvae_conv2d_fwd_packed, mode 1 for mgvae_conv2d_bwd_data_packed (one
 * packed block per stride phase).  mgvae_conv_pack_floats returns the workspace size in floats
 * (0 = geometry not supported by the direct kernel: use the unpacked entry points).  Packing is
 * ~2 HBM passes over the weight and is redone whenever the weight changed; descriptors must use
 * x_coff = y_coff = 0 (pass pointers to the first channel of a slice).                         */
size_t mgvae_conv_pack_floats(const MgvaeConvDesc* d, int mode);
int mgvae_conv_pack(const MgvaeConvDesc* d, int mode, const float* w, float* packed, void* stream);
int mgvae_conv2d_fwd_packed(const MgvaeConvDesc* d, const float* x, const float* packed, const float* bias,
                            float* y, void* stream);
int mgvae_conv2d_bwd_data_packed(const MgvaeConvDesc* d, const float* y, const float* packed, const float* bias,
                                 float* x, void* stream);
/* db[c] += sum_{n,h,w} t[n, coff+c, h, w]   (bias gradient of Conv/ConvT/Linear)     */
int mgvae_channel_sum_accum(const float* t, int N, int C, int P, int ctot, int coff, float* db,
                            void* stream);

/* ---- InstanceNorm2d(affine, eps, no running stats) + fused activation -------------
 * graph/encodingBlock.py:17,48,79,110; graph/decoder.py:17,48,81-83,124-126,173
 * x [N,C,P] contiguous; y is channel-sliced; stats[N*C*2] = (mean, rstd) saved for bwd */
int mgvae_instance_norm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                            float* stats, int N, int C, int P, int y_ctot, int y_coff,
                            float eps, int act, float slope,
                            float* pool_avg, float* pool_max, int* pool_idx,   /* optional (all or none): fused CBAM
                                channel pooling of y -- pass the avg/max/argmax regions of the CBAM `save` buffer */
                            void* stream);
/* dy is channel-sliced like y; dx contiguous; dgamma/dbeta accumulate                 */
int mgvae_instance_norm_bwd(const float* x, const float* gamma, const float* beta, const float* stats,
                            const float* dy, float* dx, float* dgamma, float* dbeta,
                            int N, int C, int P, int dy_ctot, int dy_coff, int act, float slope,
                            const float* add_const, const float* add_point, const int* add_index, /* optional (all or
                                none): dy[n,c,p] += add_const[n,c]/P + (p == add_index[n,c]) * add_point[n,c]: the
                                deferred tail of mgvae_cbam_bwd (parts & 4) */
                            void* stream);

/* ---- BatchNorm2d (graph/bar_discriminator.py:19-23,69,113-114,153; graph/refiner.py) ----------
 * training != 0: batch statistics, running_mean/var updated in place with `momentum` (unbiased
 * variance, like torch); stats[2C] = (mean, rstd) saved.  x, y, dy, dx dense [N,C,P];
 * dgamma / dbeta accumulate.                                                                     */
int mgvae_batch_norm_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* y, float* stats, int N, int C, int P, int training,
                         float momentum, float eps, int act, float slope, void* stream);
int mgvae_batch_norm_bwd(const float* x, const float* gamma, const float* beta, const float* stats,
                         const float* dy, float* dx, float* dgamma, float* dbeta, int N, int C, int P,
                         int training, int act, float slope, void* stream);

/* ---- CBAM (graph/cbam.py:22-29,43-52,63-67) fused with the residual that follows it
 * mode 0: y = cbam(u)                      (graph/cbam.py CBAM.forward)
 * mode 1: y = act(u + cbam(u))             (graph/encodingBlock.py:32,63,122; decoder.py:32,62,103,140,150,213)
 * mode 2: y = act(res + cbam(u))           (graph/encodingBlock.py:94-98)
 * `save` layout (floats): cg[NC] avg[NC] max[NC] argmax_hw[NC](int) hidden[2*N*C/16, padded to a multiple of 4] s_in[2NP] argmax_c[NP](int)
 * sg[NP]; bwd `scratch`: dt[NP] ds_in[2NP] dcg[NC] davg[NC] dmaxp[NC] dh[2*N*C/16].
 * parts | 4: forward -- avg/max/argmax were already written into `save` by mgvae_instance_norm_fwd;
 * backward -- skip the final "du += davg/P + [p==argmax] dmaxp" pass: the caller's
 * mgvae_instance_norm_bwd applies it (add_const = davg, add_point = dmaxp, add_index = argmax_hw).
 * parts: 3 = channel then spatial attention (CBAM); 1 = ChannelAttention alone (graph/cbam.py:22-29);
 * 2 = SpatialAttention alone (:43-52).  u [N,C,P] contiguous, P = H*W; w1 [C/16,C], w2 [C,C/16], wsp [1,2,3,3].
 * Workspace `save` (floats, size mgvae_cbam_save_floats) keeps what backward needs.   */
size_t mgvae_cbam_save_floats(int N, int C, int H, int W);
int mgvae_cbam_fwd(const float* u, const float* res, const float* w1, const float* w2, const float* wsp,
                   float* y, float* save, int N, int C, int H, int W, int y_ctot, int y_coff,
                   int mode, int act, float slope, int parts, void* stream);
/* du (contiguous) and dres (contiguous, mode 2 only) are written; dw1/dw2/dwsp accumulate.
 * y/dy are channel-sliced (ctot, coff).  `scratch` needs mgvae_cbam_bwd_scratch_floats.  */
size_t mgvae_cbam_bwd_scratch_floats(int N, int C, int H, int W);
int mgvae_cbam_bwd(const float* u, const float* y, const float* dy, const float* w1, const float* w2,
                   const float* wsp, const float* save, float* du, float* dres, float* dw1, float* dw2,
                   float* dwsp, float* scratch, int N, int C, int H, int W, int y_ctot, int y_coff,
                   int mode, int act, float slope, int parts, void* stream);

/* ---- pointwise / small ops ---------------------------------------------------------*/
/* ---- channels-last fused InstanceNorm2d -> CBAM -> (+residual) -> activation (graph/encodingBlock.py:48-55,93-100,
 * 110-126 on tensors stored [N, H, W, C]): x [N, P, C] dense (a conv output), res / y / dy channel slices (ctot, coff) of
 * channels-last buffers, dres / dx dense.  C in {64, 128, 256, 512, 1024}.  u = gamma * xhat + beta is recomputed from x
 * in every pass instead of being stored (csrc/norm_cbam_nhwc.inc).  `save` / `scratch`: workspaces of the sizes the two
 * functions below return (floats, 16-byte aligned).  Weight-like gradients accumulate (fp32).
 * `storage`: element type of the big tensors (x, res, y, dy, dx, dres), passed as void*: MGVAE_STORE_F32, or
 * MGVAE_STORE_BF16 for the bf16-storage mode of BASELINE.json configs 3-4 (statistics, gates, arithmetic stay fp32).    */
enum { MGVAE_STORE_F32 = 0, MGVAE_STORE_BF16 = 1 };
size_t mgvae_norm_cbam_nhwc_save_floats(int N, int C, int H, int W);
size_t mgvae_norm_cbam_nhwc_scratch_floats(int N, int C, int H, int W);
int mgvae_norm_cbam_nhwc_fwd(const void* x, const float* gamma, const float* beta, const void* res, int res_ctot,
                             int res_coff, const float* w1, const float* w2, const float* wsp, void* y, float* save,
                             int N, int C, int H, int W, int y_ctot, int y_coff, float eps, int mode, int act,
                             float slope, int storage, void* stream);
int mgvae_norm_cbam_nhwc_bwd(const void* x, const float* gamma, const float* beta, const void* y, const void* dy,
                             const float* w1, const float* w2, const float* wsp, const float* save, void* dx,
                             void* dres, float* dgamma, float* dbeta, float* dw1, float* dw2, float* dwsp,
                             float* scratch, int N, int C, int H, int W, int y_ctot, int y_coff, int mode, int act,
                             float slope, int storage, void* stream);
/* the CBAM gate MLP's weight gradients alone: after mgvae_norm_cbam_nhwc_bwd was called with dw1 = dw2 = NULL, from the
 * same `save` / `scratch` buffers, on any stream ordered after that call (the launch chains put it on the weight-gradient
 * stream: only the optimizer waits for it)                                                                             */
int mgvae_norm_cbam_nhwc_bwd_mlp_wgrad(const float* save, const float* scratch, float* dw1, float* dw2, int N, int C, int H,
                                       int W, void* stream);
/* InstanceNorm2d (+ReLU / LeakyReLU) alone on channels-last tensors (graph/decoder.py:81-83,124-126); `stats`:
 * mgvae_instance_norm_nhwc_stats_floats() floats (6 N C kept for backward + the chunked statistics pass's partials), `scratch`: 2 N C floats; and the bias gradient of a transposed conv (sum over pixel rows).   */
size_t mgvae_instance_norm_nhwc_stats_floats(int N, int C, int H, int W);   /* floats of the `stats` workspace below */
int mgvae_instance_norm_nhwc_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int N,
                                 int C, int H, int W, int y_ctot, int y_coff, float eps, int act, float slope, int storage,
                                 void* stream);
int mgvae_instance_norm_nhwc_bwd(const void* x, const float* gamma, const float* stats, const void* y, const void* dy,
                                 void* dx, float* dgamma, float* dbeta, float* scratch, int N, int C, int H, int W,
                                 int y_ctot, int y_coff, int act, float slope, int storage, void* stream);
int mgvae_channel_sum_nhwc_accum(const void* t, long rows, int C, int ctot, int coff, float* db, int storage, void* stream);
/* layout changes at the ends of a channels-last island (channel slices on both sides; the NCHW side is always fp32, the
 * channels-last side has the `storage` type) and the whole-map average of a channels-last tensor [N, P, C] -> fp32 [N, C]
 * (graph/encoder.py:35, graph/phrase_encoder.py:36) with its gradient */
int mgvae_layout_nchw_to_nhwc(const float* src, void* dst, int N, int C, int P, int src_ctot, int src_coff,
                              int dst_ctot, int dst_coff, int storage, void* stream);
int mgvae_layout_nhwc_to_nchw(const void* src, float* dst, int N, int C, int P, int src_ctot, int src_coff,
                              int dst_ctot, int dst_coff, int storage, void* stream);
int mgvae_mean_nhwc_fwd(const void* x, float* out, int N, int C, int P, int storage, void* stream);
int mgvae_mean_nhwc_bwd(const float* dout, void* dx, int N, int C, int P, int storage, void* stream);

/* The channels-last ENDS of the island (csrc/thin_nhwc.hip): the two convs with ONE channel on their other side, so the
 * encoder trunks begin and the decoder ends without a layout change.
 *  - conv2d_c1: a conv of a one-channel fp32 map x [N,1,H,W] into Cy in {16,32,64} channels, written channels-last
 *    (`storage` type) with the descriptor's activation -- the encoder stems' first convs (reference
 *    graph/encodingBlock.py:11-14,42-45: Conv2d(1, 32, (4,1)|(1,4), stride 2 on that axis, bias=False) + LeakyReLU).
 *    d->Cx = d->x_ctot = 1, KH*KW <= 8; w is the reference's [Cy,1,KH,KW] fp32.  bwd_weight ACCUMULATES
 *    dw[c][t] += sum g[pixel][c] x[tap t], g = dy, or dy * act'(ymask) when `ymask` (the forward's output) is given.
 *  - conv2d_to1: a bias-free 1x1 conv of a channels-last map (rows = N*H*W pixel rows of C in {16..256} channels, a
 *    channel slice allowed) into ONE fp32 channel with an activation -- the decoder's fit2 (graph/decoder.py:186,217:
 *    Conv2d(64, 1, 1, bias=False) + Sigmoid).  bwd: g = dy * act'(y); dx[r,c] = g w[c] (`storage` type, may be NULL);
 *    dw[c] += sum_r g x[r,c] (may be NULL).                                                                          */
int mgvae_conv2d_c1_nhwc_fwd(const MgvaeConvDesc* d, const float* x, const float* w, void* y, int storage, void* stream);
int mgvae_conv2d_c1_nhwc_bwd_weight(const MgvaeConvDesc* d, const float* x, const void* dy, const void* ymask, float* dw,
                                    int storage, void* stream);
int mgvae_conv2d_to1_nhwc_fwd(const void* x, const float* w, float* y, long rows, int C, int x_ctot, int x_coff, int act,
                              float slope, int storage, void* stream);
int mgvae_conv2d_to1_nhwc_bwd(const void* x, const float* w, const float* y, const float* dy, void* dx, float* dw, long rows,
                              int C, int x_ctot, int x_coff, int act, float slope, int storage, void* stream);
/* storage-type change of a dense tensor of n elements (n % 4 == 0), fp32 <-> bf16 (round to nearest even): the fp32 stems'
 * concat entering a bf16 island and its gradient coming back (BASELINE.json configs 3-4)                            */
int mgvae_cast_storage(const void* src, int src_storage, void* dst, int dst_storage, size_t n, void* stream);

/* dx = dy * act'(y) given the activation OUTPUT y (ReLU/LeakyReLU/Sigmoid); all three
 * tensors may be channel slices of [N, ctot, P] buffers                               */
int mgvae_act_bwd(const float* y, const float* dy, float* dx, int N, int C, int P,
                  int y_ctot, int y_coff, int dy_ctot, int dy_coff, int dx_ctot, int dx_coff,
                  int act, float slope, void* stream);
/* strided row copy (torch.cat / slicing along channels): dst[r, 0:width] = src[r, 0:width] */
int mgvae_copy2d(float* dst, size_t dpitch, const float* src, size_t spitch, size_t width, size_t rows,
                 void* stream);
/* dst[i] += src[i] */
int mgvae_add_inplace(float* dst, const float* src, size_t n, void* stream);
/* out[r] = mean(x[r, 0:L]) -- nn.AvgPool2d over the whole map, graph/encoder.py:20,35;
 * graph/phrase_encoder.py:21,36; bwd: dx[r, l] = dout[r] / L                          */
int mgvae_rowmean_fwd(const float* x, float* out, int rows, int L, void* stream);
int mgvae_rowmean_bwd(const float* dout, float* dx, int rows, int L, void* stream);
/* out[r, g] = sum_{i<gsize} x[r, g*gsize + i]: the pitch-axis folding of the BarDiscriminator
 * front-ends (graph/bar_discriminator.py:32-34: 60 -> 12 groups of 5; :86-87: sum over 60)      */
int mgvae_group_sum_fwd(const float* x, float* out, size_t rows, int groups, int gsize, void* stream);
int mgvae_group_sum_bwd(const float* dout, float* dx, size_t rows, int groups, int gsize, void* stream);
/* Refiner pieces (graph/refiner.py:11-58): MaxPool2d(2) with saved argmax (0..3), a plain
 * activation pass, and out = a*x + b*y (residual adds, the final (x + y) * 0.5)                 */
int mgvae_maxpool2_fwd(const float* x, float* y, int* idx, size_t planes, int H, int W, void* stream);
int mgvae_maxpool2_bwd(const float* dy, const int* idx, float* dx, size_t planes, int H, int W, void* stream);
int mgvae_act_fwd(const float* x, float* y, size_t n, int act, float slope, void* stream);
int mgvae_axpby(const float* x, const float* y, float* out, float a, float b, size_t n, void* stream);
/* nn.Embedding gather (graph/decoder.py:187,193) into a row-sliced destination and its
 * scatter-add gradient                                                                */
int mgvae_embedding_fwd(const int64_t* idx, const float* table, float* out, int B, int D, int rows,
                        size_t out_pitch, void* stream);
int mgvae_embedding_bwd(const int64_t* idx, const float* dout, float* dtable, int B, int D, int rows,
                        size_t dout_pitch, void* stream);
/* nn.Dropout(p) (graph/decoder.py:164,196,201): Philox4x32-10 keyed by (seed, offset);
 * mask (float {0, 1/(1-p)}) is written for backward / parity tests                    */
int mgvae_dropout_fwd(const float* x, float* y, float* mask, size_t n, float p, uint64_t seed,
                      uint64_t offset, void* stream);
int mgvae_mul(const float* a, const float* b, float* out, size_t n, void* stream);
/* Gaussian prior noise randn*sigma (agent/barGen2.py:243,250; barGen_with_gan.py:517) */
int mgvae_randn(float* out, size_t n, float sigma, uint64_t seed, uint64_t offset, void* stream);

/* ---- losses (graph/loss/bar_loss.py:23-33,41-42) -------------------------------------
 * BCE(mean) with torch's log clamp at -100.  target_mode 0: targets[i]; 1: label
 * smoothing targets[i]*0.82 + 0.1/60 + prior[i % 60]*0.08 (prior = 60 floats, already
 * multiplied by 0.08); 2: constant target `tconst` (DLoss with all-ones/zeros targets).
 * If count_term != 0 adds 0.005 * #{ targets[i] - (x[i] > 0.3) > 1e-4 } (no gradient).
 * loss_out[0] = result; partial must hold mgvae_bce_partial_floats() floats.          */
size_t mgvae_bce_partial_floats(void);
int mgvae_bce_fwd(const float* x, const float* targets, const float* prior, float tconst, size_t n,
                  int target_mode, int count_term, float* partial, float* loss_out, void* stream);
/* dx[i] = gscale[0] * (x - t) / max((1 - x) * x, 1e-12) / n   (torch's BCE backward)   */
int mgvae_bce_bwd(const float* x, const float* targets, const float* prior, float tconst, size_t n,
                  int target_mode, const float* gscale, float* dx, void* stream);

/* ---- reparameterisation sampler + KL (old/graphs/models/bar_v1/encoder.py:60-63;
 * old/graphs/losses/loss.py:14-17).  eps is supplied (mgvae_randn) so tests can pin it.
 * kl_out[0] = -0.5 * sum(1 + logvar - mean^2 - exp(logvar))                            */
int mgvae_reparam_kl_fwd(const float* mean, const float* logvar, const float* eps, float* z,
                         float* partial, float* kl_out, size_t n, void* stream);
/* dmean = dz + gkl[0]*mean ; dlogvar = dz*eps*0.5*exp(0.5*logvar) + gkl[0]*0.5*(exp(logvar)-1) */
int mgvae_reparam_kl_bwd(const float* mean, const float* logvar, const float* eps, const float* dz,
                         const float* gkl, float* dmean, float* dlogvar, size_t n, void* stream);

/* ---- fused flat Adam (torch.optim.Adam defaults, agent/barGen2.py:60-64) ------------
 * hyper (device, 4 floats): lr/bias_correction1, sqrt(bias_correction2), beta1, beta2.
 * One launch updates a whole flat parameter buffer; grad_scale multiplies g first
 * (1/world_size after an all-reduce(sum)).                                             */
/* Input pipeline (reference data/bar_dataset.py:20-25 + agent make_batch, barGen2.py:128-135, ship fp32 rolls: 138 KB per
 * sample).  Rolls are {0,1}: the host ships them bit-packed (LSB first) and the device expands them to fp32:
 * out[i] = (packed[i >> 3] >> (i & 7)) & 1, i < nbits.  `out` 16-byte aligned. */
int mgvae_unpack_bits(const unsigned char* packed, float* out, size_t nbits, void* stream);

int mgvae_adam_step(float* p, const float* g, float* m, float* v, size_t n, const float* hyper,
                    float eps, float grad_scale, void* stream);

/* ---- bf16 gradient transport of the data-parallel exchange (replaces the per-tensor Horovod all-reduce of
 * agent/barGen_horovod.py:91-99): the flat fp32 gradient is rounded to bf16 (RNE) for the wire, the `R` addends that
 * arrive at a rank are summed in fp32 in rank order and rounded once, the reduced bucket is expanded back to fp32.
 * bf16 buffers are passed as void* (2 bytes per element, 8-byte aligned); n of mgvae_bf16_rows_sum a multiple of 4. */
int mgvae_f32_to_bf16(const float* src, void* dst_bf16, size_t n, void* stream);
int mgvae_bf16_to_f32(const void* src_bf16, float* dst, size_t n, void* stream);
int mgvae_bf16_rows_sum(const void* rows_bf16, void* dst_bf16, int R, size_t n, void* stream);

/* ---- launch chains (round 3; csrc/chain.hip) -----------------------------------------------------------------------
 * One call enqueues a whole sequence of the entry points above -- a block's forward or backward -- so that the host layer
 * pays one foreign call and one autograd node per block instead of one per launch (the reference's block boundaries:
 * graph/encodingBlock.py:87-100,118-126; graph/decoder.py:91-109,135-154).  Call k invokes entry point `fn` (an id from
 * mgvae_chain_fn_id(name); ids are positions in the sorted list of int-returning entry points, mgvae_chain_fn_count() of
 * them) with `nargs` arguments read from the 8-byte words words[first .. first + nargs): pointers, size_t, long and
 * uint64_t as the 64-bit word, int as its low 32 bits, float as the IEEE bits in its low 32 bits, double as the word's
 * bits; a struct-pointer argument points wherever the caller keeps the struct (usually further words of the same array).
 * Every entry point takes its stream as an argument, so one chain may span several streams; mgvae_stream_fork orders
 * them.  Stops at the first call that fails: returns its code and, when `failed` is not null, stores its index there (-1
 * if none).  Same kernels, tuner decisions, results and stream order as issuing the calls one by one.                       */
typedef struct MgvaeChainCall { int32_t fn, nargs, first, reserved; } MgvaeChainCall;
int mgvae_chain_fn_count(void);
int mgvae_chain_fn_id(const char* name);
int mgvae_chain_run(const MgvaeChainCall* calls, int ncalls, const uint64_t* words, int* failed);
/* stream `to` waits for everything enqueued on stream `from` so far (hipEventRecord + hipStreamWaitEvent on a library-owned
 * event per ordered stream pair): how a chain forks a weight gradient onto a side stream and joins it again             */
int mgvae_stream_fork(void* from, void* to);
/* dst += src over n elements of the channels-last storage type (0 fp32, 1 bf16; n a multiple of 4): the sum of the two
 * gradient contributions of a tensor that feeds two consumers inside one chained block (residual input, the twin
 * transposed convs of a decoder block)                                                                                    */
int mgvae_add_inplace_typed(void* dst, const void* src, size_t n, int storage, void* stream);

/* ---- measurement hooks (bench.py roofline leg) ----------------------------------------
 * When enabled, every mgvae_conv2d_* launch is bracketed by hipEvents on its stream and
 * its algorithmic FLOPs are recorded.  mgvae_prof_collect synchronises the events and
 * returns per-kernel-variant totals: up to `cap` records of
 * {kind (0 fwd,1 bwd_data,2 bwd_weight igemm; 3 fwd,4 bwd_data direct), tile id, launches, total ms, total flops}.
 * Kinds 5..7 are the HBM-bound kernels (flat Adam, InstanceNorm forward / backward): their `flops` field carries the
 * ALGORITHMIC BYTES of the launch instead (Adam 7 x 4n; InstanceNorm 2 x / 3 x 4 N C P).
 * Kinds 8..10: the channels-last implicit-GEMM kernels (forward, data gradient, weight gradient); 11..13: their bf16-storage
 * forms; 14..16: their fp32-storage forms on the bf16 matrix pipe (x3). */
enum { MGVAE_PROF_ADAM = 5, MGVAE_PROF_INORM_FWD = 6, MGVAE_PROF_INORM_BWD = 7, MGVAE_PROF_NHWC_FWD = 8, MGVAE_PROF_NHWC_BF16_FWD = 11,
       MGVAE_PROF_NHWC_X3_FWD = 14, MGVAE_PROF_KINDS = 17 };
typedef struct MgvaeProfRec { int32_t kind, tile, launches; double ms, flops; } MgvaeProfRec;
int mgvae_prof_enable(int on);
int mgvae_prof_collect(MgvaeProfRec* out, int cap);
/* optional: write one CSV row per profiled launch (geometry, grid, us, TFLOP/s) at collect time */
int mgvae_prof_detail(const char* path);
/* bracket any launch with profiler events (used by the direct-conv path; kind 3/4) */
int mgvae_prof_record_begin(int kind, int tile, double flops, void* stream, void** token);
int mgvae_prof_record_end(void* token, void* stream);
const char* mgvae_kernel_name(int kind, int tile);

#ifdef __cplusplus
}
#endif
#endif /* MGVAE_H */
