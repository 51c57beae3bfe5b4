/*
 * segengine.h — C ABI of the MI355X (gfx950) segmentation hot-path library, libsegengine.so.
 *
 * The reference (A511-1103/building-detection) is 100 % Python on tf.keras; it has no FFI of its own.  The
 * seam this library plugs into is the one TensorFlow occupies there: the numeric runtime underneath the
 * tf.keras layer calls of predict_model/{v3plus,bam,scse,res34,hrnet}.py and the loss / metric / optimizer
 * ops of train_model/DeepLabv3plus.py:490-623,834-837.  Every entry point below names the reference layer
 * call (file:line, relative to the reference root) whose arithmetic it replaces.  The Python host in
 * building_detection_amd/ binds these symbols with ctypes (building_detection_amd/_lib.py) and is the only
 * caller; INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no C++/torch types cross the boundary.
 *  - every function returns int: 0 = ok, <0 = SG_E* engine error, >0 = hipError_t.  sg_last_error() returns
 *    a thread-local message for the last non-zero return on the calling thread.
 *  - all tensor pointers are DEVICE pointers owned by the caller (they may be torch-owned storage).
 *  - every launch takes the hipStream_t to launch on (as void*); no function synchronises, allocates or
 *    frees device memory (so all of them are hipGraph-capturable); scratch is a caller-provided workspace
 *    whose size the matching *_ws_bytes() query returns.
 *  - layout is fixed: activations NHWC, conv kernels HWIO ([kh][kw][Cin][Cout]), depthwise [kh][kw][C],
 *    Conv2DTranspose kernels [kh][kw][Cout][Cin], Dense [in][out]  (the tf.keras get_weights() layouts).
 *  - dtype: SG_F32 (the reference's own precision) or SG_BF16 = bf16 STORAGE of every activation and activation
 *    gradient (x, y, dy, dx, gate tensors) with fp32 arithmetic inside the kernels: convolutions multiply bf16
 *    operands on the matrix pipe and accumulate in fp32, everything else widens on load and rounds (to nearest
 *    even) on store.  Weights, biases, BatchNorm parameters and statistics, weight gradients, the loss and Adam are
 *    fp32 in both modes (fp32 master weights).  BASELINE config 3.
 *  - "pixel stride" arguments (x_ld / y_ld, in elements) let an operand be a channel slice of a wider
 *    NHWC buffer, which is how concat is aliased away; 0 means "dense" (= its channel count).
 */
#ifndef SEGENGINE_H_
#define SEGENGINE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_ABI_VERSION 1

/* engine error codes (negative returns) */
#define SG_EINVAL   (-1)  /* bad argument / unsupported shape */
#define SG_EWORKSPACE (-2) /* workspace too small */
#define SG_EUNSUPPORTED (-3)
#define SG_ECOMM    (-4)  /* RCCL reported an error (sg_comm_*) */

#define SG_F32  0
#define SG_BF16 1
#define SG_I64  2         /* sg_comm_allreduce_sum only (the four confusion counts) */
/* OR-ed into the dtype of sg_conv2d_{fwd,dgrad,wgrad} on SG_BF16 storage: the few-channel side of a thin 1x1
 * convolution (Cout <= 4: y of the forward, dy of the two backward calls) is fp32.  This is the softmax head
 * Conv2D(num_classes, 1, activation='softmax') (v3plus.py:345): logits, probabilities, loss and their gradients
 * stay in fp32 while every other activation is bf16. */
#define SG_HEAD_F32 0x100
/* OR-ed into the dtype of sg_conv2d_wgrad: x is the SOURCE of a nearest 2x up-sampling - x[N, H/2, W/2, Cin] stands for the
 * up-sampled [N, H, W, Cin] tensor the descriptor names, the kernel reads x[n, h >> 1, w >> 1, :] (same bits as the filter
 * gradient on the materialised tensor).  See SG_PRO_UP2. */
#define SG_X_UP2 0x200

/* epilogue flags for conv-like ops */
#define SG_EPI_BIAS 1
#define SG_EPI_RELU 2
/* UpSampling2D(size=2) -> Conv2D(3x3, 'same') fused (train_model/DeepLabv3plus.py:476-477, the decoder's last stage; round 5).
 * The descriptor always names the convolution on the UP-SAMPLED grid (H x W); the up-sampled tensor itself never exists.
 *   SG_PRO_UP2   (flags of sg_conv2d_fwd / _ws / _stats) x is the source x[N, H/2, W/2, Cin].  The "sub-pixel" kernel computes
 *                the four output phases from the 2 x 2 source pixels each of them sees, with the kernel's taps summed
 *                beforehand: 4/9 of the multiplications, within fp32 rounding of the unfused pair (not bit-identical to it).
 *   SG_EPI_DOWN2 (flags of sg_conv2d_dgrad) dx is the gradient of the SOURCE, dx[N, H/2, W/2, Cin]: each 2 x 2 cell of the
 *                up-sampled tensor's gradient is added in the epilogue, in sg_upsample_nearest_bwd's order (bit-identical to
 *                the unfused pair).  No bias / ReLU / collected gradient with it.
 * Geometry: 3x3, stride 1, dilation 1, SAME, Cin = 64, Cout = 32, H % 16 = 0, W % 32 = 0, SG_F32 storage, x6 arithmetic on;
 * anything else returns SG_EUNSUPPORTED (sg_conv2d_up2_supported tells beforehand). */
#define SG_PRO_UP2   16
#define SG_EPI_DOWN2 32

typedef struct sg_ctx sg_ctx;

int sg_abi_version(void);
const char* sg_last_error(void);
int sg_create(int device, sg_ctx** out);
int sg_destroy(sg_ctx* ctx);
/* number of compute units of the ctx's device (for host-side split heuristics) */
int sg_num_cus(const sg_ctx* ctx);
/* Process-wide switch of the convolution arithmetic on SG_F32 storage (csrc/conv_x6.h): 1 (default, or SG_CONV_X6 in
 * the environment) = "x6": every fp32 product as six bf16 MFMA passes over an exact 3-way split, at least as accurate
 * as the fp32 MFMA; 0 = the native fp32 MFMA kernels; 2 = ONE bf16 MFMA pass (operands rounded to bf16 on their way
 * into LDS, fp32 accumulation): the arithmetic of SG_BF16 on fp32 tensors.  Returns the previous value.  Lets a
 * caller (and the parity tests) run the paths on the same inputs. */
int sg_set_conv_x6(int on);

/* ------------------------------------------------------------------------------------------------ conv
 * Geometry of a forward convolution  y[N,Ho,Wo,Cout] = conv(x[N,H,W,Cin], w[KH,KW,Cin,Cout]).
 * pad_t / pad_l are the TF "SAME" pads *before* (SURVEY.md App. B-1); pads after follow from Ho/Wo. */
typedef struct sg_conv_desc {
  int32_t N, H, W, Cin;
  int32_t Cout, KH, KW;
  int32_t stride, dilation;
  int32_t pad_t, pad_l;
  int32_t Ho, Wo;
  int32_t x_ld; /* pixel stride of x  (0 = Cin)  */
  int32_t y_ld; /* pixel stride of y  (0 = Cout) */
} sg_conv_desc;

/* Conv2D forward: implicit-GEMM on MFMA, M = N*Ho*Wo, N = Cout, K = KH*KW*Cin.
 * Replaces tf.keras.layers.Conv2D at predict_model/v3plus.py:173,177,185,289 (incl. the ASPP / SK dilated
 * 3x3 of :83-91,:298-300), predict_model/scse.py:52-95, predict_model/res34.py:33,54,147,156,
 * predict_model/hrnet.py:21, and the pointwise half of SeparableConv2D (v3plus.py:187-278).
 * flags: SG_EPI_BIAS adds bias[Cout]; SG_EPI_RELU applies max(.,0) (Conv2D(activation='relu')). */
int sg_conv2d_fwd(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x,
                  const void* w, const void* bias, void* y, int flags);

/* The same forward with a caller-provided workspace of sg_conv2d_fwd_ws_bytes(d) bytes (16-byte aligned).
 * With it the GEMM may run on the bf16 matrix pipe as six MFMA passes over a 3-way bf16 split of the fp32
 * operands ("x6", csrc/conv_x6.h: at least as accurate as the fp32 MFMA, measured; the workspace holds the
 * split kernel); without it (or when the shape does not qualify) the native fp32 MFMA kernel runs. */
size_t sg_conv2d_fwd_ws_bytes(const sg_conv_desc* d);
int sg_conv2d_fwd_ws(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x,
                     const void* w, const void* bias, void* y, int flags, void* ws, size_t ws_bytes);
/* 1 when the convolution `d` (on the up-sampled grid) takes the fused up-sampling kernels (SG_PRO_UP2 / SG_EPI_DOWN2 /
 * SG_X_UP2) with this storage type and the current arithmetic switch, else 0. */
int sg_conv2d_up2_supported(int dtype, const sg_conv_desc* d);

/* The forward that also hands the following BatchNormalization its statistics (epilogue flag
 * bn_stats): per 128-pixel tile and output channel the sum and the centred sum of squares of y, written to
 * stats[tiles][2][Cout] (sg_conv2d_fwd_stats_bytes) from the accumulators.  *tiles_out = number of tiles written,
 * or 0 when this launch could not produce them (shape not on the x6 path): the caller then lets
 * sg_bn_train_fwd compute its own.  sg_bn_train_fwd_tiles turns them into mean / inv-std / moving statistics
 * (fp64 combination); the apply pass is sg_bn_apply.  Replaces Conv2D -> BatchNormalization (training) at
 * predict_model/v3plus.py:173-179 and every SeparableConv2D -> BatchNormalization pair (v3plus.py:187-278). */
size_t sg_conv2d_fwd_stats_bytes(const sg_conv_desc* d);
int sg_conv2d_fwd_stats(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x,
                        const void* w, const void* bias, void* y, int flags, void* ws, size_t ws_bytes,
                        void* stats, int* tiles_out);
size_t sg_bn_tiles_ws_bytes(const sg_ctx* ctx, int tiles, int C);
int sg_bn_train_fwd_tiles(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* stats,
                          int tiles, void* moving_mean, void* moving_var, void* save_mean, void* save_invstd,
                          float momentum, float eps, int unbiased_update, void* ws, size_t ws_bytes);
/* y = gamma * (x - mean) * invstd + beta (ReLU optional) with given statistics: the apply pass of the training
 * forward on its own. */
int sg_bn_apply(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* x, const void* gamma,
                const void* beta, const void* mean, const void* invstd, void* y, int relu);

/* Conv2D input gradient dx[N,H,W,Cin] (pixel stride d->x_ld) from dy[N,Ho,Wo,Cout] (pixel stride d->y_ld).
 * The same routine is Conv2DTranspose *forward* (y_T = dgrad of the SAME conv that maps the upsampled grid
 * back): v3plus.py:328,335, scse.py:71-89, res34.py:144 — hence the optional epilogue
 * (bias has d->Cin entries here).  ws: sg_conv2d_dgrad_ws_bytes(d) bytes (the per-tap transposed kernel of the
 * fp32 path, or its three bf16 planes for the x6 path, whichever is larger). */
size_t sg_conv2d_dgrad_ws_bytes(const sg_conv_desc* d);
int sg_conv2d_dgrad(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy,
                    const void* w, const void* bias, void* dx, int flags, void* ws, size_t ws_bytes);
/* sg_conv2d_dgrad with a gradient already collected for the same tensor added in the epilogue: dx = dgrad(dy) [+ bias] [relu]
 * + res (res in dx's layout; it may be dx itself).  Only launches that take the slab kernels (sg_conv2d_planes_job kind 1) or
 * the thin 1x1 kernel (Cout <= 4, not the fp32 softmax head) do this; the others return SG_EUNSUPPORTED and launch nothing. */
int sg_conv2d_dgrad_acc(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy,
                        const void* w, const void* bias, void* dx, int flags, void* ws, size_t ws_bytes,
                        const void* res);

/* Conv2D kernel gradient dw[KH,KW,Cin,Cout] (and dbias[Cout] if non-null) = sum over N*Ho*Wo.
 * Deterministic split-K: partial slabs in ws, then a fixed-order reduce (no float atomics: bit-reproducible run
 * to run).  Runs as six bf16 MFMA passes (x6) when the geometry is stride 1 / "same" / W % 32 == 0, else on the
 * fp32 MFMA; 1x1 kernels with Cout <= 4 take a streaming reduction.  ws: sg_conv2d_wgrad_ws_bytes. */
size_t sg_conv2d_wgrad_ws_bytes(const sg_ctx* ctx, const sg_conv_desc* d);
int sg_conv2d_wgrad(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x,
                    const void* dy, void* dw, void* dbias, void* ws, size_t ws_bytes);

/* A training- or inference-mode BatchNormalization (+ReLU) applied to the convolution's INPUT while the kernel loads it (round 5):
 * the normalised tensor between `BatchNormalization -> [ReLU] -> Conv2D` (conv_bn_relu chains, train_model/DeepLabv3plus.py:424-429,
 * 476-480: the 512 x 512 x 32 tensors in front of the decoder's last 3x3 convolution and of the softmax head) is never written or
 * read.  x is the BatchNormalization's RAW input; the kernel evaluates bn_apply's own expression
 * [relu](fmaf((x - mean) * invstd, gamma, beta)) on every pixel inside the image (fp32 storage: the bits of the unfused pair).
 * infer: `invstd` points at the moving VARIANCE, invstd = rsqrtf(var + eps).  Covered launches: the thin 1x1 kernels (Cout <= 4) and
 * the patch kernels (3x3, stride 1, SAME, Cin 32 / 64; forward and filter gradient) - sg_conv2d_bn_in_supported; anything else
 * returns SG_EUNSUPPORTED.  The input gradient needs nothing (it is the BatchNormalization's output gradient). */
typedef struct sg_bn_in {
  const void* mean;
  const void* invstd;
  const void* gamma;
  const void* beta;
  int32_t relu, infer;
  float eps;
} sg_bn_in;
int sg_conv2d_bn_in_supported(const sg_ctx* ctx, int dtype, const sg_conv_desc* d);
int sg_conv2d_fwd_stats_bn(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w,
                           const void* bias, void* y, int flags, void* ws, size_t ws_bytes, void* stats, int* tiles_out,
                           const sg_bn_in* bn);
int sg_conv2d_wgrad_bn(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* dy,
                       void* dw, void* dbias, void* ws, size_t ws_bytes, const sg_bn_in* bn);

/* The input gradient of a pointwise convolution whose forward output feeds a training-mode BatchNormalization, with that layer's
 * backward APPLY evaluated in the kernel's A path (round 5; csrc/conv_pw.h, BNB form; SeparableConv2D -> BatchNormalization,
 * train_model/DeepLabv3plus.py:323-416).  dy is the gradient of the BatchNormalization's OUTPUT, bnb->x its raw input (= this
 * convolution's output), dgamma / dbeta its FINISHED column sums (sg_dwconv2d_dgrad_bnsums / sg_bn_train_bwd), rows the pixels
 * it normalises over.  The kernel computes dz = gamma invstd ((g - dbeta / rows) - xhat dgamma / rows) on the fly (g = dy where the
 * fused ReLU passed: the mask is bn_apply's own expression on x), multiplies it with the kernel and also stores it to bnb->dz for
 * the filter gradient - what sg_bn_train_bwd_apply + sg_conv2d_dgrad do in two launches and three more tensor passes.
 * Launches of the wide pointwise kernel only (sg_conv2d_dgrad_bnb_supported: 1x1, stride 1, dense fp32, >= 6144 rows, wide
 * enough); same order of products as sg_conv2d_dgrad, the applied gradient within rounding of sg_bn_train_bwd_apply's. */
typedef struct sg_bn_bwd_in {
  const void* x;
  const void* mean;
  const void* invstd;
  const void* gamma;
  const void* beta;     /* null without a fused ReLU */
  const void* dgamma;
  const void* dbeta;
  void* dz;
  int32_t relu;
  int64_t rows;
} sg_bn_bwd_in;
int sg_conv2d_dgrad_bnb_supported(const sg_ctx* ctx, int dtype, const sg_conv_desc* d);
int sg_conv2d_dgrad_bnb(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w, void* dx,
                        void* ws, size_t ws_bytes, const sg_bn_bwd_in* bnb);

/* Planes-in filter gradient (round 5).  The fp32 ("x6") filter gradient splits BOTH of its operands into three bf16 planes on
 * the VALU in every launch; where the planes already exist - the forward's activation planes of a long-K convolution, kept - the
 * kernel can take them as they are:
 *   sg_split_planes          x[rows][ld] fp32 (C channels used) -> planes[3][rows][C] bf16, the exact 3-way split of conv_x6.h
 *                            (a1 + a2 + a3 = x; the layout the planes-in forward kernel conv_x6w.h reads).
 *   sg_conv2d_wgrad_planes   dw[KH,KW,Cin,Cout] of the convolution `d` from x_planes[3][N*H*W][Cin] and dy_planes[3][N*Ho*Wo][Cout];
 *                            same products in the same order as sg_conv2d_wgrad on the fp32 tensors (bit-identical), no bias
 *                            gradient (sg_bias_grad).  ws: sg_conv2d_wgrad_planes_ws_bytes(ctx, d).  Stride 1, SAME, Wo % 32 = 0,
 *                            channels % 8 = 0, not a layer of the wide-pointwise / patch families; _supported() tells.
 * Replaces the filter gradient of the ASPP / SK / decoder 3x3 convolutions (train_model/DeepLabv3plus.py:219-229, 431-443). */
int sg_split_planes(sg_ctx* ctx, void* stream, const void* x, int64_t rows, int C, int ld, void* planes);
/* The forward / input gradient with the activation's planes handed in.  The long-K multi-tap launches (csrc/conv_x6w.h: the ASPP /
 * SK / decoder 3x3 convolutions) read their A operand as sg_split_planes' planes and otherwise split it themselves in every
 * launch; x_planes / dy_planes (may be null) = the planes of the very tensor passed as x / dy, dense [3][pixels][C].  The five
 * consumers of the ASPP input share ONE split this way, and a layer's dy planes serve its dgrad and its filter gradient.
 * Launches that take another kernel ignore the planes.  sg_conv2d_planes_in(d, dgrad): 1 if the launch reads planes. */
int sg_conv2d_planes_in(const sg_conv_desc* d, int dgrad);
int sg_conv2d_fwd_stats_ap(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w,
                           const void* bias, void* y, int flags, void* ws, size_t ws_bytes, void* stats, int* tiles_out,
                           const void* x_planes);
int sg_conv2d_dgrad_ap(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w,
                       const void* bias, void* dx, int flags, void* ws, size_t ws_bytes, const void* dy_planes);
int sg_conv2d_wgrad_planes_supported(const sg_ctx* ctx, const sg_conv_desc* d);
size_t sg_conv2d_wgrad_planes_ws_bytes(const sg_ctx* ctx, const sg_conv_desc* d);
int sg_conv2d_wgrad_planes(sg_ctx* ctx, void* stream, const sg_conv_desc* d, const void* x_planes, const void* dy_planes, void* dw,
                           void* ws, size_t ws_bytes);

/* Bias gradient db[C] = column sums of dy[rows][C] (pixel stride ld, 0 = C); fixed-order two-stage reduce.
 * Used for the bias of Conv2DTranspose (v3plus.py:328,335; scse.py:71-89; res34.py:144), whose kernel
 * gradient comes from sg_conv2d_wgrad with the operand roles swapped. */
size_t sg_bias_grad_ws_bytes(const sg_ctx* ctx, int64_t rows, int C);
int sg_bias_grad(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, int ld, const void* dy,
                 void* dbias, void* ws, size_t ws_bytes);

/* Depthwise 3x3 (the first half of SeparableConv2D, depth_multiplier 1, no bias): v3plus.py:187-278.
 * d->Cout must equal d->Cin; w is [KH][KW][C].  pre_relu folds the Activation('relu') that precedes the
 * layer (v3plus.py:204,225,242,...) into the gather: y = dw(relu(x)); its dgrad masks by x > 0. */
int sg_dwconv2d_fwd(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x,
                    const void* w, void* y, int pre_relu);
int sg_dwconv2d_dgrad(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy,
                      const void* w, const void* x_for_mask, void* dx, int pre_relu);
/* The same with a gradient already collected for the layer's input added to the result: dx = dgrad(dy) [masked] + res (res may
 * be dx itself).  The input of an Xception block feeds the block AND its residual add; the block's input gradient then leaves
 * this kernel complete instead of through a separate add (v3plus.py's `add([residual, shortcut])` blocks).  Stride-1 3x3,
 * W % 4 == 0, C % 4 == 0, 16-byte aligned tensors, else SG_EUNSUPPORTED. */
int sg_dwconv2d_dgrad_acc(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy,
                          const void* w, const void* x_for_mask, void* dx, int pre_relu, const void* res);
/* sg_dwconv2d_dgrad[_acc] of a SeparableConv2D whose input is the output of a training-mode BatchNormalization(+ReLU) with no
 * other consumer (conv_bn_relu / the Xception blocks, train_model/DeepLabv3plus.py:323-416,424-429): dx is then that layer's
 * output gradient, and the kernel also produces what the layer's backward sums over the pixels - dbeta = sum g, dgamma = sum
 * g * xhat, g = dx [masked by the fused ReLU, recomputed from bn_x as sg_bn_train_bwd does], xhat = (bn_x - mean) * invstd,
 * bn_x the layer's RAW input (dense [N,H,W,C]) - in its epilogue: one more tensor read here instead of the two-read
 * reduction pass of sg_bn_train_bwd.  Follow with sg_bn_train_bwd_apply.  res (may be NULL) as in sg_dwconv2d_dgrad_acc.
 * Partial sums are added in a fixed order (deterministic); they differ from sg_bn_train_bwd's in rounding only.  Stride-1 3x3,
 * W % 4 == 0, C % 4 == 0, 16-byte aligned tensors, else SG_EUNSUPPORTED.  Workspace: sg_dwconv2d_dgrad_bnsums_ws_bytes. */
size_t sg_dwconv2d_dgrad_bnsums_ws_bytes(const sg_ctx* ctx, const sg_conv_desc* d);
int sg_dwconv2d_dgrad_bnsums(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w,
                             const void* x_for_mask, void* dx, int pre_relu, const void* res, const void* bn_x,
                             const void* bn_mean, const void* bn_invstd, const void* bn_gamma, const void* bn_beta, int bn_relu,
                             void* dgamma, void* dbeta, void* ws, size_t ws_bytes);
/* The depthwise convolution of a SeparableConv2D whose input is BatchNormalization(+ReLU) of a tensor x_raw, in training
 * mode, with that normalisation applied as the window is loaded - fmaf((x - mean) * invstd, gamma, beta), then max(., 0) if
 * relu - so that the normalised tensor is never written (the pattern BatchNormalization -> Activation('relu') ->
 * SeparableConv2D of the Xception middle flow, predict_model/v3plus.py:170-260).  x is x_raw; mean / invstd are the batch
 * statistics sg_bn_train_fwd / sg_bn_train_fwd_tiles saved.  Stride-1 3x3, W % 4 == 0, C % 4 == 0, 16-byte aligned tensors
 * only (SG_EUNSUPPORTED otherwise: apply sg_bn_apply and call the plain entry points).  sg_dwconv2d_wgrad_bn is the filter
 * gradient of the same layer (workspace: sg_dwconv2d_wgrad_ws_bytes); the input gradient is sg_dwconv2d_dgrad without a mask
 * (the ReLU's mask belongs to sg_bn_train_bwd, which recomputes it from x_raw). */
int sg_dwconv2d_fwd_bn(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w, void* y,
                       const void* gamma, const void* beta, const void* mean, const void* invstd, int relu);
int sg_dwconv2d_wgrad_bn(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* dy, void* dw,
                         const void* gamma, const void* beta, const void* mean, const void* invstd, int relu, void* ws,
                         size_t ws_bytes);
size_t sg_dwconv2d_wgrad_ws_bytes(const sg_ctx* ctx, const sg_conv_desc* d);
int sg_dwconv2d_wgrad(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x,
                      const void* dy, void* dw, int pre_relu, void* ws, size_t ws_bytes);

/* Dense on [rows,in] x [in,out] (+bias): bam.py channel_gate, res34.py:94,98.  Tiny GEMMs. */
int sg_dense_fwd(sg_ctx* ctx, void* stream, int dtype, int rows, int in, int out, const void* x,
                 const void* w, const void* bias, void* y, int flags);

/* Weight operands prepared ONCE per optimiser step instead of once per launch.  The matrix-pipe convolution kernels
 * read their weights as bf16 planes ([n][k] rows, or fragment-major for the low-channel patch kernel); without this,
 * every forward / dgrad launch first converts its kernel into the workspace (about 200 small launches per DeepLabv3+
 * step).  With it the host keeps one planes arena per model:
 *   sg_conv2d_planes_job   geometry of the planes of one convolution (dgrad = 0: forward / Conv2DTranspose backward-dx,
 *                          1: dgrad / Conv2DTranspose forward) into *out, their size into *bytes; out->kind == 0 means
 *                          this convolution does not take a prepared-planes kernel (thin, odd shapes).  The caller
 *                          fills w_off (element offset of the fp32 kernel inside the weight arena), out_off (16-byte
 *                          aligned byte offset inside the planes arena) and block0 (running sum of nblocks), copies
 *                          the job table to the device, and
 *   sg_prepare_planes      converts every job of the table in ONE launch (after Adam, after set_weights);
 *   then passes `planes_arena + out_off` as `ws` with `ws_bytes = SG_WS_PREPARED` to sg_conv2d_fwd_ws / _stats /
 *   sg_conv2d_dgrad.  A launch whose operands turn out not to qualify (unaligned, strided) returns SG_EINVAL.
 * The planes depend on the arithmetic mode (sg_get_conv_x6) and the storage dtype: prepare again after changing either. */
#define SG_WS_PREPARED ((size_t)-1)
typedef struct sg_planes_job {
  int64_t w_off, out_off;
  int32_t kind; /* 0 none, 1 bf16 operand planes of the 128-wide-tile kernels, 2 fragment-major (patch kernel), 3 planes of the
                 * wide pointwise kernel (Npad a multiple of 384); layout of 1 and 3: see kd */
  int32_t K, N, Kpad, Npad, Ck, Ckp, s_tap, s_k, s_n, npl;
  int32_t block0, nblocks;
  int32_t kd; /* kinds 1 and 3: k-block depth of the layout [npl][Kpad / kd][Npad][kd] (32 / 64 for the 128-wide tiles, 16 / 64
               * for the wide pointwise kernel): the kd reduction indices a kernel stages per row are contiguous, and so are the
               * rows, so a staged slab is one contiguous range of whole cache lines; 0 = row-major [npl][Npad][Kpad] */
} sg_planes_job;
int sg_get_conv_x6(void);
int sg_conv2d_planes_job(const sg_ctx* ctx, int dtype, const sg_conv_desc* d, int dgrad, sg_planes_job* out, size_t* bytes);
int sg_prepare_planes(sg_ctx* ctx, void* stream, const void* w_arena, void* planes_arena, const sg_planes_job* jobs_dev,
                      int njobs, int total_blocks);

/* ------------------------------------------------------------------------------------- normalisation
 * BatchNormalization, Keras defaults (momentum .99, eps 1e-3): v3plus.py:174 ... (92 sites), 2-D after
 * Dense in bam.py / res34.py:95,99.  rows = N*H*W (or N for 2-D), C channels innermost.
 * train_fwd: batch statistics (biased var for normalisation), y = gamma*xhat+beta (ReLU if relu!=0),
 *   saves mean[C], invstd[C]; updates moving_mean / moving_var in place (unbiased var iff unbiased_update,
 *   which Keras' fused 4-D path uses; the 2-D path passes 0).  ws: sg_bn_ws_bytes(rows, C).
 * train_bwd: given dy (and y when relu was fused), produces dx, dgamma[C], dbeta[C].
 * infer: y = gamma*(x-mm)/sqrt(mv+eps)+beta (ReLU optional). */
size_t sg_bn_ws_bytes(const sg_ctx* ctx, int64_t rows, int C);
int sg_bn_train_fwd(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* x,
                    const void* gamma, const void* beta, void* moving_mean, void* moving_var, void* y,
                    void* save_mean, void* save_invstd, float momentum, float eps, int relu,
                    int unbiased_update, void* ws, size_t ws_bytes);
/* Residual add with the BatchNormalization of its operands applied on the way (v3plus.py's Xception blocks:
 * add([BatchNormalization(branch), shortcut]) and add([BN(branch), BN(1x1 shortcut)])):
 * y = [relu](f_a(a) + f_b(b)), f = gamma * (x - mean) * invstd + beta for an operand whose four parameter pointers are given
 * (infer != 0: `invstd` holds the moving VARIANCE and eps is added under the root), the identity for null pointers.  The
 * normalised tensor is never materialised.  a_relu / b_relu: that operand's BatchNormalization is followed by a ReLU before the
 * add (res34.py's blocks).  C % 4 == 0 and 16-byte aligned tensors, else SG_EUNSUPPORTED. */
int sg_add2_bn(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* a, const void* b,
               const void* a_mean, const void* a_invstd, const void* a_gamma, const void* a_beta, const void* b_mean,
               const void* b_invstd, const void* b_gamma, const void* b_beta, void* y, int relu, int infer, float eps,
               int a_relu, int b_relu);
/* With a fused ReLU the backward needs the mask [y > 0].  Given beta it is recomputed from x (the same
 * fmaf((x-mean)*invstd, gamma, beta) the forward evaluated), which saves reading y in both passes; with
 * beta == NULL the mask is read from y. */
int sg_bn_train_bwd(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* x,
                    const void* y, const void* dy, const void* gamma, const void* beta, const void* save_mean,
                    const void* save_invstd, void* dx, void* dgamma, void* dbeta, int relu, void* ws,
                    size_t ws_bytes);
/* The second pass of sg_bn_train_bwd on its own: dx from FINISHED column sums dgamma / dbeta (sg_dwconv2d_dgrad_bnsums wrote
 * them).  relu: the fused ReLU's mask is recomputed from x (beta required). */
int sg_bn_train_bwd_apply(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* x, const void* dy,
                          const void* gamma, const void* beta, const void* save_mean, const void* save_invstd,
                          const void* dgamma, const void* dbeta, void* dx, int relu);
int sg_bn_infer(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* x,
                const void* gamma, const void* beta, const void* moving_mean, const void* moving_var,
                void* y, float eps, int relu);

/* ------------------------------------------------------------------------------------- element-wise
 * Activation('relu'|'sigmoid'), n-ary add, channel concat / slice copies. act: 0 relu, 1 sigmoid. */
#define SG_ACT_RELU 0
#define SG_ACT_SIGMOID 1
int sg_act_fwd(sg_ctx* ctx, void* stream, int dtype, int act, int64_t n, const void* x, void* y);
/* dx = dy * act'(.) expressed through the OUTPUT y (relu: y>0; sigmoid: y(1-y)); accumulate!=0 adds into dx */
int sg_act_bwd(sg_ctx* ctx, void* stream, int dtype, int act, int64_t n, const void* y, const void* dy,
               void* dx, int accumulate);
/* y = sum_i xs[i]  (k <= 8 device pointers passed by value in a host array); relu!=0 fuses the
 * Activation('relu') that follows a residual add (res34.py:43-44, hrnet.py:35-36). */
int sg_add_n(sg_ctx* ctx, void* stream, int dtype, int k, const void* const* xs, int64_t n, void* y,
             int relu);
/* strided channel copy: dst[r, dst_off + c] (pixel stride dst_ld) (+)= src[r, src_off + c] (stride src_ld),
 * c < C.  Used for concat forward (tf.concat, v3plus.py:306,323,...) and its backward (slice). */
int sg_copy_channels(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* src,
                     int src_ld, int src_off, void* dst, int dst_ld, int dst_off, int accumulate);
/* softmax over the last axis of size 2 (Conv2D(num_classes, 1, activation='softmax'), v3plus.py:345) and
 * its backward dz = p * (dp - sum_k dp_k p_k). */
int sg_softmax2_fwd(sg_ctx* ctx, void* stream, int dtype, int64_t rows, const void* z, void* p);
int sg_softmax2_bwd(sg_ctx* ctx, void* stream, int dtype, int64_t rows, const void* p, const void* dp,
                    void* dz);
/* Softmax(axis=-2) over the B stacked SK branch logits z[N,B,C] (v3plus.py:120-121), and backward. */
int sg_softmax_branch_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int B, int C, const void* z,
                          void* p);
int sg_softmax_branch_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int B, int C, const void* p,
                          const void* dp, void* dz);

/* Broadcast multiply  y[n,h,w,c] = x[n,h,w,c] * g  with g either per (n,c) ([N,C], cSE / SK weights /
 * attention_demo: v3plus.py:128-132,159; res34.py:104) or per (n,h,w) ([N,HW], sSE: v3plus.py:145).
 * mode 0 = channel gate g[N,C]; mode 1 = spatial gate g[N,HW].  accumulate!=0: y += x*g (SK fusion sum).
 * bwd: dx (+)= dy*g ; dg = reduce(dy*x) over HW (mode 0) or over C (mode 1).  ws for mode-0 dg partials. */
int sg_bcast_mul_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int64_t HW, int C, int mode,
                     const void* x, const void* g, void* y, int accumulate);
size_t sg_bcast_mul_bwd_ws_bytes(const sg_ctx* ctx, int N, int64_t HW, int C, int mode);
int sg_bcast_mul_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int64_t HW, int C, int mode,
                     const void* x, const void* g, const void* dy, void* dx, void* dg, int accumulate_dx,
                     void* ws, size_t ws_bytes);
/* scSE combine  y = x * (sigmoid(s) + sigmoid(c))  with s[N,HW] spatial logits and c[N,C] channel logits
 * (sSE_block + cSE + tf.add, v3plus.py:141-167 / scse.py:20-46); backward gives dx (from the product
 * only), ds[N,HW], dc[N,C]. */
int sg_scse_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int64_t HW, int C, const void* x,
                const void* s, const void* c, void* y);
size_t sg_scse_bwd_ws_bytes(const sg_ctx* ctx, int N, int64_t HW, int C);
int sg_scse_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int64_t HW, int C, const void* x,
                const void* s, const void* c, const void* dy, void* dx, void* ds, void* dc, void* ws,
                size_t ws_bytes);
/* BAM combine  y = x + x * sigmoid(mc[n,c] + ms[n,hw])  (BAM_attention, bam.py:57-71); backward. */
int sg_bam_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int64_t HW, int C, const void* x,
               const void* mc, const void* ms, void* y);
size_t sg_bam_bwd_ws_bytes(const sg_ctx* ctx, int N, int64_t HW, int C);
int sg_bam_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int64_t HW, int C, const void* x,
               const void* mc, const void* ms, const void* dy, void* dx, void* dmc, void* dms, void* ws,
               size_t ws_bytes);

/* ------------------------------------------------------------------------------------------ pooling
 * MaxPooling2D (3x3 s2 'same' v3plus.py:192; 2x2 s2 scse.py:54-66, res34.py:152,154; 2x2 window stride 4
 * res34.py:153).  pad_t/pad_l = TF SAME pad before (0 for valid); padded cells are -inf.  The backward
 * recomputes the argmax (first max in window scan order wins) and scatters dy into dx (dx zeroed first). */
int sg_maxpool_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int k, int stride,
                   int pad_t, int pad_l, int Ho, int Wo, const void* x, void* y);
int sg_maxpool_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int k, int stride,
                   int pad_t, int pad_l, int Ho, int Wo, const void* x, const void* y, const void* dy,
                   void* dx);
/* Training form of the same layer: the forward also records, one byte per output element in idx[N,Ho,Wo,C], which cell
 * (a * k + b, scan order; k <= 15) of the window held the first maximum, and the backward routes dy by that byte alone
 * (reads dy and idx, no x / y): identical dx to sg_maxpool_bwd. */
int sg_maxpool_fwd_idx(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int k, int stride,
                       int pad_t, int pad_l, int Ho, int Wo, const void* x, void* y, void* idx);
int sg_maxpool_bwd_idx(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int k, int stride,
                       int pad_t, int pad_l, int Ho, int Wo, const void* dy, const void* idx, void* dx);
/* AveragePooling2D(pool_size=k) (stride k, valid; v3plus.py:302) and GlobalAveragePooling2D (k = H = W):
 * y[N,H/kh,W/kw,C] (a wavefront/LDS two-stage reduce over the window; ws from sg_avgpool_ws_bytes);
 * backward spreads dy/(kh*kw).  accumulate!=0 adds into dx. */
size_t sg_avgpool_ws_bytes(const sg_ctx* ctx, int N, int H, int W, int C, int kh, int kw);
int sg_avgpool_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int kh, int kw,
                   const void* x, void* y, void* ws, size_t ws_bytes);
int sg_avgpool_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int kh, int kw,
                   const void* dy, void* dx, int accumulate);
/* UpSampling2D(size=s) nearest (v3plus.py:100,304,321,341; bam.py:332; hrnet.py:105-158):
 * y[N,H*s,W*s,C] with pixel stride y_ld (0 = C) so it can write straight into a concat buffer;
 * backward sums each s x s block. */
int sg_upsample_nearest_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int sh,
                            int sw, const void* x, void* y, int y_ld);
int sg_upsample_nearest_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int sh,
                            int sw, const void* dy, int dy_ld, void* dx, int accumulate);

/* -------------------------------------------------------------------------------- loss / metrics / Adam
 * The three losses of train_model/DeepLabv3plus.py:490-527 on softmax probabilities p[rows,2] and
 * y_true[rows,4] = (one-hot 2, f_edge weight, p_edge weight):
 *   kind 0 "binary_crossentropy": a_c = y_c,                      no focal term
 *   kind 1 focal_loss:            a_c = .5 * y_c,                 (1-p_c)^2
 *   kind 2 edge_focal_loss:       a_c = alpha_c * w_c * y_c,      (1-p_c)^2, alpha = (.35,.65), w = y_true[2:4]
 *   L = -(1/rows) * sum_rows sum_c a_c * f(p_c) * log(p_c + 1e-7).
 * fwd writes the scalar loss to loss_out[0] (fixed-order two-stage reduce; ws: sg_loss_ws_bytes(rows)).
 * bwd writes dL/dp[rows,2] scaled by grad_scale (=1 for the plain mean). y_cols = 2 or 4 (row length). */
#define SG_LOSS_CE2 0
#define SG_LOSS_FOCAL 1
#define SG_LOSS_EDGE_FOCAL 2
size_t sg_loss_ws_bytes(const sg_ctx* ctx, int64_t rows);
int sg_loss_fwd(sg_ctx* ctx, void* stream, int kind, int64_t rows, int y_cols, const void* p,
                const void* y_true, void* loss_out, void* ws, size_t ws_bytes);
int sg_loss_bwd(sg_ctx* ctx, void* stream, int kind, int64_t rows, int y_cols, const void* p,
                const void* y_true, void* dp, float grad_scale);
/* PA / IoU / MIoU / F1_score share one confusion count (DeepLabv3plus.py:530-623): argmax of p (ties ->
 * class 0) vs argmax of y_true[:, :2]; out[4] = int64 {TP, TN, FP, FN}, accumulated (caller zeroes). */
int sg_confusion_counts(sg_ctx* ctx, void* stream, int64_t rows, int y_cols, const void* p,
                        const void* y_true, void* out_i64x4);
/* Keras-2 Adam (compile(optimizer='adam'), DeepLabv3plus.py:835) on one flat fp32
 * parameter arena: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_t * m / (sqrt(v) + eps) with
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) precomputed by the host.  g is scaled by grad_scale first (1/world). */
int sg_adam_step(sg_ctx* ctx, void* stream, int64_t n, void* w, void* m, void* v, const void* g,
                 float lr_t, float beta1, float beta2, float eps, float grad_scale);
/* The same step with lr_t read from device memory (one float): for a training step captured into a hipGraph, whose
 * bias-corrected learning rate lr * sqrt(1 - beta2^t) / (1 - beta1^t) changes at every replay while the captured kernel
 * arguments cannot (tf.keras Adam, optimizer_v2/adam.py: the reference compiles with optimizer='adam',
 * train_model/DeepLabv3plus.py:835). */
int sg_adam_step_lr(sg_ctx* ctx, void* stream, int64_t n, void* w, void* m, void* v, const void* g, const void* lr_t_dev,
                    float beta1, float beta2, float eps, float grad_scale);

/* Label channels of train_data_gen (DeepLabv3plus.py:70-100) from label[N,H,W] (gray/255, float): y[N,H,W,4] =
 * (1-fg, fg, f_edge, p_edge); fg = 1 iff label == 1.0 (to_categorical truncation); `iterations` (5 in the
 * reference) erosions / dilations with a 3x3 kernel = (2*iterations+1)^2 min / max box ignoring out-of-image
 * cells; p_edge = 2 where label - erode == 1 else 1; f_edge = 2 where dilate - label == 1 else 1. */
int sg_edge_labels(sg_ctx* ctx, void* stream, int N, int H, int W, int iterations, const void* label,
                   void* y_true4);
/* cv.resize(img, (OW, OH)) with the default INTER_LINEAR on 8-bit pixels - decode_img / decode_lbel of
 * train_model/DeepLabv3plus.py:35,45 for tiles that are not already 512 x 512: src_u8[N,H,W,C] -> dst_u8[N,OH,OW,C] in
 * OpenCV's fixed-point arithmetic (11-bit coefficients, the vectorised vertical pass; an exact 2x downscale is the fast
 * INTER_AREA, as resize() substitutes it).  Bit-exact against oracle/input_pipeline.py:resize_linear_u8. */
int sg_resize_linear_u8(sg_ctx* ctx, void* stream, int N, int H, int W, int C, const void* src_u8, int OH, int OW,
                        void* dst_u8);

/* ------------------------------------------------------------------------------------ inference tail
 * predict.py:110-114: mask = argmax(p) (ties -> 0) added as int8 into the canvas window at (y0,x0) of a
 * [CH,CW] canvas; tile is [TH,TW] probabilities p[TH*TW,2]. */
int sg_argmax_accumulate_i8(sg_ctx* ctx, void* stream, const void* p, int TH, int TW, void* canvas,
                            int CH, int CW, int y0, int x0);
/* model_fuse.py:315,323: out = 255 where sum_i (masks[i] // 255) >= k else 0; masks are u8 [n]. */
int sg_vote_ge(sg_ctx* ctx, void* stream, int nmasks, const void* const* masks, int64_t n, int k,
               void* out_u8);
/* dst[i] = (dst_dtype) src[i]: fp32 <-> bf16 conversion (round to nearest even; NaN stays NaN), e.g. the fp32 input
 * tiles of predict.py:109 / the generator of DeepLabv3plus.py:100 entering a bf16 model. */
int sg_cast(sg_ctx* ctx, void* stream, int src_dtype, int dst_dtype, int64_t n, const void* src, void* dst);
/* Mask clean-up around the vote (model_fuse.py:9-218, 271-350) on label maps instead of OpenCV contours; what each
 * OpenCV call means at mask level is restated in oracle/cleanup.py.
 *   sg_mask_objects   fill_and_delete (model_fuse.py:9-32) of a u8 mask [H][W] (non-zero = building): holes filled (4-connected
 *                     background not reaching the image border), 8-connected objects numbered, per object the bounding
 *                     box and 2 x cv.contourArea (2*N4 + N3 over the 2x2 pixel quads) into table[k][8] = {root pixel,
 *                     area2, x0, y0, x1, y1, kept, 0}, kept = area2 > min_area2 (2000 for the reference's "area <= 1000"),
 *                     labels[H][W] = object number or -1, *count = number of objects (may exceed max_objs: enlarge and
 *                     repeat), kept_mask (optional) = the reference's gray_label.  ws: sg_mask_objects_ws_bytes(H, W).
 *   sg_mask_split     eroede_dilate_process (model_fuse.py:65-115, 173-218) for the listed objects, one workgroup each:
 *                     1x5 and 5x1 erosions (5 iterations), pieces of contourArea <= piece_area2 / 2 dropped (1000 for the
 *                     reference's 500), every kept piece dilated and filled on its own; the object is kept, replaced by
 *                     its pieces or dropped exactly as the reference's if-chain decides; 255 is written into out (which
 *                     the caller zeroes).  offsets[i] = first 32-bit word of object i's scratch inside ws, each object
 *                     needing sg_mask_split_words(H, W, its box) words. */
size_t sg_mask_objects_ws_bytes(int H, int W);
int sg_mask_objects(sg_ctx* ctx, void* stream, int H, int W, const void* mask_u8, int min_area2, void* ws, size_t ws_bytes,
                    void* labels_i32, void* table_i32, int max_objs, void* count_i32, void* kept_mask_u8);
int64_t sg_mask_split_words(int H, int W, int x0, int y0, int x1, int y1);
int sg_mask_split(sg_ctx* ctx, void* stream, int H, int W, const void* labels_i32, const void* table_i32, const void* objs_i32,
                  const void* offsets_i64, int nobj, int piece_area2, void* ws, void* out_u8);
/* dst[i] = (float)src[i] / div - sub: decode_img's `np.array(img, np.float32) / 127.5 - 1` and decode_lbel's `/ 255`
 * (train_model/DeepLabv3plus.py:36-37, 48) on the GPU, so the input pipeline ships uint8 pixels over PCIe (a quarter of
 * the fp32 bytes) and converts on the device; bit-identical to the numpy float32 arithmetic. */
int sg_u8_to_f32(sg_ctx* ctx, void* stream, int64_t n, const void* src_u8, void* dst_f32, float div, float sub);
/* fill n floats with value (workspace / gradient zeroing without leaving the stream) */
int sg_fill_f32(sg_ctx* ctx, void* stream, void* p, int64_t n, float value);
/* Profiling aid: launches an empty one-thread kernel named sg_trace_mark_kernel<tag, end> on `stream`, so that a
 * rocprofv3 --kernel-trace of a whole training step shows where a group of launches begins (end = 0) and ends
 * (end = 1).  tag 0 = the north_star dilated-convolution set, tag 1 = any GEMM convolution (scripts/trace_dilated.py). */
int sg_trace_mark(sg_ctx* ctx, void* stream, int tag, int end);
/* p[i] *= a in place (1/nranks on the summed loss and on the summed BatchNorm moving statistics of the replicas) */
int sg_scale_f32(sg_ctx* ctx, void* stream, void* p, int64_t n, float a);

/* ------------------------------------------------------------------------------------- data parallelism
 * The reference trains on one device (train_model/DeepLabv3plus.py:844 fit_generator; no tf.distribute anywhere):
 * this is the new exchange step of SURVEY.md 8e — one process per GPU, identical replicas, rank r trains on its
 * own tiles, the flat gradient arena is summed over the ranks once per step (RCCL over xGMI), 1/nranks folded
 * into sg_adam_step's grad_scale.  RCCL is resolved with dlopen at the first call (SG_EUNSUPPORTED if absent).
 *   sg_comm_probe       any rank, local: can this process load a usable RCCL (dlopen + ncclGetVersion; *version_out may
 *                       be null)?  No bootstrap root, thread or socket is created - what every rank other than 0 calls
 *                       before the ranks agree to enter sg_comm_init.
 *   sg_comm_unique_id   rank 0 only: writes SG_COMM_ID_BYTES opaque bytes the caller hands to every rank.
 *   sg_comm_init        collective over the nranks processes; binds to `device`.
 *   sg_comm_allreduce_sum  in-place sum of buf[count] (SG_F32 / SG_BF16 / SG_I64) over the ranks, queued on
 *                       `stream`; no host synchronisation (the host orders it against the compute stream with
 *                       events, building_detection_amd/dist.py).
 *   sg_comm_destroy     frees the communicator. */
#define SG_COMM_ID_BYTES 128
typedef struct sg_comm sg_comm;
int sg_comm_probe(int* version_out);
int sg_comm_unique_id(void* id_out);
int sg_comm_init(const void* id, int rank, int nranks, int device, sg_comm** out);
int sg_comm_allreduce_sum(sg_comm* comm, void* stream, int dtype, void* buf, int64_t count);
int sg_comm_rank(const sg_comm* comm);
int sg_comm_nranks(const sg_comm* comm);
int sg_comm_destroy(sg_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* SEGENGINE_H_ */
