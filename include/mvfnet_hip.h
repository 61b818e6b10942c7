/* mvfnet_hip.h -- C ABI of libmvfnet_hip.so (MI355X / gfx950 only).
 *
 * The reference (whwu95/MVFNet) is 100 % Python and has NO FFI boundary of its own: its hot path calls
 * torch.nn modules.  This header is therefore this build's own boundary; each entry point
 * cites the reference code whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensors' data_ptr()); the library
 *     never allocates or frees user-visible memory; scratch comes in through (ws, ws_bytes) with a
 *     *_workspace_bytes() query;
 *   - every entry takes the HIP stream as an opaque void* (hipStream_t; pass
 *     torch.cuda.current_stream().cuda_stream) and is asynchronous on it;
 *   - return value: 0 = OK, <0 = MVF_E*; never throws; mvf_last_error() = thread-local message;
 *   - no global mutable state besides the thread-local error string: safe from several host threads/streams;
 *   - dtype = storage type of activations (weights / BN parameters / statistics are always fp32,
 *     accumulation is always fp32).
 */
#ifndef MVFNET_HIP_H
#define MVFNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVF_ABI_VERSION 2    /* 2: mvf_conv_desc_t.x_c0 */

enum { MVF_OK = 0, MVF_EINVAL = -1, MVF_ESHAPE = -2, MVF_EWS = -3, MVF_EHIP = -4, MVF_EUNSUPPORTED = -5 };
enum { MVF_F32 = 0, MVF_BF16 = 1 };
enum { MVF_NCHW = 0, MVF_NHWC = 1 };                       /* memory order of the (N*T, C, H, W) tensor */
enum { MVF_VIEW_T = 1, MVF_VIEW_H = 2, MVF_VIEW_W = 4 };   /* mode 'T' = 1, 'TH' = 3, 'THW' = 7 (MVF.py:112-129) */

int mvf_abi_version(void);
const char* mvf_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * MVF-proper: everything in MVF.forward except self.net (codes/models/modules/MVF.py:104-137).
 * Tensor x is (nt, c, h, w) in `layout`; clips are runs of n_segment consecutive images (MVF.py:107-109).
 * Only channels [0, cs) are read/written by the stencil; channels >= cs pass through (MVF.py:110,135).
 * Tap weights are fp32 [cs][3]: tap j multiplies the element at offset (j-1) along the view axis
 * (= shift_conv.weight (cs,1,3,1,1) / h_conv.weight (cs,1,1,3,1) / w_conv.weight (cs,1,1,1,3) flattened).
 * share=True (MVF.py:114-116,125-126): pass w_h = w_w = w_t and add the three weight grads.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t nt, c, h, w;      /* tensor dims                                      */
    int32_t n_segment;        /* T: frames per clip, nt % n_segment == 0          */
    int32_t cs;               /* num_shift_channel = int(c * alpha), 0 < cs <= c  */
    int32_t mode;             /* MVF_VIEW_* bitmask, T bit mandatory              */
    int32_t layout;           /* MVF_NCHW | MVF_NHWC                              */
    int32_t dtype;            /* MVF_F32 | MVF_BF16                               */
} mvf_desc_t;

/* Inference / eval-BN forward (MVF.py:118-134 with BatchNorm3d in eval mode folded by the caller:
 * bn_scale = gamma/sqrt(running_var+eps), bn_shift = beta - running_mean*bn_scale).
 * bn_scale == NULL  <=>  use_hs=False: no BN, no activation (MVF.py:131-134).
 * out == x: in place on the slice (allowed; nothing else is touched).  out != x: channels >= cs are copied. */
int mvf_fwd_infer(const mvf_desc_t* d, const void* x, void* out,
                  const float* w_t, const float* w_h, const float* w_w,
                  const float* bn_scale, const float* bn_shift, void* stream);

/* Engine variant (MVF_NHWC only): writes ONLY the cs slice, compactly, to out_slice (nt, h, w, cs).  The wrapped
 * 1x1 conv then reads channels [0, cs) from it and the rest from x (mvf_conv_desc_t.split_c): the pass-through
 * channels are never copied and x stays intact for the residual branch (replaces cat + transpose + contiguous,
 * MVF.py:135-137). */
int mvf_fwd_infer_slice(const mvf_desc_t* d, const void* x, void* out_slice,
                        const float* w_t, const float* w_h, const float* w_w,
                        const float* bn_scale, const float* bn_shift, void* stream);

/* Training forward: BatchNorm3d with batch statistics over (n,t,h,w) (biased variance), running-stat
 * update (momentum, unbiased variance; torch defaults used at MVF.py:69), then hard-swish.
 * save_mean / save_invstd (fp32 [cs]) are outputs kept for mvf_bwd.  running_* may be NULL. */
size_t mvf_fwd_train_workspace_bytes(const mvf_desc_t* d);
int mvf_fwd_train(const mvf_desc_t* d, const void* x, void* out,
                  const float* w_t, const float* w_h, const float* w_w,
                  const float* gamma, const float* beta, float eps, float momentum,
                  float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                  void* ws, size_t ws_bytes, void* stream);

/* Backward of MVF-proper (the reference relies on autograd; formulas in SURVEY.md Appendix B).
 * g = dL/d(out) (full tensor), x = the forward input.  training != 0: batch-stat BN backward using
 * save_mean/save_invstd; training == 0: eval BN (pass mean = running_mean, invstd = 1/sqrt(var+eps)).
 * gamma == NULL <=> use_hs=False.  dx == g allowed (in place on the slice); otherwise channels >= cs
 * are copied from g.  dw_* are fp32 [cs][3], dgamma/dbeta fp32 [cs]; all are overwritten. */
size_t mvf_bwd_workspace_bytes(const mvf_desc_t* d);
int mvf_bwd(const mvf_desc_t* d, const void* g, const void* x,
            const float* w_t, const float* w_h, const float* w_w,
            const float* gamma, const float* beta, const float* mean, const float* invstd, int training,
            void* dx, float* dw_t, float* dw_h, float* dw_w, float* dgamma, float* dbeta,
            void* ws, size_t ws_bytes, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Conv / BN / ReLU stack of the backbone (codes/models/backbones/resnet.py:208-244 Bottleneck.forward,
 * :479-494 ResNet.forward) as implicit-GEMM on the matrix cores.  Activations are channels-last
 * (N, H, W, C); weights are pre-packed [cout][kh][kw][cin] in the activation dtype with the eval-mode
 * BatchNorm scale folded in (mvf_pack_conv_weight), the BN shift arrives as `bias`.
 *   y = act( conv(x, w) + bias [+ residual] )
 * replaces nn.Conv2d -> BatchNorm2d(eval) -> [+= identity] -> ReLU  (resnet.py:213-242).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n, h, w;          /* input: images, height, width                                        */
    int32_t cin, cout;        /* channels contracted per tap / output channels                       */
    int32_t kh, kw, stride, pad;
    int32_t ho, wo;           /* output height / width (validated against the formula)               */
    int32_t x_pix_stride;     /* elements between consecutive input pixels (>= cin; = C of x)        */
    int32_t dtype;            /* MVF_F32 | MVF_BF16 (storage of x, w, residual, y)                   */
    int32_t relu;             /* apply ReLU in the epilogue                                          */
    int32_t split_c;          /* 0, or: channels [0, split_c) are read from x2 instead of x (the    */
    int32_t x2_pix_stride;    /*   compact MVF slice, pitch x2_pix_stride); 1x1 convs only           */
    int32_t in_dil;           /* 0/1, or s > 1: x is read as if zero-upsampled by s (data-gradient of a       */
                              /*   stride-s conv: y = dgrad needs stride == 1 here; ho,wo up to (h-1)*s+1+... ) */
    int32_t res_c0;           /* the residual is added to output channels >= res_c0 only (multiple of 4; 0 = all).   */
                              /*   MVF block backward: channels [0, cs) of the conv1 data gradient go through the    */
                              /*   transposed stencil first, which adds their share (mvf_nhwc_stencil addend)        */
    int32_t x_c0;             /* [r5] 0, or with split_c > 0: x holds channels [split_c, cin) at column (channel - x_c0) of its rows  */
                              /*   (x_pix_stride >= cin - x_c0): the contraction runs over the CONCATENATION [x2 | x] of two compact   */
                              /*   tensors (x_c0 = split_c) instead of over x with its first split_c channels replaced               */
} mvf_conv_desc_t;

int mvf_conv2d_nhwc_fwd(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed,
                        const float* bias, const void* residual, void* y, void* stream);
/* Same, with a scratch buffer (>= mvf_conv2d_workspace_bytes) that enables the stream-K decomposition of the last,
 * partial wave of output tiles (partial accumulators + publication flags live there).  One buffer per stream. */
size_t mvf_conv2d_workspace_bytes(const mvf_conv_desc_t* d);
int mvf_conv2d_nhwc_fwd_ws(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed,
                           const float* bias, const void* residual, void* y, void* ws, size_t ws_bytes, void* stream);
/* The MVF module FUSED into its wrapped 1x1 conv (MVF.forward, codes/models/modules/MVF.py:104-138, followed by conv1 -> bn1 -> relu of
 * Bottleneck.forward, resnet.py:213-215, eval mode): y = relu(conv1x1(x') + bias) where x' = x with channels [0, cs) replaced by
 * act(scale * (T/H/W 3-tap views of x, zero padded inside the clip / image) + shift) -- computed in the conv's A-operand loader from
 * the seven neighbours of each pixel; no slice buffer, no separate stencil launch.  d: kh = kw = 1, stride 1, ho x wo = h x w,
 * x_pix_stride = cin, d->n = clips * n_segment frames, relu = 1.  mvf_coef: [cs][12] fp32 = {w_t[3], w_h[3], w_w[3], scale, shift, 0} per
 * channel (absent views: zeros); act = 1: affine + hard-swish (use_hs), 0: the bare tap sum.  cs % 64 == 0 (bf16) / 32 (fp32). */
int mvf_conv2d_nhwc_fwd_mvf(const mvf_conv_desc_t* d, const void* x, const void* w_packed, const float* bias, const float* mvf_coef,
                            int cs, int n_segment, int act, void* y, void* ws, size_t ws_bytes, void* stream);
/* [r3] Second pass of a bottleneck's last conv in TRAINING (Bottleneck.forward, codes/models/backbones/resnet.py:229-244: out = conv3(out);
 * out = norm3(out); out += identity; out = relu(out)): the conv is recomputed from its narrow input (a quarter of z3's bytes) and the epilogue
 * applies the BatchNorm whose batch statistics the first pass (mvf_conv2d_nhwc_fwd_stats) produced: the accumulators are rounded to the storage
 * type (= the z3 that first pass stored), out = relu(bn_scale * z3 + bn_shift + r), r = residual or -- a downsample block -- res_scale * residual
 * + res_shift (that branch's BatchNorm applied to its raw conv output); sign_bits [n*ho*wo][cout/4] bytes as mvf_bn_apply_bits writes them.
 * Same result as mvf_bn_apply_bits on the z3 THIS entry point's first pass (mvf_conv2d_nhwc_fwd_stats through the same kernel) stored, bit for bit,
 * without reading z3; a statistics pass through another kernel (csrc/pw_sums.hip) may have seen individual z3 elements one storage-type ulp away.
 * d->relu = 0, in_dil <= 1. */
int mvf_conv2d_nhwc_fwd_bnapply(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const float* bn_scale,
                                const float* bn_shift, const void* residual, const float* res_scale, const float* res_shift, void* out,
                                unsigned char* sign_bits, void* ws, size_t ws_bytes, void* stream);
/* [r3] BatchNorm backward of that bn3 WITHOUT a stored z3 (autograd of resnet.py:229-244): the conv is recomputed from its input and its
 * accumulators, rounded to the storage type, are z3; g = gradient of the block output, sign_bits = the bits the forward wrote, gm = g * bit.
 *   _sums : sums_part CHANNEL-MAJOR [cout][mvf_conv2d_stats_rows(d)][2] = per-128-row column sums of gm and gm * (z3 - mean) * invstd;
 *           mvf_bn_bwd_finalize turns them into dbeta / dgamma.  Nothing else is written.
 *   _apply: dz = gamma * invstd * (gm - dbeta / M - (z3 - mean) * invstd * dgamma / M), M = n*ho*wo (bn_bwd_apply's formula, mask mode 4).
 * With these two and mvf_conv2d_nhwc_fwd_bnapply the first pass may run as a statistics-only pass: mvf_conv2d_nhwc_fwd_stats(..., y = NULL). */
int mvf_conv2d_nhwc_fwd_bnbwd_sums(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const void* g,
                                   const unsigned char* sign_bits, const float* bn_mean, const float* bn_invstd, float* sums_part, void* ws,
                                   size_t ws_bytes, void* stream);
int mvf_conv2d_nhwc_fwd_bnbwd_apply(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const void* g,
                                    const unsigned char* sign_bits, const float* bn_gamma, const float* bn_mean, const float* bn_invstd,
                                    const float* dgamma, const float* dbeta, void* dz, void* ws, size_t ws_bytes, void* stream);
/* ... with the residual gated per element by sign bits ([n*ho*wo][cout/4] bytes, see mvf_bn_apply_bits); stride-1 launches */
int mvf_conv2d_nhwc_fwd_resmask(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed,
                                const float* bias, const void* residual, const unsigned char* res_sign_bits, void* y,
                                void* ws, size_t ws_bytes, void* stream);
/* [r5] ... and with the OUTPUT gated as well: y[.., c] = (conv + bias + gated residual) * [out_gate bit] for c >= d->res_c0 (channels below res_c0 --
 * the MVF slice of a block's conv1 data gradient -- are written ungated).  out_gate_bits: the sign bits of the block output whose gradient y is
 * ([n*ho*wo][cout/4] bytes).  The block below then receives gm = g * [out > 0] (torch autograd's relu backward, resnet.py:244) as a tensor: its
 * BatchNorm backward, data gradient and weight gradient read it without the bits.  res_sign_bits may be NULL (an ungated residual). */
int mvf_conv2d_nhwc_fwd_resmask_gate(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const float* bias,
                                     const void* residual, const unsigned char* res_sign_bits, const unsigned char* out_gate_bits, void* y,
                                     void* ws, size_t ws_bytes, void* stream);
/* [r5] ... and the column sums of the gated output in the same epilogue: with gm = y (as stored), sums_part CHANNEL-MAJOR [cout][mvf_conv2d_stats_rows(d)][2] =
 * per-128-row column sums of gm and gm^2.  The dz3-free block below (resnet.py:236-244 under autograd) takes bn3's dbeta from them and dgamma from its
 * weight-gradient GEMM (mvf_bn_bwd_dzfree_sums): no pass over (gm, z3), no read of z3.  (Channels below res_c0 belong to the MVF stencil's launch.)
 * ([r6] the form that read z3 here for the complete sums measured neutral for two rounds and was removed.) */
int mvf_conv2d_nhwc_fwd_resmask_gate_colsums(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const void* residual,
                                             const unsigned char* res_sign_bits, const unsigned char* out_gate_bits, void* y, float* sums_part, void* ws,
                                             size_t ws_bytes, void* stream);
/* Stride-1 data gradient whose output da feeds the backward of a = ReLU(BN(z)): besides y = dgrad(dz) it accumulates that
 * BatchNorm's backward sums in the epilogue -- sums_part, CHANNEL-MAJOR [cout][mvf_conv2d_stats_rows(d)][2] = per-128-row column sums of gm and
 * gm * xhat with gm = y * [bn_scale*z + bn_shift > 0], xhat = (z - bn_mean) * bn_invstd (z: the forward conv output the BN
 * normalised, same shape as y).  mvf_bn_bwd_finalize (nblk = that row count) turns them into dgamma / dbeta; no separate
 * mvf_bn_bwd_reduce pass.  (Channel-major: a channel's rows are contiguous, the finalize reads whole cache lines.) */
int mvf_conv2d_nhwc_dgrad_bnsums(const mvf_conv_desc_t* d, const void* dz, const void* w_packed_dgrad, void* y, const void* bn_z,
                                 const float* bn_mean, const float* bn_invstd, const float* bn_scale, const float* bn_shift,
                                 float* sums_part, void* ws, size_t ws_bytes, void* stream);
int mvf_bn_bwd_finalize(const float* sums_part, int nblk, int c, float* dgamma, float* dbeta, void* stream);
/* Training forward: plain conv (no bias / residual / ReLU) whose epilogue also accumulates the BatchNorm batch statistics of
 * the tensor it writes: stats_part, CHANNEL-MAJOR [cout][mvf_conv2d_stats_rows(d)][2] = per-128-row column sums of (y-K), (y-K)^2 with
 * K = stats_shift[cout] (pass the BN's running_mean; NULL = 0).  Feed stats_part to mvf_bn_train_finalize. */
int mvf_conv2d_stats_rows(const mvf_conv_desc_t* d);
int mvf_conv2d_nhwc_fwd_stats(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, void* y,
                              float* stats_part, const float* stats_shift, void* ws, size_t ws_bytes, void* stream);

/* w_oihw fp32 (cout, cin, kh, kw) [x scale[cout]] -> packed [cout][kh][kw_pad][cin_pad] in `dtype`
 * (zero padded; kw_pad >= kw, cin_pad >= cin).  bias_out[co] = shift[co] (copied) when given.
 * scale/shift = folded eval BatchNorm2d: scale = gamma/sqrt(var+eps), shift = beta - mean*scale. */
int mvf_pack_conv_weight(const float* w_oihw, int cout, int cin, int kh, int kw, int kw_pad, int cin_pad,
                         const float* scale, void* w_packed, int dtype, void* stream);
/* scale = gamma/sqrt(var+eps), shift = beta - mean*scale  (resnet eval BN, norm.py:59 eps) */
int mvf_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int c,
                float* scale, float* shift, void* stream);

/* Stem input: (n,3,h,w) fp32 NCHW -> zero-padded channels-last (n, h+2*pad, wp, 4) in `dtype`
 * (wp >= w + 2*pad + 2) so that the 7x7/2 stem (resnet.py:424-425,481) becomes 7 K-chunks of 8 px x 4 ch. */
int mvf_stem_prep(const float* x_nchw, int n, int c, int h, int w, int pad, int wp, void* out, int dtype, void* stream);

/* The device side of the input pipeline, fused: decoded uint8 frames (n, hs, ws, 3) HWC -> the h x w window at
 * (y0, x0) of each frame [crop: augmentations.py CenterCrop / ThreeCrop :465-540 offsets], mirrored when flip != 0 [Flip :196-228],
 * channels reversed when to_rgb [Normalize.imnormalize :365-374], value = (float(px) [/ 255 when div_255] - mean[c]) * (1/std[c])
 * in two rounded fp32 steps as the reference's subtract-then-multiply, stacked channels-first [FormatShape formating.py:146-160].
 * window = device int32 (n,3) rows (y0, x0, flip) or NULL (= 0,0,0).  mean3 / std3 are HOST pointers (3 floats each).
 * Outputs (either may be NULL): out_stem = (n, h+2*pad, wp, 4) `dtype`, zero padded, the 7x7 stem's operand exactly as
 * mvf_stem_prep lays it out; out_nchw = (n,3,h,w) fp32, the tensor the reference's collate would hand to the model.
 * The host then ships uint8 frames (1/4 of the fp32 bytes, distributed.py:40-62 scatter) instead of normalised floats. */
int mvf_frames_prep_u8(const unsigned char* frames_hwc, int n, int hs, int ws, const int* window, int h, int w,
                       const float* mean3, const float* std3, int to_rgb, int div_255, int pad, int wp, void* out_stem,
                       float* out_nchw, int dtype, void* stream);

/* MaxPool2d(3, stride 2, pad 1) on NHWC (resnet.py:431,484). */
int mvf_maxpool3x3s2_nhwc(const void* x, int n, int h, int w, int c, void* y, int dtype, void* stream);

/* Head (codes/models/heads/tsn_clshead.py:71-117): per clip, mean over (T,H,W) of the NHWC features
 * (AdaptiveAvgPool2d + consensus mean are both linear, so they commute with the FC; the fcn_testing
 * branch :99-117 is the same expression), then new_fc.  feat (clips*T, H, W, c) -> scores (clips, classes). */
int mvf_head_pool_fc(const void* feat, int clips, int t, int hw, int c, const float* fc_w, const float* fc_b,
                     int classes, float* pooled_ws, float* scores, int dtype, void* stream);
/* average_clip (codes/models/recognizers/base.py:43-74): kind 0 = None (copy), 1 = 'score', 2 = 'prob'. */
int mvf_average_clip(const float* scores, int clips, int classes, int kind, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training step (fp32 this round).  Mirrors, in order: Bottleneck.forward with batch-statistics BatchNorm2d
 * (codes/models/backbones/resnet.py:208-244; torch defaults momentum 0.1, eps from norm.py:59), its autograd
 * backward, TSNClsHead.forward + BaseHead.loss (heads/tsn_clshead.py:71-98, heads/base.py:40-45) and
 * DistOptimizerHook.after_train_iter (core/dist_utils.py:61-67: backward, all-reduce/world, clip_grad_norm_(40),
 * SGD-nesterov step; optimizer cfg mvf_kinetics400_2d_rgb_r50_dense.py:152-154).
 * All tensors are channels-last matrices [m][c] (m = n*h*w), c % 4 == 0.
 * ---------------------------------------------------------------------------------------------- */
/* [r5] mvf_conv2d_nhwc_dgrad_bnsums over a SPLIT operand and with a bias: y = [x2 | x] * w_packed^T + bias (d->split_c channels from x2, the rest
 * from x, see mvf_conv_desc_t.x_c0), pointwise, stride 1 -- the data gradient of a bottleneck's conv3 taken on (gm, a2) instead of on dz3
 * (mvf_bn_bwd_dzfree_prep makes w_packed and bias). */
int mvf_conv2d_nhwc_dgrad_bnsums_split(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const float* bias, void* y,
                                       const void* bn_z, const float* bn_mean, const float* bn_invstd, const float* bn_scale, const float* bn_shift,
                                       float* sums_part, void* ws, size_t ws_bytes, void* stream);
/* [r5] BatchNorm backward of out = relu(bn3(conv3(a2)) + identity) (Bottleneck.forward, codes/models/backbones/resnet.py:229-244, under torch autograd)
 * WITHOUT the dz3 tensor (csrc/bn_dzfree.hip).  dz3 = a (gm - d0 - kx (z3 - mean)) with a = gamma invstd, d0 = dbeta / m, kx = invstd dgamma / m is
 * affine in (gm, z3) and z3 = a2 W^T, so
 *   da2 = dz3 W    = [gm | a2] [a.W ; -G] - v      G = W^T diag(a kx) W,  v_k = sum_c a_c (d0_c - kx_c mean_c) W[c][k]
 *   dW  = dz3^T a2 = a . (Q - d0 (x) sa - kx . (W A2 - mean (x) sa))     Q = gm^T a2, A2 = a2^T a2, sa = m * a_mean
 * _prep : w_out [k][c + k] (the data-gradient pack's layout: row = conv input channel) and bias_out [k] = -v for mvf_conv2d_nhwc_dgrad_bnsums_split,
 *         from w_packed_dgrad [k][c] (mvf_pack_conv_weight_dgrad of the c x k pointwise weights) and the BatchNorm's dgamma / dbeta
 *         (mvf_bn_bwd_reduce on gm and the stored z3; zeros for a BatchNorm with frozen statistics).
 * _wgrad: dw [c][k] holds Q on entry (mvf_conv2d_nhwc_wgrad with dz = gm) and dW on return; gram [k][k] = A2 (the same call with dz = x = a2),
 *         a_mean [k] = column means of a2, w_packed [c][k] = the forward pack.  bf16 storage. */
int mvf_bn_bwd_dzfree_prep(const void* w_packed_dgrad, int c, int k, const float* gamma, const float* mean, const float* invstd, const float* dgamma,
                           const float* dbeta, long m, void* w_out, float* bias_out, int dtype, void* stream);
int mvf_bn_bwd_dzfree_wgrad(float* dw, const void* w_packed, const float* gram, const float* a_mean, const float* gamma, const float* mean,
                            const float* invstd, const float* dgamma, const float* dbeta, long m, int c, int k, int dtype, void* stream);
/* _sums : dgamma / dbeta WITHOUT the pass over (gm, z3) either: sum_m gm z3 = sum_k W[c][k] Q[c][k] (Q = the weight-gradient GEMM of _wgrad, taken FIRST) and
 *         sum_m gm from the column sums the kernels that stored gm took in their epilogues -- part_lo [c_split][rows_lo][2] (the MVF stencil's slice,
 *         mvf_nhwc_stencil_gate_colsums) and part_hi [c][rows_hi][2] indexed by the absolute channel (mvf_conv2d_nhwc_fwd_resmask_gate_colsums); element
 *         [.][.][0] is read.  dgamma[c] = invstd (W[c].Q[c] - mean sum gm), dbeta[c] = sum gm. */
int mvf_bn_bwd_dzfree_sums(const float* q, const void* w_packed, int c, int k, const float* mean, const float* invstd, const float* part_lo, int rows_lo,
                           int c_split, const float* part_hi, int rows_hi, float* dgamma, float* dbeta, int dtype, void* stream);
size_t mvf_bn_workspace_bytes(long m, int c);
/* batch mean / biased var of z over m -> save_mean, save_invstd, scale = gamma*invstd, shift = beta - mean*scale;
 * running_mean/var updated in place (unbiased var).  Shifted single-pass sums (shift = old running_mean). */
int mvf_bn_train_stats(const void* z, long m, int c, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* scale,
                       float* shift, void* ws, size_t ws_bytes, int dtype, void* stream);
/* second half of mvf_bn_train_stats for the channel-major partial sums [c][nblk][2] produced by mvf_conv2d_nhwc_fwd_stats
 * (nblk = mvf_conv2d_stats_rows(d); K must be the same running_mean) */
int mvf_bn_train_finalize(const float* part, int nblk, long m, int c, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                          float* scale, float* shift, void* stream);
/* [r5] The same outputs for z = a W^T, the output of a POINTWISE conv (Bottleneck.conv3 -> norm3, resnet.py:236-237), WITHOUT computing z: mean_c = W[c].abar,
 * var_c = W[c] (A2 / m - abar abar^T) W[c]^T from the Gram matrix gram [k][k] = a^T a of the conv's input (mvf_conv2d_nhwc_wgrad with dz = x = a), its
 * column means a_mean [k] (mvf_bn_train_stats on a) and the forward pack w_packed [c][k].  Replaces the statistics pass of a block that applies this
 * BatchNorm in a second conv pass (mvf_conv2d_nhwc_fwd_bnapply); the dz3-free backward reuses gram / a_mean.  bf16 storage, c and k multiples of 32. */
int mvf_bn_train_stats_gram(const float* gram, const float* a_mean, const void* w_packed, long m, int c, int k, const float* gamma, const float* beta,
                            float eps, float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* scale,
                            float* shift, int dtype, void* stream);
/* out = act(z * scale + shift) (mvf_bn_apply without a residual) + the column means of what it stores: a_mean for mvf_bn_train_stats_gram from the kernel that
 * WRITES conv3's input (norm2 + relu, resnet.py:233-234) instead of a pass over it.  ws >= max(mvf_bn_workspace_bytes(m, c), 4608 * c * 8) bytes
 * (one partial row per workgroup row band of the apply plan). */
int mvf_bn_apply_colmeans(const void* z, long m, int c, const float* scale, const float* shift, int act, void* out, float* mean_out, void* ws,
                          size_t ws_bytes, int dtype, void* stream);
/* out = act(z*scale + shift [+ residual | + residual*rscale + rshift]); act: 0 none, 1 ReLU, 2 hard-swish */
int mvf_bn_apply(const void* z, long m, int c, const float* scale, const float* shift, const void* residual,
                 const float* rscale, const float* rshift, int act, void* out, int dtype, void* stream);
/* gm = g * act'(.) ; dbeta = sum gm ; dgamma = sum gm * xhat.  mask_mode: 0 none, 1 ReLU via ymask > 0 (ymask = the
 * forward output), 2 ReLU via scale*z+shift > 0, 3 hard-swish'(scale*z+shift), 4 sign bits (see mvf_bn_apply_bits).
 * g rows are g_pitch apart (>= c).
 * gm_out (optional, pitch c) receives gm. */
int mvf_bn_bwd_reduce(const void* g, int g_pitch, const void* z, const void* ymask, long m, int c, const float* mean,
                      const float* invstd, const float* scale, const float* shift, int mask_mode, void* gm_out,
                      float* dgamma, float* dbeta, void* ws, size_t ws_bytes, int dtype, void* stream);
/* dz = gamma*invstd*(gm - dbeta/m - xhat*dgamma/m), gm recomputed from g with mask_mode 0 / 2 / 3 */
int mvf_bn_bwd_apply(const void* g, int g_pitch, const void* z, long m, int c, const float* gamma, const float* mean,
                     const float* invstd, const float* scale, const float* shift, const float* dgamma,
                     const float* dbeta, int mask_mode, void* dz, int dtype, void* stream);
/* The same with the ReLU mask of a block output kept as SIGN BITS instead of re-reading the tensor (1/8 of its bf16 bytes):
 * mvf_bn_apply_bits also writes sign_bits [m][c/4] bytes (bit j of byte k <=> out[.., 4k+j] > 0); mask_mode 4 of
 * mvf_bn_bwd_reduce / mvf_bn_bwd_apply_masked takes them as ymask (mask_mode 1 there = the tensor itself).  The skip-connection
 * gradient g*mask is then never materialised: mvf_conv2d_nhwc_fwd_resmask / mvf_nhwc_stencil gate their residual / addend
 * with the same bits (torch autograd's relu backward, Bottleneck.forward codes/models/backbones/resnet.py:238-244). */
int mvf_bn_apply_bits(const void* z, long m, int c, const float* scale, const float* shift, const void* residual,
                      const float* rscale, const float* rshift, int relu, void* out, unsigned char* sign_bits, int dtype,
                      void* stream);
int mvf_bn_bwd_apply_masked(const void* g, int g_pitch, const void* z, const void* ymask, long m, int c, const float* gamma,
                            const float* mean, const float* invstd, const float* scale, const float* shift,
                            const float* dgamma, const float* dbeta, int mask_mode, void* dz, int dtype, void* stream);
/* PAIRED backward of the two BatchNorms of a downsample block, out = relu(bn3(z3) + bnd(zd)) (Bottleneck.forward, resnet.py:227-233): both
 * receive the same masked gradient g * [out > 0] (sign_bits as written by mvf_bn_apply_bits), so g and the bits are read once for
 * both: dgamma / dbeta of a and b, then dz_a, dz_b.  Results are bit-identical to two mvf_bn_bwd_reduce + mvf_bn_bwd_apply_masked
 * (mask_mode 4) calls.  ws: 2 x mvf_bn_workspace_bytes(m, c). */
int mvf_bn_bwd_pair(const void* g, int g_pitch, const void* z_a, const void* z_b, const unsigned char* sign_bits, long m, int c,
                    const float* gamma_a, const float* mean_a, const float* invstd_a, float* dgamma_a, float* dbeta_a,
                    const float* gamma_b, const float* mean_b, const float* invstd_b, float* dgamma_b, float* dbeta_b,
                    void* dz_a, void* dz_b, void* ws, size_t ws_bytes, int dtype, void* stream);
/* [r4] BatchNorm backward APPLY fused with the weight gradient of the pointwise (1x1, stride 1) conv that produced the BatchNorm's input
 * (autograd of Bottleneck.forward, resnet.py:213-244: z = conv(x); out = relu(bn(z) [+ identity])): the pass that forms
 * dz = gamma * invstd * (gm - dbeta / m - xhat * dgamma / m) also contracts it with the conv input, dW[co][ci] = sum_m dz[m][co] * x[m][ci],
 * so dz is read once afterwards (by the data gradient) instead of twice.  dz is BIT-identical to mvf_bn_bwd_apply_masked; the weight gradient
 * is left as `splits` fp32 partial slabs [splits][c][k] that mvf_wgrad_slab_reduce sums in fixed order into dw (c, k, 1, 1) -- on any stream
 * that is ordered behind this call.  dgamma / dbeta must be final (mvf_bn_bwd_reduce / mvf_bn_bwd_finalize ran).  bf16 storage only; built for
 * the byte-bound shapes whose c x k accumulator fits one workgroup: mask_mode 4 (sign bits) with c % 256 == 0 and k = 64 / 128, mask_mode 2
 * (gate = scale * z + shift > 0) with c = 64 / 128 / 256 and k % 256 == 0; mvf_bn_bwd_wgrad_splits returns 0 for every other shape (use the
 * separate calls).  nbn = 1 here, 2 for the paired form below. */
int mvf_bn_bwd_wgrad_splits(long m, int c, int k, int nbn, int mask_mode);
size_t mvf_bn_bwd_wgrad_slab_bytes(long m, int c, int k, int nbn, int mask_mode);
int mvf_bn_bwd_apply_wgrad(const void* g, int g_pitch, const void* z, const void* ymask, long m, int c, const float* gamma, const float* mean,
                           const float* invstd, const float* scale, const float* shift, const float* dgamma, const float* dbeta, int mask_mode,
                           void* dz, const void* x, int x_pitch, int k, float* slabs, size_t slab_bytes, int dtype, void* stream);
/* [r4] The same for the PAIRED backward of a downsample block (mvf_bn_bwd_pair: sums of both BatchNorms, then both dz): dz_a / dz_b are
 * contracted with x_a (conv3's input) / x_b (the downsample conv's input = the block input; stride 1 only).  x_b = NULL: only conv a's weight
 * gradient is fused (k = 128; a stride-2 downsample conv keeps its GEMM).  ws: 2 x mvf_bn_workspace_bytes(m, c); slabs_*: slab_bytes each. */
int mvf_bn_bwd_pair_wgrad(const void* g, int g_pitch, const void* z_a, const void* z_b, const unsigned char* sign_bits, long m, int c,
                          const float* gamma_a, const float* mean_a, const float* invstd_a, float* dgamma_a, float* dbeta_a,
                          const float* gamma_b, const float* mean_b, const float* invstd_b, float* dgamma_b, float* dbeta_b,
                          void* dz_a, void* dz_b, const void* x_a, int xa_pitch, const void* x_b, int xb_pitch, int k,
                          float* slabs_a, float* slabs_b, size_t slab_bytes, void* ws, size_t ws_bytes, int dtype, void* stream);
/* [r4] The WHOLE backward of the last conv of a bottleneck that never stored its conv output (autograd of Bottleneck.forward, resnet.py:229-244:
 * out = relu(bn3(conv3(a2)) + identity), a2 = relu(bn2(z2))) in ONE pass: the 64 -> 256 channel pointwise conv is recomputed per 64-pixel chunk,
 * dz3 = gamma invstd (gm - dbeta / m - xhat dgamma / m) (gm = g gated by the block output's sign bits; bn3's dgamma / dbeta must be final) stays in
 * LDS and is contracted three ways: the weight gradient (fp32 slabs [splits][256][64] for mvf_wgrad_slab_reduce), the data gradient
 * dx = round_bf16(dz3 W) (m, 64), and the backward sums of bn2 over dx gated by scale2 z2 + shift2 > 0 (partial rows [64][sums_rows][2] for
 * mvf_bn_bwd_finalize, sums_rows = 2 x splits).  Replaces mvf_conv2d_nhwc_fwd_bnbwd_apply + mvf_conv2d_nhwc_dgrad_bnsums + mvf_conv2d_nhwc_wgrad
 * (dz3 is neither written nor re-read).  z_in = NULL: the conv input is not a BatchNorm's activation (a downsample branch, resnet.py:227-228, reading the block
 * input): dx is the plain data gradient, no sums (in_* / sums_part unused).  bf16 storage, c = 256, k = 64 only: mvf_conv1x1_bwd_fused_splits returns 0 for
 * anything else.  sign_bits: 16-byte aligned. */
int mvf_conv1x1_bwd_fused_splits(long m, int c, int k);
int mvf_conv1x1_bwd_fused(const void* a_in, int a_pitch, const void* w_packed, const void* g, int g_pitch, const unsigned char* sign_bits, long m, int c,
                          int k, const float* gamma, const float* mean, const float* invstd, const float* dgamma, const float* dbeta, const void* z_in,
                          const float* in_mean, const float* in_invstd, const float* in_scale, const float* in_shift, void* dx, float* sums_part,
                          int sums_rows, float* slabs, size_t slab_bytes, int dtype, void* stream);
/* [r4] BatchNorm-backward sums of BOTH branches of a downsample bottleneck that stored no z3 (resnet.py:227-244: out = relu(bn3(conv3(a2)) + bn_d(conv_d(x)))) in one
 * pass over g: sum gm and sum gm xhat for bn3 (a: conv3 on a_in) and bn_d (b: the downsample conv on x_in), both convs recomputed, g and the sign bits read once
 * (mvf_conv2d_nhwc_fwd_bnbwd_sums twice reads them twice).  Partial rows [256][rows][2] each for mvf_bn_bwd_finalize, rows = 2 x mvf_conv1x1_bwd_fused_splits.
 * x_in = NULL: the sums of bn3 alone (a plain block: conv a only; w_b / mean_b / invstd_b / part_b unused).  bf16 storage, c = 256, k = 64 for both convs. */
int mvf_conv1x1_bnbwd_sums_pair(const void* a_in, int a_pitch, const void* w_a, const void* x_in, int x_pitch, const void* w_b, const void* g, int g_pitch,
                                const unsigned char* sign_bits, long m, int c, int k, const float* mean_a, const float* invstd_a, const float* mean_b,
                                const float* invstd_b, float* part_a, float* part_b, int rows, int dtype, void* stream);
int mvf_wgrad_slab_reduce(const float* slabs, int nsplit, int cout, int k, float* dw_oihw, void* stream);
/* stem: y = maxpool3x3/2(relu(z*scale+shift)) (resnet.py:482-484).  argmax (optional, one byte per element of y) receives
 * the window position dy*3+dx of the first maximum; the backward routes g to it: ga = dL/d relu(bn(z)). */
int mvf_maxpool_bn_relu_fwd(const void* z, int n, int h, int w, int c, const float* scale, const float* shift, void* y,
                            unsigned char* argmax, int dtype, void* stream);
int mvf_maxpool_bn_relu_bwd(const unsigned char* argmax, const void* g, int n, int h, int w, int c, void* ga, int dtype,
                            void* stream);
/* The same scatter that also accumulates the backward sums of the BatchNorm under the pool (stem: pool(relu(bn(z))), reference
 * resnet.py:461-466): gm = ga * [scale*z + shift > 0], sums of gm and gm * (z - mean) * invstd per channel into the channel-major
 * partials sums_part [c][mvf_maxpool_bwd_sums_rows(n, h)][2]; mvf_bn_bwd_finalize turns them into dgamma / dbeta, so the
 * BatchNorm backward only needs its apply pass (mask_mode 2 over ga).  Needs c/4 to divide 256.  ga may be NULL. */
int mvf_maxpool_bwd_sums_rows(int n, int h);
/* ... and the matching apply pass that re-gathers ga instead of reading it (pass ga = NULL to mvf_maxpool_bn_relu_bwd_sums then):
 * dz = gamma*invstd * (gm - dbeta/M - xhat*dgamma/M), M = n*h*w, gm as above.  dgamma / dbeta: the finalized sums, or zeros for
 * a BatchNorm in eval mode (no batch-statistics term). */
int mvf_maxpool_bn_relu_bwd_apply(const unsigned char* argmax, const void* g, int n, int h, int w, int c, const void* z, const float* gamma,
                                  const float* mean, const float* invstd, const float* scale, const float* shift, const float* dgamma,
                                  const float* dbeta, void* dz, int dtype, void* stream);
int mvf_maxpool_bn_relu_bwd_sums(const unsigned char* argmax, const void* g, int n, int h, int w, int c, void* ga, const void* z, const float* mean,
                                 const float* invstd, const float* scale, const float* shift, float* sums_part, int dtype, void* stream);
/* head: avg-pool per frame -> new_fc -> mean over the clip's t frames -> cross-entropy (mean over clips).
 * pooled (clips*t, c), scores (clips, classes), dscores = dloss/dscores, loss_part (clips), loss (1): all fp32. */
int mvf_head_train_fwd(const void* feat, int clips, int t, int hw, int c, const float* fc_w, const float* fc_b, int classes,
                       const long long* labels, const float* drop_mask /* (clips*t, c) pre-scaled keep mask or NULL */,
                       float* pooled, float* scores, float* dscores, float* loss_part, float* loss, int dtype, void* stream);
/* BaseHead.loss (codes/models/heads/base.py:40-45): loss[0] = mean over clips of softmax cross-entropy(scores[clip], labels[clip]);
 * loss_part (clips) = the per-clip terms; dscores (clips, classes) = dloss/dscores or NULL.  All fp32. */
int mvf_ce_loss(const float* scores, const long long* labels, int clips, int classes, float* dscores, float* loss_part, float* loss,
                void* stream);
int mvf_head_train_bwd(const float* dscores, const float* pooled, const float* fc_w, const float* drop_mask, int clips, int t,
                       int hw, int c, int classes, float* dfc_w, float* dfc_b, float* dpool_ws /* clips*c */, void* dfeat, int dtype,
                       void* stream);
/* conv weight gradient dw_oihw (cout, cin_real, kh, kw_real) fp32; d describes the FORWARD conv (x dims, ho/wo = dz dims).
 * kw_packed*cin_packed == d->kw*d->cin; they differ from (kw_real, cin_real) only for the stem view (8x4 vs 7x3).
 * fp32 accumulation over the pixels in a fixed order (per-split partial tiles, then a split-lane reduce): deterministic, no atomics.  fp32
 * operands are multiplied on the bf16 matrix cores as exact three-term bf16 splits, six partial products per product -- as accurate against
 * an fp64 weight gradient as the exact-fp32 matrix instruction, which MVF_WGRAD_X3=0 (or MVF_F32_X3=0) selects. */
size_t mvf_conv2d_wgrad_workspace_bytes(const mvf_conv_desc_t* d);
int mvf_conv2d_nhwc_wgrad(const mvf_conv_desc_t* d, const void* dz, const void* x, const void* x2, int kw_real,
                          int cin_real, int kw_packed, int cin_packed, float* dw_oihw, void* ws, size_t ws_bytes,
                          void* stream);
/* [r5] the same GEMM with the workgroup count to aim at named by the caller (the library's own plans size weight gradients for a side stream: half the
 * chip); results differ from mvf_conv2d_nhwc_wgrad only by the fp32 summation order of the pixel split. */
int mvf_conv2d_nhwc_wgrad_wgs(const mvf_conv_desc_t* d, const void* dz, const void* x, const void* x2, int kw_real,
                              int cin_real, int kw_packed, int cin_packed, float* dw_oihw, void* ws, size_t ws_bytes, int wgs, void* stream);
/* Every weight pack of a training step in one launch.  jobs_dev = DEVICE array of njobs records sorted by first_block;
 * job k owns workgroups [first_block, first_block + ceil(elements / 2048)), total_blocks = their sum.  kind 0 = the forward
 * pack of mvf_pack_conv_weight (no scale), kind 1 = the data-gradient pack of mvf_pack_conv_weight_dgrad.  kind 2 / 3 = the
 * same two packs as an LDS-tiled transpose for cout % 32 == 0, cin % 32 == 0, kh * kw <= 9, kw_pad == kw, cin_pad == cin:
 * such a job owns (cout / 32) * (cin / 32) workgroups. */
typedef struct {
    const float* w;      /* fp32 OIHW parameter */
    void* out;           /* packed operand in `dtype` */
    int cout, cin, kh, kw, kw_pad, cin_pad;
    int kind, first_block;
} mvf_pack_job_t;
int mvf_pack_conv_weights_batched(const mvf_pack_job_t* jobs_dev, int njobs, int total_blocks, int dtype, void* stream);

/* data gradient = mvf_conv2d_nhwc_fwd on dz with these weights: packed[ci][kh'][kw'][co] = w[co][ci][KH-1-kh'][KW-1-kw'],
 * pad' = k-1-pad, stride 1, in_dil = forward stride */
int mvf_pack_conv_weight_dgrad(const float* w_oihw, int cout, int cin, int kh, int kw, void* w_packed, int dtype,
                               void* stream);
/* MVF training primitives, channels-last (the engine composes MVF fwd/bwd from these + the BN calls above) */
/* out[..., :cs] = f(stencil(x[..., :cs])) [+ addend[..., :cs] (pitch addend_c; NULL = none), gated per channel by
 * addend_sign_bits ([pixels][addend_c/4] bytes as written by mvf_bn_apply_bits; NULL = ungated)]; flip = transposed stencil */
int mvf_nhwc_stencil(const mvf_desc_t* d, const void* x, int x_c, void* out, int out_c, const float* w_t,
                     const float* w_h, const float* w_w, const float* scale, const float* shift, int flip,
                     const void* addend, int addend_c, const unsigned char* addend_sign_bits, void* stream);
/* [r5] ... with the OUTPUT gated per channel by out_gate_bits ([pixels][out_c/4] bytes): out = (f(stencil(x)) + gated addend) * [bit] */
int mvf_nhwc_stencil_gate(const mvf_desc_t* d, const void* x, int x_c, void* out, int out_c, const float* w_t,
                          const float* w_h, const float* w_w, const float* scale, const float* shift, int flip,
                          const void* addend, int addend_c, const unsigned char* addend_sign_bits,
                          const unsigned char* out_gate_bits, void* stream);
/* [r5] mvf_nhwc_stencil_gate + the column sums of the gated slice: sums_part CHANNEL-MAJOR [cs][mvf_nhwc_stencil_stats_rows(d, x_c, out_c)][2] = per-workgroup
 * sums of gm and gm^2 -- the slice's share of mvf_conv2d_nhwc_fwd_resmask_gate_colsums. */
int mvf_nhwc_stencil_gate_colsums(const mvf_desc_t* d, const void* x, int x_c, void* out, int out_c, const float* w_t, const float* w_h,
                                  const float* w_w, int flip, const void* addend, int addend_c, const unsigned char* addend_sign_bits,
                                  const unsigned char* out_gate_bits, float* sums_part, void* stream);
/* [r5] the plain stencil (no activation) that also accumulates the batch statistics of MVF's BatchNorm3d (MVF.py:131-134, training mode) over the
 * values it stores: stats_part CHANNEL-MAJOR [cs][mvf_nhwc_stencil_stats_rows(d, x_c, out_c)][2] = per-workgroup sums of (y - K), (y - K)^2, K =
 * stats_shift (the old running mean; NULL = 0) -> mvf_bn_train_finalize.  Replaces the statistics pass over y. */
int mvf_nhwc_stencil_stats_rows(const mvf_desc_t* d, int x_c, int out_c);
/* [r6] plan query (nothing is launched): 1 = a stencil launch of this shape (16-byte aligned operands) runs on the LDS-tiled bf16 kernel, with
 * rows_per_band x chan_per_wg (both optional) its tile; 0 = the register-chunked kernel (fp32, slices not in 16-channel chunks, one clip of the source
 * larger than a 32-bit buffer descriptor, fewer workgroups than the tile needs). */
int mvf_nhwc_stencil_tile_plan(const mvf_desc_t* d, int x_c, int out_c, int* rows_per_band, int* chan_per_wg);
int mvf_nhwc_stencil_stats(const mvf_desc_t* d, const void* x, int x_c, void* out, int out_c, const float* w_t, const float* w_h,
                           const float* w_w, float* stats_part, const float* stats_shift, void* stream);
size_t mvf_nhwc_tapgrad_workspace_bytes(const mvf_desc_t* d);
int mvf_nhwc_tapgrad(const mvf_desc_t* d, const void* x, int x_c, const void* dy, int dy_c, float* dw_t, float* dw_h,
                     float* dw_w, void* ws, size_t ws_bytes, void* stream);
/* clip_grad_norm_(max_norm) on grad_scale*grads, then torch.optim.SGD(nesterov) on flat fp32 buffers.
 * norm_out[0] = total norm, norm_out[1] = clip coefficient. */
size_t mvf_sgd_workspace_bytes(long n);
int mvf_sgd_nesterov_step(float* params, const float* grads, float* momentum_buf, long n, float grad_scale, float max_norm,
                          float lr, float momentum, float weight_decay, int first_step, float* norm_out, void* ws,
                          size_t ws_bytes, void* stream);
/* The same step with per-segment multipliers = build_optimizer's paramwise_options (codes/core/train.py:117-156: bias_lr_mult,
 * bias_decay_mult, norm_decay_mult give every parameter its own lr / weight_decay) on the flat buffers: segment k covers elements
 * [first_k, first_{k+1}) (sorted, first_0 = 0) with lr * lr_mult, weight_decay * decay_mult.  nesterov = 0 gives the plain
 * momentum update (p -= lr * buf).  The gradient norm / clip coefficient are global, as clip_grad_norm_ over all parameters.
 * lr_mult < 0 marks a segment EXCLUDED from training (requires_grad False: norm_frozen / partial_norm, resnet.py:496-527): its
 * gradient does not enter the norm, its parameters and momentum are left untouched. */
typedef struct mvf_sgd_segment {
    long long first;
    float lr_mult, decay_mult;
} mvf_sgd_segment_t;
int mvf_sgd_step_segments(float* params, const float* grads, float* momentum_buf, long n, float grad_scale, float max_norm, float lr,
                          float momentum, float weight_decay, int first_step, int nesterov, const mvf_sgd_segment_t* segments, int nseg,
                          float* norm_out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * [r6] Launch-table replay.  One training step (the reference's batch_processor + DistOptimizerHook.after_train_iter, codes/core/train.py:45-60,
 * codes/core/dist_utils.py:61-67) is a fixed sequence of this library's entry points on two HIP streams; the host code records it once and replays it from C
 * instead of re-issuing ~650 calls from Python (mvfnet_amd/launch_plan.py).  An op is
 *   MVF_PLAN_CALL    fn(words[word0 .. word0 + n_int), floats[float0 .. float0 + n_flt)): fn = the address of ANY int-returning entry point of this header,
 *                    its integer-class arguments (pointers, int, long, size_t, the stream) in prototype order as 64-bit words, its float arguments in order
 *   MVF_PLAN_RECORD  hipEventRecord(event = words[word0], stream = words[word0 + 1])
 *   MVF_PLAN_WAIT    hipStreamWaitEvent(stream = words[word0], event = words[word0 + 1])
 * Events are the caller's.  The run stops at the first op that fails: its index goes to *failed_op (else -1), its status is returned and mvf_last_error()
 * describes it.  Arguments that change between runs are patched in `words` by the caller.  x86-64 System V hosts only. */
enum { MVF_PLAN_CALL = 0, MVF_PLAN_RECORD = 1, MVF_PLAN_WAIT = 2 };
typedef struct mvf_plan_op {
    int kind, n_int, n_flt, reserved;
    void* fn;
    long long word0, float0;
} mvf_plan_op_t;
int mvf_plan_run(const mvf_plan_op_t* ops, int n_ops, const unsigned long long* words, const float* floats, int* failed_op);

#ifdef __cplusplus
}
#endif
#endif /* MVFNET_HIP_H */
