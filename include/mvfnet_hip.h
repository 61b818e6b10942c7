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

#define MVF_ABI_VERSION 1

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

#ifdef __cplusplus
}
#endif
#endif /* MVFNET_HIP_H */
