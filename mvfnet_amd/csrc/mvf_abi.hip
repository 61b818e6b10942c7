// C-ABI entry points of libmvfnet_hip.so: argument validation + dispatch (see include/mvfnet_hip.h).
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mvf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// layout-specific implementations (mvf_nchw.hip / mvf_nhwc.hip)
int mvf_nchw_check(const mvf_desc_t* d, bool bwd);
size_t mvf_nchw_ws_fwd_train(const mvf_desc_t* d);
size_t mvf_nchw_ws_bwd(const mvf_desc_t* d);
int mvf_nchw_fwd_infer(const mvf_desc_t*, const void*, void*, const float*, const float*, const float*, const float*,
                       const float*, hipStream_t);
int mvf_nchw_fwd_train(const mvf_desc_t*, const void*, void*, const float*, const float*, const float*, const float*,
                       const float*, float, float, float*, float*, float*, float*, void*, hipStream_t);
int mvf_nchw_bwd(const mvf_desc_t*, const void*, const void*, const float*, const float*, const float*, const float*,
                 const float*, const float*, const float*, int, void*, float*, float*, float*, float*, float*, void*,
                 hipStream_t);
size_t mvf_nhwc_ws_fwd_train(const mvf_desc_t* d);
size_t mvf_nhwc_ws_bwd(const mvf_desc_t* d);
int mvf_nhwc_fwd_infer(const mvf_desc_t*, const void*, void*, const float*, const float*, const float*, const float*,
                       const float*, hipStream_t);
int mvf_nhwc_fwd_infer_impl(const mvf_desc_t*, const void*, void*, int, const float*, const float*, const float*,
                            const float*, const float*, hipStream_t);
int mvf_nhwc_fwd_train(const mvf_desc_t*, const void*, void*, const float*, const float*, const float*, const float*,
                       const float*, float, float, float*, float*, float*, float*, void*, hipStream_t);
int mvf_nhwc_bwd(const mvf_desc_t*, const void*, const void*, const float*, const float*, const float*, const float*,
                 const float*, const float*, const float*, int, void*, float*, float*, float*, float*, float*, void*,
                 hipStream_t);

static int check_desc(const mvf_desc_t* d) {
    MVF_REQUIRE(d != nullptr, MVF_EINVAL, "mvf: desc is NULL");
    MVF_REQUIRE(d->nt > 0 && d->c > 0 && d->h > 0 && d->w > 0, MVF_ESHAPE, "mvf: bad dims nt=%d c=%d h=%d w=%d", d->nt,
                d->c, d->h, d->w);
    MVF_REQUIRE(d->n_segment > 0 && d->nt % d->n_segment == 0, MVF_ESHAPE,
                "mvf: nt=%d is not a multiple of n_segment=%d (MVF.py:107-109)", d->nt, d->n_segment);
    MVF_REQUIRE(d->cs > 0 && d->cs <= d->c, MVF_ESHAPE, "mvf: cs=%d must be in (0, c=%d]", d->cs, d->c);
    MVF_REQUIRE((d->mode & MVF_VIEW_T) && (d->mode & ~7) == 0 && d->mode != (MVF_VIEW_T | MVF_VIEW_W), MVF_EINVAL,
                "mvf: mode=%d must be T(1), TH(3) or THW(7)", d->mode);
    MVF_REQUIRE(d->layout == MVF_NCHW || d->layout == MVF_NHWC, MVF_EINVAL, "mvf: bad layout %d", d->layout);
    MVF_REQUIRE(d->dtype == MVF_F32 || d->dtype == MVF_BF16, MVF_EINVAL, "mvf: bad dtype %d", d->dtype);
    MVF_REQUIRE((long)d->nt * d->c * d->h * d->w < (1L << 40), MVF_ESHAPE, "mvf: tensor too large");
    return MVF_OK;
}

extern "C" {

int mvf_abi_version(void) { return MVF_ABI_VERSION; }
const char* mvf_last_error(void) { return g_err; }

int mvf_fwd_infer(const mvf_desc_t* d, const void* x, void* out, const float* w_t, const float* w_h, const float* w_w,
                  const float* bn_scale, const float* bn_shift, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    MVF_REQUIRE(x && out && w_t, MVF_EINVAL, "mvf_fwd_infer: NULL x/out/w_t");
    MVF_REQUIRE(!(d->mode & MVF_VIEW_H) || w_h, MVF_EINVAL, "mvf_fwd_infer: mode has H view but w_h is NULL");
    MVF_REQUIRE(!(d->mode & MVF_VIEW_W) || w_w, MVF_EINVAL, "mvf_fwd_infer: mode has W view but w_w is NULL");
    MVF_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), MVF_EINVAL, "mvf_fwd_infer: bn_scale/bn_shift must both be set or both NULL");
    hipStream_t st = (hipStream_t)stream;
    if (d->layout == MVF_NCHW) return mvf_nchw_fwd_infer(d, x, out, w_t, w_h, w_w, bn_scale, bn_shift, st);
    return mvf_nhwc_fwd_infer(d, x, out, w_t, w_h, w_w, bn_scale, bn_shift, st);
}

int mvf_fwd_infer_slice(const mvf_desc_t* d, const void* x, void* out_slice, const float* w_t, const float* w_h,
                        const float* w_w, const float* bn_scale, const float* bn_shift, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    MVF_REQUIRE(d->layout == MVF_NHWC, MVF_EUNSUPPORTED, "mvf_fwd_infer_slice: NHWC only");
    MVF_REQUIRE(x && out_slice && w_t, MVF_EINVAL, "mvf_fwd_infer_slice: NULL x/out/w_t");
    MVF_REQUIRE(!(d->mode & MVF_VIEW_H) || w_h, MVF_EINVAL, "mvf_fwd_infer_slice: mode has H view but w_h is NULL");
    MVF_REQUIRE(!(d->mode & MVF_VIEW_W) || w_w, MVF_EINVAL, "mvf_fwd_infer_slice: mode has W view but w_w is NULL");
    MVF_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), MVF_EINVAL, "mvf_fwd_infer_slice: bn_scale/bn_shift must both be set or both NULL");
    return mvf_nhwc_fwd_infer_impl(d, x, out_slice, d->cs, w_t, w_h, w_w, bn_scale, bn_shift, (hipStream_t)stream);
}

size_t mvf_fwd_train_workspace_bytes(const mvf_desc_t* d) {
    if (check_desc(d)) return 0;
    return d->layout == MVF_NCHW ? mvf_nchw_ws_fwd_train(d) : mvf_nhwc_ws_fwd_train(d);
}

int mvf_fwd_train(const mvf_desc_t* d, const void* x, void* out, const float* w_t, const float* w_h, const float* w_w,
                  const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                  float* running_var, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    MVF_REQUIRE(x && out && w_t && gamma && beta && save_mean && save_invstd, MVF_EINVAL, "mvf_fwd_train: NULL argument");
    MVF_REQUIRE(x != out, MVF_EINVAL, "mvf_fwd_train: in-place is inference-only (x is needed by mvf_bwd)");
    MVF_REQUIRE(!(d->mode & MVF_VIEW_H) || w_h, MVF_EINVAL, "mvf_fwd_train: mode has H view but w_h is NULL");
    MVF_REQUIRE(!(d->mode & MVF_VIEW_W) || w_w, MVF_EINVAL, "mvf_fwd_train: mode has W view but w_w is NULL");
    MVF_REQUIRE(ws && ws_bytes >= mvf_fwd_train_workspace_bytes(d), MVF_EWS, "mvf_fwd_train: workspace %zu B < %zu B",
                ws_bytes, mvf_fwd_train_workspace_bytes(d));
    hipStream_t st = (hipStream_t)stream;
    if (d->layout == MVF_NCHW)
        return mvf_nchw_fwd_train(d, x, out, w_t, w_h, w_w, gamma, beta, eps, momentum, running_mean, running_var,
                                  save_mean, save_invstd, ws, st);
    return mvf_nhwc_fwd_train(d, x, out, w_t, w_h, w_w, gamma, beta, eps, momentum, running_mean, running_var,
                              save_mean, save_invstd, ws, st);
}

size_t mvf_bwd_workspace_bytes(const mvf_desc_t* d) {
    if (check_desc(d)) return 0;
    return d->layout == MVF_NCHW ? mvf_nchw_ws_bwd(d) : mvf_nhwc_ws_bwd(d);
}

int mvf_bwd(const mvf_desc_t* d, const void* g, const void* x, const float* w_t, const float* w_h, const float* w_w,
            const float* gamma, const float* beta, const float* mean, const float* invstd, int training, void* dx,
            float* dw_t, float* dw_h, float* dw_w, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
            void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    MVF_REQUIRE(g && x && dx && w_t && dw_t, MVF_EINVAL, "mvf_bwd: NULL argument");
    MVF_REQUIRE(!(d->mode & MVF_VIEW_H) || w_h, MVF_EINVAL, "mvf_bwd: mode has H view but w_h is NULL");
    MVF_REQUIRE(!(d->mode & MVF_VIEW_W) || w_w, MVF_EINVAL, "mvf_bwd: mode has W view but w_w is NULL");
    MVF_REQUIRE(gamma == nullptr || (beta && mean && invstd), MVF_EINVAL, "mvf_bwd: gamma given but beta/mean/invstd NULL");
    MVF_REQUIRE(ws && ws_bytes >= mvf_bwd_workspace_bytes(d), MVF_EWS, "mvf_bwd: workspace %zu B < %zu B", ws_bytes,
                mvf_bwd_workspace_bytes(d));
    hipStream_t st = (hipStream_t)stream;
    if (d->layout == MVF_NCHW)
        return mvf_nchw_bwd(d, g, x, w_t, w_h, w_w, gamma, beta, mean, invstd, training, dx, dw_t, dw_h, dw_w, dgamma,
                            dbeta, ws, st);
    return mvf_nhwc_bwd(d, g, x, w_t, w_h, w_w, gamma, beta, mean, invstd, training, dx, dw_t, dw_h, dw_w, dgamma, dbeta,
                        ws, st);
}

}  // extern "C"
