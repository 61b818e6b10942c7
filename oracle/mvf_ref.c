/* ORACLE (test infrastructure, not product code): plain-C restatement of MVF-proper, forward and backward.
 *
 * Restates codes/models/modules/MVF.py:104-137 of the reference (whwu95/MVFNet): on the (N*T, C, H, W) tensor, channels
 * [0, cs) get  y = shift_conv(x) + h_conv(x) + w_conv(x)  (three depthwise 3-tap cross-correlations along T, H, W with zero
 * padding, MVF.py:65-81,118-120), then BatchNorm3d(cs) (torch defaults: biased variance to normalise, unbiased into
 * running_var, momentum 0.1, eps 1e-5; MVF.py:69,133) and HardSwish x*relu6(x+3)/6 (codes/models/common/se_module.py:5-24);
 * channels >= cs pass through (MVF.py:110,135).  The backward is the analytic gradient of exactly that.
 *
 * Deliberately naive: scalar loops in double precision, one element at a time -- a second, independent oracle next to
 * oracle/mvf_numpy.py.  Pinned by tests/test_oracle_golden.py against tests/golden/mvf_cases.npz (reference outputs).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX(n, t, c, y, x) ((((size_t)((n) * T + (t)) * C + (c)) * H + (y)) * W + (x))

static double hswish(double u) {
    double r = u + 3.0;
    r = r < 0.0 ? 0.0 : (r > 6.0 ? 6.0 : r);
    return u * (r / 6.0);
}
static double hswish_grad(double u) {
    double r = u + 3.0, inner = (u > -3.0 && u < 3.0) ? 1.0 : 0.0;
    r = r < 0.0 ? 0.0 : (r > 6.0 ? 6.0 : r);
    return r / 6.0 + u * inner / 6.0;
}

/* 9-tap view sum at one position; mode bits: 1 = T, 2 = H, 4 = W */
static double stencil(const float* x, int N, int T, int C, int H, int W, int n, int t, int c, int y, int xx, int mode,
                      const float* wt, const float* wh, const float* ww) {
    double s = 0.0;
    (void)N;
    for (int j = 0; j < 3; ++j) {
        int tt = t + j - 1, yy = y + j - 1, xj = xx + j - 1;
        if (tt >= 0 && tt < T) s += (double)wt[c * 3 + j] * x[IDX(n, tt, c, y, xx)];
        if ((mode & 2) && yy >= 0 && yy < H) s += (double)wh[c * 3 + j] * x[IDX(n, t, c, yy, xx)];
        if ((mode & 4) && xj >= 0 && xj < W) s += (double)ww[c * 3 + j] * x[IDX(n, t, c, y, xj)];
    }
    return s;
}

/* out (NT,C,H,W); use_hs: apply BN + hswish; training: batch statistics (save_mean/save_invstd out, running_* updated
 * in place when non-NULL), else running statistics.  ypre (NT,cs,H,W doubles, may be NULL) receives y for the backward. */
int mvf_ref_forward(const float* x, int NT, int C, int H, int W, int T, int cs, int mode, const float* wt, const float* wh,
                    const float* ww, int use_hs, int training, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, double eps, double momentum, float* out, double* save_mean, double* save_invstd,
                    double* ypre) {
    if (NT % T) return -1;
    const int N = NT / T;
    const size_t m = (size_t)N * T * H * W;
    memcpy(out, x, sizeof(float) * (size_t)NT * C * H * W);
    double* y = ypre ? ypre : (double*)malloc(sizeof(double) * m * (cs > 0 ? cs : 1));
    for (int c = 0; c < cs; ++c) {
        double sum = 0.0;
        size_t k = 0;
        for (int n = 0; n < N; ++n)
            for (int t = 0; t < T; ++t)
                for (int yy = 0; yy < H; ++yy)
                    for (int xx = 0; xx < W; ++xx, ++k) {
                        double v = stencil(x, N, T, C, H, W, n, t, c, yy, xx, mode, wt, wh, ww);
                        y[(size_t)c * m + k] = v;
                        sum += v;
                    }
        if (!use_hs) {
            k = 0;
            for (int n = 0; n < N; ++n)
                for (int t = 0; t < T; ++t)
                    for (int yy = 0; yy < H; ++yy)
                        for (int xx = 0; xx < W; ++xx, ++k) out[IDX(n, t, c, yy, xx)] = (float)y[(size_t)c * m + k];
            continue;
        }
        double mean, var;
        if (training) {
            mean = sum / (double)m;
            var = 0.0;
            for (k = 0; k < m; ++k) var += (y[(size_t)c * m + k] - mean) * (y[(size_t)c * m + k] - mean);
            var /= (double)m;
            if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * (double)m / (double)(m > 1 ? m - 1 : 1));
        } else {
            mean = running_mean[c];
            var = running_var[c];
        }
        const double invstd = 1.0 / sqrt(var + eps);
        if (save_mean) save_mean[c] = mean;
        if (save_invstd) save_invstd[c] = invstd;
        k = 0;
        for (int n = 0; n < N; ++n)
            for (int t = 0; t < T; ++t)
                for (int yy = 0; yy < H; ++yy)
                    for (int xx = 0; xx < W; ++xx, ++k) {
                        double u = (y[(size_t)c * m + k] - mean) * invstd * gamma[c] + beta[c];
                        out[IDX(n, t, c, yy, xx)] = (float)hswish(u);
                    }
    }
    if (!ypre) free(y);
    return 0;
}

/* g = dL/d(out) (NT,C,H,W) -> dx (NT,C,H,W), dwt/dwh/dww [cs][3] doubles, dgamma/dbeta [cs] doubles.
 * ypre, mean, invstd from mvf_ref_forward (training: batch statistics; eval: the running ones). */
int mvf_ref_backward(const float* g, const float* x, const double* ypre, int NT, int C, int H, int W, int T, int cs, int mode,
                     const float* wt, const float* wh, const float* ww, int use_hs, int training, const float* gamma,
                     const float* beta, const double* mean, const double* invstd, float* dx, double* dwt, double* dwh,
                     double* dww, double* dgamma, double* dbeta) {
    if (NT % T) return -1;
    const int N = NT / T;
    const size_t m = (size_t)N * T * H * W;
    memcpy(dx, g, sizeof(float) * (size_t)NT * C * H * W);
    double* dy = (double*)malloc(sizeof(double) * m);
    for (int c = 0; c < cs; ++c) {
        size_t k = 0;
        double sb = 0.0, sg = 0.0;
        for (int n = 0; n < N; ++n)
            for (int t = 0; t < T; ++t)
                for (int yy = 0; yy < H; ++yy)
                    for (int xx = 0; xx < W; ++xx, ++k) {
                        double go = g[IDX(n, t, c, yy, xx)];
                        if (use_hs) {
                            double xh = (ypre[(size_t)c * m + k] - mean[c]) * invstd[c];
                            double du = go * hswish_grad(xh * gamma[c] + beta[c]);
                            dy[k] = du;
                            sb += du;
                            sg += du * xh;
                        } else {
                            dy[k] = go;
                        }
                    }
        if (use_hs) {
            dbeta[c] = sb;
            dgamma[c] = sg;
            for (k = 0; k < m; ++k) {
                double xh = (ypre[(size_t)c * m + k] - mean[c]) * invstd[c];
                double corr = training ? (sb / (double)m + xh * sg / (double)m) : 0.0;
                dy[k] = gamma[c] * invstd[c] * (dy[k] - corr);
            }
        }
        for (int j = 0; j < 3; ++j) dwt[c * 3 + j] = dwh[c * 3 + j] = dww[c * 3 + j] = 0.0;
        k = 0;
        for (int n = 0; n < N; ++n)
            for (int t = 0; t < T; ++t)
                for (int yy = 0; yy < H; ++yy)
                    for (int xx = 0; xx < W; ++xx, ++k) {
                        double ds = 0.0;
                        for (int j = 0; j < 3; ++j) {
                            int tt = t + j - 1, y2 = yy + j - 1, x2 = xx + j - 1;
                            /* weight gradients: dw[j] += dy[p] * x[p + (j-1)] */
                            if (tt >= 0 && tt < T) dwt[c * 3 + j] += dy[k] * x[IDX(n, tt, c, yy, xx)];
                            if ((mode & 2) && y2 >= 0 && y2 < H) dwh[c * 3 + j] += dy[k] * x[IDX(n, t, c, y2, xx)];
                            if ((mode & 4) && x2 >= 0 && x2 < W) dww[c * 3 + j] += dy[k] * x[IDX(n, t, c, yy, x2)];
                            /* data gradient: ds[p] = sum_j w[j] * dy[p - (j-1)] */
                            int tb = t - (j - 1), yb = yy - (j - 1), xb = xx - (j - 1);
                            if (tb >= 0 && tb < T) ds += (double)wt[c * 3 + j] * dy[(((size_t)n * T + tb) * H + yy) * W + xx];
                            if ((mode & 2) && yb >= 0 && yb < H) ds += (double)wh[c * 3 + j] * dy[(((size_t)n * T + t) * H + yb) * W + xx];
                            if ((mode & 4) && xb >= 0 && xb < W) ds += (double)ww[c * 3 + j] * dy[(((size_t)n * T + t) * H + yy) * W + xb];
                        }
                        dx[IDX(n, t, c, yy, xx)] = (float)ds;
                    }
    }
    free(dy);
    return 0;
}
