// MVF-proper on the channels-last (N*T, H, W, C) tensor -- the layout of the fused network engine (gfx950).
//
// In NHWC the slice channels [0, cs) of one pixel are cs contiguous elements (256 B .. 1 KiB for the
// MVFNet shapes), so lanes run along channels (16 B per lane) and then along pixels: every load/store
// instruction covers whole contiguous slice rows.  A thread owns (pixel, 4 channels), keeps the 9 tap weights
// and the folded BN of its 4 channels in registers and slides a (prev, cur, next) register window along t, so
// the T-view costs no extra loads; the H- and W-view neighbours (+-W*C, +-C elements in the same frame) are
// re-reads of lines the same workgroup is streaming and are served by L1/L2.  HBM traffic = slice read + slice
// write.  Codes/models/modules/MVF.py:104-137 is the arithmetic being replaced.
//
// Training / backward in this layout arrive with the NHWC training engine (they return MVF_EUNSUPPORTED now;
// the NCHW implementation in mvf_nchw.hip is complete).
#include <algorithm>

#include "common.h"

namespace {

constexpr int kThreads = 256;

struct NhwcArgs {
    const void* x;
    void* out;
    const float* wt;
    const float* wh;
    const float* ww;
    const float* scale;
    const float* shift;
    int nt, c, h, w, T, cs, mode, n_clips;
    int cg;        // channel groups (of VEC channels) in the slice
    int cgp;       // threads along channels per block (power of two <= 256)
    int pixw;      // pixels per workgroup
    int bands;     // workgroups per clip along pixels
    int out_c;     // channel pitch of `out` (c for a full tensor, cs for a compact slice buffer)
};

template <typename ET, int VEC>
struct Vec;
template <typename ET>
struct Vec<ET, 4> {
    static __device__ __forceinline__ void load(const ET* p, float (&v)[4]) {
        float4 q = ld4(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
    static __device__ __forceinline__ void store(ET* p, const float (&v)[4]) { st4(p, make_float4(v[0], v[1], v[2], v[3])); }
};
template <typename ET>
struct Vec<ET, 1> {
    static __device__ __forceinline__ void load(const ET* p, float (&v)[1]) { v[0] = ldf(p); }
    static __device__ __forceinline__ void store(ET* p, const float (&v)[1]) { stf(p, v[0]); }
};

template <typename ET, int VEC>
__global__ __launch_bounds__(kThreads) void mvf_nhwc_apply(NhwcArgs a) {
    const int HW = a.h * a.w, W = a.w, H = a.h, T = a.T, C = a.c;
    const int n = blockIdx.x / a.bands, band = blockIdx.x % a.bands;
    const int cgi = blockIdx.y * a.cgp + (threadIdx.x % a.cgp);
    const int plane = threadIdx.x / a.cgp, nplanes = kThreads / a.cgp;
    if (cgi >= a.cg) return;
    const int c0 = cgi * VEC;
    const bool vh = a.mode & MVF_VIEW_H, vw = a.mode & MVF_VIEW_W;
    const bool hs = a.scale != nullptr;
    float wt[VEC][3], wh[VEC][3], ww[VEC][3], sc[VEC], sh[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            wt[i][j] = a.wt[(c0 + i) * 3 + j];
            wh[i][j] = vh ? a.wh[(c0 + i) * 3 + j] : 0.f;
            ww[i][j] = vw ? a.ww[(c0 + i) * 3 + j] : 0.f;
        }
        sc[i] = hs ? a.scale[c0 + i] : 1.f;
        sh[i] = hs ? a.shift[c0 + i] : 0.f;
    }
    const ET* x = reinterpret_cast<const ET*>(a.x);
    ET* out = reinterpret_cast<ET*>(a.out);
    const long fstride = (long)HW * C;                         // one frame
    const int pend = min(HW, (band + 1) * a.pixw);
    for (int pix = band * a.pixw + plane; pix < pend; pix += nplanes) {
        const int hh = pix / W, wv = pix - hh * W;
        const long e0 = ((long)n * T * HW + pix) * C + c0;     // (n, t=0, pix, c0)
        const long o0 = ((long)n * T * HW + pix) * a.out_c + c0;
        float prev[VEC], cur[VEC], next[VEC], up[VEC], dn[VEC], lf[VEC], rt[VEC], y[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) prev[i] = 0.f;
        Vec<ET, VEC>::load(x + e0, cur);
        for (int t = 0; t < T; ++t) {
            const ET* f = x + e0 + (long)t * fstride;
            if (t + 1 < T) Vec<ET, VEC>::load(f + fstride, next);
            else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) next[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) up[i] = dn[i] = lf[i] = rt[i] = 0.f;
            if (vh && hh > 0) Vec<ET, VEC>::load(f - (long)W * C, up);
            if (vh && hh < H - 1) Vec<ET, VEC>::load(f + (long)W * C, dn);
            if (vw && wv > 0) Vec<ET, VEC>::load(f - C, lf);
            if (vw && wv < W - 1) Vec<ET, VEC>::load(f + C, rt);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float yt = wt[i][0] * prev[i] + wt[i][1] * cur[i] + wt[i][2] * next[i];
                float yh = wh[i][0] * up[i] + wh[i][1] * cur[i] + wh[i][2] * dn[i];
                float yw = ww[i][0] * lf[i] + ww[i][1] * cur[i] + ww[i][2] * rt[i];
                float v = (yt + yh) + yw;
                if (hs) {
                    float u = sc[i] * v + sh[i];
                    v = u * (fminf(fmaxf(u + 3.0f, 0.0f), 6.0f) / 6.0f);
                }
                y[i] = v;
            }
            Vec<ET, VEC>::store(out + o0 + (long)t * HW * a.out_c, y);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                prev[i] = cur[i];
                cur[i] = next[i];
            }
        }
    }
}

// copy channels [cs, c) of every pixel (out != x case)
template <typename ET>
__global__ void copy_tail_nhwc(const ET* src, ET* dst, long npix, int c, int cs) {
    const int tail = c - cs;
    const long total = npix * tail;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p = i / tail;
        int k = (int)(i - p * tail);
        dst[p * c + cs + k] = src[p * c + cs + k];
    }
}

}  // namespace

size_t mvf_nhwc_ws_fwd_train(const mvf_desc_t*) { return 256; }
size_t mvf_nhwc_ws_bwd(const mvf_desc_t*) { return 256; }

int mvf_nhwc_fwd_infer_impl(const mvf_desc_t* d, const void* x, void* out, int out_c, const float* wt, const float* wh,
                            const float* ww, const float* scale, const float* shift, hipStream_t st) {
    // In-place hazard: a workgroup re-reads neighbour pixels that another workgroup may already have overwritten.
    MVF_REQUIRE(x != out, MVF_EINVAL,
                "mvf_fwd_infer(NHWC): in-place is not supported in this layout (neighbour pixels are re-read from "
                "global memory); pass a separate out buffer");
    NhwcArgs a = {};
    a.x = x; a.out = out; a.wt = wt; a.wh = wh; a.ww = ww; a.scale = scale; a.shift = shift;
    a.nt = d->nt; a.c = d->c; a.h = d->h; a.w = d->w; a.T = d->n_segment; a.cs = d->cs; a.mode = d->mode;
    a.n_clips = d->nt / d->n_segment;
    a.out_c = out_c;
    const int esz = d->dtype == MVF_F32 ? 4 : 2;
    const bool vec = (d->cs % 4 == 0) && (d->c % 4 == 0) && (out_c % 4 == 0) && (((uintptr_t)x | (uintptr_t)out) % (4 * esz) == 0);
    a.cg = vec ? d->cs / 4 : d->cs;
    int cgp = 1;
    while (cgp < a.cg && cgp < kThreads) cgp <<= 1;
    a.cgp = cgp;
    const int HW = d->h * d->w;
    const int nplanes = kThreads / cgp;
    // enough workgroups to fill 256 CUs several times over, but >= 4 pixels per thread-plane to amortise weights
    int pixw = std::max(nplanes * 4, 1);
    while ((long)a.n_clips * ((HW + pixw - 1) / pixw) > 8192 && pixw < HW) pixw *= 2;
    a.pixw = std::min(pixw, HW);
    a.bands = (HW + a.pixw - 1) / a.pixw;
    dim3 grid(a.n_clips * a.bands, (a.cg + cgp - 1) / cgp);
    if (d->dtype == MVF_F32) {
        if (vec) hipLaunchKernelGGL((mvf_nhwc_apply<float, 4>), grid, dim3(kThreads), 0, st, a);
        else hipLaunchKernelGGL((mvf_nhwc_apply<float, 1>), grid, dim3(kThreads), 0, st, a);
    } else {
        if (vec) hipLaunchKernelGGL((mvf_nhwc_apply<bf16_t, 4>), grid, dim3(kThreads), 0, st, a);
        else hipLaunchKernelGGL((mvf_nhwc_apply<bf16_t, 1>), grid, dim3(kThreads), 0, st, a);
    }
    MVF_LAUNCH_CHECK();
    if (d->cs < d->c && out_c == d->c) {
        const long npix = (long)d->nt * HW;
        const int blocks = (int)std::min<long>((npix * (d->c - d->cs) + 255) / 256, 256L * 16);
        if (d->dtype == MVF_F32)
            hipLaunchKernelGGL(copy_tail_nhwc<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)out, npix, d->c, d->cs);
        else
            hipLaunchKernelGGL(copy_tail_nhwc<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, npix, d->c, d->cs);
        MVF_LAUNCH_CHECK();
    }
    return MVF_OK;
}

int mvf_nhwc_fwd_infer(const mvf_desc_t* d, const void* x, void* out, const float* wt, const float* wh,
                       const float* ww, const float* scale, const float* shift, hipStream_t st) {
    return mvf_nhwc_fwd_infer_impl(d, x, out, d->c, wt, wh, ww, scale, shift, st);
}

int mvf_nhwc_fwd_train(const mvf_desc_t*, const void*, void*, const float*, const float*, const float*, const float*,
                       const float*, float, float, float*, float*, float*, float*, void*, hipStream_t) {
    mvf_set_error("mvf_fwd_train: NHWC layout not implemented yet (use MVF_NCHW)");
    return MVF_EUNSUPPORTED;
}

int mvf_nhwc_bwd(const mvf_desc_t*, const void*, const void*, const float*, const float*, const float*, const float*,
                 const float*, const float*, const float*, int, void*, float*, float*, float*, float*, float*, void*,
                 hipStream_t) {
    mvf_set_error("mvf_bwd: NHWC layout not implemented yet (use MVF_NCHW)");
    return MVF_EUNSUPPORTED;
}
