// Bandwidth-bound companions of the conv stack: stem input re-layout, max-pool, TSN head, clip averaging.
#include <algorithm>
#include <math.h>

#include "common.h"

namespace {

// (n,c<=4,h,w) fp32 NCHW -> zero-padded NHWC4: out[n][ih+pad][iw+pad][0..3]; lanes run along w (coalesced reads
// of each colour plane, 16-B/8-B stores).
// grid.x = (image, padded row), threads along the padded row: no per-element 64-bit division, and the colour-plane loads are
// unconditional on a clamped address (zeroed by a select) so the three of a pixel share one round trip
template <typename ET>
__global__ void stem_prep_kernel(const float* x, int n, int c, int h, int w, int pad, int hp, int wp, ET* out) {
    const int row = blockIdx.x, img = row / hp, yo = row - img * hp;
    const int ih = yo - pad;
    const bool rok = ih >= 0 && ih < h;
    const float* xi = x + ((long)img * c * h + (rok ? ih : 0)) * w;
    for (int xo = threadIdx.x; xo < wp; xo += blockDim.x) {
        const int iw = xo - pad;
        const bool ok = rok && iw >= 0 && iw < w;
        const int iwc = ok ? iw : 0;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = x ? xi[(long)(k < c ? k : 0) * h * w + iwc] : 0.f;
            v[k] = (ok && k < c) ? t : 0.f;
        }
        st4(out + ((long)row * wp + xo) * 4, make_float4(v[0], v[1], v[2], v[3]));
    }
}

// Decoded uint8 HWC frames -> normalised stem operand (and / or the reference's fp32 NCHW tensor): window (crop), mirror (flip),
// channel order (to_rgb), (float(px) [/ 255] - mean) * stdinv in two rounded fp32 steps (the reference subtracts, then multiplies
// by 1/std: no FMA), channels-first stacking.  Thread = one pixel of the padded stem image; 3 byte loads, one 8/16-byte store.
struct FramePrep {
    float mean[3], stdinv[3];
    int to_rgb, div_255;
};
template <typename ET>
__global__ void frames_prep_kernel(const unsigned char* frames, int n, int hs, int ws, const int* win, int h, int w, FramePrep fp,
                                   int pad, int hp, int wp, ET* out_stem, float* out_nchw) {
    const long total = (long)n * hp * wp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % wp);
        const long t = i / wp;
        const int yo = (int)(t % hp);
        const int img = (int)(t / hp);
        const int ih = yo - pad, iw = xo - pad;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const bool inside = ih >= 0 && ih < h && iw >= 0 && iw < w;
        if (inside) {
            const int y0 = win ? win[img * 3 + 0] : 0, x0 = win ? win[img * 3 + 1] : 0, flip = win ? win[img * 3 + 2] : 0;
            const int sy = y0 + ih, sx = x0 + (flip ? w - 1 - iw : iw);
            const unsigned char* px = frames + (((long)img * hs + sy) * ws + sx) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float f = (float)px[fp.to_rgb ? 2 - k : k];
                if (fp.div_255) f = __fdiv_rn(f, 255.f);
                v[k] = __fmul_rn(__fsub_rn(f, fp.mean[k]), fp.stdinv[k]);
            }
            if (out_nchw) {
#pragma unroll
                for (int k = 0; k < 3; ++k) out_nchw[(((long)img * 3 + k) * h + ih) * w + iw] = v[k];
            }
        }
        if (out_stem) st4(out_stem + i * 4, make_float4(v[0], v[1], v[2], v[3]));
    }
}

// MaxPool2d(3, 2, 1) NHWC; thread = (output pixel, 4 channels).  The nine window loads are UNCONDITIONAL on clamped coordinates (a
// clamped tap re-reads a pixel that is inside the window anyway, so the maximum is unchanged): a load inside an `if` is waited
// for on the spot, nine serial round trips per output instead of one.
template <typename ET>
__global__ void maxpool_kernel(const ET* x, int n, int h, int w, int c, int ho, int wo, ET* y) {
    const int c4 = c >> 2;
    const long total = (long)n * ho * wo * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % c4);
        long t = i / c4;
        const int ow = (int)(t % wo); t /= wo;
        const int oh = (int)(t % ho);
        const int img = (int)(t / ho);
        const ET* base = x + (long)img * h * w * c + cq * 4;
        float4 v[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ih = min(max(oh * 2 - 1 + dy, 0), h - 1);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int iw = min(max(ow * 2 - 1 + dx, 0), w - 1);
                v[dy * 3 + dx] = ld4(base + ((long)ih * w + iw) * c);
            }
        }
        float4 m = v[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) {
            m.x = fmaxf(m.x, v[k].x); m.y = fmaxf(m.y, v[k].y); m.z = fmaxf(m.z, v[k].z); m.w = fmaxf(m.w, v[k].w);
        }
        st4(y + (((long)img * ho + oh) * wo + ow) * c + cq * 4, m);
    }
}

// pooled[clip][ch] = mean over the clip's T*HW feature rows; lanes along channels (coalesced rows)
// workgroup = 64 channels (16 lanes x 4) x 16 row lanes: 8/16-byte loads, four rows in flight per thread, the 16 row-lane sums
// combined in lane order through LDS (fixed order).  The one-thread-per-channel form walked the clip's 392 rows serially with
// 2-byte loads from 128 workgroups: 134 us for 51 MB.
template <typename ET>
__global__ __launch_bounds__(256) void head_pool_kernel(const ET* feat, int rows_per_clip, int c, float* pooled) {
    __shared__ float4 red[256];
    const int clip = blockIdx.y, q = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int ch = (blockIdx.x * 16 + q) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch < c) {
        const ET* p = feat + (long)clip * rows_per_clip * c + ch;
        int r = rl;
        for (; r + 48 < rows_per_clip; r += 64) {
            const float4 a = ld4(p + (long)r * c), b = ld4(p + (long)(r + 16) * c), d = ld4(p + (long)(r + 32) * c), e = ld4(p + (long)(r + 48) * c);
            s.x += (a.x + b.x) + (d.x + e.x); s.y += (a.y + b.y) + (d.y + e.y);
            s.z += (a.z + b.z) + (d.z + e.z); s.w += (a.w + b.w) + (d.w + e.w);
        }
        for (; r < rows_per_clip; r += 16) {
            const float4 a = ld4(p + (long)r * c);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && ch < c) {
        float4 t = red[q];
        for (int l = 1; l < 16; ++l) {
            const float4 v = red[l * 16 + q];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        const float inv = 1.f / (float)rows_per_clip;
        *reinterpret_cast<float4*>(pooled + (long)clip * c + ch) = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
}

// scores[clip][k] = b[k] + pooled[clip] . W[k]; one wave per (clip, class)
__global__ void head_fc_kernel(const float* pooled, const float* w, const float* b, int clips, int c, int classes,
                               float* scores) {
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (gw >= clips * classes) return;
    const int clip = gw / classes, k = gw - clip * classes;
    const float* p = pooled + (long)clip * c;
    const float* wr = w + (long)k * c;
    float s = 0.f;
    for (int i = lane; i < c; i += 64) s += p[i] * wr[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) scores[gw] = s + (b ? b[k] : 0.f);
}

// average_clip: kind 1 = mean of scores over clips, kind 2 = mean of softmax(scores) over clips; one block
__global__ void average_clip_kernel(const float* scores, int clips, int classes, int kind, float* out) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    for (int k = tid; k < classes; k += blockDim.x) out[k] = 0.f;
    __syncthreads();
    for (int cl = 0; cl < clips; ++cl) {
        const float* s = scores + (long)cl * classes;
        float mx = -INFINITY, den = 1.f;
        if (kind == 2) {
            for (int k = tid; k < classes; k += blockDim.x) mx = fmaxf(mx, s[k]);
            red[tid] = mx;
            __syncthreads();
            for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
                if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
                __syncthreads();
            }
            mx = red[0];
            __syncthreads();
            float e = 0.f;
            for (int k = tid; k < classes; k += blockDim.x) e += expf(s[k] - mx);
            red[tid] = e;
            __syncthreads();
            for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
                if (tid < o) red[tid] += red[tid + o];
                __syncthreads();
            }
            den = red[0];
            __syncthreads();
        }
        for (int k = tid; k < classes; k += blockDim.x) {
            const float v = kind == 2 ? expf(s[k] - mx) / den : s[k];
            out[k] += v;
        }
    }
    for (int k = tid; k < classes; k += blockDim.x) out[k] = out[k] / (float)clips;
}

inline int grid_for(long total, int per_block = 256, int cap = 256 * 32) {
    return (int)std::min<long>((total + per_block - 1) / per_block, cap);
}

}  // namespace

extern "C" {

int mvf_stem_prep(const float* x_nchw, int n, int c, int h, int w, int pad, int wp, void* out, int dtype, void* stream) {
    MVF_REQUIRE(x_nchw && out && n > 0 && c > 0 && c <= 4 && h > 0 && w > 0 && pad >= 0 && wp >= w + 2 * pad, MVF_EINVAL, "stem_prep: bad argument");
    const int hp = h + 2 * pad;
    MVF_REQUIRE((long)n * hp < (1L << 31), MVF_ESHAPE, "stem_prep: too many rows");
    const dim3 grid(n * hp);
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(stem_prep_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x_nchw, n, c, h, w, pad, hp, wp, (float*)out);
    else
        hipLaunchKernelGGL(stem_prep_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, x_nchw, n, c, h, w, pad, hp, wp, (bf16_t*)out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_frames_prep_u8(const unsigned char* frames_hwc, int n, int hs, int ws, const int* window, int h, int w,
                       const float* mean3, const float* std3, int to_rgb, int div_255, int pad, int wp, void* out_stem,
                       float* out_nchw, int dtype, void* stream) {
    MVF_REQUIRE(frames_hwc && mean3 && std3 && (out_stem || out_nchw) && n > 0 && hs > 0 && ws > 0 && h > 0 && w > 0 && h <= hs && w <= ws && pad >= 0,
                MVF_EINVAL, "frames_prep_u8: bad argument");
    MVF_REQUIRE(!out_stem || wp >= w + 2 * pad, MVF_EINVAL, "frames_prep_u8: wp=%d < w + 2*pad", wp);
    MVF_REQUIRE(dtype == MVF_F32 || dtype == MVF_BF16, MVF_EINVAL, "frames_prep_u8: bad dtype");
    FramePrep fp;
    for (int k = 0; k < 3; ++k) {
        MVF_REQUIRE(std3[k] != 0.f, MVF_EINVAL, "frames_prep_u8: std[%d] is zero", k);
        fp.mean[k] = mean3[k];
        fp.stdinv[k] = (float)(1.0 / (double)std3[k]);      // the reference multiplies by 1 / float64(std)
    }
    fp.to_rgb = to_rgb;
    fp.div_255 = div_255;
    const int p = out_stem ? pad : 0, wpp = out_stem ? wp : w;
    const int hp = h + 2 * p;
    const long total = (long)n * hp * wpp;
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(frames_prep_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, frames_hwc, n, hs, ws, window, h, w, fp,
                           p, hp, wpp, (float*)out_stem, out_nchw);
    else
        hipLaunchKernelGGL(frames_prep_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, frames_hwc, n, hs, ws, window, h, w, fp,
                           p, hp, wpp, (bf16_t*)out_stem, out_nchw);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_maxpool3x3s2_nhwc(const void* x, int n, int h, int w, int c, void* y, int dtype, void* stream) {
    MVF_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, MVF_EINVAL, "maxpool: bad argument (c %% 4 != 0?)");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long total = (long)n * ho * wo * (c / 4);
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, n, h, w, c, ho, wo, (float*)y);
    else
        hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, n, h, w, c, ho, wo, (bf16_t*)y);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_head_pool_fc(const void* feat, int clips, int t, int hw, int c, const float* fc_w, const float* fc_b, int classes,
                     float* pooled_ws, float* scores, int dtype, void* stream) {
    MVF_REQUIRE(feat && fc_w && pooled_ws && scores && clips > 0 && t > 0 && hw > 0 && c > 0 && classes > 0, MVF_EINVAL, "head_pool_fc: bad argument");
    hipStream_t st = (hipStream_t)stream;
    MVF_REQUIRE(c % 4 == 0, MVF_ESHAPE, "head_pool_fc: c=%d must be a multiple of 4", c);
    dim3 g((c + 63) / 64, clips);
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(head_pool_kernel<float>, g, dim3(256), 0, st, (const float*)feat, t * hw, c, pooled_ws);
    else
        hipLaunchKernelGGL(head_pool_kernel<bf16_t>, g, dim3(256), 0, st, (const bf16_t*)feat, t * hw, c, pooled_ws);
    MVF_LAUNCH_CHECK();
    const long waves = (long)clips * classes;
    hipLaunchKernelGGL(head_fc_kernel, dim3((int)((waves * 64 + 255) / 256)), dim3(256), 0, st, pooled_ws, fc_w, fc_b, clips, c, classes, scores);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_average_clip(const float* scores, int clips, int classes, int kind, float* out, void* stream) {
    MVF_REQUIRE(scores && out && clips > 0 && classes > 0 && kind >= 0 && kind <= 2, MVF_EINVAL, "average_clip: bad argument");
    if (kind == 0) {
        MVF_HIP_OK(hipMemcpyAsync(out, scores, sizeof(float) * clips * classes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return MVF_OK;
    }
    hipLaunchKernelGGL(average_clip_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scores, clips, classes, kind, out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // extern "C"
