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
// Training forward / backward in this layout (mvf_fwd_train / mvf_bwd with MVF_NHWC, cs % 4 == 0) are composed below from
// the engine's primitives: stencil -> batch statistics -> stencil with BN + hard-swish; backward = recomputed stencil, BN
// backward with the hard-swish mask, tap gradients, transposed stencil.
#include <algorithm>

#include <stdlib.h>

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
    int flip;      // use taps reversed (transposed stencil: the data-gradient of the views)
    const void* add;   // optional addend (same pixels, channel pitch add_c), added after the activation
    int add_c;
    const unsigned char* add_mask;   // optional gate bits of the addend: [pixels][add_c/4] bytes, bit j of byte k = channel 4k+j
    float* stats_part;               // [r5] optional (chunked kernel): CHANNEL-MAJOR [cs][gridDim.x][2] column sums of (y - K), (y - K)^2 over the workgroup's outputs AS STORED
    int rb, cw, lbands;   // [r5] LDS-tiled kernel (mvf_nhwc_apply_lds): rows per band, channels per workgroup, bands per clip
    unsigned fdw_mul, fdw_shr, fdrw_mul, fdrw_shr;      // host-made magic for n / w and n / ((rb + 2) w)
    const float* stats_shift;        // K per channel for the statistics sums (the BatchNorm's old running mean; NULL = 0): MVF's BatchNorm3d statistics without a pass over y
    const unsigned char* out_gate;   // [r5] optional gate bits of the OUTPUT (after the addend): [pixels][out_c/4] bytes, same layout (VEC = 4 kernels only)
};

// one 3-tap view with the contraction spelled out: every stencil kernel of this file rounds the same way whatever the compiler would fuse on its own
__device__ __forceinline__ float tap3(float w0, float w1, float w2, float a, float b, float c) { return __builtin_fmaf(w2, c, __builtin_fmaf(w1, b, w0 * a)); }

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
            const int js = a.flip ? 2 - j : j;
            wt[i][j] = a.wt[(c0 + i) * 3 + js];
            wh[i][j] = vh ? a.wh[(c0 + i) * 3 + js] : 0.f;
            ww[i][j] = vw ? a.ww[(c0 + i) * 3 + js] : 0.f;
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
        // sliding: t runs 0..T-1 with (prev, cur, next) carried in registers
        const int t_begin = 0, t_end = T;
        Vec<ET, VEC>::load(x + e0, cur);
#pragma unroll
        for (int i = 0; i < VEC; ++i) prev[i] = 0.f;
        // neighbour validity is per thread, constant over t: out-of-range neighbours re-read the centre pixel (always valid)
        // and are zeroed afterwards, so the five loads of a t-step are unconditional and go out back to back (a conditional
        // load each cost its own memory round trip: the compiler waits at every join)
        const bool ok_up = vh && hh > 0, ok_dn = vh && hh < H - 1, ok_lf = vw && wv > 0, ok_rt = vw && wv < W - 1;
        const long d_up = ok_up ? -(long)W * C : 0, d_dn = ok_dn ? (long)W * C : 0, d_lf = ok_lf ? -(long)C : 0, d_rt = ok_rt ? (long)C : 0;
        for (int t = t_begin; t < t_end; ++t) {
            const ET* f = x + e0 + (long)t * fstride;
            const bool ok_nx = t + 1 < T;
            Vec<ET, VEC>::load(f + (ok_nx ? fstride : 0), next);
            Vec<ET, VEC>::load(f + d_up, up);
            Vec<ET, VEC>::load(f + d_dn, dn);
            Vec<ET, VEC>::load(f + d_lf, lf);
            Vec<ET, VEC>::load(f + d_rt, rt);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                next[i] = ok_nx ? next[i] : 0.f;
                up[i] = ok_up ? up[i] : 0.f; dn[i] = ok_dn ? dn[i] : 0.f;
                lf[i] = ok_lf ? lf[i] : 0.f; rt[i] = ok_rt ? rt[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float yt = tap3(wt[i][0], wt[i][1], wt[i][2], prev[i], cur[i], next[i]);
                float yh = tap3(wh[i][0], wh[i][1], wh[i][2], up[i], cur[i], dn[i]);
                float yw = tap3(ww[i][0], ww[i][1], ww[i][2], lf[i], cur[i], rt[i]);
                float v = (yt + yh) + yw;
                if (hs) {
                    float u = __builtin_fmaf(sc[i], v, sh[i]);
                    v = u * (fminf(fmaxf(u + 3.0f, 0.0f), 6.0f) / 6.0f);
                }
                y[i] = v;
            }
            if (a.add) {
                float ad[VEC];
                const long apix = ((long)n * T + t) * HW + pix;
                Vec<ET, VEC>::load(reinterpret_cast<const ET*>(a.add) + apix * a.add_c + c0, ad);
                unsigned mb = 0xfu;
                if (a.add_mask) mb = a.add_mask[apix * (a.add_c / 4) + c0 / 4] >> (c0 & 3);
#pragma unroll
                for (int i = 0; i < VEC; ++i) y[i] += ((mb >> i) & 1u) ? ad[i] : 0.f;
            }
            if (a.out_gate) {
                const unsigned gb = a.out_gate[(((long)n * T + t) * HW + pix) * (a.out_c / 4) + c0 / 4] >> (c0 & 3);
#pragma unroll
                for (int i = 0; i < VEC; ++i) y[i] = ((gb >> i) & 1u) ? y[i] : 0.f;
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

// The same stencil with the serial walk over t removed: a thread still owns (pixel, VEC channels) for every frame of its clip, but
// loads a chunk of TB frames -- centre t0-1 .. t0+TB and the four in-plane neighbours of t0 .. t0+TB-1 (+ the addend and its gate byte) --
// with ALL loads issued before the first use: one memory round trip per chunk instead of one per frame (the walk's (prev, cur, next)
// window makes every frame wait for its own loads: 8 dependent round trips at T = 8 were most of the launch's 22 us for 25 MB).
// The tap weights of the thread's VEC = 4 channels are 12 consecutive floats per view: three 16-byte loads instead of twelve scalars.
template <typename ET, int TB>
__global__ __launch_bounds__(kThreads) void mvf_nhwc_apply_chunked(NhwcArgs a) {
    constexpr int VEC = 4;
    const int HW = a.h * a.w, W = a.w, H = a.h, T = a.T, C = a.c;
    const int n = blockIdx.x / a.bands, band = blockIdx.x % a.bands;
    const int cgi = blockIdx.y * a.cgp + (threadIdx.x % a.cgp);
    const int plane = threadIdx.x / a.cgp, nplanes = kThreads / a.cgp;
    const bool active = cgi < a.cg;
    if (!active && !a.stats_part) return;
    const int c0 = active ? cgi * VEC : 0;
    const bool vh = a.mode & MVF_VIEW_H, vw = a.mode & MVF_VIEW_W;
    const bool hs = a.scale != nullptr;
    float wt[VEC][3], wh[VEC][3], ww[VEC][3], sc[VEC], sh[VEC];
    float st1[VEC], st2[VEC], kk[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        st1[i] = st2[i] = 0.f;
        kk[i] = (a.stats_part && a.stats_shift) ? a.stats_shift[c0 + i] : 0.f;
    }
    {
        auto load12 = [&](const float* p, bool on, float (&dst)[VEC][3]) {
            float f[12];
            if (on) {
                const float4 q0 = *reinterpret_cast<const float4*>(p + c0 * 3), q1 = *reinterpret_cast<const float4*>(p + c0 * 3 + 4),
                             q2 = *reinterpret_cast<const float4*>(p + c0 * 3 + 8);
                f[0] = q0.x; f[1] = q0.y; f[2] = q0.z; f[3] = q0.w; f[4] = q1.x; f[5] = q1.y; f[6] = q1.z; f[7] = q1.w;
                f[8] = q2.x; f[9] = q2.y; f[10] = q2.z; f[11] = q2.w;
            } else {
#pragma unroll
                for (int k = 0; k < 12; ++k) f[k] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                dst[i][0] = a.flip ? f[i * 3 + 2] : f[i * 3];
                dst[i][1] = f[i * 3 + 1];
                dst[i][2] = a.flip ? f[i * 3] : f[i * 3 + 2];
            }
        };
        load12(a.wt, true, wt);
        load12(a.wh, vh, wh);
        load12(a.ww, vw, ww);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sc[i] = hs ? a.scale[c0 + i] : 1.f;
            sh[i] = hs ? a.shift[c0 + i] : 0.f;
        }
    }
    const ET* x = reinterpret_cast<const ET*>(a.x);
    ET* out = reinterpret_cast<ET*>(a.out);
    const long fstride = (long)HW * C;                         // one frame
    const int pend = active ? min(HW, (band + 1) * a.pixw) : 0;
    for (int pix = band * a.pixw + plane; pix < pend; pix += nplanes) {
        const int hh = pix / W, wv = pix - hh * W;
        const long e0 = ((long)n * T * HW + pix) * C + c0;     // (n, t=0, pix, c0)
        const long o0 = ((long)n * T * HW + pix) * a.out_c + c0;
        // neighbour validity is per thread, constant over t: out-of-range neighbours re-read the centre pixel and are zeroed afterwards
        const bool ok_up = vh && hh > 0, ok_dn = vh && hh < H - 1, ok_lf = vw && wv > 0, ok_rt = vw && wv < W - 1;
        const long d_up = ok_up ? -(long)W * C : 0, d_dn = ok_dn ? (long)W * C : 0, d_lf = ok_lf ? -(long)C : 0, d_rt = ok_rt ? (long)C : 0;
        for (int t0 = 0; t0 < T; t0 += TB) {
            float cen[TB + 2][VEC], up[TB][VEC], dn[TB][VEC], lf[TB][VEC], rt[TB][VEC], ad[TB][VEC];
            unsigned mb[TB], gb[TB];
#pragma unroll
            for (int k = 0; k < TB + 2; ++k) {                 // frames t0 - 1 .. t0 + TB (clamped into the clip; zeroed below where outside)
                int t = t0 - 1 + k;
                t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
                Vec<ET, VEC>::load(x + e0 + (long)t * fstride, cen[k]);
            }
#pragma unroll
            for (int k = 0; k < TB; ++k) {
                const int t = min(t0 + k, T - 1);
                const ET* f = x + e0 + (long)t * fstride;
                Vec<ET, VEC>::load(f + d_up, up[k]);
                Vec<ET, VEC>::load(f + d_dn, dn[k]);
                Vec<ET, VEC>::load(f + d_lf, lf[k]);
                Vec<ET, VEC>::load(f + d_rt, rt[k]);
                mb[k] = 0xfu;
                gb[k] = 0xfu;
                if (a.out_gate) gb[k] = a.out_gate[(((long)n * T + t) * HW + pix) * (a.out_c / 4) + c0 / 4] >> (c0 & 3);
                if (a.add) {
                    const long apix = ((long)n * T + t) * HW + pix;
                    Vec<ET, VEC>::load(reinterpret_cast<const ET*>(a.add) + apix * a.add_c + c0, ad[k]);
                    if (a.add_mask) mb[k] = a.add_mask[apix * (a.add_c / 4) + c0 / 4] >> (c0 & 3);
                }
            }
#pragma unroll
            for (int k = 0; k < TB; ++k) {
                const int t = t0 + k;
                if (t >= T) break;
                float y[VEC];
                const bool ok_pv = t > 0, ok_nx = t + 1 < T;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float pv = ok_pv ? cen[k][i] : 0.f, cu = cen[k + 1][i], nx = ok_nx ? cen[k + 2][i] : 0.f;
                    const float u_ = ok_up ? up[k][i] : 0.f, d_ = ok_dn ? dn[k][i] : 0.f, l_ = ok_lf ? lf[k][i] : 0.f, r_ = ok_rt ? rt[k][i] : 0.f;
                    float yt = tap3(wt[i][0], wt[i][1], wt[i][2], pv, cu, nx);
                    float yh = tap3(wh[i][0], wh[i][1], wh[i][2], u_, cu, d_);
                    float yw = tap3(ww[i][0], ww[i][1], ww[i][2], l_, cu, r_);
                    float v = (yt + yh) + yw;
                    if (hs) {
                        float u = __builtin_fmaf(sc[i], v, sh[i]);
                        v = u * (fminf(fmaxf(u + 3.0f, 0.0f), 6.0f) / 6.0f);
                    }
                    if (a.add) v += ((mb[k] >> i) & 1u) ? ad[k][i] : 0.f;
                    y[i] = ((gb[k] >> i) & 1u) ? v : 0.f;      // (all ones without an output gate)
                }
                Vec<ET, VEC>::store(out + o0 + (long)t * HW * a.out_c, y);
                if (a.stats_part) {                            // statistics of what is STORED, as bn_stats_kernel would read it back
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        float v = y[i];
                        if constexpr (sizeof(ET) == 2) v = bf16_to_f32(f32_to_bf16(v));
                        const float dlt = v - kk[i];
                        st1[i] += dlt;
                        st2[i] += dlt * dlt;
                    }
                }
            }
        }
    }
    if (a.stats_part) {
        // the workgroup's column sums: the planes of a channel group in order (fixed), one partial row per workgroup
        __shared__ float red[kThreads * 2 * VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            red[(threadIdx.x * VEC + i) * 2] = st1[i];
            red[(threadIdx.x * VEC + i) * 2 + 1] = st2[i];
        }
        __syncthreads();
        if (plane == 0 && active) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float t1 = 0.f, t2 = 0.f;
                for (int q = 0; q < nplanes; ++q) {
                    t1 += red[((q * a.cgp + (threadIdx.x % a.cgp)) * VEC + i) * 2];
                    t2 += red[((q * a.cgp + (threadIdx.x % a.cgp)) * VEC + i) * 2 + 1];
                }
                reinterpret_cast<float2*>(a.stats_part)[(long)(c0 + i) * gridDim.x + blockIdx.x] = make_float2(t1, t2);
            }
        }
    }
}

// [r5] The same stencil through an LDS tile (bf16): the chunked kernel above needs every operand of a chunk in registers at once (66 8-byte loads in flight per
// thread, 256 VGPRs + 44 AGPRs, ONE wave per SIMD) and still fetches every element 5.25 times through L1 -- 32 / 48 us per layer3 launch for 26 / 52 MB
// (0.8 TB/s), three occupancy-1 rounds of "issue, wait, compute".  Here a workgroup owns (clip, band of RB rows, CW channels, ALL T frames): phase 1 copies the
// band + one halo row on each side into LDS with 16-byte loads (<= 16 per thread, all in flight, rows clamped into the image), phase 2 gives every thread
// (pixel, 4 channels) items that walk t with a (prev, cur, next) window and read the four in-plane neighbours from LDS (ds_read_b64, lanes along channels:
// conflict-free).  The addend / gate bytes of an item are streamed from global memory, issued before the walk.  ~100 VGPRs, <= 64 KB of LDS: two workgroups per
// CU, 448 workgroups on the 14 x 14 stages.  Same arithmetic, same order per output element: results bit-identical to the chunked kernel; the statistics'
// partial rows follow this kernel's own grid (mvf_nhwc_stencil_stats_rows asks the same plan).
// n / d for small n with host-made magic (as conv_nhwc.hip's fd_div): l = ceil(log2 d), mul = floor(2^32 (2^l - d) / d) + 1, q = (mulhi(n, mul) + n) >> l
__device__ __forceinline__ int fdiv(int n, unsigned mul, unsigned shr) { return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> shr); }
inline void fdiv_make(unsigned d, unsigned& mul, unsigned& shr) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    shr = l;
}

template <int TT, int CW>
__global__ __launch_bounds__(kThreads, 2) void mvf_nhwc_apply_lds(NhwcArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds_tile[];
    constexpr int VEC = 4, NG = CW / VEC, PPT = CW / 8;        // 4-channel groups per pixel, 16-byte pieces per pixel
    const int W = a.w, H = a.h, HW = H * W, C = a.c, RB = a.rb, RP = RB + 2;
    const int n = blockIdx.x / a.lbands, band = blockIdx.x % a.lbands;
    const int cbase = blockIdx.y * CW, h0 = band * RB;
    const int tid = threadIdx.x;
    const bf16_t* x = reinterpret_cast<const bf16_t*>(a.x);
    bf16_t* out = reinterpret_cast<bf16_t*>(a.out);
    // ---- phase 1: global -> LDS by LDS-DMA (no staging registers), piece index = ((t * RP + r) * W + w) * PPT + q, 16 bytes each; the LDS image is linear
    // in it, so wave v's k-th instruction fills bytes [(k * 256 + 64 v) * 16, + 1 KiB).  Pieces past the tile read out of the descriptor's range (zeros).
    {
        const int npieces = TT * RP * W * PPT;
        const bf16_t* clip = x + (long)n * TT * HW * C + cbase;
        const i32x4 rs = rsrc_words(clip, (unsigned)(((long)TT * HW * C - cbase) * 2));
        const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_tile + (unsigned)(tid >> 6) * 1024u));
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (__builtin_amdgcn_readfirstlane(k * kThreads + (tid & ~63)) < npieces) {          // (wave-uniform)
                const int idx = tid + k * kThreads;
                const int q = idx % PPT, pw = idx / PPT;
                const int t = fdiv(pw, a.fdrw_mul, a.fdrw_shr), rem = pw - t * RP * W;
                const int r = fdiv(rem, a.fdw_mul, a.fdw_shr), w = rem - r * W;
                int h = h0 - 1 + r;
                h = h < 0 ? 0 : (h > H - 1 ? H - 1 : h);           // (rows outside the image are never used: ok_up / ok_dn below)
                const unsigned voff = idx < npieces ? (unsigned)(((t * HW + h * W + w) * C + q * 8) * 2) : 0xffffffffu;
                glds16(rs, lds0 + (unsigned)(k * kThreads * 16), voff);
            }
        }
    }
    // ---- per-thread constants: every item of a thread has the same channel group (kThreads % NG == 0)
    const int g = tid % NG, c0 = cbase + g * VEC;
    const bool vh = a.mode & MVF_VIEW_H, vw = a.mode & MVF_VIEW_W;
    const bool hs = a.scale != nullptr;
    float wt[VEC][3], wh[VEC][3], ww[VEC][3], sc[VEC], sh[VEC], st1[VEC], st2[VEC], kk[VEC];
    {
        auto load12 = [&](const float* p, bool on, float (&dst)[VEC][3]) {
            float f[12];
            if (on) {
                const float4 q0 = *reinterpret_cast<const float4*>(p + c0 * 3), q1 = *reinterpret_cast<const float4*>(p + c0 * 3 + 4),
                             q2 = *reinterpret_cast<const float4*>(p + c0 * 3 + 8);
                f[0] = q0.x; f[1] = q0.y; f[2] = q0.z; f[3] = q0.w; f[4] = q1.x; f[5] = q1.y; f[6] = q1.z; f[7] = q1.w;
                f[8] = q2.x; f[9] = q2.y; f[10] = q2.z; f[11] = q2.w;
            } else {
#pragma unroll
                for (int k = 0; k < 12; ++k) f[k] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                dst[i][0] = a.flip ? f[i * 3 + 2] : f[i * 3];
                dst[i][1] = f[i * 3 + 1];
                dst[i][2] = a.flip ? f[i * 3] : f[i * 3 + 2];
            }
        };
        load12(a.wt, true, wt);
        load12(a.wh, vh, wh);
        load12(a.ww, vw, ww);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sc[i] = hs ? a.scale[c0 + i] : 1.f;
            sh[i] = hs ? a.shift[c0 + i] : 0.f;
            st1[i] = st2[i] = 0.f;
            kk[i] = (a.stats_part && a.stats_shift) ? a.stats_shift[c0 + i] : 0.f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- phase 2
    const bf16_t* tile = reinterpret_cast<const bf16_t*>(lds_tile);
    const int nitems = RB * W * NG;
    const int fs = RP * W * CW;                                    // one frame of the tile, in elements
#pragma unroll 1
    for (int it = tid; it < nitems; it += kThreads) {
        const int p = it / NG, r = fdiv(p, a.fdw_mul, a.fdw_shr), w = p - r * W, h = h0 + r;
        if (h >= H) break;                                         // (items are ordered by row: the rest of this thread's items are below the image too)
        const long pix0 = (long)n * TT * HW + h * W + w;           // (n, t = 0, h, w)
        constexpr int TC = 4;                                      // frames per batch of streamed operands (addend, gate bytes, bn_z): one batch ahead
        struct Side { uint2 ad[TC]; unsigned mb[TC], gb[TC]; };
        auto load_side = [&](int tc, Side& sd) {
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                const long apix = pix0 + (long)(tc + k) * HW;
                sd.mb[k] = 0xfu;
                sd.gb[k] = 0xfu;
                sd.ad[k] = make_uint2(0u, 0u);
                if (a.add) {
                    sd.ad[k] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a.add) + apix * a.add_c + c0);
                    if (a.add_mask) sd.mb[k] = a.add_mask[apix * (a.add_c / 4) + c0 / 4];
                }
                if (a.out_gate) sd.gb[k] = a.out_gate[apix * (a.out_c / 4) + c0 / 4];
            }
        };
        Side sd, sn;
        load_side(0, sd);
        const bool ok_up = vh && h > 0, ok_dn = vh && h < H - 1, ok_lf = vw && w > 0, ok_rt = vw && w < W - 1;
        const bf16_t* ctr = tile + ((r + 1) * W + w) * CW + g * VEC;   // (t = 0, this pixel)
        const int d_up = ok_up ? -W * CW : 0, d_dn = ok_dn ? W * CW : 0, d_lf = ok_lf ? -CW : 0, d_rt = ok_rt ? CW : 0;
        float4 prev = make_float4(0.f, 0.f, 0.f, 0.f), cur = ld4(ctr);
        const int t_end = a.T;                                     // (== TT; a run-time bound keeps the batches a loop: unrolled, two batches of operands per frame group stay live)
#pragma unroll 1
        for (int tc = 0; tc < t_end; tc += TC) {
            if (tc + TC < t_end) load_side(tc + TC, sn);
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                const int t = tc + k;
                const bf16_t* f = ctr + t * fs;
                float4 next = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t + 1 < t_end) next = ld4(f + fs);
                const float4 u4 = ld4(f + d_up), d4 = ld4(f + d_dn), l4 = ld4(f + d_lf), r4 = ld4(f + d_rt);
                const float pv[VEC] = {prev.x, prev.y, prev.z, prev.w}, cu[VEC] = {cur.x, cur.y, cur.z, cur.w}, nx[VEC] = {next.x, next.y, next.z, next.w};
                const float uu[VEC] = {u4.x, u4.y, u4.z, u4.w}, dd[VEC] = {d4.x, d4.y, d4.z, d4.w}, ll[VEC] = {l4.x, l4.y, l4.z, l4.w}, rr[VEC] = {r4.x, r4.y, r4.z, r4.w};
                const float av[VEC] = {__uint_as_float(sd.ad[k].x << 16), __uint_as_float(sd.ad[k].x & 0xffff0000u), __uint_as_float(sd.ad[k].y << 16),
                                       __uint_as_float(sd.ad[k].y & 0xffff0000u)};
                float y[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float u_ = ok_up ? uu[i] : 0.f, d_ = ok_dn ? dd[i] : 0.f, l_ = ok_lf ? ll[i] : 0.f, r_ = ok_rt ? rr[i] : 0.f;
                    float yt = tap3(wt[i][0], wt[i][1], wt[i][2], pv[i], cu[i], nx[i]);
                    float yh = tap3(wh[i][0], wh[i][1], wh[i][2], u_, cu[i], d_);
                    float yw = tap3(ww[i][0], ww[i][1], ww[i][2], l_, cu[i], r_);
                    float v = (yt + yh) + yw;
                    if (hs) {
                        float u = __builtin_fmaf(sc[i], v, sh[i]);
                        v = u * (fminf(fmaxf(u + 3.0f, 0.0f), 6.0f) / 6.0f);
                    }
                    if (a.add) v += ((sd.mb[k] >> i) & 1u) ? av[i] : 0.f;
                    y[i] = ((sd.gb[k] >> i) & 1u) ? v : 0.f;
                }
                st4(out + (pix0 + (long)t * HW) * a.out_c + c0, make_float4(y[0], y[1], y[2], y[3]));
                if (a.stats_part) {                                // statistics of what is STORED
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float v = bf16_to_f32(f32_to_bf16(y[i]));
                        const float dlt = v - kk[i];
                        st1[i] += dlt;
                        st2[i] += dlt * dlt;
                    }
                }
                prev = cur;
                cur = next;
            }
            sd = sn;
        }
    }
    if (a.stats_part) {
        __syncthreads();                                           // the tile is dead: its first bytes carry the reduction
        float* red = reinterpret_cast<float*>(lds_tile);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            red[(tid * VEC + i) * 2] = st1[i];
            red[(tid * VEC + i) * 2 + 1] = st2[i];
        }
        __syncthreads();
        if (tid < NG) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll 4
                for (int q = 0; q < kThreads / NG; ++q) {           // the threads of this channel group, in order
                    t1 += red[((q * NG + tid) * VEC + i) * 2];
                    t2 += red[((q * NG + tid) * VEC + i) * 2 + 1];
                }
                reinterpret_cast<float2*>(a.stats_part)[(long)(c0 + i) * gridDim.x + blockIdx.x] = make_float2(t1, t2);
            }
        }
    }
}

struct LdsPlan { int rb, cw, bands; size_t lds; };
// the LDS-tiled kernel's plan: the widest channel chunk (64 / 32 / 16) whose band of >= 2 rows (or the whole image) + halo fits 64 KB over all T frames, taken when
// it makes >= policy stencil_lds_minwg (150) workgroups; a stage whose whole planes fit (7 x 7: 128 workgroups) may halve the budget once if the chunk stays 64 wide.
// Everything smaller stays on the chunked kernel: with few workgroups the two-phase tile (load all, then compute) has nothing to overlap with.  Measured in the
// step (ms, chunked / tiled): C3 18.52 / 18.24-18.29, C4 31.30 / 31.03-31.05.  The threshold (400 / 150; three alternations): 12 clips per GPU (168 workgroups on
// layer3) 9.11-9.13 / 9.06-9.09, C5 video 4.97-5.00 / 4.94-4.95, C4 30.84-30.89 / 30.76-30.86, the 16-clip inference chains (224) 4.00-4.03 / 4.02-4.05;
// narrower 32-channel tiles forced on those small launches (budget halving without the 64-wide rule) measured 9.38 -> 9.42 and 4.04 -> 4.15.
static bool lds_plan(int T, int H, int W, int cs, int n_clips, LdsPlan& best) {
    static const int on = mvf_policy_int("stencil_lds", 1);
    static const int min_wg = mvf_policy_int("stencil_lds_minwg", 150);
    if (!on || (T != 4 && T != 8 && T != 16)) return false;
    auto bytes = [&](int rb, int cw) { return (((size_t)T * (rb + 2) * W * cw * 2 / 16 + 63) / 64) * 1024; };      // whole 64-piece DMA instructions
    for (size_t budget = 65536; budget >= 32768; budget >>= 1) {
        for (int cw = 64; cw >= 16; cw >>= 1) {
            if (cs % cw) continue;
            int rb = 0;
            for (int cand = std::min(H, 8); cand >= 1; --cand)
                if (bytes(cand, cw) <= budget) { rb = cand; break; }
            if (!(rb >= 2 || (rb >= 1 && rb == H))) continue;
            // even bands: the same number of them with the fewest rows each (a 14-row image in bands of 4 would end on a 2-row band)
            const int bands = (H + rb - 1) / rb;
            rb = (H + bands - 1) / bands;
            const LdsPlan p = {rb, cw, bands, bytes(rb, cw)};
            if (p.lds < (size_t)kThreads * 4 * 2 * sizeof(float)) continue;      // (the statistics' reduction reuses the tile)
            if ((long)n_clips * bands * (cs / cw) >= min_wg && (budget == 65536 || cw == 64)) {
                best = p;
                return true;
            }
            break;
        }
    }
    return false;
}

// the shape rule of the LDS-tiled kernel (operand alignment is checked at the launch): bf16, a slice in whole 16-channel chunks, ONE CLIP of the source tensor
// inside a 32-bit buffer descriptor (the tile's DMA offsets are clip-relative 32-bit byte offsets), and a plan
static bool lds_shape_plan(const mvf_desc_t* d, int out_c, LdsPlan& lp) {
    return d->dtype == MVF_BF16 && d->cs % 16 == 0 && d->c % 8 == 0 && out_c % 4 == 0 && d->n_segment > 0 &&
           (long)d->n_segment * d->h * d->w * d->c * 2 < 0x7ffffff0L &&
           lds_plan(d->n_segment, d->h, d->w, d->cs, d->nt / d->n_segment, lp);
}

template <int TT>
static void launch_lds(const NhwcArgs& a, const LdsPlan& p, hipStream_t st) {
    dim3 grid(a.n_clips * p.bands, a.cs / p.cw);
    if (p.cw == 64) hipLaunchKernelGGL((mvf_nhwc_apply_lds<TT, 64>), grid, dim3(kThreads), p.lds, st, a);
    else if (p.cw == 32) hipLaunchKernelGGL((mvf_nhwc_apply_lds<TT, 32>), grid, dim3(kThreads), p.lds, st, a);
    else hipLaunchKernelGGL((mvf_nhwc_apply_lds<TT, 16>), grid, dim3(kThreads), p.lds, st, a);
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

struct NhwcFlip { int flip; const void* add; int add_c; const unsigned char* add_mask; const unsigned char* out_gate; float* stats_part; const float* stats_shift; int rows_only; };
int mvf_nhwc_fwd_infer_impl2(const mvf_desc_t* d, const void* x, void* out, int out_c, const float* wt, const float* wh,
                             const float* ww, const float* scale, const float* shift, NhwcFlip fl, hipStream_t st) {
    // In-place hazard: a workgroup re-reads neighbour pixels that another workgroup may already have overwritten.
    MVF_REQUIRE(x != out, MVF_EINVAL,
                "mvf_fwd_infer(NHWC): in-place is not supported in this layout (neighbour pixels are re-read from "
                "global memory); pass a separate out buffer");
    NhwcArgs a = {};
    a.x = x; a.out = out; a.wt = wt; a.wh = wh; a.ww = ww; a.scale = scale; a.shift = shift;
    a.nt = d->nt; a.c = d->c; a.h = d->h; a.w = d->w; a.T = d->n_segment; a.cs = d->cs; a.mode = d->mode;
    a.n_clips = d->nt / d->n_segment;
    a.out_c = out_c;
    a.flip = fl.flip;
    a.add = fl.add;
    a.add_c = fl.add_c;
    a.add_mask = fl.add_mask;
    a.out_gate = fl.out_gate;
    a.stats_part = fl.stats_part;
    a.stats_shift = fl.stats_shift;
    const int esz = d->dtype == MVF_F32 ? 4 : 2;
    const bool vec = (d->cs % 4 == 0) && (d->c % 4 == 0) && (out_c % 4 == 0) && (((uintptr_t)x | (uintptr_t)out) % (4 * esz) == 0) &&
                     (!fl.add || (fl.add_c % 4 == 0 && (uintptr_t)fl.add % (4 * esz) == 0));
    MVF_REQUIRE(!fl.out_gate || vec, MVF_EUNSUPPORTED, "nhwc_stencil: the output gate needs the 4-channel kernels (cs, pitches %% 4 == 0, aligned operands)");
    a.cg = vec ? d->cs / 4 : d->cs;
    int cgp = 1;
    while (cgp < a.cg && cgp < kThreads) cgp <<= 1;
    a.cgp = cgp;
    const int HW = d->h * d->w;
    const int nplanes = kThreads / cgp;
    // enough workgroups to fill 256 CUs several times over: 4 pixels per thread-plane (amortises the tap registers) only when that
    // still leaves >= 1024 workgroups -- the 14x14 / 7x7 stages of a 16-32 clip batch got 112-224 workgroups and ran latency-bound
    int per = 4;
    while (per > 1 && (long)a.n_clips * ((HW + nplanes * per - 1) / (nplanes * per)) * ((a.cg + cgp - 1) / cgp) < 1024) per >>= 1;
    int pixw = std::max(nplanes * per, 1);
    while ((long)a.n_clips * ((HW + pixw - 1) / pixw) > 8192 && pixw < HW) pixw *= 2;
    a.pixw = std::min(pixw, HW);
    a.bands = (HW + a.pixw - 1) / a.pixw;
    dim3 grid(a.n_clips * a.bands, (a.cg + cgp - 1) / cgp, 1);
    // [r5] bf16: the LDS-tiled kernel wherever its plan exists
    LdsPlan lp = {};
    // (the plan depends on the shape only -- mvf_nhwc_stencil_stats_rows must name the same partial rows as the launch; operands that break its alignment
    // rules fall back to the chunked kernel, which is an error when partial rows were asked for)
    const bool lds_shape = (!fl.add || fl.add_c % 4 == 0) && lds_shape_plan(d, out_c, lp);
    const bool use_lds = lds_shape && vec && (uintptr_t)x % 16 == 0 && ((uintptr_t)wt | (uintptr_t)(wh ? wh : wt) | (uintptr_t)(ww ? ww : wt)) % 16 == 0;
    MVF_REQUIRE(use_lds || !lds_shape || !fl.stats_part, MVF_EINVAL, "nhwc_stencil: operands of a statistics launch must be 16-byte aligned (x, taps) / 8-byte (out, addend, bn_z)");
    if (use_lds) {
        if (fl.rows_only) return a.n_clips * lp.bands;
        a.rb = lp.rb; a.cw = lp.cw; a.lbands = lp.bands;
        fdiv_make((unsigned)d->w, a.fdw_mul, a.fdw_shr);
        fdiv_make((unsigned)((lp.rb + 2) * d->w), a.fdrw_mul, a.fdrw_shr);
        if (a.T == 4) launch_lds<4>(a, lp, st);
        else if (a.T == 8) launch_lds<8>(a, lp, st);
        else launch_lds<16>(a, lp, st);
        MVF_LAUNCH_CHECK();
    }
    if (fl.rows_only) return (int)grid.x;                  // mvf_nhwc_stencil_stats_rows: the partial-row count of this plan, nothing is launched
    // [r3] all of a chunk's loads in flight at once instead of the serial walk over t (policy stencil_chunked=0: the walk)
    static const int chunked_env = mvf_policy_int("stencil_chunked", 1);
    const bool chunked = chunked_env != 0 && vec && ((uintptr_t)wt | (uintptr_t)(wh ? wh : wt) | (uintptr_t)(ww ? ww : wt)) % 16 == 0;
    MVF_REQUIRE(!fl.stats_part || chunked, MVF_EUNSUPPORTED, "nhwc_stencil_stats: needs the chunked 4-channel kernel (cs, pitches %% 4 == 0, 16-byte aligned taps, policy stencil_chunked != 0)");
    if (use_lds) {
        // launched above
    } else if (d->dtype == MVF_F32) {
        if (chunked) hipLaunchKernelGGL((mvf_nhwc_apply_chunked<float, 4>), grid, dim3(kThreads), 0, st, a);
        else if (vec) hipLaunchKernelGGL((mvf_nhwc_apply<float, 4>), grid, dim3(kThreads), 0, st, a);
        else hipLaunchKernelGGL((mvf_nhwc_apply<float, 1>), grid, dim3(kThreads), 0, st, a);
    } else {
        if (chunked) hipLaunchKernelGGL((mvf_nhwc_apply_chunked<bf16_t, 8>), grid, dim3(kThreads), 0, st, a);
        else if (vec) hipLaunchKernelGGL((mvf_nhwc_apply<bf16_t, 4>), grid, dim3(kThreads), 0, st, a);
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

int mvf_nhwc_fwd_infer_impl(const mvf_desc_t* d, const void* x, void* out, int out_c, const float* wt, const float* wh,
                            const float* ww, const float* scale, const float* shift, hipStream_t st) {
    NhwcFlip f = {0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0};
    return mvf_nhwc_fwd_infer_impl2(d, x, out, out_c, wt, wh, ww, scale, shift, f, st);
}

int mvf_nhwc_fwd_infer(const mvf_desc_t* d, const void* x, void* out, const float* wt, const float* wh,
                       const float* ww, const float* scale, const float* shift, hipStream_t st) {
    return mvf_nhwc_fwd_infer_impl(d, x, out, d->c, wt, wh, ww, scale, shift, st);
}

// ---- channels-last training entry points: composed from the engine's primitives (same arithmetic as the NCHW kernels) ----
namespace {
struct NhwcTrainWs {            // byte offsets into the caller's workspace
    size_t y, dz, coef, bn, tap, total;
};
NhwcTrainWs nhwc_train_ws(const mvf_desc_t* d) {
    const size_t esz = d->dtype == MVF_F32 ? 4 : 2;
    const size_t m = (size_t)d->nt * d->h * d->w;
    NhwcTrainWs w;
    size_t o = 0;
    w.y = o;    o += align_up(m * d->cs * esz, 256);
    w.dz = o;   o += align_up(m * d->cs * esz, 256);
    w.coef = o; o += align_up((size_t)4 * d->cs * sizeof(float), 256);      // scale, shift, zero dgamma, zero dbeta
    w.bn = o;   o += align_up(mvf_bn_workspace_bytes((long)m, d->cs), 256);
    w.tap = o;  o += align_up(mvf_nhwc_tapgrad_workspace_bytes(d), 256);
    w.total = o;
    return w;
}
__global__ void bn_coef_kernel(const float* gamma, const float* beta, const float* mean, const float* invstd, int c, float* coef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const float s = gamma[i] * invstd[i];
    coef[i] = s;                       // scale
    coef[c + i] = beta[i] - mean[i] * s;   // shift
    coef[2 * c + i] = 0.f;             // zero dgamma / dbeta for the eval-mode (no batch-statistics term) apply
    coef[3 * c + i] = 0.f;
}
}  // namespace

size_t mvf_nhwc_ws_fwd_train(const mvf_desc_t* d) { return d->cs % 4 ? 256 : nhwc_train_ws(d).total; }
size_t mvf_nhwc_ws_bwd(const mvf_desc_t* d) { return d->cs % 4 ? 256 : nhwc_train_ws(d).total; }

int mvf_nhwc_fwd_train(const mvf_desc_t* d, const void* x, void* out, const float* wt, const float* wh, const float* ww,
                       const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                       float* save_mean, float* save_invstd, void* ws, hipStream_t st) {
    MVF_REQUIRE(d->cs % 4 == 0 && d->c % 4 == 0, MVF_EUNSUPPORTED, "mvf_fwd_train(NHWC): needs cs %% 4 == 0 and c %% 4 == 0 (use MVF_NCHW)");
    const NhwcTrainWs w = nhwc_train_ws(d);
    char* base = (char*)ws;
    const long m = (long)d->nt * d->h * d->w;
    float* scale = (float*)(base + w.coef);
    float* shift = scale + d->cs;
    // pass 1: y = stencil(x) (compact slice) -> batch statistics (and the folded scale / shift); pass 2: the stencil again with
    // BN + hard-swish fused, written into `out` (the pass-through channels are copied there): 2 slice reads + 2 slice writes
    int rc = mvf_nhwc_stencil(d, x, d->c, base + w.y, d->cs, wt, wh, ww, nullptr, nullptr, 0, nullptr, 0, nullptr, st);
    if (rc) return rc;
    rc = mvf_bn_train_stats(base + w.y, m, d->cs, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale, shift,
                            base + w.bn, mvf_bn_workspace_bytes(m, d->cs), d->dtype, st);
    if (rc) return rc;
    return mvf_nhwc_fwd_infer_impl(d, x, out, d->c, wt, wh, ww, scale, shift, st);
}

int mvf_nhwc_bwd(const mvf_desc_t* d, const void* g, const void* x, const float* wt, const float* wh, const float* ww,
                 const float* gamma, const float* beta, const float* mean, const float* invstd, int training, void* dx, float* dwt,
                 float* dwh, float* dww, float* dgamma, float* dbeta, void* ws, hipStream_t st) {
    MVF_REQUIRE(d->cs % 4 == 0 && d->c % 4 == 0, MVF_EUNSUPPORTED, "mvf_bwd(NHWC): needs cs %% 4 == 0 and c %% 4 == 0 (use MVF_NCHW)");
    MVF_REQUIRE(gamma == nullptr || (dgamma && dbeta), MVF_EINVAL, "mvf_bwd(NHWC): dgamma / dbeta are NULL");
    const NhwcTrainWs w = nhwc_train_ws(d);
    char* base = (char*)ws;
    const long m = (long)d->nt * d->h * d->w;
    const void* dy = g;                 // gradient w.r.t. the stencil output, pitch dy_c
    int dy_c = d->c;
    int rc;
    if (gamma) {
        float* coef = (float*)(base + w.coef);
        hipLaunchKernelGGL(bn_coef_kernel, dim3((d->cs + 255) / 256), dim3(256), 0, st, gamma, beta, mean, invstd, d->cs, coef);
        MVF_LAUNCH_CHECK();
        rc = mvf_nhwc_stencil(d, x, d->c, base + w.y, d->cs, wt, wh, ww, nullptr, nullptr, 0, nullptr, 0, nullptr, st);   // recompute y
        if (rc) return rc;
        rc = mvf_bn_bwd_reduce(g, d->c, base + w.y, nullptr, m, d->cs, mean, invstd, coef, coef + d->cs, 3, nullptr, dgamma, dbeta,
                               base + w.bn, mvf_bn_workspace_bytes(m, d->cs), d->dtype, st);
        if (rc) return rc;
        // eval-mode BN has no batch-statistics term: the apply sees zero sums
        const float* ag = training ? dgamma : coef + 2 * d->cs;
        const float* ab = training ? dbeta : coef + 3 * d->cs;
        rc = mvf_bn_bwd_apply_masked(g, d->c, base + w.y, nullptr, m, d->cs, gamma, mean, invstd, coef, coef + d->cs, ag, ab, 3, base + w.dz,
                                     d->dtype, st);
        if (rc) return rc;
        dy = base + w.dz;
        dy_c = d->cs;
    }
    rc = mvf_nhwc_tapgrad(d, x, d->c, dy, dy_c, dwt, dwh, dww, base + w.tap, mvf_nhwc_tapgrad_workspace_bytes(d), st);
    if (rc) return rc;
    if (dx != g && d->cs < d->c) {      // pass-through channels
        const long npix = m;
        const int blocks = (int)std::min<long>((npix * (d->c - d->cs) + 255) / 256, 256L * 16);
        if (d->dtype == MVF_F32)
            hipLaunchKernelGGL(copy_tail_nhwc<float>, dim3(blocks), dim3(256), 0, st, (const float*)g, (float*)dx, npix, d->c, d->cs);
        else
            hipLaunchKernelGGL(copy_tail_nhwc<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)g, (bf16_t*)dx, npix, d->c, d->cs);
        MVF_LAUNCH_CHECK();
    }
    if (dy == g && dx == g) {           // use_hs == False and in place: the transposed stencil must not read what it overwrites
        MVF_HIP_OK(hipMemcpy2DAsync(base + w.dz, (size_t)d->cs * (d->dtype == MVF_F32 ? 4 : 2), g, (size_t)d->c * (d->dtype == MVF_F32 ? 4 : 2),
                                    (size_t)d->cs * (d->dtype == MVF_F32 ? 4 : 2), (size_t)m, hipMemcpyDeviceToDevice, st));
        dy = base + w.dz;
        dy_c = d->cs;
    }
    return mvf_nhwc_stencil(d, dy, dy_c, dx, d->c, wt, wh, ww, nullptr, nullptr, 1, nullptr, 0, nullptr, st);
}

namespace {

// 7 distinct tap-gradient sums per channel: dw[j] = sum_p dy[p] * s[p + (j-1)] for the t/h/w views (centre shared).
// thread = (pixel lane, 4 channels); t slides in registers for s; partials [block][c][7] reduced by a second kernel.
template <typename ET>
__global__ __launch_bounds__(kThreads) void mvf_nhwc_tapgrad_kernel(const ET* x, int x_c, const ET* dy, int dy_c, int nt, int h, int w, int T,
                                                                    int cs, int cgp, int pixw, int bands, float* part) {
    __shared__ float4 red[kThreads];
    const int HW = h * w;
    const int n = blockIdx.x / bands, band = blockIdx.x % bands;
    const int cgi = blockIdx.y * cgp + (threadIdx.x % cgp);
    const int plane = threadIdx.x / cgp, nplanes = kThreads / cgp;
    const bool ok = cgi * 4 < cs;
    const int c0 = cgi * 4;
    float4 s[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) s[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int pend = min(HW, (band + 1) * pixw);
    if (ok) {
        for (int pix = band * pixw + plane; pix < pend; pix += nplanes) {
            const int hh = pix / w, wv = pix - hh * w;
            const long p0 = (long)n * T * HW + pix;
            float4 xprev = make_float4(0.f, 0.f, 0.f, 0.f), xcur = ld4(x + p0 * x_c + c0), xnext;
            for (int t = 0; t < T; ++t) {
                const long pp = p0 + (long)t * HW;
                const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
                xnext = (t + 1 < T) ? ld4(x + (pp + HW) * x_c + c0) : zero;
                const float4 xup = hh > 0 ? ld4(x + (pp - w) * x_c + c0) : zero;
                const float4 xdn = hh < h - 1 ? ld4(x + (pp + w) * x_c + c0) : zero;
                const float4 xlf = wv > 0 ? ld4(x + (pp - 1) * x_c + c0) : zero;
                const float4 xrt = wv < w - 1 ? ld4(x + (pp + 1) * x_c + c0) : zero;
                const float4 d = ld4(dy + pp * dy_c + c0);
#define ACC(k, v) s[k].x += d.x * v.x; s[k].y += d.y * v.y; s[k].z += d.z * v.z; s[k].w += d.w * v.w;
                ACC(0, xprev) ACC(1, xnext) ACC(2, xup) ACC(3, xdn) ACC(4, xlf) ACC(5, xrt) ACC(6, xcur)
#undef ACC
                xprev = xcur;
                xcur = xnext;
            }
        }
    }
    // reduce over the pixel lanes (planes) of the block, one accumulator at a time
    for (int i = 0; i < 7; ++i) {
        __syncthreads();
        red[threadIdx.x] = s[i];
        __syncthreads();
        if (plane == 0 && ok) {
            float4 tsum = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int r = 0; r < nplanes; ++r) {
                const float4 v = red[r * cgp + (threadIdx.x % cgp)];
                tsum.x += v.x; tsum.y += v.y; tsum.z += v.z; tsum.w += v.w;
            }
            float* p = part + ((long)blockIdx.x * cs + c0) * 7;
            p[i] = tsum.x; p[7 + i] = tsum.y; p[14 + i] = tsum.z; p[21 + i] = tsum.w;
        }
    }
}

// Row-parallel form of the same sums: the rows (n, t, h, w) of the slice are cut into `nblk` contiguous ranges, a thread owns 4
// channels of one row lane and walks its range with stride `nplanes`; the six neighbours (t-1, t+1, up, down, left, right) are
// loaded UNCONDITIONALLY from a clamped row (the row itself when the neighbour is outside) and zeroed by a select afterwards, so the
// eight loads of an iteration share one memory round trip (the (clip, band) kernel above waits for five of its seven loads one by
// one and launches only clips x bands = 128 workgroups on the 14x14 layers: 116 us for 26 MB of reads).  Same partial layout
// [block][c][7], same fixed-order finalize.
template <typename ET>
__global__ __launch_bounds__(kThreads) void mvf_nhwc_tapgrad_rows_kernel(const ET* x, int x_c, const ET* dy, int dy_c, int rows, int h, int w, int T,
                                                                         int cs, int cgp, int rows_per_blk, float* part) {
    __shared__ float4 red[kThreads];
    const int HW = h * w;
    const int cgi = blockIdx.y * cgp + (threadIdx.x % cgp);
    const int plane = threadIdx.x / cgp, nplanes = kThreads / cgp;
    const bool ok = cgi * 4 < cs;
    const int c0 = cgi * 4;
    float4 s[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) s[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int r_begin = blockIdx.x * rows_per_blk, r_end = min(rows, r_begin + rows_per_blk);
    if (ok) {
        for (int r = r_begin + plane; r < r_end; r += nplanes) {
            const int f = r / HW, pix = r - f * HW;          // frame (n*T + t), pixel
            const int t = f % T, hh = pix / w, wv = pix - hh * w;
            const bool vp = t > 0, vn = t + 1 < T, vu = hh > 0, vd = hh + 1 < h, vl = wv > 0, vr = wv + 1 < w;
            const long b = (long)r * x_c + c0;
            const float4 xc = ld4(x + b);
            const float4 xp = ld4(x + (vp ? b - (long)HW * x_c : b));
            const float4 xn = ld4(x + (vn ? b + (long)HW * x_c : b));
            const float4 xu = ld4(x + (vu ? b - (long)w * x_c : b));
            const float4 xd = ld4(x + (vd ? b + (long)w * x_c : b));
            const float4 xl = ld4(x + (vl ? b - x_c : b));
            const float4 xr = ld4(x + (vr ? b + x_c : b));
            const float4 d = ld4(dy + (long)r * dy_c + c0);
#define ACCM(k, v, on) { const float m = (on) ? 1.f : 0.f; s[k].x += d.x * (v.x * m); s[k].y += d.y * (v.y * m); s[k].z += d.z * (v.z * m); s[k].w += d.w * (v.w * m); }
            ACCM(0, xp, vp) ACCM(1, xn, vn) ACCM(2, xu, vu) ACCM(3, xd, vd) ACCM(4, xl, vl) ACCM(5, xr, vr) ACCM(6, xc, true)
#undef ACCM
        }
    }
    for (int i = 0; i < 7; ++i) {
        __syncthreads();
        red[threadIdx.x] = s[i];
        __syncthreads();
        if (plane == 0 && ok) {
            float4 tsum = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = 0; q < nplanes; ++q) {
                const float4 v = red[q * cgp + (threadIdx.x % cgp)];
                tsum.x += v.x; tsum.y += v.y; tsum.z += v.z; tsum.w += v.w;
            }
            float* p = part + ((long)blockIdx.x * cs + c0) * 7;
            p[i] = tsum.x; p[7 + i] = tsum.y; p[14 + i] = tsum.z; p[21 + i] = tsum.w;
        }
    }
}

// 16 (channel, tap sum) outputs x 16 block-lanes per workgroup: lane l sums blocks l, l+16, ... (two accumulators), the 16 lane
// sums are combined in lane order through LDS -> fixed summation order, 64-byte coalesced reads, cs*7/16 workgroups
__global__ __launch_bounds__(256) void mvf_tapgrad_finalize_rows_kernel(const float* part, int nblk, int cs, int mode, float* dwt, float* dwh, float* dww) {
    __shared__ double red[16][16];
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + e;
    const int n = cs * 7;
    double a0 = 0, a1 = 0;
    if (j < n) {
        int b = sl;
        for (; b + 16 < nblk; b += 32) {
            a0 += part[(long)b * n + j];
            a1 += part[(long)(b + 16) * n + j];
        }
        if (b < nblk) a0 += part[(long)b * n + j];
    }
    red[sl][e] = a0 + a1;
    __syncthreads();
    if (sl != 0 || j >= n) return;
    double t = 0;
    for (int q = 0; q < 16; ++q) t += red[q][e];
    const float v = (float)t;
    const int c = j / 7, i = j - c * 7;
    const bool on_h = mode & MVF_VIEW_H, on_w = mode & MVF_VIEW_W;
    switch (i) {
        case 0: dwt[c * 3 + 0] = v; break;
        case 1: dwt[c * 3 + 2] = v; break;
        case 2: if (dwh) dwh[c * 3 + 0] = on_h ? v : 0.f; break;
        case 3: if (dwh) dwh[c * 3 + 2] = on_h ? v : 0.f; break;
        case 4: if (dww) dww[c * 3 + 0] = on_w ? v : 0.f; break;
        case 5: if (dww) dww[c * 3 + 2] = on_w ? v : 0.f; break;
        default:
            dwt[c * 3 + 1] = v;
            if (dwh) dwh[c * 3 + 1] = on_h ? v : 0.f;
            if (dww) dww[c * 3 + 1] = on_w ? v : 0.f;
    }
}

__global__ void mvf_tapgrad_finalize_kernel(const float* part, int nblk, int cs, int mode, float* dwt, float* dwh, float* dww) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cs) return;
    double s[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < nblk; ++b)
        for (int i = 0; i < 7; ++i) s[i] += part[((long)b * cs + c) * 7 + i];
    dwt[c * 3 + 0] = (float)s[0]; dwt[c * 3 + 1] = (float)s[6]; dwt[c * 3 + 2] = (float)s[1];
    if (dwh) { const bool on = mode & MVF_VIEW_H; dwh[c * 3 + 0] = on ? (float)s[2] : 0.f; dwh[c * 3 + 1] = on ? (float)s[6] : 0.f; dwh[c * 3 + 2] = on ? (float)s[3] : 0.f; }
    if (dww) { const bool on = mode & MVF_VIEW_W; dww[c * 3 + 0] = on ? (float)s[4] : 0.f; dww[c * 3 + 1] = on ? (float)s[6] : 0.f; dww[c * 3 + 2] = on ? (float)s[5] : 0.f; }
}

struct TapRows { int cgp, gy, rows_per_blk, nblk; };
TapRows tap_rows_plan(long rows, int cs) {
    TapRows p;
    const int cg = cs / 4;
    int cgp = 1;
    while (cgp < cg && cgp < kThreads) cgp <<= 1;
    p.cgp = cgp;
    p.gy = (cg + cgp - 1) / cgp;
    const int nplanes = kThreads / cgp;
    // ~512 workgroups (2 per CU) per channel slab, at least 4 rows per row lane
    long rpb = std::max<long>((rows + 511) / 512, (long)nplanes * 4);
    rpb = (rpb + nplanes - 1) / nplanes * nplanes;
    p.rows_per_blk = (int)rpb;
    p.nblk = (int)((rows + rpb - 1) / rpb);
    return p;
}
int g_tap_rows = -1;      // policy tapgrad_rows=0 keeps the (clip, band) kernel (A/B switch)
bool tap_rows_on() {
    if (g_tap_rows < 0) {
        g_tap_rows = mvf_policy_int("tapgrad_rows", 1) != 0 ? 1 : 0;
    }
    return g_tap_rows != 0;
}

struct TapPlan { int cgp, pixw, bands, gy; };
TapPlan tap_plan(int n_clips, int hw, int cs) {
    TapPlan p;
    const int cg = cs / 4;
    int cgp = 1;
    while (cgp < cg && cgp < kThreads) cgp <<= 1;
    p.cgp = cgp;
    const int nplanes = kThreads / cgp;
    int pixw = std::max(nplanes * 8, 1);
    while ((long)n_clips * ((hw + pixw - 1) / pixw) > 2048 && pixw < hw) pixw *= 2;
    p.pixw = std::min(pixw, hw);
    p.bands = (hw + p.pixw - 1) / p.pixw;
    p.gy = (cg + cgp - 1) / cgp;
    return p;
}

}  // namespace

extern "C" {

// Engine primitive (channels-last): out[..., :cs] (pitch out_c) = f( stencil_{T,H,W}( x[..., :cs] (pitch x_c) ) ),
// f = hswish(scale*y+shift) when scale != NULL else identity; flip != 0 uses reversed taps (the transposed stencil,
// i.e. the data-gradient of the three views).  x and out must not alias.
int mvf_nhwc_stencil(const mvf_desc_t* d, const void* x, int x_c, void* out, int out_c, const float* w_t, const float* w_h,
                     const float* w_w, const float* scale, const float* shift, int flip, const void* addend, int addend_c,
                     const unsigned char* addend_sign_bits, void* stream) {
    MVF_REQUIRE(d && x && out && w_t && x != out && x_c >= d->cs && out_c >= d->cs, MVF_EINVAL, "nhwc_stencil: bad argument");
    MVF_REQUIRE(!addend || addend_c >= d->cs, MVF_EINVAL, "nhwc_stencil: addend pitch < cs");
    MVF_REQUIRE(!addend_sign_bits || (addend && addend_c % 4 == 0), MVF_EINVAL, "nhwc_stencil: gate bits need an addend with pitch % 4 == 0");
    mvf_desc_t dd = *d;
    dd.c = x_c;
    NhwcFlip f = {flip, addend, addend_c, addend_sign_bits, nullptr, nullptr, nullptr, 0};
    return mvf_nhwc_fwd_infer_impl2(&dd, x, out, out_c, w_t, w_h, w_w, scale, shift, f, (hipStream_t)stream);
}

// [r5] ... with the OUTPUT gated per channel by out_gate_bits ([pixels][out_c/4] bytes, the sign bits of the block output this gradient belongs to):
// out[..., :cs] = (f(stencil(x)) + gated addend) * [bit] -- the transposed stencil then hands the block below gm = g * [out > 0] for the slice, as
// mvf_conv2d_nhwc_fwd_resmask_gate does for the channels >= cs
int mvf_nhwc_stencil_gate(const mvf_desc_t* d, const void* x, int x_c, void* out, int out_c, const float* w_t, const float* w_h,
                          const float* w_w, const float* scale, const float* shift, int flip, const void* addend, int addend_c,
                          const unsigned char* addend_sign_bits, const unsigned char* out_gate_bits, void* stream) {
    MVF_REQUIRE(d && x && out && w_t && x != out && x_c >= d->cs && out_c >= d->cs && out_gate_bits, MVF_EINVAL, "nhwc_stencil_gate: bad argument");
    MVF_REQUIRE(!addend || addend_c >= d->cs, MVF_EINVAL, "nhwc_stencil_gate: addend pitch < cs");
    MVF_REQUIRE(!addend_sign_bits || (addend && addend_c % 4 == 0), MVF_EINVAL, "nhwc_stencil_gate: gate bits need an addend with pitch % 4 == 0");
    mvf_desc_t dd = *d;
    dd.c = x_c;
    NhwcFlip f = {flip, addend, addend_c, addend_sign_bits, out_gate_bits, nullptr, nullptr, 0};
    return mvf_nhwc_fwd_infer_impl2(&dd, x, out, out_c, w_t, w_h, w_w, scale, shift, f, (hipStream_t)stream);
}

// [r5] mvf_nhwc_stencil_gate that ALSO accumulates the column sums of the gated slice it stores: sums_part CHANNEL-MAJOR
// [cs][mvf_nhwc_stencil_stats_rows(d, x_c, out_c)][2] = per-workgroup sums of gm and gm^2 -- the slice's share of mvf_conv2d_nhwc_fwd_resmask_gate_colsums
// (the dz3-free block below takes bn3's dbeta from them and dgamma from its weight-gradient GEMM: mvf_bn_bwd_dzfree_sums)
int mvf_nhwc_stencil_gate_colsums(const mvf_desc_t* d, const void* x, int x_c, void* out, int out_c, const float* w_t, const float* w_h,
                                  const float* w_w, int flip, const void* addend, int addend_c, const unsigned char* addend_sign_bits,
                                  const unsigned char* out_gate_bits, float* sums_part, void* stream) {
    MVF_REQUIRE(d && x && out && w_t && x != out && x_c >= d->cs && out_c >= d->cs && out_gate_bits && sums_part, MVF_EINVAL, "nhwc_stencil_gate_colsums: bad argument");
    MVF_REQUIRE(!addend || addend_c >= d->cs, MVF_EINVAL, "nhwc_stencil_gate_colsums: addend pitch < cs");
    MVF_REQUIRE(!addend_sign_bits || (addend && addend_c % 4 == 0), MVF_EINVAL, "nhwc_stencil_gate_colsums: gate bits need an addend with pitch % 4 == 0");
    mvf_desc_t dd = *d;
    dd.c = x_c;
    NhwcFlip f = {flip, addend, addend_c, addend_sign_bits, out_gate_bits, sums_part, nullptr, 0};
    return mvf_nhwc_fwd_infer_impl2(&dd, x, out, out_c, w_t, w_h, w_w, nullptr, nullptr, f, (hipStream_t)stream);
}

// [r5] the plain stencil (y = taps * x-slice, no activation) that ALSO accumulates the batch statistics of MVF's BatchNorm3d (MVF.py:131-134, training
// mode) over what it stores: stats_part CHANNEL-MAJOR [cs][mvf_nhwc_stencil_stats_rows(d, ...)][2] = per-workgroup sums of (y - K), (y - K)^2 with
// K = stats_shift (the old running mean; NULL = 0) -> mvf_bn_train_finalize.  Replaces the separate statistics pass over y (mvf_bn_train_stats).
// [r6] Which kernel a stencil launch of this shape takes (16-byte aligned operands assumed): 1 = the LDS-tiled kernel (rows_per_band / chan_per_wg = its tile),
// 0 = the register-chunked kernel (fp32, slices not in 16-channel chunks, a clip larger than a 32-bit buffer descriptor, launches below the workgroup threshold).
// A plan query for tests and tooling; nothing is launched.
int mvf_nhwc_stencil_tile_plan(const mvf_desc_t* d, int x_c, int out_c, int* rows_per_band, int* chan_per_wg) {
    if (!d || d->cs <= 0 || d->nt <= 0 || d->n_segment <= 0) return 0;
    mvf_desc_t dd = *d;
    dd.c = x_c;
    LdsPlan lp = {};
    if (!lds_shape_plan(&dd, out_c, lp)) return 0;
    if (rows_per_band) *rows_per_band = lp.rb;
    if (chan_per_wg) *chan_per_wg = lp.cw;
    return 1;
}

int mvf_nhwc_stencil_stats_rows(const mvf_desc_t* d, int x_c, int out_c) {
    if (!d || d->cs <= 0 || d->nt <= 0) return 0;
    mvf_desc_t dd = *d;
    dd.c = x_c;
    NhwcFlip f = {0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 1};
    return mvf_nhwc_fwd_infer_impl2(&dd, (const void*)16, (void*)32, out_c, (const float*)16, (const float*)16, (const float*)16, nullptr, nullptr, f, nullptr);
}
int mvf_nhwc_stencil_stats(const mvf_desc_t* d, const void* x, int x_c, void* out, int out_c, const float* w_t, const float* w_h,
                           const float* w_w, float* stats_part, const float* stats_shift, void* stream) {
    MVF_REQUIRE(d && x && out && w_t && x != out && x_c >= d->cs && out_c >= d->cs && stats_part, MVF_EINVAL, "nhwc_stencil_stats: bad argument");
    mvf_desc_t dd = *d;
    dd.c = x_c;
    NhwcFlip f = {0, nullptr, 0, nullptr, nullptr, stats_part, stats_shift, 0};
    return mvf_nhwc_fwd_infer_impl2(&dd, x, out, out_c, w_t, w_h, w_w, nullptr, nullptr, f, (hipStream_t)stream);
}

size_t mvf_nhwc_tapgrad_workspace_bytes(const mvf_desc_t* d) {
    if (!d || d->cs <= 0 || d->cs % 4) return 0;
    TapPlan p = tap_plan(d->nt / d->n_segment, d->h * d->w, d->cs);
    TapRows q = tap_rows_plan((long)d->nt * d->h * d->w, d->cs);
    const size_t a = (size_t)(d->nt / d->n_segment) * p.bands * d->cs * 7 * sizeof(float), b = (size_t)q.nblk * d->cs * 7 * sizeof(float);
    return align_up(std::max(a, b), 256);
}

// dw_t/dw_h/dw_w [cs][3] = tap gradients of the three views given dy (pitch dy_c) and the forward input x (pitch x_c)
int mvf_nhwc_tapgrad(const mvf_desc_t* d, const void* x, int x_c, const void* dy, int dy_c, float* dw_t, float* dw_h, float* dw_w,
                     void* ws, size_t ws_bytes, void* stream) {
    MVF_REQUIRE(d && x && dy && dw_t && d->cs % 4 == 0 && x_c % 4 == 0 && dy_c % 4 == 0, MVF_EINVAL, "nhwc_tapgrad: bad argument (cs %% 4?)");
    MVF_REQUIRE(ws && ws_bytes >= mvf_nhwc_tapgrad_workspace_bytes(d), MVF_EWS, "nhwc_tapgrad: workspace too small");
    const int n_clips = d->nt / d->n_segment;
    hipStream_t st = (hipStream_t)stream;
    if (tap_rows_on() && (long)d->nt * d->h * d->w < (1L << 31)) {
        const int rows = d->nt * d->h * d->w;
        TapRows q = tap_rows_plan(rows, d->cs);
        dim3 grid(q.nblk, q.gy);
        if (d->dtype == MVF_F32)
            hipLaunchKernelGGL(mvf_nhwc_tapgrad_rows_kernel<float>, grid, dim3(kThreads), 0, st, (const float*)x, x_c, (const float*)dy, dy_c, rows, d->h, d->w,
                               d->n_segment, d->cs, q.cgp, q.rows_per_blk, (float*)ws);
        else
            hipLaunchKernelGGL(mvf_nhwc_tapgrad_rows_kernel<bf16_t>, grid, dim3(kThreads), 0, st, (const bf16_t*)x, x_c, (const bf16_t*)dy, dy_c, rows, d->h, d->w,
                               d->n_segment, d->cs, q.cgp, q.rows_per_blk, (float*)ws);
        MVF_LAUNCH_CHECK();
        hipLaunchKernelGGL(mvf_tapgrad_finalize_rows_kernel, dim3((d->cs * 7 + 15) / 16), dim3(256), 0, st, (const float*)ws, q.nblk, d->cs, d->mode, dw_t, dw_h, dw_w);
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    TapPlan p = tap_plan(n_clips, d->h * d->w, d->cs);
    dim3 grid(n_clips * p.bands, p.gy);
    if (d->dtype == MVF_F32)
        hipLaunchKernelGGL(mvf_nhwc_tapgrad_kernel<float>, grid, dim3(kThreads), 0, st, (const float*)x, x_c, (const float*)dy, dy_c, d->nt, d->h, d->w, d->n_segment, d->cs, p.cgp, p.pixw, p.bands, (float*)ws);
    else
        hipLaunchKernelGGL(mvf_nhwc_tapgrad_kernel<bf16_t>, grid, dim3(kThreads), 0, st, (const bf16_t*)x, x_c, (const bf16_t*)dy, dy_c, d->nt, d->h, d->w, d->n_segment, d->cs, p.cgp, p.pixw, p.bands, (float*)ws);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(mvf_tapgrad_finalize_kernel, dim3((d->cs + 63) / 64), dim3(64), 0, st, (const float*)ws, n_clips * p.bands, d->cs, d->mode, dw_t, dw_h, dw_w);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // extern "C"
