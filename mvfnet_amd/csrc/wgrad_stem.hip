// Weight gradient of the stem conv (reference codes/models/backbones/resnet.py:448-452, _make_stem_layer: Conv2d(3, 64, 7, stride 2, pad 3); autograd's
// weight gradient) as a DIRECT kernel for gfx950, bf16 storage, fp32 accumulation.  The forward sees the stem as 7 x 1 taps over the 32-"channel" view
// of the zero-padded NHWC4 operand (mvf_stem_prep: 8 pixels x 4 channels per output pixel and kernel row, stride 2), so
//
//   dW[co][kh][kw'][c] = sum over (n, oh, ow) of dZ[n][oh][ow][co] * Xp[n][2 oh + kh][2 ow + kw'][c]          kw' < 8 (the 8th is a zero tap), c < 4
//
// Why: the implicit-GEMM weight gradient (wgrad_nhwc.hip) runs this as M = N Ho Wo, N = 64, K = 224 and re-stages the padded input once per kernel row:
// 224 us per launch at the C3 shape (32 clips x 8 frames of 224^2) for ~100 us of bytes (dZ 411 MB + input 103 MB), and it is the LAST kernel of the
// backward pass: nothing is left to overlap it with.  Here (the construction of wgrad3x3_c64.hip)
//   * a workgroup (4 waves, three workgroups per CU) walks bands of R = 2 output rows.  The band's dZ rows (R Wo pixels x 128 bytes) and the 2 R + 5
//     input rows under them are each ONE contiguous range of global memory and are staged once by LDS-DMA: dZ as a [pixel][64 channels] image with
//     XOR-swizzled 16-byte units, the input rows raw;
//   * the contraction runs over the band's pixels in steps of 16 (Wo % 16 == 0: a step never straddles rows).  dZ comes out of its image through the
//     transpose read ds_read_b64_tr_b16 as in wgrad_bf16_kernel; for kernel row kh the 32 "channels" (kw', c) of output pixel (oh, ow) are the 64
//     contiguous bytes at input pixel (2 oh + kh, 2 ow): the transpose read takes a per-lane address, so the same instruction reads this virtual
//     [pixel][32] image whose 64-byte rows OVERLAP (pixel pitch 16 bytes) straight out of the raw rows -- seven kernel rows = seven row offsets;
//   * a wave owns 32 output channels x 4 (or 3) kernel rows: 64 accumulator registers, one dZ fragment feeds its four matrix instructions;
//   * workgroups are persistent and write ONE fp32 partial slab [64][7][8][4] each in wgrad_reduce_kernel's layout (deterministic, no atomics).
// Arithmetic: the same bf16 products as the implicit GEMM, accumulated in fp32 in a different (band-major) order.
#include <algorithm>
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));

struct KArgs {
    const char* dz;                           // (N, Ho, Wo, 64) bf16
    const char* x;                            // (N, H, W, 4) bf16, padded
    float* part;                              // [nwg][64][224]
    int N, H, W, Ho, Wo;
    int bands_per_frame, bands;
    int rowb;                                 // input row pitch in bytes (W * 8)
    int xbytes, dpix;                         // staged input bytes (2 R + 5 rows), dZ pixels (R Wo)
    int ksr;                                  // k-steps per output row (Wo / 16)
    unsigned fd_bpf_mul, fd_bpf_shr, fd_ksr_mul, fd_ksr_shr;
};

__device__ __forceinline__ int fdiv(int n, unsigned mul, unsigned shr) { return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> shr); }
__device__ __forceinline__ int swzf(int pixel) { return ((pixel >> 1) & 1) << 2; }      // (wgrad_bf16_kernel's swz16<8>)

template <int kR>                             // output rows per band
__global__ __launch_bounds__(256, 2) void wgrad_stem_kernel(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xs = smem;                          // raw input rows
    char* Ds = smem + (a.xbytes + 1023) / 1024 * 1024;      // [dpix][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform: the LDS-DMA destinations below go through M0)
    const int wm = wave & 1, kg = wave >> 1;  // output-channel half; kernel rows 0-3 / 4-6
    constexpr unsigned kOOB = 0x80000000u;
    const i32x4 gs_x = rsrc_words(a.x, (unsigned)((long)a.N * a.H * a.rowb));               // (< 2 GB: checked on the host)
    const i32x4 gs_d = rsrc_words(a.dz, (unsigned)((long)a.N * a.Ho * a.Wo * 128));
    const unsigned lds_x = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Xs);
    const unsigned lds_d = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Ds);
    const int nx = (a.xbytes + 1023) >> 10, nd = a.dpix >> 3;      // wave instructions (1 KB each) per image
    const int lp = lane >> 3, lu = lane & 7;

    // transpose-read lane geometry (wgrad_bf16_kernel): 16-lane group g supplies pixel rows (i >> 2) + 8 (g >> 1), channel quad 16 (g & 1) + 4 (i & 3)
    const int tg = lane >> 4, ti = lane & 15;
    const int trow = (ti >> 2) + 8 * (tg >> 1);
    const int tunit = 2 * (tg & 1) + ((ti & 3) >> 1), thalf = ti & 1;
    const int offA = trow * 128 + (((wm * 4 + tunit) ^ swzf(trow)) << 4) + thalf * 8;
    const int offB = 16 * trow + 32 * (tg & 1) + 8 * (ti & 3);      // pixel pitch 16 B (stride 2 x 8-byte pixels), channel quad = one input pixel
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s* lds_v4s;
    auto gather = [&](const char* lo_p, const char* hi_p) {          // pixels +0..3 and +4..7 of the lane's channel
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(lo_p));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(hi_p));
        bf16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nks = a.dpix >> 4;
    for (int b = blockIdx.x; b < a.bands; b += gridDim.x) {
        const int img = fdiv(b, a.fd_bpf_mul, a.fd_bpf_shr);
        const int oh0 = (b - img * a.bands_per_frame) * kR;
        __syncthreads();                      // the previous band's reads are done
        // ---- stage: both sources are contiguous; wave w issues instructions w, w + 4, ... ----
        {
            const long xbase = ((long)img * a.H + 2 * oh0) * a.rowb;
            for (int i = wave; i < nx; i += 4) {
                const int o = i * 1024 + lane * 16;
                glds16(gs_x, lds_x + (unsigned)(i * 1024), o < a.xbytes ? (unsigned)(xbase + o) : kOOB);
            }
            const long dbase = ((long)img * a.Ho + oh0) * a.Wo * 128;
            for (int i = wave; i < nd; i += 4) {
                const int p = i * 8 + lp;
                glds16(gs_d, lds_d + (unsigned)(i * 1024), (unsigned)(dbase + p * 128 + ((lu ^ swzf(p)) << 4)));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- contract over the band's pixels ----
        for (int ks = 0; ks < nks; ++ks) {
            const int r = fdiv(ks, a.fd_ksr_mul, a.fd_ksr_shr), ow0 = (ks - r * a.ksr) * 16;
            const bf16x8_t fa = gather(Ds + ks * 2048 + offA, Ds + ks * 2048 + offA + 4 * 128);
            const char* xb = Xs + (2 * r + 4 * kg) * a.rowb + ow0 * 16 + offB;
            bf16x8_t fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < 3 || kg == 0) fb[t] = gather(xb + t * a.rowb, xb + t * a.rowb + 64);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < 3 || kg == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[t], acc[t], 0, 0, 0);
            }
        }
    }
    // ---- one partial slab per workgroup: part[wg][co][kh * 32 + (kw', c)] ----
    const int lr = lane >> 5, lc = lane & 31;
    float* out = a.part + (long)blockIdx.x * 64 * 224;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < 3 || kg == 0) {
            const int col = (4 * kg + t) * 32 + lc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lr;
                out[row * 224 + col] = acc[t][r];
            }
        }
    }
}

inline void fd_make(unsigned d, unsigned& mul, unsigned& shr) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    shr = l;
}

// output rows per band: 2 -> 46 KB of LDS per workgroup at the 224^2 shape (three workgroups per CU: two stage while one contracts); 1 -> 27 KB
static int band_rows() {
    static const int r = mvf_policy_int("wgrad_stem_r", 2);
    return r == 1 ? 1 : 2;
}

}  // namespace

namespace mvf_internal {

// the stem view (7 x 1 taps over 32 "channels" of the padded NHWC4 operand, stride 2) at sizes the band walk covers
bool wgrad_stem_ok(int n, int h, int w, int ho, int wo) {
    static const bool on = (mvf_policy_int("wgrad_stem_direct", 1) != 0);
    return on && n > 0 && wo >= 16 && wo % 16 == 0 && ho % 2 == 0 && (w * 8) % 16 == 0 && 2 * (ho - 1) + 6 < h && 2 * (wo - 1) + 7 < w &&
           (long)n * h * w * 8 < 0x7ffffff0L && (long)n * ho * wo * 128 < 0x7ffffff0L &&
           (2 * 2 + 5) * w * 8 + 1024 + 2 * wo * 128 <= 72 * 1024;
}

// workgroups (= partial slabs) of a launch: three per CU (policy wgrad_stem_wgs overrides), never more than bands
int wgrad_stem_wgs(int n, int ho) {
    static const int env = std::max(1, mvf_policy_int("wgrad_stem_wgs", 768));      // measured at the C3 shape: 256 / 512 / 768 / 1024 workgroups 235 / 163 / 143 / 175 us
    return (int)std::min<long>(env, (long)n * (ho / band_rows()));
}

int wgrad_stem_launch(const WgradStemArgs& w, hipStream_t st) {
    KArgs a = {};
    a.dz = (const char*)w.dz; a.x = (const char*)w.x; a.part = w.part;
    a.N = w.N; a.H = w.H; a.W = w.W; a.Ho = w.Ho; a.Wo = w.Wo;
    const int kR = band_rows();
    a.bands_per_frame = w.Ho / kR;
    a.bands = w.N * a.bands_per_frame;
    a.rowb = w.W * 8;
    a.xbytes = (2 * kR + 5) * a.rowb;
    a.dpix = kR * w.Wo;
    a.ksr = w.Wo / 16;
    fd_make((unsigned)a.bands_per_frame, a.fd_bpf_mul, a.fd_bpf_shr);
    fd_make((unsigned)a.ksr, a.fd_ksr_mul, a.fd_ksr_shr);
    const int lds = (a.xbytes + 1023) / 1024 * 1024 + a.dpix * 128;
    static bool attr = false;
    if (!attr) {
        MVF_HIP_OK(hipFuncSetAttribute((const void*)wgrad_stem_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        MVF_HIP_OK(hipFuncSetAttribute((const void*)wgrad_stem_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        attr = true;
    }
    if (kR == 1) hipLaunchKernelGGL(wgrad_stem_kernel<1>, dim3(w.nwg), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(wgrad_stem_kernel<2>, dim3(w.nwg), dim3(256), lds, st, a);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // namespace mvf_internal
