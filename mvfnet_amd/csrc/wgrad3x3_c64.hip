// Weight gradient of the 3x3 stride-1 pad-1 conv with 64 input and 64 output channels (layer1's conv2: reference
// codes/models/backbones/resnet.py:213-224 conv2 of a Bottleneck with planes = 64, autograd's weight gradient) as a DIRECT kernel for gfx950,
// bf16 storage, fp32 accumulation.
//
//   dW[co][kh][kw][ci] = sum over (n, oh, ow) of dZ[n][oh][ow][co] * X[n][oh + kh - 1][ow + kw - 1][ci]      (zero outside the image)
//
// Why: the implicit-GEMM weight gradient (wgrad_nhwc.hip) sees this as M = N*H*W, N = 64, K = 576 and re-stages the input once per tap -- nine
// shifted copies of the same 64-channel rows through L2 -> LDS: 137 us per launch at the C3 shape against 33 us for its bytes
// (profiles/r04_per_layer_bf16_train.txt: the launch furthest above its bound in the weight-gradient group, 432 TF/s).  Here
//   * a workgroup (4 waves, two workgroups per CU) walks row bands of R = 4 output rows.  The band's dZ rows and the ZERO-PADDED (R + 2) x (W + 2)
//     input window are staged ONCE by LDS-DMA, both as [pixel][64 channels] images of 128-byte rows with the SAME row pitch W + 2 (dZ's two pad
//     columns are zeros), so that for tap (kh, kw) the input pixel that meets dZ pixel p is simply window pixel p + kh (W + 2) + kw: all nine
//     taps read ONE staged window at nine constant pixel shifts (out-of-range offsets DMA zeros: no tap masks, no per-tap staging);
//   * the contraction runs over the band's pixels in steps of 16; both operands come out of their pixel-major images through the gfx950
//     transpose read ds_read_b64_tr_b16 exactly as in wgrad_bf16_kernel (16-byte units XOR-swizzled by bit 1 of the pixel index on the DMA's
//     source side; the nine shifts only move the per-lane base address);
//   * a wave owns a 32 (co) x 32 (ci) block of ALL NINE taps: 9 accumulator tiles = 144 VGPRs, one dZ fragment feeds nine matrix instructions;
//   * workgroups are persistent (each sums its bands in registers) and write ONE fp32 partial slab [64][576] each, in wgrad_reduce_kernel's
//     layout, summed in fixed order (deterministic, no atomics).
// Arithmetic: the same bf16 products as the implicit GEMM, accumulated in fp32 in a different (band-major) order.
// Measured (C3 shape, 256 frames of 56 x 56; kernel + slab reduce, tools/kbench.py wgrad16 l1.c2): 146 -> 90-94 us; in the training step the
// weight-gradient group alone 3.42 -> 3.25 ms and the step 19.84 -> 19.81 ms (four alternations, every pair the same sign: it runs on the side
// stream).  Ablations (-DMVF_WGRAD_ABLATE, policy wgrad_abl bits): matrix loop 27 us (= the matrix rate: 2.2 PF/s while it runs), staging 21,
// slab stores 8, the reduce of 512 slabs + two launches 21; they overlap little.  Measured without effect: a second fragment set (spills at 256
// registers), the head of the next k-step fetched ahead, starting the second half of the grid late, 16-byte loads in the reduce; 256 workgroups
// (one per CU, half the slabs) lose the second wave per SIMD: 110 us.
#include <algorithm>
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));

struct KArgs {
    const char* dz;
    const char* x;
    float* part;
    int N, H, W, xps;                         // xps = pixel pitch of x in elements (>= 64)
    int WP;                                   // W + 2: row pitch of both staged images, in pixels
    int bands_per_frame, bands;
    int xpix, dpix;                           // staged pixels (x window incl. the zero tail the last k-step's shifted reads run into; dZ band): multiples of 16
    unsigned fd_bpf_mul, fd_bpf_shr, fd_wp_mul, fd_wp_shr;
    int abl;                                  // -DMVF_WGRAD_ABLATE builds (timing ablation, wrong results): bit 0 no staging after the first band, bit 1 no matrix loop, bit 2 no slab store
};

__device__ __forceinline__ int fdiv(int n, unsigned mul, unsigned shr) { return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> shr); }
__device__ __forceinline__ int swzf(int pixel) { return ((pixel >> 1) & 1) << 2; }      // (wgrad_bf16_kernel's swz16<8>)

template <int kR>                             // output rows per band
__global__ __launch_bounds__(256, 2) void wgrad3x3_c64_kernel(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xs = smem;                          // [xpix][128 B]
    char* Ds = smem + a.xpix * 128;           // [dpix][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform: the LDS-DMA destinations below go through M0)
    const int wm = wave >> 1, wn = wave & 1;  // co half, ci half
    constexpr unsigned kOOB = 0x80000000u;
    const i32x4 gs_x = rsrc_words(a.x, (unsigned)((long)a.N * a.H * a.W * a.xps * 2));      // (< 2 GB: checked on the host)
    const i32x4 gs_d = rsrc_words(a.dz, (unsigned)((long)a.N * a.H * a.W * 128));
    const unsigned lds_x = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Xs);
    const unsigned lds_d = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Ds);
    const int nx = a.xpix >> 3, nd = a.dpix >> 3;            // wave instructions (8 pixels x 128 B each) per image
    const int lp = lane >> 3, lu = lane & 7;

    // transpose-read lane geometry (wgrad_bf16_kernel): 16-lane group g supplies pixel rows (i >> 2) + 8 (g >> 1), channel quad 16 (g & 1) + 4 (i & 3)
    const int tg = lane >> 4, ti = lane & 15;
    const int trow = (ti >> 2) + 8 * (tg >> 1);
    const int tunit = 2 * (tg & 1) + ((ti & 3) >> 1), thalf = ti & 1;
    const int offA = trow * 128 + (((wm * 4 + tunit) ^ swzf(trow)) << 4) + thalf * 8;
    int offB[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int sh = (t / 3) * a.WP + (t % 3);
        offB[t] = (sh + trow) * 128 + (((wn * 4 + tunit) ^ swzf(sh + trow)) << 4) + thalf * 8;
    }
    auto gather = [&](const char* p) {        // pixels +0..3 and +4..7 of the lane's channel
        typedef short v4s __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4s* lds_v4s;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p + 4 * 128));
        bf16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nks = a.dpix >> 4;
    for (int b = blockIdx.x; b < a.bands; b += gridDim.x) {
        const int img = fdiv(b, a.fd_bpf_mul, a.fd_bpf_shr);
        const int row0 = (b - img * a.bands_per_frame) * kR;
        __syncthreads();                      // the previous band's reads are done
#ifdef MVF_WGRAD_ABLATE
        if (!((a.abl & 1) && b != (int)blockIdx.x))
#endif
        {
        // ---- stage: wave w issues instructions w, w + 4, ...; lane -> (pixel 8 i + (lane >> 3), source unit (lane & 7) ^ swizzle) ----
        for (int i = wave; i < nx; i += 4) {
            const int p = i * 8 + lp;
            const int wr = fdiv(p, a.fd_wp_mul, a.fd_wp_shr), wc = p - wr * a.WP;
            const int ih = row0 - 1 + wr, iw = wc - 1;
            const bool ok = wr < kR + 2 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)(((img * a.H + ih) * a.W + iw) * a.xps * 2 + ((lu ^ swzf(p)) << 4)) : kOOB;
            glds16(gs_x, lds_x + (unsigned)(i * 1024), off);
        }
        for (int i = wave; i < nd; i += 4) {
            const int p = i * 8 + lp;
            const int r = fdiv(p, a.fd_wp_mul, a.fd_wp_shr), ow = p - r * a.WP;
            const int oh = row0 + r;
            const bool ok = r < kR && oh < a.H && ow < a.W;
            const unsigned off = ok ? (unsigned)(((img * a.H + oh) * a.W + ow) * 128 + ((lu ^ swzf(p)) << 4)) : kOOB;
            glds16(gs_d, lds_d + (unsigned)(i * 1024), off);
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- contract over the band's pixels (a second fragment set, or the head of the next k-step fetched ahead, measured the same 93 us: the
        // other workgroup of the CU already covers the LDS round trips) ----
#ifdef MVF_WGRAD_ABLATE
        if (!(a.abl & 2))
#endif
        for (int ks = 0; ks < nks; ++ks) {
            const bf16x8_t fa = gather(Ds + ks * 2048 + offA);
            bf16x8_t fb[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) fb[t] = gather(Xs + ks * 2048 + offB[t]);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[t], acc[t], 0, 0, 0);
        }
    }
    // ---- one partial slab per workgroup: part[wg][co][tap * 64 + ci] ----
#ifdef MVF_WGRAD_ABLATE
    if ((a.abl & 4) && acc[0][0] != 12345.678f) return;
#endif
    const int lr = lane >> 5, lc = lane & 31;
    float* out = a.part + (long)blockIdx.x * 64 * 576;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int col = t * 64 + wn * 32 + lc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lr;
            out[row * 576 + col] = acc[t][r];
        }
    }
}

inline void fd_make(unsigned d, unsigned& mul, unsigned& shr) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    shr = l;
}

}  // namespace

namespace mvf_internal {

// output rows per band: 3 -> 61 KB of LDS per workgroup at W = 56 (two workgroups per CU: one stages while the other contracts); 4 -> 78 KB
static int band_rows() {
    static const int r = mvf_policy_int("wgrad3x3_r", 3);
    return r == 4 ? 4 : 3;
}

// workgroups (= partial slabs) of a launch: two per CU (policy wgrad3x3_wgs overrides), never more than bands
int wgrad3x3_c64_wgs(int n, int h) {
    static const int env = std::max(1, mvf_policy_int("wgrad3x3_wgs", 512));
    const int kR = band_rows();
    const long bands = (long)n * ((h + kR - 1) / kR);
    return (int)std::min<long>(env, bands);
}

bool wgrad3x3_c64_ok(int n, int h, int w, int xps) {
    static const bool on = (mvf_policy_int("wgrad3x3_direct", 1) != 0);
    return on && w >= 4 && w <= 56 && h >= 1 && xps >= 64 && xps % 8 == 0 && (long)n * h * w * xps * 2 < 0x7ffffff0L;
}

int wgrad3x3_c64_launch(const Wgrad3x3C64Args& w, hipStream_t st) {
    const int kR = band_rows();
    KArgs a = {};
    a.dz = (const char*)w.dz; a.x = (const char*)w.x; a.part = w.part;
    a.N = w.N; a.H = w.H; a.W = w.W; a.xps = w.xps;
    a.WP = w.W + 2;
    a.bands_per_frame = (w.H + kR - 1) / kR;
    a.bands = w.N * a.bands_per_frame;
    a.dpix = (kR * a.WP + 15) / 16 * 16;
    a.xpix = (std::max((kR + 2) * a.WP, a.dpix + 2 * a.WP + 2) + 15) / 16 * 16;
    fd_make((unsigned)a.bands_per_frame, a.fd_bpf_mul, a.fd_bpf_shr);
    fd_make((unsigned)a.WP, a.fd_wp_mul, a.fd_wp_shr);
#ifdef MVF_WGRAD_ABLATE
    a.abl = mvf_policy_int("wgrad_abl", 0);
#endif
    const int lds = (a.xpix + a.dpix) * 128;
    static bool attr = false;
    if (!attr) {
        MVF_HIP_OK(hipFuncSetAttribute((const void*)wgrad3x3_c64_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        MVF_HIP_OK(hipFuncSetAttribute((const void*)wgrad3x3_c64_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr = true;
    }
    if (kR == 4) hipLaunchKernelGGL(wgrad3x3_c64_kernel<4>, dim3(w.nwg), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(wgrad3x3_c64_kernel<3>, dim3(w.nwg), dim3(256), lds, st, a);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // namespace mvf_internal
