// Column sums of a pointwise conv whose output is never stored (gfx950, bf16 storage, K = 64 input channels, N <= 256 output channels):
//   MODE 0  the statistics-only first pass of a z3-free bottleneck (reference resnet.py:229-231: conv3 -> bn3 in train mode): per output
//           channel sum (z - K), sum (z - K)^2 of z = round_bf16(conv1x1(a)), K = the BatchNorm's running mean;
//   MODE 1  bn3's backward sums on the recomputed conv (autograd of resnet.py:229-244 out = relu(bn3(conv3(a)) + identity)):
//           sum gm and sum gm * (z - mean) * invstd with gm = g * [sign bit of out].
// The implicit-GEMM kernel's epilogues for these (conv_nhwc.hip EPI 1 with y = NULL, EPI 10) spend 83 / 210 us on 19 / 98 us of bytes
// (here: 70 / 150-160 us; what is left is the second read of the conv input by the other channel half and ~7 VALU instructions per element):
// the accumulators sit pixel = lane, channels in registers (D^T, the layout the STORE wants), so every column sum is a cross-lane
// reduction through LDS.  Nothing is stored here, so the product is taken the other way round -- channel = lane, 16 pixels of a 32-pixel
// block in the registers -- and a channel's sums are plain per-lane accumulations: 2 VGPRs per 32 channels for the whole launch, the
// BatchNorm constants one register each, one cross-lane step (the two pixel halves) and two partial rows (value + remainder of the fp64 sum) per WORKGROUP at the very end.
//   * a wave owns 32-pixel blocks: the operand fragments of its pixels are four 16-byte global loads per lane (no LDS), the whole weight
//     matrix (N x 64) is 16 VGPRs per 32 output channels;
//   * MODE 1: g arrives pixel-major; the wave gates its [32 pixel][N] tile with the sign bits in registers, writes it to a private LDS
//     slab (16-byte units XOR-swizzled by the row) and reads it back TRANSPOSED with ds_read_b64_tr_b16 (4 consecutive pixels of one
//     channel per read = one register quad of the accumulator layout; lane mapping as in wgrad_nhwc.hip, probed in tools/probes).
// Arithmetic: the same bf16 products, fp32 accumulation in the same k order, rounded once; the sums in another order than EPI 1 / 10.
#include <algorithm>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct KArgs {
    const char* a;             // (M, xps) bf16, channels [0, 64)
    const char* w;             // packed [N][64] bf16
    const char* g;             // MODE 1: (M, N) bf16
    const unsigned char* bits; // MODE 1: (M, N / 4) bytes, bit j of byte k = channel 4k + j
    const float* c0;           // MODE 0: shift K (or NULL); MODE 1: mean
    const float* c1;           // MODE 1: invstd
    float* part;               // [N][rows][2]
    int rows, M, N, xps;
    int nblocks;               // 32-pixel blocks
};

// A workgroup covers NB x 32 = 128 output channels (the 64 weight VGPRs of 4 channel blocks; all 256 at once spill): with N = 256 the
// workgroups alternate between the two channel halves and each half reads the conv input again (103 of 643 MB in MODE 1).
template <int MODE, int NB>
__global__ __launch_bounds__(256, 3) void pw_sums_kernel(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int N = NB * 32, PITCH = N * 2;                  // slab row = one pixel's gradients of this workgroup's N channels
    // workgroup -> (pixel-block walker wg, channel half): the halves of one walker sit 8 workgroup ids apart = on the SAME XCD (ids round-robin
    // over the 8 XCDs), so the second half's read of the conv input is served by that XCD's L2 instead of HBM
    const int nhalf = a.N / N, nwg = gridDim.x / nhalf;
    const int chalf = (blockIdx.x >> 3) % nhalf, wg = (blockIdx.x / (8 * nhalf)) * 8 + (blockIdx.x & 7);
    const int cb = chalf * N;                                   // first output channel of this workgroup
    // ---- weights: fragment (channel block nb, k-step ks) = 8 input channels of output channel 32 nb + (lane & 31) ----
    bf16x8 wf[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 v = *reinterpret_cast<const uint4*>(a.w + ((long)(cb + nb * 32 + l31) * 64 + ks * 16 + half * 8) * 2);
            __builtin_memcpy(&wf[nb][ks], &v, 16);
        }
    float k0[NB], k1[NB];
    double s1[NB], s2[NB];                                      // a lane adds ~25 block sums per channel: kept in fp64 (the implicit-GEMM path sums 128-row partials in fp64 too)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        k0[nb] = a.c0 ? a.c0[cb + nb * 32 + l31] : 0.f;
        k1[nb] = MODE == 1 ? a.c1[cb + nb * 32 + l31] : 0.f;
        s1[nb] = 0.0;
        s2[nb] = 0.0;
    }
    char* slab = smem + wave * (32 * PITCH);                    // MODE 1: this wave's gated gradient tile
    // transposed read of block nb, pixel quad q: lane i of a 16-lane group hands in pixel (i >> 2), channels 4 (i & 3) .. + 3 and
    // receives channel i of the four pixels; group G = lane >> 4 covers channels 16 (G & 1) .. + 15 of the block, pixel half G >> 1
    const int ti = lane & 15, tgc = (lane >> 4) & 1;
    const int trow = ti >> 2;                                   // + 8 q + 4 half
    const int tch = 16 * tgc + 4 * (ti & 3);                    // + 32 nb
    const int nwaves = nwg * 4;
    for (int blk = wg * 4 + wave; blk < a.nblocks; blk += nwaves) {
        const long m0 = (long)blk * 32;
        const bool tail = m0 + 32 > a.M;
        // ---- this block's operand fragments: pixel = lane & 31, 8 input channels per k-step half (fetching block b + 1 under block b was
        // measured: no gain for MODE 0, spills and -15 % for MODE 1 at 3 waves per SIMD) ----
        uint4 fa[4];
        {
            const long m = min(m0 + l31, (long)a.M - 1);
            const char* ap = a.a + (m * a.xps + half * 8) * 2;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[ks] = *reinterpret_cast<const uint4*>(ap + ks * 32);
            if (tail && m0 + l31 >= a.M) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[ks] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        if constexpr (MODE == 1) {
            // gate g with the sign bits and lay the tile down pixel-major: 32 rows x (N / 8) 16-byte units, lane = (row parity, unit)
            constexpr int UPR = N / 8;                          // units per row
            constexpr int RPI = 64 / UPR;                       // rows per wave instruction
            static_assert(64 % UPR == 0, "a wave instruction covers whole rows");
            const int u = lane % UPR, r0 = lane / UPR;
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int row = it * RPI + r0;
                const long m = min(m0 + row, (long)a.M - 1);
                uint4 v = *reinterpret_cast<const uint4*>(a.g + (m * a.N + cb + u * 8) * 2);
                const unsigned b = *reinterpret_cast<const unsigned short*>(a.bits + m * (a.N / 4) + cb / 4 + u * 2);      // channels 8u .. 8u + 7: two nibbles
                const bool in = m0 + row < a.M;
                auto gate = [&](unsigned x, unsigned lo, unsigned hi) {
                    return (lo ? (x & 0x0000ffffu) : 0u) | (hi ? (x & 0xffff0000u) : 0u);
                };
                v.x = gate(v.x, in && (b & 1u), in && (b & 2u));
                v.y = gate(v.y, in && (b & 4u), in && (b & 8u));
                v.z = gate(v.z, in && (b & 0x100u), in && (b & 0x200u));
                v.w = gate(v.w, in && (b & 0x400u), in && (b & 0x800u));
                *reinterpret_cast<uint4*>(slab + row * PITCH + ((u ^ ((row & 3) << 1)) * 16)) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            __builtin_amdgcn_sched_barrier(0);                  // one channel block at a time (interleaved, the 8 accumulator sets spill)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 av;
                __builtin_memcpy(&av, &fa[ks], 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wf[nb][ks], acc, 0, 0, 0);   // D: channel = lane & 31, pixel = 8 (r >> 2) + 4 half + (r & 3)
            }
            float b1 = 0.f, b2 = 0.f;                           // this block's 16 pixels of the lane's channel
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // z as the forward pass rounded it
                const unsigned p0 = pack_bf16x2(acc[4 * q], acc[4 * q + 1]), p1 = pack_bf16x2(acc[4 * q + 2], acc[4 * q + 3]);
                float z[4] = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float d = z[e] - k0[nb];
                        if (tail && m0 + 8 * q + 4 * half + e >= a.M) d = 0.f;
                        b1 += d;
                        b2 += d * d;
                    }
                } else {
                    typedef short v4s __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(3))) v4s* lds_v4s;
                    const int row = 8 * q + 4 * half + trow, ch = nb * 32 + tch;
                    const v4s gq = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_v4s)(slab + row * PITCH + (((ch >> 3) ^ ((row & 3) << 1)) * 16) + (ch & 7) * 2));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gm = __uint_as_float(((unsigned)(unsigned short)gq[e]) << 16);      // (rows past M were gated to zero)
                        b1 += gm;
                        b2 += gm * ((z[e] - k0[nb]) * k1[nb]);
                    }
                }
            }
            s1[nb] += (double)b1;
            s2[nb] += (double)b2;
        }
        if constexpr (MODE == 1) __builtin_amdgcn_wave_barrier();
    }
    // ---- one partial row per workgroup: the two pixel halves of a channel (lane, lane + 32), then the 4 waves in order ----
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        s1[nb] += __shfl_xor(s1[nb], 32, 64);
        s2[nb] += __shfl_xor(s2[nb], 32, 64);
    }
    __syncthreads();
    double2* red = reinterpret_cast<double2*>(smem);            // [wave][N]
    if (lane < 32) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) red[wave * N + nb * 32 + lane] = make_double2(s1[nb], s2[nb]);
    }
    __syncthreads();
    for (int c = tid; c < N; c += 256) {
        const double2 p0 = red[c], p1 = red[N + c], p2 = red[2 * N + c], p3 = red[3 * N + c];
        float2* dst = reinterpret_cast<float2*>(a.part) + (long)(cb + c) * a.rows;
        // the workgroup's fp64 sums leave as TWO fp32 partial rows (value and remainder): the finalize adds all rows in fp64, so the channel's
        // statistic keeps the precision of the per-tile partials of the implicit-GEMM path (and its invariance under a permutation of the clips)
        const double t1 = (p0.x + p1.x) + (p2.x + p3.x), t2 = (p0.y + p1.y) + (p2.y + p3.y);
        const float h1 = (float)t1, h2 = (float)t2;
        dst[wg] = make_float2(h1, h2);
        dst[nwg + wg] = make_float2((float)(t1 - (double)h1), (float)(t2 - (double)h2));
        for (int r = 2 * nwg + wg; r < a.rows; r += nwg) dst[r] = make_float2(0.f, 0.f);
    }
}

template <int MODE, int NB>
int launch(const KArgs& a, int grid, hipStream_t st) {
    auto k = pw_sums_kernel<MODE, NB>;
    constexpr int lds = MODE == 1 ? 4 * 32 * NB * 64 : 4 * NB * 32 * 16;
    static bool attr = false;
    if (!attr) {
        MVF_HIP_OK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // namespace

namespace mvf_internal {

// MVF_OK when launched, -1 when the shape is not this kernel's (the caller falls back to the implicit-GEMM epilogues)
int pw_sums_launch(const PwSumsArgs& s, hipStream_t st) {
    if (mvf_policy_int("pw_sums", 1) == 0) return -1;      // A/B switch, read per call: 0 = the implicit-GEMM epilogues
    if (s.K != 64 || (s.N != 256 && s.N != 128) || s.xps < 64 || s.xps % 8 || s.M <= 0) return -1;
    const long nblocks = ((long)s.M + 31) / 32;
    if (nblocks >= (1L << 31)) return -1;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    const int nhalf = s.N / 128;
    const long per_cu = mvf_policy_int("pw_sums_wgs", 3);
    int nwg = (int)std::min<long>((nblocks + 3) / 4, per_cu * cus / nhalf);      // resident workgroups per CU, every wave walks its blocks
    if (nwg > s.rows / 2) nwg = s.rows / 2;                         // two partial rows per workgroup (and channel half)
    if (nwg < 1) return -1;
    nwg = nwg / 8 * 8;                                              // whole groups of 8 (see the XCD pairing in the kernel)
    if (nwg < 8) return -1;
    const int grid = nwg * nhalf;
    KArgs a = {};
    a.a = (const char*)s.a; a.w = (const char*)s.w; a.g = (const char*)s.g; a.bits = s.bits; a.c0 = s.c0; a.c1 = s.c1;
    a.part = s.part; a.rows = s.rows; a.M = s.M; a.N = s.N; a.xps = s.xps; a.nblocks = (int)nblocks;
    return s.mode == 0 ? launch<0, 4>(a, grid, st) : launch<1, 4>(a, grid, st);
}

}  // namespace mvf_internal
