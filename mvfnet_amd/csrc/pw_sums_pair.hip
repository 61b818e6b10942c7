// BatchNorm-backward sums of BOTH branches of a z3-free downsample bottleneck in one pass over the block-output gradient (gfx950, bf16 storage,
// both convs 64 -> 256 channels pointwise: layer1.0 of the ResNet) -- and, with x = NULL, of the one branch of a plain z3-free block (NBR = 1).
//
// Reference arithmetic: autograd of Bottleneck.forward (codes/models/backbones/resnet.py:227-244): out = relu(bn3(conv3(a2)) + bn_d(conv_d(x))).  With
// g = dL/dout and gm = g * [out > 0] both BatchNorms receive the same gated gradient:
//     dbeta_3 = dbeta_d = sum_m gm,   dgamma_3 = sum_m gm (z3 - mean_3) invstd_3,   dgamma_d = sum_m gm (z_d - mean_d) invstd_d
// with z3 = round_bf16(conv3(a2)), z_d = round_bf16(conv_d(x)) RECOMPUTED (neither is read; z3 was never stored).  Two pw_sums.hip MODE 1 launches
// would read g and the sign bits twice (2 x 462 MB of their 2 x 565 MB at the R50 8x8 shape); here g is read and gated once.
// Construction (csrc/pw_bwd_fused.hip's): a persistent workgroup of 8 waves walks 64-pixel chunks; a2, x, g and the sign bits arrive by LDS-DMA one
// chunk ahead (double buffers; ONE barrier per chunk: nothing is written to LDS by the waves); wave w owns output channels [32 w, 32 w + 32) of both
// convs -- its W fragments of both live in registers --, forms z3^T and z_d^T of the chunk (lane = pixel, four consecutive channels per register quad)
// and accumulates the three sums per (lane, channel) in registers from 8-byte LDS reads of the swizzled g tile and one byte of sign bits per quad.
// At the end the 32 pixel-lanes of a channel are summed in fp64 (shuffles); two fp32 partial rows per workgroup and BatchNorm (value + remainder),
// channel-major [256][2 nsplit][2] for mvf_bn_bwd_finalize.
#include <algorithm>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kT = 512, CH = 64, NC = 256, NK = 64;
constexpr int PD = NC * 2, PX = NK * 2, PB = NC / 4;
constexpr unsigned kOOB = 0x80000000u;
constexpr int kOffA = 0, kOffX = kOffA + 2 * CH * PX, kOffG = kOffX + 2 * CH * PX, kOffM = kOffG + 2 * CH * PD, kLds = kOffM + 2 * CH * PB;

__device__ __forceinline__ int swz_d(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int swz_x(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ u32x4 lds16(const char* p) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    u32x4 r;
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
}

template <int NBR>                                                    // 2 = both branches of a downsample block; 1 = conv a only (a plain z3-free block)
__global__ __launch_bounds__(kT) void pw_sums_pair_kernel(const mvf_internal::PwSumsPairArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem + kOffA;                                         // [2][CH][PX]   a2 (conv a's input)
    char* Xs = smem + kOffX;                                         // [2][CH][PX]   x  (conv b's input)
    char* Gs = smem + kOffG;                                         // [2][CH][PD]   g, 16-byte units XOR-swizzled by the row
    char* Ms = smem + kOffM;                                         // [2][CH][PB]   sign bits

    const int split = blockIdx.x;
    const int m_begin = split * a.rows_per_split, m_end = min(a.M, m_begin + a.rows_per_split);
    const int nchunks = (m_end - m_begin + CH - 1) / CH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    // ---- weights (A operands, row = channel 32 wave + l31) and the BatchNorm constants of this lane's 16 channels 32 wave + 8 g4 + 4 half + e ----
    bf16x8_t wa[4], wb[4];
    {
        const char* pa = reinterpret_cast<const char*>(a.w_a);
        const char* pb = reinterpret_cast<const char*>(a.w_b);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const long o = ((long)(32 * wave + l31) * NK + ks * 16 + half * 8) * 2;
            const uint4 va = *reinterpret_cast<const uint4*>(pa + o);
            __builtin_memcpy(&wa[ks], &va, 16);
            if constexpr (NBR == 2) {
                const uint4 vb = *reinterpret_cast<const uint4*>(pb + o);
                __builtin_memcpy(&wb[ks], &vb, 16);
            }
        }
    }
    float mua[16], rsa[16], mub[16], rsb[16];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int c = 32 * wave + 8 * g4 + 4 * half;
        const float4 m0 = *reinterpret_cast<const float4*>(a.mean_a + c), r0 = *reinterpret_cast<const float4*>(a.invstd_a + c);
        float4 m1 = m0, r1 = r0;
        if constexpr (NBR == 2) { m1 = *reinterpret_cast<const float4*>(a.mean_b + c); r1 = *reinterpret_cast<const float4*>(a.invstd_b + c); }
        mua[4 * g4] = m0.x; mua[4 * g4 + 1] = m0.y; mua[4 * g4 + 2] = m0.z; mua[4 * g4 + 3] = m0.w;
        rsa[4 * g4] = r0.x; rsa[4 * g4 + 1] = r0.y; rsa[4 * g4 + 2] = r0.z; rsa[4 * g4 + 3] = r0.w;
        mub[4 * g4] = m1.x; mub[4 * g4 + 1] = m1.y; mub[4 * g4 + 2] = m1.z; mub[4 * g4 + 3] = m1.w;
        rsb[4 * g4] = r1.x; rsb[4 * g4 + 1] = r1.y; rsb[4 * g4 + 2] = r1.z; rsb[4 * g4 + 3] = r1.w;
    }
    float s1[16], s2a[16], s2b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2a[i] = 0.f; s2b[i] = 0.f; }

    // ---- LDS-DMA of a chunk: a2 / x one transfer per thread (source unit p ^ swz_x(row)), g four (source unit p ^ swz_d(row)), bits waves 0-3 ----
    const int rbx = tid >> 3;
    const int qbx = (tid & 7) ^ swz_x(rbx);
    const i32x4 gs_a = rsrc_words(a.a, (unsigned)min((long)a.M * a.aps * 2, 0x7ffffff0L));
    const i32x4 gs_x = rsrc_words(a.x, (unsigned)min((long)a.M * a.xps * 2, 0x7ffffff0L));
    const i32x4 gs_g = rsrc_words(a.g, (unsigned)min((long)a.M * a.g_pitch * 2, 0x7ffffff0L));
    const i32x4 gs_m = rsrc_words(a.bits, (unsigned)min((long)a.M * PB, 0x7ffffff0L));
    auto lds_base = [&](char* p) { return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)p); };
    const unsigned lds_a = lds_base(As) + wave * 8 * PX, lds_x = lds_base(Xs) + wave * 8 * PX;
    const unsigned lds_g = lds_base(Gs) + wave * 2 * PD, lds_m = lds_base(Ms) + wave * 16 * PB;
    auto dma = [&](int cc) {
        const int mc = m_begin + cc * CH, buf = cc & 1;
        {
            const int m = mc + rbx;
            const bool ok = m < m_end;
            glds16(gs_a, lds_a + (unsigned)(buf * CH * PX), ok ? (unsigned)(m * a.aps + qbx * 8) * 2u : kOOB);
            if constexpr (NBR == 2) glds16(gs_x, lds_x + (unsigned)(buf * CH * PX), ok ? (unsigned)(m * a.xps + qbx * 8) * 2u : kOOB);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                // a wave lays down rows 16 i + 2 wave, + 1 (32 units each)
            const int row = 16 * i + 2 * wave + half, m = mc + row;
            glds16(gs_g, lds_g + (unsigned)(buf * CH * PD + 16 * i * PD), m < m_end ? (unsigned)(m * a.g_pitch + (l31 ^ swz_d(row)) * 8) * 2u : kOOB);
        }
        if (wave < 4) {                                              // bits: rows 16 wave + (lane >> 2), unit lane & 3 (rows past the end: zeros = gated off)
            const int m = mc + 16 * wave + (lane >> 2);
            glds16(gs_m, lds_m + (unsigned)(buf * CH * PB), m < m_end ? (unsigned)(m * PB + (lane & 3) * 16) : kOOB);
        }
    };

    auto compute = [&](int buf) {
        const char* as = As + buf * CH * PX;
        const char* xs = Xs + buf * CH * PX;
        const char* gs = Gs + buf * CH * PD;
        const char* ms = Ms + buf * CH * PB;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const int row = pb * 32 + l31;
            f32x16 za, zb;
#pragma unroll
            for (int r = 0; r < 16; ++r) { za[r] = 0.f; zb[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int u = ((ks * 2 + half) ^ swz_x(row)) * 16;
                const u32x4 va = lds16(as + row * PX + u);
                bf16x8_t fa;
                __builtin_memcpy(&fa, &va, 16);
                za = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks], fa, za, 0, 0, 0);      // D[i = channel][j = pixel]: lane = pixel, registers = channels 8 (r >> 2) + 4 half + (r & 3)
                if constexpr (NBR == 2) {
                    const u32x4 vb = lds16(xs + row * PX + u);
                    bf16x8_t fb;
                    __builtin_memcpy(&fb, &vb, 16);
                    zb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ks], fb, zb, 0, 0, 0);
                }
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const uint2 gq = *reinterpret_cast<const uint2*>(gs + row * PD + (((4 * wave + g4) ^ swz_d(row)) * 16) + half * 8);
                const unsigned mb = *reinterpret_cast<const unsigned char*>(ms + row * PB + 8 * wave + 2 * g4 + half);
                // the conv outputs as the forward pass rounded them
                const unsigned pa0 = pack_bf16x2(za[4 * g4], za[4 * g4 + 1]), pa1 = pack_bf16x2(za[4 * g4 + 2], za[4 * g4 + 3]);
                const unsigned pb0 = pack_bf16x2(zb[4 * g4], zb[4 * g4 + 1]), pb1 = pack_bf16x2(zb[4 * g4 + 2], zb[4 * g4 + 3]);
                const float z3[4] = {__uint_as_float(pa0 << 16), __uint_as_float(pa0 & 0xffff0000u), __uint_as_float(pa1 << 16), __uint_as_float(pa1 & 0xffff0000u)};
                const float zd[4] = {__uint_as_float(pb0 << 16), __uint_as_float(pb0 & 0xffff0000u), __uint_as_float(pb1 << 16), __uint_as_float(pb1 & 0xffff0000u)};
                const float gv[4] = {__uint_as_float(gq.x << 16), __uint_as_float(gq.x & 0xffff0000u), __uint_as_float(gq.y << 16), __uint_as_float(gq.y & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gm = __uint_as_float(__float_as_uint(gv[e]) & (unsigned)__builtin_amdgcn_sbfe((int)mb, e, 1));      // bit set ? g : +0 (rows past the end: g and the bits are zeros)
                    const int i = 4 * g4 + e;
                    s1[i] += gm;
                    s2a[i] += gm * ((z3[e] - mua[i]) * rsa[i]);
                    if constexpr (NBR == 2) s2b[i] += gm * ((zd[e] - mub[i]) * rsb[i]);
                }
            }
        }
    };

    // ---- the chunk loop: wait for chunk c, ONE barrier (chunk c visible, chunk c - 1 consumed by everyone), start chunk c + 1, compute chunk c ----
    if (nchunks > 0) dma(0);
    for (int cc = 0; cc < nchunks; ++cc) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (cc + 1 < nchunks) dma(cc + 1);
        compute(cc & 1);
    }

    // ---- a channel's 32 pixel-lanes in fp64; two fp32 partial rows per workgroup and BatchNorm (value + remainder) ----
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double t1 = (double)s1[i], ta = (double)s2a[i], tb = (double)s2b[i];
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
            t1 += __shfl_xor(t1, o, 64);
            ta += __shfl_xor(ta, o, 64);
            tb += __shfl_xor(tb, o, 64);
        }
        if (l31 == 0) {
            const int c = 32 * wave + 8 * (i >> 2) + 4 * half + (i & 3);
            const float h1 = (float)t1, ha = (float)ta, hb = (float)tb;
            float2* pa = reinterpret_cast<float2*>(a.part_a) + (long)c * a.rows;
            pa[split] = make_float2(h1, ha);
            pa[a.nsplit + split] = make_float2((float)(t1 - (double)h1), (float)(ta - (double)ha));
            if constexpr (NBR == 2) {
                float2* pb = reinterpret_cast<float2*>(a.part_b) + (long)c * a.rows;
                pb[split] = make_float2(h1, hb);
                pb[a.nsplit + split] = make_float2((float)(t1 - (double)h1), (float)(tb - (double)hb));
            }
        }
    }
}

}  // namespace

namespace mvf_internal {

int pw_sums_pair_launch(const PwSumsPairArgs& a, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        MVF_HIP_OK(hipFuncSetAttribute((const void*)pw_sums_pair_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
        MVF_HIP_OK(hipFuncSetAttribute((const void*)pw_sums_pair_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
        attr = true;
    }
    if (a.x) hipLaunchKernelGGL(pw_sums_pair_kernel<2>, dim3(a.nsplit), dim3(kT), kLds, st, a);
    else hipLaunchKernelGGL(pw_sums_pair_kernel<1>, dim3(a.nsplit), dim3(kT), kLds, st, a);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // namespace mvf_internal
