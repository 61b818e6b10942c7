// The whole backward of a z3-free bottleneck's last conv in ONE pass (gfx950, bf16 storage, 64 -> 256 channels): dz3 never leaves the chip.
//
// Reference arithmetic: autograd of Bottleneck.forward (codes/models/backbones/resnet.py:229-244): out = relu(bn3(conv3(a2)) + identity),
// a2 = relu(bn2(z2)).  With g = dL/dout, gm = g * [out > 0] and bn3's final dgamma / dbeta (pw_sums.hip MODE 1 + mvf_bn_bwd_finalize):
//     z3[m][c]  = round_bf16(sum_k a2[m][k] W[c][k])                                   (recomputed: the block never stored it)
//     dz3[m][c] = round_bf16(gamma invstd (gm - dbeta / M - (z3 - mean) invstd dgamma / M))   (conv_nhwc.hip EPI 9 = bn_bwd_apply_kernel)
//     dW[c][k]  = sum_m dz3[m][c] a2[m][k]                                               (wgrad_bf16_kernel; fp32 slabs, fixed-order reduce)
//     da2[m][k] = round_bf16(sum_c dz3[m][c] W[c][k])                                    (the data gradient, conv_nhwc.hip EPI 6)
//     bn2's backward sums: sum_m gq, sum_m gq (z2 - mean2) invstd2 with gq = da2 * [scale2 z2 + shift2 > 0]      (z_in = NULL: a conv whose input is
//     not a BatchNorm's activation -- a downsample branch reading the block input -- skips them)
// The un-fused step runs three launches for this (conv + BatchNorm-backward apply, data gradient + sums, weight gradient): dz3 -- a 4 x planes
// tensor, 411 MB in layer1 of the R50 8x8 step -- is written once and read twice.  Here a persistent workgroup (8 waves, one per CU) walks
// 64-pixel chunks of its row range and keeps the chunk's dz3 tile in LDS between three small matrix products:
//   GEMM 1  z3^T = W a2^T          wave w: channels [32 w, 32 w + 32) x 64 pixels; its W fragments live in registers
//   (VALU)  dz3 in place, thread = (8-channel unit, 4 rows): the arithmetic of bn_bwd_apply_kernel
//   then the workgroup splits: waves 4-7 run GEMM 2, waves 0-3 GEMM 3 (a wave of each kind per SIMD, 16 matrix instructions each):
//   GEMM 2  dW += dz3^T a2          transpose reads (ds_read_b64_tr_b16) of both tiles; wave 4 + q owns channels [64 q, 64 q + 64) x 64
//   GEMM 3  da2^T = W^T dz3^T       wave = (pixel block, input-channel half): ONE accumulation chain over the 256 channels in ascending order --
//                                   da2 is bit-identical to the un-fused data gradient -- rounded and written to a bf16 LDS tile
//   (VALU)  thread = (row, 8-channel unit of da2): 16-byte store of the tile, bn2's gated sums in registers
//   The 64 registers of a lane hold GEMM 3's sixteen W^T fragments in waves 0-3 and GEMM 2's four accumulator blocks in waves 4-7.
// Every operand (a2, g, sign bits, z2) arrives by LDS-DMA: no staging registers and no compiler-visible loads in the loop (the first version
// prefetched g through registers, spilled 13 of them and lost 80 us per launch to the scratch reloads' waits).  a2 / bits / z2 are one chunk
// ahead, g -- four fifths of the bytes -- two: its buffer is dead as soon as dz3 is formed.  LDS tiles are XOR-swizzled in 16-byte units so that
// the row-per-lane 16-byte reads (GEMM 1 / 3 operands), the transpose reads (GEMM 2) and the unit-per-lane accesses are all conflict-free.
// Three LDS barriers per chunk; the wait before the last one is counted (vmcnt(4): the four g transfers of chunk c + 2 stay in flight).
// HBM traffic per pixel: g 512 B + bits 64 B + a2 128 B + z2 128 B in, da2 128 B out (960 B; the three launches it replaces move 2.6 KB).
// Outputs: da2, one fp32 weight-gradient slab per workgroup ([nsplit][256][64], mvf_wgrad_slab_reduce), two partial rows of bn2's sums per
// workgroup (value + remainder of an fp64 sum, channel-major [64][2 nsplit][2], mvf_bn_bwd_finalize).
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kT = 512;                  // 8 waves, two per SIMD
constexpr int CH = 64;                   // pixels per chunk
constexpr int NC = 256, NK = 64;         // conv output / input channels
constexpr int PD = NC * 2;               // dz3 / g tile row pitch (bytes): 32 units of 16 bytes
constexpr int PX = NK * 2;               // a2 / z2 / da2 tile row pitch: 8 units
constexpr int PB = NC / 4;               // sign-bit tile row pitch (bytes)
constexpr unsigned kOOB = 0x80000000u;
constexpr int kOffD = 0, kOffX = CH * PD, kOffG = kOffX + 2 * CH * PX, kOffZ = kOffG + 2 * CH * PD, kOffM = kOffZ + 2 * CH * PX, kOffT = kOffM + 2 * CH * PB,
              kOffC = kOffT + CH * PX, kOffB = kOffC + 4 * NK * 4, kLds = kOffB + 4 * NC * 4;
static_assert(CH * 8 * 16 * 4 <= CH * PD, "the final reduction of the sums reuses a g buffer");
static_assert(kLds <= 160 * 1024, "LDS");

// row -> XOR mask of the 16-byte unit index.  Both are invariant under row += 16 (a transpose read's k-step) and give 16 distinct bank groups
// over any 16 consecutive rows (ds_read_b128 services 16 lanes per cycle) as well as over the row sets {0-3, 8-11} / {4-7, 12-15} of a
// transpose read.
__device__ __forceinline__ int swz_d(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int swz_x(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ void unpack8(const u32x4 r, float (&f)[8]) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 r;
    r.x = pack_bf16x2(f[0], f[1]); r.y = pack_bf16x2(f[2], f[3]); r.z = pack_bf16x2(f[4], f[5]); r.w = pack_bf16x2(f[6], f[7]);
    return r;
}
__device__ __forceinline__ u32x4 lds16(const char* p) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    u32x4 r;
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
}
__device__ __forceinline__ void ldp8(const float* p, int c, float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 8; j += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + c + j);
        f[j] = v.x; f[j + 1] = v.y; f[j + 2] = v.z; f[j + 3] = v.w;
    }
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void full_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void g_ahead_barrier() { asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef policy pwbf_ablate
#define PWBF_ON(bit) if (!(a.ablate & (bit)))
#else
#define PWBF_ON(bit)
#endif

__global__ __launch_bounds__(kT) void pw_bwd_fused_kernel(const mvf_internal::PwBwdFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ds = smem + kOffD;                                         // [CH][PD]      z3, then dz3 (bf16)
    char* Xs = smem + kOffX;                                         // [2][CH][PX]   a2 (bf16)
    char* Gs = smem + kOffG;                                         // [2][CH][PD]   g (bf16)
    char* Zs = smem + kOffZ;                                         // [2][CH][PX]   z2 (bf16)
    char* Ms = smem + kOffM;                                         // [2][CH][PB]   sign bits of the block output
    char* Ts = smem + kOffT;                                         // [CH][PX]      da2 (bf16, as it is stored)
    float* Cs = reinterpret_cast<float*>(smem + kOffC);              // [4][NK]       bn2: scale, shift, mean, invstd
    float* Bs = reinterpret_cast<float*>(smem + kOffB);              // [4][NC]       bn3: gamma invstd, dbeta / M, invstd dgamma / M, mean

    const int split = blockIdx.x;
    const int m_begin = split * a.rows_per_split, m_end = min(a.M, m_begin + a.rows_per_split);
    const int nchunks = (m_end - m_begin + CH - 1) / CH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const bool role_a = wave < 4;                                    // GEMM 3 (waves 0-3) / GEMM 2 (waves 4-7)
    const bool sums = a.z_in != nullptr;                             // the conv input is relu(bn_in(z_in)): gate the gradient and take bn_in's backward sums

    // ---- this thread's dz3 unit: channels [8 q, 8 q + 8), rows r0 + 16 i; bn3's backward coefficients (bn_bwd_apply_kernel's folding) ----
    const int q = tid & 31, r0 = tid >> 5;
    if (tid < NC) {
        const float inv_m = 1.0f / (float)a.M;
        const float rs = a.invstd[tid];
        Bs[tid] = a.gamma[tid] * rs; Bs[NC + tid] = a.dbeta[tid] * inv_m; Bs[2 * NC + tid] = rs * a.dgamma[tid] * inv_m; Bs[3 * NC + tid] = a.mean[tid];
    }
    if (sums && tid < NK) {
        Cs[tid] = a.in_scale[tid]; Cs[NK + tid] = a.in_shift[tid]; Cs[2 * NK + tid] = a.in_mean[tid]; Cs[3 * NK + tid] = a.in_invstd[tid];
    }

    // ---- weights in registers.  GEMM 1 (A operand, row = channel 32 wave + l31): 8 input channels per k-step half.  GEMM 3 (A operand,
    // row = input channel 32 kb + l31 of W^T): 8 output channels per k-step half, all 16 k-steps, gathered once (waves 0-3, in run() below);
    // waves 4-7 keep GEMM 2's four accumulator blocks in those registers instead ----
    const int pb3 = wave & 1, kb3 = (wave >> 1) & 1;
    bf16x8_t w1[4];
    {
        const char* wp = reinterpret_cast<const char*>(a.w);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 v = *reinterpret_cast<const uint4*>(wp + ((long)(32 * wave + l31) * NK + ks * 16 + half * 8) * 2);
            __builtin_memcpy(&w1[ks], &v, 16);
        }
    }

    const __amdgpu_buffer_rsrc_t rs_dx = __builtin_amdgcn_make_buffer_rsrc(a.dx, 0, (unsigned)min((long)a.M * NK * 2, 0x7ffffff0L), 0x00020000);

    // ---- LDS-DMA: a2 and z2 one 16-byte transfer per thread and chunk (lane at LDS position (row, p) fetches a2's source unit p ^ swz_x(row)),
    // g four, the sign bits one per thread of waves 0-3 ----
    const int rbx = tid >> 3;
    const int qbx = (tid & 7) ^ swz_x(rbx);
    const i32x4 gs_x = rsrc_words(a.a, (unsigned)min((long)a.M * a.aps * 2, 0x7ffffff0L));
    const i32x4 gs_g = rsrc_words(a.g, (unsigned)min((long)a.M * a.g_pitch * 2, 0x7ffffff0L));
    const i32x4 gs_z = rsrc_words(a.z_in, (unsigned)min((long)a.M * NK * 2, 0x7ffffff0L));
    const i32x4 gs_m = rsrc_words(a.bits, (unsigned)min((long)a.M * PB, 0x7ffffff0L));
    auto lds_base = [&](char* p) { return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)p); };
    const unsigned lds_x = lds_base(Xs) + wave * 8 * PX, lds_z = lds_base(Zs) + wave * 8 * PX;
    const unsigned lds_g = lds_base(Gs) + wave * 2 * PD, lds_m = lds_base(Ms) + wave * 16 * PB;
    auto dma_x = [&](int cc) {
        const int m = m_begin + cc * CH + rbx;
        glds16(gs_x, lds_x + (unsigned)((cc & 1) * CH * PX), m < m_end ? (unsigned)(m * a.aps + qbx * 8) * 2u : kOOB);
    };
    auto dma_g = [&](int cc) {                                       // exactly four transfers per wave (g_ahead_barrier counts them)
        const int mc = m_begin + cc * CH;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                // a wave lays down rows 16 i + 2 wave, + 1 (32 units each)
            const int m = mc + 16 * i + 2 * wave + half;
            glds16(gs_g, lds_g + (unsigned)((cc & 1) * CH * PD + 16 * i * PD), m < m_end ? (unsigned)(m * a.g_pitch + l31 * 8) * 2u : kOOB);
        }
    };
    auto dma_mz = [&](int cc) {
        const int mc = m_begin + cc * CH;
        if (sums) {                                                  // z2: rows 8 wave + (lane >> 3), unit lane & 7
            const int m = mc + rbx;
            glds16(gs_z, lds_z + (unsigned)((cc & 1) * CH * PX), m < m_end ? (unsigned)(m * NK + (tid & 7) * 8) * 2u : kOOB);
        }
        if (role_a) {                                                // bits: rows 16 wave + (lane >> 2), unit lane & 3
            const int m = mc + 16 * wave + (lane >> 2);
            glds16(gs_m, lds_m + (unsigned)((cc & 1) * CH * PB), m < m_end ? (unsigned)(m * PB + (lane & 3) * 16) : kOOB);
        }
    };
    const int re = tid >> 3, ue = tid & 7;                           // da2 / z2 unit of the last phase: row re, channels [8 ue, 8 ue + 8)

    // ---- GEMM 1: z3^T tile.  D[i = channel][j = pixel]: lane = pixel l31 of block pb, registers = channels 8 (r >> 2) + 4 half + (r & 3) ----
    auto gemm1 = [&](int buf) {
        const char* xs = Xs + buf * CH * PX;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const int row = pb * 32 + l31;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 v = lds16(xs + row * PX + (((ks * 2 + half) ^ swz_x(row)) * 16));
                bf16x8_t bv;
                __builtin_memcpy(&bv, &v, 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[ks], bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                         // z3 as the forward pass rounded it
                uint2 pk;
                pk.x = pack_bf16x2(acc[4 * g4], acc[4 * g4 + 1]);
                pk.y = pack_bf16x2(acc[4 * g4 + 2], acc[4 * g4 + 3]);
                *reinterpret_cast<uint2*>(Ds + row * PD + (((4 * wave + g4) ^ swz_d(row)) * 16) + half * 8) = pk;
            }
        }
    };

    // ---- dz3 in place: this thread's unit q of rows r0 + 16 i ----
    auto form_dz = [&](int buf) {
        float ca[8], cd[8], ck[8], cmu[8];
        ldp8(Bs, q * 8, ca); ldp8(Bs + NC, q * 8, cd); ldp8(Bs + 2 * NC, q * 8, ck); ldp8(Bs + 3 * NC, q * 8, cmu);
        const char* gs = Gs + buf * CH * PD;
        const char* ms = Ms + buf * CH * PB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 16 * i;
            char* p = Ds + row * PD + ((q ^ swz_d(row)) * 16);
            const u32x4 zu = lds16(p), gu = lds16(gs + row * PD + q * 16);
            const unsigned mb = *reinterpret_cast<const unsigned short*>(ms + row * PB + q * 2);
            float gv[8], zv[8], o[8];
            unpack8(gu, gv);
            unpack8(zu, zv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gv[j] = __uint_as_float(__float_as_uint(gv[j]) & (unsigned)__builtin_amdgcn_sbfe((int)mb, j + (j >= 4 ? 4 : 0), 1));      // bit set ? g : +0 (v_bfe_i32 + v_and)
                o[j] = ca[j] * (gv[j] - cd[j] - (zv[j] - cmu[j]) * ck[j]);
            }
            const u32x4 pk = pack8(o);
            *reinterpret_cast<uint4*>(p) = make_uint4(pk.x, pk.y, pk.z, pk.w);
        }
    };

    // ---- GEMM 2 (waves 4-7): dW tile (wgrad_bf16_kernel's transpose reads): wave 4 + wq owns channels [64 wq, 64 wq + 64) x the 64 input channels;
    // group tg = lane >> 4 supplies pixel rows (ti >> 2) + 8 (tg >> 1) [+ 4 for the second read] and channel quad 16 (tg & 1) + 4 (ti & 3) ----
    const int wq = wave & 3;
    int offA[2][2], offB[2][2];
    {
        const int tg = lane >> 4, ti = lane & 15;
        const int trow = (ti >> 2) + 8 * (tg >> 1);
        const int tunit = 2 * (tg & 1) + ((ti & 3) >> 1), thalf = ti & 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = trow + 4 * h;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                offA[i][h] = row * PD + ((((wq * 2 + i) * 4 + tunit) ^ swz_d(row)) * 16) + thalf * 8;
                offB[i][h] = row * PX + (((i * 4 + tunit) ^ swz_x(row)) * 16) + thalf * 8;
            }
        }
    }
    auto gather = [&](const char* lo_p, const char* hi_p) {
        typedef short v4s __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4s* lds_v4s;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(lo_p));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(hi_p));
        bf16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };
    // ---- last phase of a chunk: da2 stored; bn2's gated sums of what was stored ----
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    auto finish = [&](int cc) {                                      // (cc = -1: nothing to finish yet -- every effect is masked)
        const int m = m_begin + cc * CH + re;
        const bool ok = cc >= 0 && m < m_end;
        const int buf = cc & 1;
        const u32x4 pk = lds16(Ts + re * PX + ((ue ^ swz_x(re)) * 16));
        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_dx, ok ? (unsigned)(m * NK + ue * 8) * 2u : kOOB, 0, 0);
        if (!sums) return;
        const u32x4 zq = lds16(Zs + buf * CH * PX + re * PX + ue * 16);
        float vr[8], zv[8], sc[8], sh[8], mu[8], rs[8];
        unpack8(pk, vr);
        unpack8(zq, zv);
        ldp8(Cs, ue * 8, sc); ldp8(Cs + NK, ue * 8, sh); ldp8(Cs + 2 * NK, ue * 8, mu); ldp8(Cs + 3 * NK, ue * 8, rs);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gq = (ok && (zv[j] * sc[j] + sh[j]) > 0.f) ? vr[j] : 0.f;
            s1[j] += gq;
            s2[j] += gq * (ok ? (zv[j] - mu[j]) * rs[j] : 0.f);      // (a masked row may hold anything, NaN included)
        }
    };

    // ---- the chunk loop: three LDS barriers per chunk.  Transfers: a2(c + 1) from the top of iteration c (its buffer was last read by GEMM 2 of
    // chunk c - 1); bits / z2(c + 1) from the first barrier on (z2(c - 1) was read by the finish at the top); g(c + 2) from the second barrier on
    // (g(c) is dead once dz3(c) is formed).  The wait before the last barrier lets exactly those four g transfers stay in flight: loads complete
    // in order, so vmcnt(4) implies everything issued before them -- a2 / bits / z2(c + 1) and, from the previous iteration, g(c + 1) -- has landed
    // (the one store in the window, da2(c - 1), can only make the wait longer).  (In the first pass the finish reads an uninitialised tile: masked.) ----
    // The loop exists twice, once per role, so that the 64 role-specific registers of a lane (GEMM 3's W^T fragments / GEMM 2's accumulators)
    // are allocated on top of each other.
    auto run = [&](auto role_c) {
        constexpr bool A = decltype(role_c)::value;
        bf16x8_t wt[A ? 16 : 1];
        f32x16 accw[A ? 1 : 4];
        if constexpr (A) {
            const short* ws = reinterpret_cast<const short*>(a.w);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) wt[ks][e] = ws[(ks * 16 + half * 8 + e) * NK + kb3 * 32 + l31];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) accw[i][r] = 0.f;
        }
        auto gemm2 = [&](int buf) {
            if constexpr (!A) {
                const char* xs = Xs + buf * CH * PX;
#pragma unroll
                for (int ks = 0; ks < CH / 16; ++ks) {
                    bf16x8_t fa[2], fb[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[j] = gather(xs + ks * 16 * PX + offB[j][0], xs + ks * 16 * PX + offB[j][1]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[i] = gather(Ds + ks * 16 * PD + offA[i][0], Ds + ks * 16 * PD + offA[i][1]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) accw[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], accw[i * 2 + j], 0, 0, 0);
                }
            }
        };
        // GEMM 3: da2^T tile of (pixel block pb3, input-channel block kb3), the 256 channels in one chain.  D[i = input channel][j = pixel]
        auto gemm3 = [&]() {
            if constexpr (A) {
                const int row = pb3 * 32 + l31;
                int vx = half ^ swz_d(row);                          // (2 ks + half) ^ swz = (2 ks) ^ vx; opaque, so that the sixteen addresses are
                asm volatile("" : "+v"(vx));                         // recomputed here (one v_xor each) instead of living in sixteen registers
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const u32x4 v = lds16(Ds + row * PD + (((ks * 2) ^ vx) * 16));
                    bf16x8_t bv;
                    __builtin_memcpy(&bv, &v, 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[ks], bv, acc, 0, 0, 0);
                    if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four operand fragments in flight, not sixteen (64 registers: spills)
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {                     // input channels 32 kb3 + 8 g4 + 4 half ..+3 = unit 4 kb3 + g4, second half of it for half = 1
                    uint2 pk;
                    pk.x = pack_bf16x2(acc[4 * g4], acc[4 * g4 + 1]);
                    pk.y = pack_bf16x2(acc[4 * g4 + 2], acc[4 * g4 + 3]);
                    *reinterpret_cast<uint2*>(Ts + row * PX + (((kb3 * 4 + g4) ^ swz_x(row)) * 16) + half * 8) = pk;
                }
            }
        };
        if (nchunks > 0) {
            dma_x(0);
            dma_mz(0);
            dma_g(0);
            if (nchunks > 1) dma_g(1);
        }
        full_barrier();
        for (int cc = 0; cc < nchunks; ++cc) {
            if (cc + 1 < nchunks) dma_x(cc + 1);
            PWBF_ON(32) finish(cc - 1);
            PWBF_ON(2) gemm1(cc & 1);
            lds_barrier();                                           // z3 tile complete; chunk c - 1's da2 tile and z2 read by everyone
            if (cc + 1 < nchunks) dma_mz(cc + 1);
            PWBF_ON(1) form_dz(cc & 1);
            lds_barrier();                                           // dz3 tile complete, g(c) dead
            const bool g2 = cc + 2 < nchunks;
            PWBF_ON(16) if (g2) dma_g(cc + 2);
            PWBF_ON(8) gemm3();
            PWBF_ON(4) gemm2(cc & 1);
#ifdef policy pwbf_ablate
            full_barrier();
#else
            if (g2) g_ahead_barrier(); else full_barrier();          // da2 tile complete; every read of the dz3 / a2 tiles done; chunk c + 1 landed
#endif
        }
        finish(nchunks - 1);
        // one fp32 weight-gradient slab per workgroup: part[split][256][64] (wgrad_reduce_kernel's layout)
        if constexpr (!A) {
            float* out = a.part + (long)split * NC * NK;
            const int lr = lane >> 5, lc = lane & 31;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = j * 32 + lc;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (wq * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lr;
                        out[(long)row * NK + col] = accw[i * 2 + j][r];
                    }
            }
        }
    };
    if (role_a) run(std::true_type());
    else run(std::false_type());

    // ---- bn2's sums: the 64 row-threads of a channel unit in fixed order, in fp64; two fp32 partial rows per workgroup (value + remainder) ----
    __syncthreads();
    if (sums) {
        float* red = reinterpret_cast<float*>(Gs);                   // [row re][unit ue][16]
        float* dst = red + (re * 8 + ue) * 16;
        *reinterpret_cast<float4*>(dst) = make_float4(s1[0], s1[1], s1[2], s1[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(s1[4], s1[5], s1[6], s1[7]);
        *reinterpret_cast<float4*>(dst + 8) = make_float4(s2[0], s2[1], s2[2], s2[3]);
        *reinterpret_cast<float4*>(dst + 12) = make_float4(s2[4], s2[5], s2[6], s2[7]);
        __syncthreads();
        if (tid < NK) {
            const int u = tid >> 3, j = tid & 7;
            double t1 = 0.0, t2 = 0.0;
            for (int r = 0; r < CH; ++r) {
                t1 += (double)red[(r * 8 + u) * 16 + j];
                t2 += (double)red[(r * 8 + u) * 16 + 8 + j];
            }
            float2* p = reinterpret_cast<float2*>(a.sums_part) + (long)tid * a.sums_rows;
            const float h1 = (float)t1, h2 = (float)t2;
            p[split] = make_float2(h1, h2);
            p[a.nsplit + split] = make_float2((float)(t1 - (double)h1), (float)(t2 - (double)h2));
        }
    }
}

}  // namespace

namespace mvf_internal {

// workgroups (= weight-gradient slabs; bn2's sums take 2 x this many partial rows) for an [m][256] gradient over a 64-channel conv input; 0 = not built
int pw_bwd_fused_plan(long m, int c, int k, int* rows_per_split) {
    if (c != NC || k != NK || m <= 0 || m * (long)NC * 2 >= 0x7ffffff0L) return 0;
    static const int wgs_env = std::max(1, mvf_policy_int("pwbf_wgs", 256));      // one persistent workgroup per CU
    long rows = (m + wgs_env - 1) / wgs_env;
    rows = (rows + CH - 1) / CH * CH;
    if (rows < 2 * CH) rows = 2 * CH;
    if (rows_per_split) *rows_per_split = (int)rows;
    return (int)((m + rows - 1) / rows);
}

int pw_bwd_fused_launch(const PwBwdFusedArgs& a0, hipStream_t st) {
    auto k = pw_bwd_fused_kernel;
    static bool attr = false;
    if (!attr) {
        MVF_HIP_OK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
        attr = true;
    }
    PwBwdFusedArgs a = a0;
#ifdef policy pwbf_ablate
    a.ablate = mvf_policy_int("pwbf_ablate", 0);
#endif
    hipLaunchKernelGGL(k, dim3(a.nsplit), dim3(kT), kLds, st, a);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // namespace mvf_internal
