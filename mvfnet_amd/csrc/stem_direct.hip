// The 7x7 stride-2 stem conv (reference: codes/models/backbones/resnet.py:420-431 conv1 -> norm1 -> relu -> maxpool; :479-484 forward)
// as a DIRECT convolution for gfx950, bf16 storage, fp32 accumulation.
//
// The implicit-GEMM kernel (conv_nhwc.hip) stages an im2col row of 224 K-values (7 kh x 8 kw x 4 ch = 448 B) per output pixel through
// L2 -> LDS: 1.44 GB per launch at the C3 shape for an input that is 108 MB, and a per-tile set-up / epilogue chain that a 128 x 64 tile
// with K = 224 cannot amortise: 315 us against 83 us for the bytes of the output (profiles/r03_per_layer_bf16_train.txt).  Here
//   * a workgroup owns R output rows of ONE frame (R = 8: 896 pixels at 112 x 112) and stages the (2R + 5) x W x 4 input patch they read
//     ONCE: one contiguous 39 KB block of the padded NHWC4 operand (mvf_stem_prep), copied by LDS-DMA;
//   * the operand fragment of output pixel (oh, ow), tap row kh, k-step ks is the 16 bytes at ((2 oh + kh) W + 2 ow + 4 ks + 2 half) x 8 B
//     of that patch -- the im2col row is never materialised, consecutive pixels read consecutive 16-byte units (no swizzle needed);
//   * the whole packed weight matrix (64 x 224) lives in REGISTERS (112 VGPRs per wave: 14 k-steps x 2 column blocks), so a k-step is one
//     ds_read_b128 for two matrix instructions;
//   * waves are independent after the patch barrier: each walks its own 32-pixel blocks and transposes its outputs through a PRIVATE LDS
//     slab (16-byte coalesced stores), so one wave's epilogue runs under the other waves' matrix instructions;
//   * EPI 1 (training): BatchNorm batch-statistic partial sums of the ROUNDED outputs, one partial row per workgroup (the other partial rows
//     the tile covers are written as zeros: the finalize sums all rows); EPI 4 (inference): folded-BN bias + ReLU.
// Arithmetic = the implicit-GEMM kernel's: the same bf16 products accumulated in fp32 in the same k order (kh-major), rounded once.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kKS = 14;            // k-steps of 16: 7 kh x (8 kw x 4 ch = 32 K-values)
constexpr int kStgPitch = 136;     // bf16 C slab pitch: 128 B + 2 dwords -> conflict-free 8-byte writes
constexpr int kStgBytes = 32 * kStgPitch;

__device__ __forceinline__ int fdiv(int n, unsigned mul, unsigned shr) { return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> shr); }

struct KArgs {
    const char* x;
    const char* w;
    char* y;
    const float* bias;
    float* stats_part;
    const float* stats_shift;
    int stats_rows;
    int H, W, Ho, Wo, R;
    int tiles_per_frame, nblocks;      // 32-pixel blocks per tile
    int tiles, tiles_per_wg;           // a workgroup walks tiles_per_wg consecutive tiles (the weights are loaded once)
    int patch_units;                   // 16-byte units of the patch
    int patch_lds;                     // bytes reserved for it (whole 1 KB DMA instructions)
    unsigned fd_wo_mul, fd_wo_shr, fd_tpf_mul, fd_tpf_shr;
    long wK;
};

template <int EPI>
__global__ __launch_bounds__(256, 2) void stem_direct_kernel(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int row_bytes = a.W * 8;
    auto stage_patch = [&](int t) {          // rows [2 oh0, 2 oh0 + 2R + 5) of the tile's frame (padded image), contiguous in memory
        const int frame = fdiv(t, a.fd_tpf_mul, a.fd_tpf_shr), oh0 = (t - frame * a.tiles_per_frame) * a.R;
        const char* src = a.x + ((long)frame * a.H + 2 * oh0) * row_bytes;
        const i32x4 rs = rsrc_words(src, (unsigned)a.patch_units * 16u);
        for (int u0 = wave * 64; u0 < a.patch_units; u0 += 256)         // (units past the patch: out-of-range offsets DMA zeros)
            glds16(rs, (unsigned)(u0 * 16), (unsigned)(u0 + lane) * 16u);
    };
    const int t_begin = blockIdx.x * a.tiles_per_wg, t_end = min(t_begin + a.tiles_per_wg, a.tiles);
    stage_patch(t_begin);
    // ---- the weights: fragment (k-step t, column block j) = 8 K-values of output channel 32 j + (lane & 31) ----
    bf16x8 wf[kKS][2];
    {
        const char* wp = a.w + ((long)l31 * a.wK + half * 8) * 2;
#pragma unroll
        for (int t = 0; t < kKS; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint4 v = *reinterpret_cast<const uint4*>(wp + ((long)j * 32 * a.wK + t * 16) * 2);
                __builtin_memcpy(&wf[t][j], &v, 16);
            }
    }
    float bias[2][4][4];
    (void)bias;
    if constexpr (EPI == 4) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(a.bias + j * 32 + 8 * g + 4 * half);
                bias[j][g][0] = b.x; bias[j][g][1] = b.y; bias[j][g][2] = b.z; bias[j][g][3] = b.w;
            }
    }
    const int cq = lane & 7, rq = lane >> 3;                     // read-back: 16-byte channel group, row within 8
    float kk[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { kk[k] = 0.f; s1[k] = 0.f; s2[k] = 0.f; }
    if constexpr (EPI == 1) {
        if (a.stats_shift) {
            const float4 u = *reinterpret_cast<const float4*>(a.stats_shift + cq * 8), v = *reinterpret_cast<const float4*>(a.stats_shift + cq * 8 + 4);
            kk[0] = u.x; kk[1] = u.y; kk[2] = u.z; kk[3] = u.w; kk[4] = v.x; kk[5] = v.y; kk[6] = v.z; kk[7] = v.w;
        }
    }
    char* stg = smem + a.patch_lds + wave * kStgBytes;
    for (int t = t_begin; t < t_end; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frame = fdiv(t, a.fd_tpf_mul, a.fd_tpf_shr), oh0 = (t - frame * a.tiles_per_frame) * a.R;
    const long m0 = ((long)frame * a.Ho + oh0) * a.Wo;           // first output pixel of the tile
    char* ytile = a.y + m0 * 128;
    for (int mb = wave; mb < a.nblocks; mb += 4) {
        const int p = mb * 32 + l31;
        const int orow = fdiv(p, a.fd_wo_mul, a.fd_wo_shr), ow = p - orow * a.Wo;
        const char* ap = smem + (2 * orow) * row_bytes + (2 * ow + 2 * half) * 8;
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        uint4 fa[kKS];
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            fa[2 * kh] = *reinterpret_cast<const uint4*>(ap + kh * row_bytes);
            fa[2 * kh + 1] = *reinterpret_cast<const uint4*>(ap + kh * row_bytes + 32);
        }
        __builtin_amdgcn_sched_barrier(0);       // all 14 fragment reads in flight before the first matrix instruction (left alone, the scheduler issues them pair by pair)
#pragma unroll
        for (int t2 = 0; t2 < kKS; ++t2) {
            bf16x8 av;
            __builtin_memcpy(&av, &fa[t2], 16);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t2][0], av, acc[0], 0, 0, 0);     // D^T: pixel = lane & 31, channel = 8 (r >> 2) + 4 half + (r & 3)
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t2][1], av, acc[1], 0, 0, 0);
        }
        // ---- transpose through the wave's slab: rows = pixels, 128 B of channels ----
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = acc[j][4 * g], v1 = acc[j][4 * g + 1], v2 = acc[j][4 * g + 2], v3 = acc[j][4 * g + 3];
                if constexpr (EPI == 4) {
                    v0 = fmaxf(v0 + bias[j][g][0], 0.f); v1 = fmaxf(v1 + bias[j][g][1], 0.f);
                    v2 = fmaxf(v2 + bias[j][g][2], 0.f); v3 = fmaxf(v3 + bias[j][g][3], 0.f);
                }
                uint2 pk;
                pk.x = pack_bf16x2(v0, v1);
                pk.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(stg + l31 * kStgPitch + (j * 32 + 8 * g + 4 * half) * 2) = pk;
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rq + 8 * i;
            const char* sp = stg + row * kStgPitch + cq * 16;
            const uint2 lo = *reinterpret_cast<const uint2*>(sp), hi = *reinterpret_cast<const uint2*>(sp + 8);
            const uint4 pk = make_uint4(lo.x, lo.y, hi.x, hi.y);
            *reinterpret_cast<uint4*>(ytile + (long)(mb * 32 + row) * 128 + cq * 16) = pk;
            if constexpr (EPI == 1) {                             // statistics of what is STORED
                float v[8];
                v[0] = __uint_as_float(pk.x << 16); v[1] = __uint_as_float(pk.x & 0xffff0000u);
                v[2] = __uint_as_float(pk.y << 16); v[3] = __uint_as_float(pk.y & 0xffff0000u);
                v[4] = __uint_as_float(pk.z << 16); v[5] = __uint_as_float(pk.z & 0xffff0000u);
                v[6] = __uint_as_float(pk.w << 16); v[7] = __uint_as_float(pk.w & 0xffff0000u);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float d = v[k] - kk[k];
                    s1[k] += d;
                    s2[k] += d * d;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                              // every wave is done with the patch (and its slab)
    if (t + 1 < t_end) stage_patch(t + 1);
    if constexpr (EPI == 1) {
        // column sums of the tile: over the 8 row lanes of a channel group (fixed butterfly), then the 4 waves in order; one partial row
        // per tile, the other partial rows it covers are zeros
#pragma unroll
        for (int off = 8; off < 64; off <<= 1)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s1[k] += __shfl_xor(s1[k], off, 64);
                s2[k] += __shfl_xor(s2[k], off, 64);
            }
        float2* red = reinterpret_cast<float2*>(smem + a.patch_lds);
        if (lane < 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) red[wave * 64 + lane * 8 + k] = make_float2(s1[k], s2[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
        __syncthreads();
        const int rows_per_tile = a.nblocks / 4;                  // partial rows (one per 128 output pixels) this tile covers
        const long row0 = m0 >> 7;
        if (tid < 64) {
            const float2 p0 = red[tid], p1 = red[64 + tid], p2 = red[128 + tid], p3 = red[192 + tid];
            float2* dst = reinterpret_cast<float2*>(a.stats_part) + (long)tid * a.stats_rows + row0;
            dst[0] = make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
        }
        for (int e = tid; e < 64 * (rows_per_tile - 1); e += 256) {
            const int c = e / (rows_per_tile - 1), r = e - c * (rows_per_tile - 1) + 1;
            reinterpret_cast<float2*>(a.stats_part)[(long)c * a.stats_rows + row0 + r] = make_float2(0.f, 0.f);
        }
    }
    }
}

inline void fd_make_local(unsigned d, unsigned& mul, unsigned& shr) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    shr = l;
}

}  // namespace

namespace mvf_internal {

// MVF_OK when launched, -1 when the shape is not this kernel's (the caller falls back to the implicit-GEMM path)
int stem_direct_launch(const StemDirectArgs& s, hipStream_t st) {
    if (mvf_policy_int("stem_direct", 1) == 0) return -1;      // A/B switch, read per call (one stem launch per pass): 0 = the implicit-GEMM kernel
    if (s.epi != 1 && s.epi != 4) return -1;
    if (s.W % 2 || s.Ho <= 0 || s.Wo <= 0 || s.wK != 224) return -1;
    if (2 * (s.Ho - 1) + 7 > s.H || 2 * (s.Wo - 1) + 8 > s.W) return -1;          // every fragment read stays inside the image row
    int R = 0;
    const int px_unit = s.epi == 1 ? 128 : 32;          // whole statistic partial rows / whole 32-pixel blocks
    for (int r = 8; r >= 1; r >>= 1) {
        const long px = (long)r * s.Wo;
        const long bytes = (long)(2 * r + 5) * s.W * 8;
        if (s.Ho % r == 0 && px % px_unit == 0 && bytes <= 44 * 1024) { R = r; break; }
    }
    if (!R) return -1;
    KArgs a = {};
    a.x = (const char*)s.x; a.w = (const char*)s.w; a.y = (char*)s.y; a.bias = s.bias;
    a.stats_part = s.stats_part; a.stats_shift = s.stats_shift; a.stats_rows = s.stats_rows;
    a.H = s.H; a.W = s.W; a.Ho = s.Ho; a.Wo = s.Wo; a.R = R; a.wK = s.wK;
    a.tiles_per_frame = s.Ho / R;
    a.nblocks = R * s.Wo / 32;
    a.patch_units = (2 * R + 5) * s.W * 8 / 16;
    a.patch_lds = (a.patch_units + 63) / 64 * 1024;
    fd_make_local((unsigned)s.Wo, a.fd_wo_mul, a.fd_wo_shr);
    fd_make_local((unsigned)a.tiles_per_frame, a.fd_tpf_mul, a.fd_tpf_shr);
    const size_t lds = (size_t)a.patch_lds + 4 * kStgBytes;
    const long tiles = (long)s.N * a.tiles_per_frame;
    if (tiles >= (1L << 31)) return -1;
    a.tiles = (int)tiles;
    a.tiles_per_wg = mvf_policy_int("stem_tpw", (int)((tiles + 511) / 512));           // ~2 workgroups per CU, each reusing its weight registers
    if (a.tiles_per_wg < 1) a.tiles_per_wg = 1;
    const long grid = (tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    if (s.epi == 1) {
        static bool attr = false;
        if (!attr) { MVF_HIP_OK(hipFuncSetAttribute((const void*)stem_direct_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); attr = true; }
        hipLaunchKernelGGL(stem_direct_kernel<1>, dim3((unsigned)grid), dim3(256), lds, st, a);
    } else {
        static bool attr = false;
        if (!attr) { MVF_HIP_OK(hipFuncSetAttribute((const void*)stem_direct_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); attr = true; }
        hipLaunchKernelGGL(stem_direct_kernel<4>, dim3((unsigned)grid), dim3(256), lds, st, a);
    }
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // namespace mvf_internal
