// Shared host/device helpers for libmvfnet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <stdint.h>
#include <stdio.h>

#include "mvfnet_hip.h"

void mvf_set_error(const char* fmt, ...);

#define MVF_HIP_OK(call)                                                                        \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            mvf_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return MVF_EHIP;                                                                    \
        }                                                                                       \
    } while (0)

#define MVF_REQUIRE(cond, code, ...)    \
    do {                                \
        if (!(cond)) {                  \
            mvf_set_error(__VA_ARGS__); \
            return (code);              \
        }                               \
    } while (0)

#define MVF_LAUNCH_CHECK()                                                              \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            mvf_set_error("%s:%d: launch -> %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return MVF_EHIP;                                                            \
        }                                                                               \
    } while (0)

// ---- storage types ---------------------------------------------------------------------------
struct bf16_t {
    uint16_t v;
};

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even via the gfx950 packed converter (v_cvt_pk_bf16_f32): one instruction per PAIR of values
typedef __bf16 mvf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mvf_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const mvf_f32x2 v = {lo, hi};
    const mvf_bf16x2 b = __builtin_convertvector(v, mvf_bf16x2);
    uint32_t u;
    __builtin_memcpy(&u, &b, 4);
    return u;
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return bf16_to_f32(p->v); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { p->v = f32_to_bf16(v); }

// 4-element vector access (16 B for f32, 8 B for bf16); pointer must be aligned to the vector size
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    uint2 r = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                       __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    uint2 r;
    r.x = pack_bf16x2(v.x, v.y);
    r.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = r;
}

__device__ __forceinline__ float hswish_f(float u) { return u * fminf(fmaxf(u + 3.0f, 0.0f), 6.0f) * (1.0f / 6.0f); }
// d/du [u * relu6(u+3)/6]; relu6' = 0 at the kinks (torch hardtanh_backward uses strict inequalities)
__device__ __forceinline__ float hswish_grad_f(float u) {
    float r6 = fminf(fmaxf(u + 3.0f, 0.0f), 6.0f);
    float inner = (u > -3.0f && u < 3.0f) ? 1.0f : 0.0f;
    return r6 * (1.0f / 6.0f) + u * inner * (1.0f / 6.0f);
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// [r6] Policy switches: ONE environment variable, MVF_POLICY="name=value,name=value" (integers), is the only thing the library reads from the environment.
// The defaults at the call sites are the measured policy (DESIGN.md section 4.5 lists every name with the measurement behind its default); an override is for
// A/B runs and for the tests that drive both sides of a switch.  Call sites that want the value once wrap the call in a function-local static.
static inline bool mvf_policy_lookup(const char* name, long* out) {
    const char* p = getenv("MVF_POLICY");
    if (!p) return false;
    const size_t n = strlen(name);
    while (*p) {
        while (*p == ',' || *p == ';' || *p == ' ') ++p;
        const char* e = p;
        while (*e && *e != ',' && *e != ';') ++e;
        const char* eq = (const char*)memchr(p, '=', (size_t)(e - p));
        if (eq && (size_t)(eq - p) == n && strncasecmp(p, name, n) == 0) {
            *out = strtol(eq + 1, nullptr, 10);
            return true;
        }
        p = e;
    }
    return false;
}
static inline int mvf_policy_int(const char* name, int dflt) {
    long v;
    return mvf_policy_lookup(name, &v) ? (int)v : dflt;
}
static inline bool mvf_policy_has(const char* name) {
    long v;
    return mvf_policy_lookup(name, &v);
}

// the direct 7x7 stride-2 stem conv (stem_direct.hip), reached from conv_nhwc.hip's conv_fwd_impl when the launch is the stem's
struct StemDirectArgs {
    const void* x;              // padded NHWC4 operand (mvf_stem_prep), (N, H, W, 4) bf16
    const void* w;              // packed weights [64][7][8][4] bf16
    void* y;                    // (N, Ho, Wo, 64) bf16
    const float* bias;          // epi 4
    float* stats_part;          // epi 1: [64][stats_rows][2]
    const float* stats_shift;
    int stats_rows, epi;        // epi 1: + BatchNorm statistics (training), 4: bias + ReLU (inference)
    int N, H, W, Ho, Wo;
    long wK;
};
// the direct 3x3 conv with 64 input / 64 output channels (conv3x3_c64.hip), reached from conv_fwd_impl for layer1's conv2 and its data gradient
struct Conv3x3C64Args {
    const void* x;              // (N, H, W, xps >= 64) bf16, channels [0, 64)
    const void* w;              // packed weights [64][3][3][64] bf16 (forward pack, or the data-gradient pack)
    void* y;                    // (N, H, W, 64) bf16
    const float* bias;          // epi 4
    float* stats_part;          // epi 1 / 6: [64][stats_rows][2]
    const float* stats_shift;   // epi 1
    const void* bn_z;           // epi 6: the pre-activation the gradient's ReLU gate is taken from, (N, H, W, 64)
    const float *bn_mean, *bn_invstd, *bn_scale, *bn_shift;
    int stats_rows, epi;        // epi 1: + BatchNorm statistics, 2: plain, 4: bias + ReLU, 6: + BatchNorm-backward sums
    int N, H, W, xps;
    long wK;
};
// column sums of a pointwise conv that stores nothing (pw_sums.hip): the statistics-only pass and the BatchNorm-backward sums of a z3-free bottleneck
struct PwSumsArgs {
    const void* a;              // (M, xps >= 64) bf16 conv input, channels [0, 64)
    const void* w;              // packed weights [N][64] bf16
    const void* g;              // mode 1: block-output gradient (M, N) bf16
    const unsigned char* bits;  // mode 1: sign bits of the block output, (M, N / 4) bytes
    const float* c0;            // mode 0: statistics shift (or NULL); mode 1: BatchNorm mean
    const float* c1;            // mode 1: BatchNorm invstd
    float* part;                // [N][rows][2]
    int rows, mode, M, N, K, xps;
};
// weight gradient of the 64 -> 64 channel 3x3 stride-1 pad-1 conv as a direct kernel (wgrad3x3_c64.hip)
struct Wgrad3x3C64Args {
    const void* dz;             // (N, H, W, 64) bf16
    const void* x;              // (N, H, W, xps >= 64) bf16, channels [0, 64)
    float* part;                // [nwg][64][576] fp32 partial slabs (wgrad_reduce_kernel's layout: [co][(kh * 3 + kw) * 64 + ci])
    int N, H, W, xps, nwg;
};
// weight gradient of the stem conv (7 x 1 taps over the 32-"channel" view of the padded NHWC4 operand, stride 2) as a direct kernel (wgrad_stem.hip)
struct WgradStemArgs {
    const void* dz;             // (N, Ho, Wo, 64) bf16
    const void* x;              // (N, H, W, 4) bf16, padded (mvf_stem_prep)
    float* part;                // [nwg][64][224] fp32 partial slabs (wgrad_reduce_kernel's layout: [co][(kh * 8 + kw) * 4 + c])
    int N, H, W, Ho, Wo, nwg;
};
namespace mvf_internal {
// BatchNorm backward apply fused with the pointwise conv's weight gradient (bnbwd_wgrad.hip); index 0 / 1 = the one or two BatchNorms
// (1 = the downsample branch of a paired backward) that share g and the sign bits
struct BnBwdWgradArgs {
    const void* g;              // (M, g_pitch >= C) bf16 gradient of the BatchNorm output(s) before the ReLU gate
    const void* z[2];           // (M, C) bf16 BatchNorm inputs = conv outputs
    const void* bits;           // mask mode 4: sign bits of the block output, (M, C / 4) bytes
    const float *gamma[2], *mean[2], *invstd[2], *dgamma[2], *dbeta[2];
    const float *scale, *shift; // mask mode 2: folded coefficients of BatchNorm 0 (gate = scale * z + shift > 0)
    void* dz[2];                // (M, C) bf16 out
    const void* x[2];           // (M, xps >= K) bf16 conv inputs, channels [0, K); x[1] may be NULL (that conv is not contracted here)
    int xps[2];
    float* part[2];             // partial slabs [nsplit][C][K] fp32
    int g_pitch, M, C, K;
    int rows_per_split, nsplit, ctiles, ktiles;
};
// the whole backward of a z3-free bottleneck's last conv in one pass (pw_bwd_fused.hip): recomputed conv + BatchNorm-backward apply + data gradient
// (+ the next BatchNorm's backward sums) + weight gradient; 64 -> 256 channels, bf16
struct PwBwdFusedArgs {
    const void* a;              // (M, aps >= 64) bf16 conv input, channels [0, 64)
    const void* w;              // packed weights [256][64] bf16
    const void* g;              // (M, g_pitch >= 256) bf16 gradient of the block output
    const unsigned char* bits;  // sign bits of the block output, (M, 64) bytes
    const float *gamma, *mean, *invstd, *dgamma, *dbeta;       // the BatchNorm behind the conv (dgamma / dbeta final)
    const void* z_in;           // (M, 64) bf16: the pre-activation the conv input was made from (a = relu(bn_in(z_in)))
    const float *in_scale, *in_shift, *in_mean, *in_invstd;    // that BatchNorm
    void* dx;                   // (M, 64) bf16 out: gradient of the conv input
    float* sums_part;           // [64][sums_rows][2] out: partial rows of bn_in's backward sums (2 x nsplit rows)
    float* part;                // [nsplit][256][64] fp32 out: weight-gradient slabs
    int aps, g_pitch, M, sums_rows, rows_per_split, nsplit;
    int ablate;                 // -Dpolicy pwbf_ablate builds only: phases to skip (timing experiments)
};
// BatchNorm-backward sums of both branches of a z3-free downsample bottleneck in one pass over g (pw_sums_pair.hip); splits as pw_bwd_fused_plan
struct PwSumsPairArgs {
    const void *a, *x;          // (M, aps / xps >= 64) bf16 inputs of conv a (conv3) / conv b (the downsample conv), channels [0, 64)
    const void *w_a, *w_b;      // packed weights [256][64] bf16
    const void* g;              // (M, g_pitch >= 256) bf16
    const unsigned char* bits;  // (M, 64) bytes
    const float *mean_a, *invstd_a, *mean_b, *invstd_b;
    float *part_a, *part_b;     // [256][rows][2] out, rows = 2 x nsplit
    int aps, xps, g_pitch, M, rows, rows_per_split, nsplit;
};
int pw_sums_pair_launch(const PwSumsPairArgs& a, hipStream_t st);
int pw_bwd_fused_plan(long m, int c, int k, int* rows_per_split);
int pw_bwd_fused_launch(const PwBwdFusedArgs& a, hipStream_t st);
bool bnbwd_wgrad_tile(int c, int k, int nbn, int mask_mode, int* ct, int* kt);
int bnbwd_wgrad_plan(long m, int c, int k, int nbn, int mask_mode, int* rows_per_split, int* ctiles, int* ktiles);
int bnbwd_wgrad_launch(const BnBwdWgradArgs& a, int nbn, int mask_mode, hipStream_t st);
int wgrad_slab_reduce_launch(const float* part, int nsplit, int cout, int k, float* dw_oihw, hipStream_t st);
bool wgrad3x3_c64_ok(int n, int h, int w, int xps);
int wgrad3x3_c64_wgs(int n, int h);
int wgrad3x3_c64_launch(const Wgrad3x3C64Args& a, hipStream_t st);
bool wgrad_stem_ok(int n, int h, int w, int ho, int wo);
int wgrad_stem_wgs(int n, int ho);
int wgrad_stem_launch(const WgradStemArgs& a, hipStream_t st);
int stem_direct_launch(const StemDirectArgs& a, hipStream_t st);
int conv3x3_c64_launch(const Conv3x3C64Args& a, hipStream_t st);
int pw_sums_launch(const PwSumsArgs& a, hipStream_t st);
}

#ifdef __HIPCC__
// LDS-DMA issued from inline asm: hipcc counts a builtin LDS-DMA as a pending LDS write and drains it (vmcnt(0)) before the
// next ds_read -- exactly the overlap this variant exists for -- so the statement is hidden from its bookkeeping and the loop
// waits for it explicitly (cdna_hip_programming.md, "What hipcc does not do" item 1).  M0 = wave-uniform LDS byte address.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 rsrc_words(const void* base, unsigned nbytes) {
    const unsigned long p = (unsigned long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)nbytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void glds16(const i32x4 rs, const unsigned lds_dst, const unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}

#endif  // __HIPCC__


