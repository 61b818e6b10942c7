// BatchNorm backward of a bottleneck's bn3 WITHOUT the dz3 tensor (reference codes/models/backbones/resnet.py:229-244: out = conv3(a2); out = norm3(out);
// out += identity; out = relu(out) -- and torch autograd's backward of it), for the stages where dz3 is the widest tensor of the block
// (planes >= 256: [pixels][4 planes] bf16, written once and read by the data gradient and by the weight gradient).
//
// The BatchNorm backward is AFFINE in (gm, z3):  dz3 = a (gm - d0 - kx (z3 - mu)),  a = gamma invstd, d0 = dbeta / M, kx = invstd dgamma / M,
// and z3 = a2 W^T is itself linear in the conv input, so both consumers of dz3 can be written on the tensors that already exist:
//   data gradient    da2 = dz3 W        = gm (a.W) - a2 G - v        G = W^T diag(a kx) W  (K x K),  v_k = sum_c a_c (d0_c - kx_c mu_c) W[c][k]
//   weight gradient  dW  = dz3^T a2     = a . (Q - d0 (x) sa - kx . (W A2 - mu (x) sa))     Q = gm^T a2,  A2 = a2^T a2 (K x K),  sa = sum_m a2[m]
// i.e. ONE GEMM over the concatenated operand [gm | a2] (K' = C + K) with the weights [a.W ; -G] and a bias, and the weight-gradient GEMM on gm
// followed by a small correction -- no pass that forms dz3 (reads g, z3, the sign bits: 212 MB per layer3 block; writes 103 MB) and no dz3
// reads (2 x 103 MB).  The two kernels here are the glue: (1) the data gradient's operand [a.W ; -G] in the data-gradient pack's layout, and its
// bias; (2) the weight gradient's correction.  gm = g * [out > 0] arrives materialised (the producing data gradient gates it:
// mvf_conv2d_nhwc_fwd_resmask_gate / mvf_nhwc_stencil_gate), dgamma / dbeta come from the usual sums pass (mvf_bn_bwd_reduce on gm and z3).
// Rounding: a.W and G are rounded to bf16 once (dz3 used to be rounded per element); the accumulation is fp32 as before.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct PrepArgs {
    const uint16_t* wd;         // [K][C] bf16: the data-gradient pack of the conv's weights, wd[k][c] = W[c][k]
    const float *gamma, *mean, *invstd, *dgamma, *dbeta;
    uint16_t* bd;               // [K][C + K] bf16: bd[k][c] = a_c W[c][k], bd[k][C + j] = -G[j][k]
    float* bias;                // [K]: -v
    int C, K;
    float inv_m;
    int g_wgs, s_wgs;           // workgroups of the G part / of the scaling part (the bias part follows)
};

__device__ __forceinline__ float bf16lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// Every workgroup first folds the per-channel coefficients into LDS (3 x C floats): a = gamma invstd, e = a kx, t = a (d0 - kx mu).
__global__ __launch_bounds__(256) void dzfree_prep_kernel(PrepArgs p) {
    extern __shared__ __attribute__((aligned(16))) float coef[];      // [3][C] + [4][16][64] (the G workgroups' partial blocks)
    const int C = p.C, K = p.K, tid = threadIdx.x;
    float* ca = coef;
    float* ce = coef + C;
    float* ct = coef + 2 * C;
    for (int c = tid; c < C; c += 256) {
        const float r = p.invstd[c], a = p.gamma[c] * r, d0 = p.dbeta[c] * p.inv_m, kx = r * p.dgamma[c] * p.inv_m;
        ca[c] = a;
        ce[c] = a * kx;
        ct[c] = a * (d0 - kx * p.mean[c]);
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    if (b < p.g_wgs) {
        // ---- G: one WORKGROUP per 32 x 32 block, G[j][k] = sum_c e_c W[c][j] W[c][k]; the channel range is cut into four quarters, one per wave (a
        // wave's trip count -- C / 256 -- is what bounds this launch: it sits on the critical path between bn3's sums and the data gradient), the
        // four partial blocks meet in LDS; operands straight from wd (rows of wd are contiguous in c), the next trip's loads issued ahead of the matrix
        // instructions
        const int kb = K / 32;
        const int j0 = (b / kb) * 32, k0 = (b % kb) * 32;
        const int row = lane & 31, half = lane >> 5;
        const int cq = C / 4, cbeg = wave * cq;
        const uint16_t* pa = p.wd + (long)(j0 + row) * C + half * 8;
        const uint16_t* pb = p.wd + (long)(k0 + row) * C + half * 8;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        uint4 va[4], vb[4], na[4], nb[4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            va[s_] = *reinterpret_cast<const uint4*>(pa + cbeg + s_ * 16);
            vb[s_] = *reinterpret_cast<const uint4*>(pb + cbeg + s_ * 16);
        }
        for (int c0 = cbeg; c0 < cbeg + cq; c0 += 64) {
            const bool more = c0 + 64 < cbeg + cq;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                na[s_] = more ? *reinterpret_cast<const uint4*>(pa + c0 + 64 + s_ * 16) : make_uint4(0, 0, 0, 0);
                nb[s_] = more ? *reinterpret_cast<const uint4*>(pb + c0 + 64 + s_ * 16) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const float4 e0 = *reinterpret_cast<const float4*>(ce + c0 + s_ * 16 + half * 8), e1 = *reinterpret_cast<const float4*>(ce + c0 + s_ * 16 + half * 8 + 4);
                uint4 sa;
                sa.x = pack_bf16x2(bf16lo(va[s_].x) * e0.x, bf16hi(va[s_].x) * e0.y);
                sa.y = pack_bf16x2(bf16lo(va[s_].y) * e0.z, bf16hi(va[s_].y) * e0.w);
                sa.z = pack_bf16x2(bf16lo(va[s_].z) * e1.x, bf16hi(va[s_].z) * e1.y);
                sa.w = pack_bf16x2(bf16lo(va[s_].w) * e1.z, bf16hi(va[s_].w) * e1.w);
                bf16x8 fa, fb;
                __builtin_memcpy(&fa, &sa, 16);
                __builtin_memcpy(&fb, &vb[s_], 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);     // D[row j = 8 (r >> 2) + 4 half + (r & 3)][col k = lane & 31]
            }
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) { va[s_] = na[s_]; vb[s_] = nb[s_]; }
        }
        float* red = coef + 3 * C;                                   // [4 waves][16 registers][64 lanes]
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
        __syncthreads();
        // wave q adds up register quad q (rows j0 + 8 q + 4 half + 0..3) of the four partial blocks, in wave order, and stores it
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (red[(0 * 16 + 4 * wave + i) * 64 + lane] + red[(1 * 16 + 4 * wave + i) * 64 + lane]) +
                                           (red[(2 * 16 + 4 * wave + i) * 64 + lane] + red[(3 * 16 + 4 * wave + i) * 64 + lane]);
        uint2 pk;
        pk.x = pack_bf16x2(-v[0], -v[1]);
        pk.y = pack_bf16x2(-v[2], -v[3]);
        *reinterpret_cast<uint2*>(p.bd + (long)(k0 + row) * (C + K) + C + j0 + 4 * half + 8 * wave) = pk;
        return;
    }
    b -= p.g_wgs;
    if (b < p.s_wgs) {
        // ---- bd[k][c] = a_c wd[k][c]: 8 channels per thread ----
        const long u = (long)b * 256 + tid, per_row = C / 8;
        if (u >= (long)K * per_row) return;
        const int k = (int)(u / per_row), c = (int)(u - (long)k * per_row) * 8;
        const uint4 v = *reinterpret_cast<const uint4*>(p.wd + (long)k * C + c);
        const float4 a0 = *reinterpret_cast<const float4*>(ca + c), a1 = *reinterpret_cast<const float4*>(ca + c + 4);
        uint4 o;
        o.x = pack_bf16x2(bf16lo(v.x) * a0.x, bf16hi(v.x) * a0.y);
        o.y = pack_bf16x2(bf16lo(v.y) * a0.z, bf16hi(v.y) * a0.w);
        o.z = pack_bf16x2(bf16lo(v.z) * a1.x, bf16hi(v.z) * a1.y);
        o.w = pack_bf16x2(bf16lo(v.w) * a1.z, bf16hi(v.w) * a1.w);
        *reinterpret_cast<uint4*>(p.bd + (long)k * (C + K) + c) = o;
        return;
    }
    b -= p.s_wgs;
    {
        // ---- bias_k = -sum_c t_c wd[k][c]: one wave per k, fixed order (lane-strided partial sums, butterfly) ----
        const int k = b * 4 + wave;
        if (k >= K) return;
        float s = 0.f;
        for (int c = lane * 8; c < C; c += 512) {
            const uint4 v = *reinterpret_cast<const uint4*>(p.wd + (long)k * C + c);
            const float4 t0 = *reinterpret_cast<const float4*>(ct + c), t1 = *reinterpret_cast<const float4*>(ct + c + 4);
            s += bf16lo(v.x) * t0.x + bf16hi(v.x) * t0.y + bf16lo(v.y) * t0.z + bf16hi(v.y) * t0.w;
            s += bf16lo(v.z) * t1.x + bf16hi(v.z) * t1.y + bf16lo(v.w) * t1.z + bf16hi(v.w) * t1.w;
        }
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) p.bias[k] = -s;
    }
}

// dW[c][k] = a_c (Q[c][k] - d0_c sa_k - kx_c ((W A2)[c][k] - mu_c sa_k)), in place on dw (which holds Q).  W = the forward pack [C][K] bf16, A2 [K][K]
// fp32 (symmetric), sa_k = M a_mean[k].  W A2 is a C x K x K product: on the matrix cores, one wave per 32 x 32 block of dW, A2's fp32 values split
// exactly into two bf16 terms (hi + lo: 16 mantissa bits; W is bf16 already) -> two matrix instructions per 16 contraction indices, fp32 accumulate.
// Both operand fragments are contiguous rows: W[c][j .. j + 8] and, A2 being symmetric, A2[k][j .. j + 8].  (Two earlier forms on the vector ALUs --
// a thread per column with A2 read straight from L2, then A2 streamed through LDS by every workgroup -- took 75 us per conv on average: the second one
// moved workgroups x K^2 x 4 bytes through L2, 1 GB for layer4's 2048 x 512 conv.)
__global__ __launch_bounds__(256) void dzfree_wgrad_fix_kernel(float* dw, const uint16_t* w, const float* gram, const float* a_mean, const float* gamma,
                                                               const float* mean, const float* invstd, const float* dgamma, const float* dbeta, int C, int K,
                                                               float m_f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kb = K / 32, blk = blockIdx.x * 4 + wave;
    if (blk >= (C / 32) * kb) return;
    const int c0 = (blk / kb) * 32, k0 = (blk % kb) * 32;
    const int row = lane & 31, half = lane >> 5;
    const uint16_t* pw = w + (long)(c0 + row) * K + half * 8;
    const float* pg = gram + (long)(k0 + row) * K + half * 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int U = 4;                                       // contraction steps per trip: their 12 loads are issued together
    for (int j = 0; j < K; j += 16 * U) {
        uint4 vw[U];
        float4 g0[U], g1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vw[u] = *reinterpret_cast<const uint4*>(pw + j + 16 * u);
            g0[u] = *reinterpret_cast<const float4*>(pg + j + 16 * u);
            g1[u] = *reinterpret_cast<const float4*>(pg + j + 16 * u + 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint4 hi, lo;
            hi.x = pack_bf16x2(g0[u].x, g0[u].y); hi.y = pack_bf16x2(g0[u].z, g0[u].w);
            hi.z = pack_bf16x2(g1[u].x, g1[u].y); hi.w = pack_bf16x2(g1[u].z, g1[u].w);
            lo.x = pack_bf16x2(g0[u].x - bf16lo(hi.x), g0[u].y - bf16hi(hi.x)); lo.y = pack_bf16x2(g0[u].z - bf16lo(hi.y), g0[u].w - bf16hi(hi.y));
            lo.z = pack_bf16x2(g1[u].x - bf16lo(hi.z), g1[u].y - bf16hi(hi.z)); lo.w = pack_bf16x2(g1[u].z - bf16lo(hi.w), g1[u].w - bf16hi(hi.w));
            bf16x8 fw, fh, fl;
            __builtin_memcpy(&fw, &vw[u], 16);
            __builtin_memcpy(&fh, &hi, 16);
            __builtin_memcpy(&fl, &lo, 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fh, acc, 0, 0, 0);       // D[row c = 8 (r >> 2) + 4 half + (r & 3)][col k = lane & 31]
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fl, acc, 0, 0, 0);
        }
    }
    const float inv_m = 1.0f / m_f;
    const int k = k0 + row;
    const float sa = a_mean[k] * m_f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = c0 + 8 * (r >> 2) + 4 * half + (r & 3);
        const float rs = invstd[c], a = gamma[c] * rs, d0 = dbeta[c] * inv_m, kx = rs * dgamma[c] * inv_m;
        const long i = (long)c * K + k;
        dw[i] = a * (dw[i] - d0 * sa - kx * (acc[r] - mean[c] * sa));
    }
}

// [r5] bn's backward sums WITHOUT a pass over (gm, z): z = a_in W^T, so sum_m gm[m][c] z[m][c] = sum_k W[c][k] Q[c][k] with Q = gm^T a_in -- the weight-gradient
// GEMM the dz-free path computes anyway (now BEFORE the data gradient, on the launch stream) -- and sum_m gm[m][c] comes from the column sums the kernels that
// STORED gm took in their epilogues (two runs of per-workgroup partials: channels < c_split from the MVF stencil, the rest from the conv).  One wave per channel.
__global__ __launch_bounds__(256) void dzfree_sums_kernel(const float* q, const uint16_t* w, int C, int K, const float* mean,
                                                          const float* invstd, const float* part_lo, int rows_lo, int c_split, const float* part_hi, int rows_hi,
                                                          float* dgamma, float* dbeta) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const bool lo = c < c_split;
    const int rows = lo ? rows_lo : rows_hi;
    const float2* p = reinterpret_cast<const float2*>(lo ? part_lo : part_hi) + (long)c * rows;
    float s1 = 0.f, dot = 0.f;
    for (int r = lane; r < rows; r += 64) s1 += p[r].x;
    const uint16_t* pw = w + (long)c * K;
    const float* pq = q + (long)c * K;
    for (int k = lane * 4; k < K; k += 256) {
        const uint2 wv = *reinterpret_cast<const uint2*>(pw + k);
        const float4 qv = *reinterpret_cast<const float4*>(pq + k);
        dot += bf16lo(wv.x) * qv.x + bf16hi(wv.x) * qv.y + bf16lo(wv.y) * qv.z + bf16hi(wv.y) * qv.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {          // fixed order
        s1 += __shfl_xor(s1, o, 64);
        dot += __shfl_xor(dot, o, 64);
    }
    if (lane == 0) {
        dbeta[c] = s1;
        dgamma[c] = invstd[c] * (dot - mean[c] * s1);
    }
}

// [r5] BatchNorm batch statistics of z = a W^T (a pointwise conv's output) WITHOUT the conv: sum_m z[m][c] = M W[c].abar and
// sum_m (z[m][c] - mean_c)^2 = M W[c] Gc W[c]^T with Gc = A2 / M - abar abar^T, A2 = a^T a the Gram matrix of the conv's (narrow) input and abar its column
// means.  One wave per 32 channels: T = W Gc on the matrix cores (Gc centred element by element in fp32, then split into three bf16 terms: 24 bits), the row-wise
// dot with W and the reduction over k by shuffles; then what bn_stats_finalize_kernel does (train_ops.hip): save_mean / save_invstd / scale / shift and the
// running statistics (unbiased variance).  The statistics are those of the UNROUNDED z (the stored-z path measures the bf16-rounded tensor: 1e-6 apart).
struct GramStatsArgs {
    const float* gram; const float* a_mean; const uint16_t* w; const float* gamma; const float* beta;
    float* running_mean; float* running_var; float* save_mean; float* save_invstd; float* scale; float* shift;
    int C, K; float inv_m, eps, momentum, unbias;
};
__global__ __launch_bounds__(256) void gram_stats_kernel(GramStatsArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, K = p.K;
    const int blk = blockIdx.x * 4 + wave;
    if (blk >= p.C / 32) return;
    const int c0 = blk * 32, row = lane & 31, half = lane >> 5;
    const uint16_t* pw = p.w + (long)(c0 + row) * K + half * 8;
    float qacc[16], macc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) qacc[r] = macc[r] = 0.f;
    for (int kb = 0; kb < K / 32; ++kb) {
        const int k = kb * 32 + row;
        const float* pg = p.gram + (long)k * K + half * 8;
        const float* pa = p.a_mean + half * 8;
        const float ak = p.a_mean[k];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int j = 0; j < K; j += 16) {
            const uint4 vw = *reinterpret_cast<const uint4*>(pw + j);
            const float4 g0 = *reinterpret_cast<const float4*>(pg + j), g1 = *reinterpret_cast<const float4*>(pg + j + 4);
            const float4 a0 = *reinterpret_cast<const float4*>(pa + j), a1 = *reinterpret_cast<const float4*>(pa + j + 4);
            float gc[8] = {g0.x * p.inv_m - ak * a0.x, g0.y * p.inv_m - ak * a0.y, g0.z * p.inv_m - ak * a0.z, g0.w * p.inv_m - ak * a0.w,
                           g1.x * p.inv_m - ak * a1.x, g1.y * p.inv_m - ak * a1.y, g1.z * p.inv_m - ak * a1.z, g1.w * p.inv_m - ak * a1.w};
            unsigned hi[4], mid[4], lo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x0 = gc[2 * q], x1 = gc[2 * q + 1];
                hi[q] = pack_bf16x2(x0, x1);
                const float r0 = x0 - bf16lo(hi[q]), r1 = x1 - bf16hi(hi[q]);
                mid[q] = pack_bf16x2(r0, r1);
                lo[q] = pack_bf16x2(r0 - bf16lo(mid[q]), r1 - bf16hi(mid[q]));
            }
            bf16x8 fw, fh, fm, fl;
            __builtin_memcpy(&fw, &vw, 16);
            __builtin_memcpy(&fh, hi, 16);
            __builtin_memcpy(&fm, mid, 16);
            __builtin_memcpy(&fl, lo, 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fh, acc, 0, 0, 0);       // D[row c = 8 (r >> 2) + 4 half + (r & 3)][col k = lane & 31]
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fl, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + 8 * (r >> 2) + 4 * half + (r & 3);
            const float wv = __uint_as_float((unsigned)p.w[(long)c * K + k] << 16);
            qacc[r] += acc[r] * wv;
            macc[r] += wv * ak;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {          // over the 32 lanes (k columns) of this half: fixed order
            qacc[r] += __shfl_xor(qacc[r], o, 64);
            macc[r] += __shfl_xor(macc[r], o, 64);
        }
    }
    if (row == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + 8 * (r >> 2) + 4 * half + (r & 3);
            const float mean = macc[r], var = fmaxf(qacc[r], 0.f);
            const float invstd = 1.0f / sqrtf(var + p.eps);
            p.save_mean[c] = mean;
            p.save_invstd[c] = invstd;
            const float s = p.gamma[c] * invstd;
            p.scale[c] = s;
            p.shift[c] = p.beta[c] - mean * s;
            if (p.running_mean) p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * mean;
            if (p.running_var) p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (var * p.unbias);
        }
    }
}

}  // namespace

extern "C" {

int mvf_bn_train_stats_gram(const float* gram, const float* a_mean, const void* w_packed, long m, int c, int k, const float* gamma, const float* beta, float eps,
                            float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* scale, float* shift,
                            int dtype, void* stream) {
    MVF_REQUIRE(gram && a_mean && w_packed && gamma && beta && save_mean && save_invstd && scale && shift && m > 0, MVF_EINVAL, "bn_train_stats_gram: NULL argument");
    MVF_REQUIRE(dtype == MVF_BF16, MVF_EUNSUPPORTED, "bn_train_stats_gram: bf16 storage only");
    MVF_REQUIRE(c > 0 && k > 0 && c % 32 == 0 && k % 32 == 0, MVF_ESHAPE, "bn_train_stats_gram: c=%d and k=%d must be multiples of 32", c, k);
    MVF_REQUIRE(((uintptr_t)gram | (uintptr_t)a_mean | (uintptr_t)w_packed) % 16 == 0, MVF_EINVAL, "bn_train_stats_gram: operands must be 16-byte aligned");
    GramStatsArgs p = {gram, a_mean, (const uint16_t*)w_packed, gamma, beta, running_mean, running_var, save_mean, save_invstd, scale, shift,
                       c, k, 1.0f / (float)m, eps, momentum, (float)((double)m / (double)(m > 1 ? m - 1 : 1))};
    hipLaunchKernelGGL(gram_stats_kernel, dim3((c / 32 + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_bn_bwd_dzfree_sums(const float* q, const void* w_packed, int c, int k, const float* mean, const float* invstd,
                           const float* part_lo, int rows_lo, int c_split, const float* part_hi, int rows_hi, float* dgamma, float* dbeta, int dtype, void* stream) {
    MVF_REQUIRE(q && w_packed && mean && invstd && dgamma && dbeta, MVF_EINVAL, "bn_bwd_dzfree_sums: NULL argument");
    MVF_REQUIRE(dtype == MVF_BF16, MVF_EUNSUPPORTED, "bn_bwd_dzfree_sums: bf16 storage only");
    MVF_REQUIRE(c > 0 && k > 0 && k % 4 == 0 && c_split >= 0 && c_split <= c, MVF_ESHAPE, "bn_bwd_dzfree_sums: c=%d, k=%d (a multiple of 4), 0 <= c_split=%d <= c", c, k, c_split);
    MVF_REQUIRE((c_split == 0 || (part_lo && rows_lo > 0)) && (c_split == c || (part_hi && rows_hi > 0)), MVF_EINVAL, "bn_bwd_dzfree_sums: a channel range without partial sums");
    MVF_REQUIRE(((uintptr_t)q % 16 == 0) && ((uintptr_t)w_packed % 8 == 0), MVF_EINVAL, "bn_bwd_dzfree_sums: q must be 16-byte, w_packed 8-byte aligned");
    hipLaunchKernelGGL(dzfree_sums_kernel, dim3((c + 3) / 4), dim3(256), 0, (hipStream_t)stream, q, (const uint16_t*)w_packed, c, k, mean, invstd, part_lo, rows_lo,
                       c_split, part_hi, rows_hi, dgamma, dbeta);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_bn_bwd_dzfree_prep(const void* w_packed_dgrad, int c, int k, const float* gamma, const float* mean, const float* invstd, const float* dgamma,
                           const float* dbeta, long m, void* w_out, float* bias_out, int dtype, void* stream) {
    MVF_REQUIRE(w_packed_dgrad && gamma && mean && invstd && dgamma && dbeta && w_out && bias_out && m > 0, MVF_EINVAL, "bn_bwd_dzfree_prep: NULL argument");
    MVF_REQUIRE(dtype == MVF_BF16, MVF_EUNSUPPORTED, "bn_bwd_dzfree_prep: bf16 storage only");
    MVF_REQUIRE(c > 0 && k > 0 && c % 64 == 0 && k % 32 == 0 && c <= 4096, MVF_ESHAPE, "bn_bwd_dzfree_prep: c=%d must be a multiple of 64 (<= 4096), k=%d a multiple of 32", c, k);
    MVF_REQUIRE(((uintptr_t)w_packed_dgrad | (uintptr_t)w_out) % 16 == 0, MVF_EINVAL, "bn_bwd_dzfree_prep: operands must be 16-byte aligned");
    PrepArgs p = {};
    p.wd = (const uint16_t*)w_packed_dgrad; p.gamma = gamma; p.mean = mean; p.invstd = invstd; p.dgamma = dgamma; p.dbeta = dbeta;
    p.bd = (uint16_t*)w_out; p.bias = bias_out; p.C = c; p.K = k; p.inv_m = 1.0f / (float)m;
    MVF_REQUIRE(c % 256 == 0, MVF_ESHAPE, "bn_bwd_dzfree_prep: c=%d must be a multiple of 256 (four waves x 64-channel trips)", c);
    const int kb = k / 32;
    p.g_wgs = kb * kb;
    p.s_wgs = (int)(((long)k * (c / 8) + 255) / 256);
    const int b_wgs = (k + 3) / 4;
    const size_t lds = (size_t)3 * c * sizeof(float) + 4 * 16 * 64 * sizeof(float);
    hipLaunchKernelGGL(dzfree_prep_kernel, dim3(p.g_wgs + p.s_wgs + b_wgs), dim3(256), lds, (hipStream_t)stream, p);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_bn_bwd_dzfree_wgrad(float* dw, const void* w_packed, const float* gram, const float* a_mean, const float* gamma, const float* mean,
                            const float* invstd, const float* dgamma, const float* dbeta, long m, int c, int k, int dtype, void* stream) {
    MVF_REQUIRE(dw && w_packed && gram && a_mean && gamma && mean && invstd && dgamma && dbeta && m > 0, MVF_EINVAL, "bn_bwd_dzfree_wgrad: NULL argument");
    MVF_REQUIRE(dtype == MVF_BF16, MVF_EUNSUPPORTED, "bn_bwd_dzfree_wgrad: bf16 storage only");
    MVF_REQUIRE(c > 0 && c % 32 == 0 && k > 0 && k % 64 == 0, MVF_ESHAPE, "bn_bwd_dzfree_wgrad: c=%d must be a multiple of 32, k=%d of 64", c, k);
    MVF_REQUIRE(((uintptr_t)w_packed | (uintptr_t)gram) % 16 == 0, MVF_EINVAL, "bn_bwd_dzfree_wgrad: w_packed / gram must be 16-byte aligned");
    const int blocks = (c / 32) * (k / 32);
    hipLaunchKernelGGL(dzfree_wgrad_fix_kernel, dim3((blocks + 3) / 4), dim3(256), 0, (hipStream_t)stream, dw, (const uint16_t*)w_packed, gram, a_mean, gamma, mean,
                       invstd, dgamma, dbeta, c, k, (float)m);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // extern "C"
