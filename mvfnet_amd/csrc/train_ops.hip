// Training-mode companions of the conv stack (all HBM-bound, channels-last [M][C] matrices):
// batch-statistics BatchNorm forward/backward, residual/ReLU, max-pool fwd/bwd, TSN head + cross-entropy fwd/bwd,
// clip-norm + SGD-nesterov.  Reference semantics: torch BatchNorm2d defaults (biased var to normalise, unbiased into
// running_var, momentum 0.1; codes/models/common/norm.py:59 sets eps), Bottleneck.forward
// (codes/models/backbones/resnet.py:208-244), TSNClsHead.forward + BaseHead.loss (heads/tsn_clshead.py:71-98,
// heads/base.py:40-45), DistOptimizerHook.after_train_iter (core/dist_utils.py:61-67) with SGD(nesterov)
// (configs/.../mvf_kinetics400_2d_rgb_r50_dense.py:152-154).
//
// Reductions over M are two-stage and atomic-free: per-block column partials [blocks][C][k] in fp32, then a
// finalize kernel that sums them in fp64 in block order -> deterministic run to run.
#include <algorithm>
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kThreads = 256;

struct ColPlan {
    int cqb;     // channel groups (VN elements each) handled by one block (threads along channels)
    int rl;      // row lanes = 256 / cqb
    int gx;      // blocks along channels
    int gy;      // blocks along rows
    int rows;    // rows per block
};

// vn = elements per lane vector (16 B: 4 fp32 or 8 bf16; 8 B: 4 bf16); blocks = total workgroups aimed at
ColPlan col_plan(long M, int C, int vn = 4, int blocks = 2048) {
    ColPlan p;
    const int cq = C / vn;
    int cqb = 1;
    while (cqb < cq && cqb < 64) cqb <<= 1;
    p.cqb = cqb;
    p.rl = kThreads / cqb;
    p.gx = (cq + cqb - 1) / cqb;
    long want = std::max<long>(1, blocks / p.gx);                // ~8 blocks per CU: these loops are latency-bound
    long rows = std::max<long>((M + want - 1) / want, (long)p.rl * 8);
    rows = (rows + p.rl - 1) / p.rl * p.rl;
    p.rows = (int)rows;
    p.gy = (int)((M + rows - 1) / rows);
    return p;
}

// lane vectors: VN elements of ET <-> VN floats.  (float,4) and (bf16,8) are 16-byte accesses, (bf16,4) is 8 bytes.
template <typename ET, int VN>
__device__ __forceinline__ void ldv(const ET* p, float (&f)[VN]) {
    if constexpr (sizeof(ET) == 4) {
        static_assert(VN == 4, "fp32 lane vector is 4 wide");
        const float4 v = *reinterpret_cast<const float4*>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else if constexpr (VN == 4) {
        const uint2 r = *reinterpret_cast<const uint2*>(p);
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    } else {
        static_assert(VN == 8, "bf16 lane vector is 4 or 8 wide");
        const uint4 r = *reinterpret_cast<const uint4*>(p);
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
    }
}
template <typename ET, int VN>
__device__ __forceinline__ void stv(ET* p, const float (&f)[VN]) {
    if constexpr (sizeof(ET) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
    } else if constexpr (VN == 4) {
        uint2 r;
        r.x = pack_bf16x2(f[0], f[1]);
        r.y = pack_bf16x2(f[2], f[3]);
        *reinterpret_cast<uint2*>(p) = r;
    } else {
        uint4 r;
        r.x = pack_bf16x2(f[0], f[1]);
        r.y = pack_bf16x2(f[2], f[3]);
        r.z = pack_bf16x2(f[4], f[5]);
        r.w = pack_bf16x2(f[6], f[7]);
        *reinterpret_cast<uint4*>(p) = r;
    }
}
// per-channel fp32 parameters of the lane's VN channels (null -> fill)
template <int VN>
__device__ __forceinline__ void ldp(const float* p, int c, float (&f)[VN], float fill = 0.f) {
#pragma unroll
    for (int j = 0; j < VN; j += 4) {
        const float4 v = p ? *reinterpret_cast<const float4*>(p + c + j) : make_float4(fill, fill, fill, fill);
        f[j] = v.x; f[j + 1] = v.y; f[j + 2] = v.z; f[j + 3] = v.w;
    }
}

// reduce 2*VN float accumulators over the row lanes of a block (fixed order); result valid in row lane 0 (threads < cqb).
// In-wave butterfly over the lanes that share a channel group (lane stride cqb), then the 4 waves through LDS.
// red: 2*VN*kThreads/64*64 floats are enough (4 waves x cqb<=64 lanes x 2*VN)
template <int VN>
__device__ __forceinline__ void rowlane_reduce(float (&a)[VN], float (&b)[VN], int cqb, int rl, float* red) {
    (void)rl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cq = lane % cqb;
    for (int off = cqb; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            a[j] += __shfl_xor(a[j], off, 64);
            b[j] += __shfl_xor(b[j], off, 64);
        }
    }
    if (lane < cqb) {
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            red[((wave * 2 * VN) + 2 * j) * cqb + cq] = a[j];
            red[((wave * 2 * VN) + 2 * j + 1) * cqb + cq] = b[j];
        }
    }
    __syncthreads();
    if (threadIdx.x < cqb) {
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            float s = 0.f, t = 0.f;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) {
                s += red[((w * 2 * VN) + 2 * j) * cqb + cq];
                t += red[((w * 2 * VN) + 2 * j + 1) * cqb + cq];
            }
            a[j] = s;
            b[j] = t;
        }
    }
}

// ---- BN forward statistics: shifted sums  S1 = sum(z-K), S2 = sum((z-K)^2), K = running_mean (any K is exact) ----
template <typename ET, int VN>
__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const ET* z, long M, int C, const float* kshift, int cqb, int rows,
                                                            float* part) {
    __shared__ float red[2 * VN * kThreads];
    const int rl = kThreads / cqb;
    const int cq = blockIdx.x * cqb + threadIdx.x % cqb, lane_r = threadIdx.x / cqb;
    const bool ok = cq * VN < C;
    const int c = cq * VN;
    float k[VN], s1[VN], s2[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) k[j] = s1[j] = s2[j] = 0.f;
    if (ok && kshift) ldp<VN>(kshift, c, k);
    const long r0 = (long)blockIdx.y * rows, r1 = min(M, r0 + rows);
    if (ok) {
        auto accum = [&](const float (&a)[VN]) {
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                const float d = a[j] - k[j];
                s1[j] += d;
                s2[j] += d * d;
            }
        };
        long r = r0 + lane_r;
        for (; r + 3 * rl < r1; r += 4 * rl) {      // 4 independent loads in flight per lane
            float a0[VN], a1[VN], a2[VN], a3[VN];
            ldv<ET, VN>(z + r * C + c, a0); ldv<ET, VN>(z + (r + rl) * C + c, a1);
            ldv<ET, VN>(z + (r + 2 * rl) * C + c, a2); ldv<ET, VN>(z + (r + 3 * rl) * C + c, a3);
            accum(a0); accum(a1); accum(a2); accum(a3);
        }
        for (; r < r1; r += rl) {
            float a0[VN];
            ldv<ET, VN>(z + r * C + c, a0);
            accum(a0);
        }
    }
    rowlane_reduce<VN>(s1, s2, cqb, rl, red);
    if (ok && lane_r == 0) {
        float* p = part + ((long)blockIdx.y * C + c) * 2;
#pragma unroll
        for (int j = 0; j < VN; ++j) { p[2 * j] = s1[j]; p[2 * j + 1] = s2[j]; }
    }
}

// partial sums [nblk][C][2] -> per-channel (s1, s2) in fp64, fixed order.  A workgroup owns kFinCh = 4 channels (C % 4 == 0); ALL 256
// threads are row lanes, each reading the 32 contiguous bytes (4 channels x 2 sums) of its rows with U rows in flight: the loop is
// pure L2 latency, and layer1's 3136-6272 partial rows are 2-4 trips of 2048 rows (12 trips of 512 with 64 row lanes x 4 channel
// lanes and 8-byte loads: finalize kernels 9-12 us -> see DESIGN 4.2).  Then a butterfly over the wave and the 4 waves through LDS.
constexpr int kFinCh = 4;
__device__ __forceinline__ bool reduce_partials_at(const float* part, int C, int nblk, int blk, int& c, double& s1, double& s2) {
    constexpr int CH = kFinCh, U = 4;
    __shared__ double sh[4][2 * CH];
    const int c0 = blk * CH, t = threadIdx.x;
    double acc[2][2 * CH];
#pragma unroll
    for (int j = 0; j < 2 * CH; ++j) acc[0][j] = acc[1][j] = 0.0;
    int k = t;
    for (; k + (U - 1) * 256 < nblk; k += U * 256) {
        float4 v[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4* p = reinterpret_cast<const float4*>(part + ((long)(k + u * 256) * C + c0) * 2);
            v[u][0] = p[0];
            v[u][1] = p[1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double* a = acc[u & 1];
            a[0] += v[u][0].x; a[1] += v[u][0].y; a[2] += v[u][0].z; a[3] += v[u][0].w;
            a[4] += v[u][1].x; a[5] += v[u][1].y; a[6] += v[u][1].z; a[7] += v[u][1].w;
        }
    }
    for (; k < nblk; k += 256) {
        const float4* p = reinterpret_cast<const float4*>(part + ((long)k * C + c0) * 2);
        const float4 v0 = p[0], v1 = p[1];
        double* a = acc[0];
        a[0] += v0.x; a[1] += v0.y; a[2] += v0.z; a[3] += v0.w;
        a[4] += v1.x; a[5] += v1.y; a[6] += v1.z; a[7] += v1.w;
    }
    double r[2 * CH];
#pragma unroll
    for (int j = 0; j < 2 * CH; ++j) {
        r[j] = acc[0][j] + acc[1][j];
        for (int off = 1; off < 64; off <<= 1) r[j] += __shfl_xor(r[j], off, 64);
    }
    if ((t & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 2 * CH; ++j) sh[t >> 6][j] = r[j];
    }
    __syncthreads();
    c = c0 + t;
    if (t >= CH || c >= C) return false;
    s1 = ((sh[0][2 * t] + sh[1][2 * t]) + (sh[2][2 * t] + sh[3][2 * t]));
    s2 = ((sh[0][2 * t + 1] + sh[1][2 * t + 1]) + (sh[2][2 * t + 1] + sh[3][2 * t + 1]));
    return true;
}
__device__ __forceinline__ bool reduce_partials(const float* part, int C, int nblk, int& c, double& s1, double& s2) {
    return reduce_partials_at(part, C, nblk, blockIdx.x, c, s1, s2);
}

// CHANNEL-MAJOR partial sums [C][nblk][2] (what the conv epilogues write: mvf_conv2d_nhwc_fwd_stats / _dgrad_bnsums): a channel's
// rows are contiguous, so the reads are whole cache lines (row-major: every 32-byte piece sits in its own 128-byte line, and
// layer1's 6272 rows made a 16-workgroup finalize a 20-70 us kernel).  MODE 1: one workgroup per channel (256 row lanes, long row
// counts); MODE 2: one WAVE per channel, 4 channels per workgroup.  Fixed order either way.
template <int MODE>
__device__ __forceinline__ bool reduce_partials_cm(const float* part, int C, int nblk, int& c, double& s1, double& s2) {
    constexpr int U = 8;
    constexpr int LN = MODE == 1 ? 256 : 64;
    __shared__ double sh[4][2];
    const int t = threadIdx.x, lane = MODE == 1 ? t : (t & 63);
    c = MODE == 1 ? (int)blockIdx.x : (int)blockIdx.x * 4 + (t >> 6);
    double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0;
    if (c < C) {
        const float2* p = reinterpret_cast<const float2*>(part) + (long)c * nblk;
        int k = lane;
        for (; k + (U - 1) * LN < nblk; k += U * LN) {
            float2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = p[k + u * LN];
#pragma unroll
            for (int u = 0; u < U; u += 2) { a0 += v[u].x; b0 += v[u].y; a1 += v[u + 1].x; b1 += v[u + 1].y; }
        }
        for (; k < nblk; k += LN) {
            const float2 v = p[k];
            a0 += v.x;
            b0 += v.y;
        }
    }
    double a = a0 + a1, b = b0 + b1;
    for (int off = 1; off < 64; off <<= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
    }
    if constexpr (MODE == 1) {
        if ((t & 63) == 0) { sh[t >> 6][0] = a; sh[t >> 6][1] = b; }
        __syncthreads();
        if (t != 0 || c >= C) return false;
        s1 = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]);
        s2 = (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]);
        return true;
    } else {
        if ((t & 63) != 0 || c >= C) return false;
        s1 = a;
        s2 = b;
        return true;
    }
}

template <int MODE = 0>          // 0: row-major partials [nblk][C][2]; 1 / 2: channel-major [C][nblk][2] (reduce_partials_cm)
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(int C, int nblk, long M, const float* part, const float* gamma, const float* beta,
                                         float eps, float momentum, float* running_mean, float* running_var,
                                         float* save_mean, float* save_invstd, float* scale, float* shift) {
    int c;
    double s1, s2;
    if constexpr (MODE == 0) {
        if (!reduce_partials(part, C, nblk, c, s1, s2)) return;
    } else {
        if (!reduce_partials_cm<MODE>(part, C, nblk, c, s1, s2)) return;
    }
    const double K = running_mean ? (double)running_mean[c] : 0.0;
    const double d = s1 / (double)M;
    const double mean = K + d;
    double var = s2 / (double)M - d * d;
    if (var < 0.0) var = 0.0;
    const float invstd = 1.0f / sqrtf((float)var + eps);
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    const float s = gamma[c] * invstd;
    scale[c] = s;
    shift[c] = beta[c] - (float)mean * s;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * (double)M / (double)(M > 1 ? M - 1 : 1));
}

// [r5] column means from bn_apply_cs_kernel's partial rows ([nblk][C][2], element 0): reduce_partials' fp64 fixed-order sum, nothing else
__global__ __launch_bounds__(256) void colmean_finalize_kernel(int C, int nblk, long M, const float* part, float* mean_out) {
    int c;
    double s1, s2;
    if (!reduce_partials(part, C, nblk, c, s1, s2)) return;
    mean_out[c] = (float)(s1 / (double)M);
}

// ---- BN apply (+ residual) (+ ReLU):  out = act(z*scale + shift [+ r] [+ r*rscale + rshift]) ----
// Column-tiled: a lane owns VN channels (its scale/shift live in registers) and walks rows; no per-element index math.
// CS ([r5], mvf_bn_apply_colmeans): the kernel also leaves the column sums of what it STORES, one partial row per workgroup row band ([gy][C][2], element 0:
// bn_stats_kernel's layout) -- the column means of a2 for the Gram form of bn3's statistics without a pass over a2 (bn_dzfree.hip: gram_stats_kernel).
template <typename ET, int VN, bool CS>
__device__ __forceinline__ void bn_apply_body(const ET* z, long M, int C, const float* scale, const float* shift, const ET* r,
                                              const float* rscale, const float* rshift, int act, ET* out,
                                              unsigned char* bits, int cqb, int rows, float* csum_part, float* red) {
    const int rl = kThreads / cqb;
    const int cq = blockIdx.x * cqb + threadIdx.x % cqb, lane_r = threadIdx.x / cqb;
    const bool ok = cq * VN < C;
    if (!CS && !ok) return;
    const int c = ok ? cq * VN : 0;
    float s[VN], b[VN], rs[VN], rb[VN], cs1[VN], cs2[VN];
    ldp<VN>(scale, c, s);
    ldp<VN>(shift, c, b);
    ldp<VN>(rscale, c, rs, 1.f);
    ldp<VN>(rshift, c, rb, 0.f);
#pragma unroll
    for (int j = 0; j < VN; ++j) cs1[j] = cs2[j] = 0.f;
    const long r0 = (long)blockIdx.y * rows, r1 = ok ? min(M, r0 + rows) : r0;
    auto one = [&](long row, float (&v)[VN], const float (&q)[VN]) {
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            float t = v[j] * s[j] + b[j];
            if (r) t += q[j] * rs[j] + rb[j];
            if (act == 1) t = fmaxf(t, 0.f);
            else if (act == 2) t = hswish_f(t);             // MVF activation (se_module.py:5-24)
            v[j] = t;
            if constexpr (CS) cs1[j] += sizeof(ET) == 2 ? bf16_to_f32(f32_to_bf16(t)) : t;
        }
        stv<ET, VN>(out + row * C + c, v);
        if (bits) {                                         // sign bits of the result, one byte per 4 channels: [M][C/4]
            unsigned mb = 0;
#pragma unroll
            for (int j = 0; j < VN; ++j) mb |= (v[j] > 0.f ? 1u : 0u) << (j + (j >= 4 ? 4 : 0));
            if constexpr (VN == 8) *reinterpret_cast<unsigned short*>(bits + row * (C / 4) + c / 4) = (unsigned short)mb;
            else bits[row * (C / 4) + c / 4] = (unsigned char)mb;
        }
    };
    long row = r0 + lane_r;
    for (; row + rl < r1; row += 2 * rl) {                  // two rows per trip: 2-4 independent loads in flight per lane
        float z0[VN], z1[VN], q0[VN], q1[VN];
        ldv<ET, VN>(z + row * C + c, z0);
        ldv<ET, VN>(z + (row + rl) * C + c, z1);
        if (r) {
            ldv<ET, VN>(r + row * C + c, q0);
            ldv<ET, VN>(r + (row + rl) * C + c, q1);
        }
        one(row, z0, q0);
        one(row + rl, z1, q1);
    }
    for (; row < r1; row += rl) {
        float z0[VN], q0[VN];
        ldv<ET, VN>(z + row * C + c, z0);
        if (r) ldv<ET, VN>(r + row * C + c, q0);
        one(row, z0, q0);
    }
    if constexpr (CS) {
        rowlane_reduce<VN>(cs1, cs2, cqb, rl, red);
        if (ok && lane_r == 0) {
            float* p = csum_part + ((long)blockIdx.y * C + c) * 2;
#pragma unroll
            for (int j = 0; j < VN; ++j) { p[2 * j] = cs1[j]; p[2 * j + 1] = 0.f; }
        }
    }
}
template <typename ET, int VN>
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const ET* z, long M, int C, const float* scale, const float* shift, const ET* r,
                                                            const float* rscale, const float* rshift, int act, ET* out,
                                                            unsigned char* bits, int cqb, int rows) {
    bn_apply_body<ET, VN, false>(z, M, C, scale, shift, r, rscale, rshift, act, out, bits, cqb, rows, nullptr, nullptr);
}
template <typename ET, int VN>
__global__ __launch_bounds__(kThreads) void bn_apply_cs_kernel(const ET* z, long M, int C, const float* scale, const float* shift, int act, ET* out,
                                                               int cqb, int rows, float* csum_part) {
    __shared__ float red[2 * VN * kThreads];
    bn_apply_body<ET, VN, true>(z, M, C, scale, shift, nullptr, nullptr, nullptr, act, out, nullptr, cqb, rows, csum_part, red);
}

// ---- BN backward reductions: gm = g * mask ; sums of gm and gm*xhat.  mask from y>0 (block output, mode 1), from its sign
// ---- bits (mode 4, ymask = the [M][C/4] bytes bn_apply wrote), from scale*z+shift > 0 (recomputed ReLU of this BN's own
// ---- output, mode 2), hard-swish' (mode 3) or none.  Optionally writes gm. ----
// mb: mask bits of the lane's channels in the [M][C/4]-byte layout (bit j of byte j/4 -> bit j + 4*(j/4) of mb), mask_mode 4
template <typename ET, int VN>
__device__ __forceinline__ void bn_mask(float (&gv)[VN], const float (&zv)[VN], const float (&yv)[VN], const float (&sc)[VN],
                                        const float (&sh)[VN], int mask_mode, unsigned mb = 0u) {
#pragma unroll
    for (int j = 0; j < VN; ++j) {
        if (mask_mode == 4) gv[j] = ((mb >> (j + (j >= 4 ? 4 : 0))) & 1u) ? gv[j] : 0.f;
        else if (mask_mode == 1) gv[j] = yv[j] > 0.f ? gv[j] : 0.f;
        else if (mask_mode == 2) gv[j] = (zv[j] * sc[j] + sh[j]) > 0.f ? gv[j] : 0.f;
        else if (mask_mode == 3) gv[j] *= hswish_grad_f(zv[j] * sc[j] + sh[j]);   // MVF: o = hswish(bn(y))
    }
}

template <int VN>
__device__ __forceinline__ unsigned ld_maskbits(const void* bits, long row, int C, int c) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(bits) + row * (C / 4) + c / 4;
    if constexpr (VN == 8) return *reinterpret_cast<const unsigned short*>(p);
    else return *p;
}

// MM >= 0 fixes the mask mode at compile time (the per-element mask code is then a single case instead of a select chain)
template <typename ET, int VN, int MM = -1>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(const ET* g, int g_pitch, const ET* z, const ET* ymask, long M, int C,
                                                                 const float* mean, const float* invstd, const float* scale,
                                                                 const float* shift, int mask_mode_rt, ET* gm_out, int cqb, int rows,
                                                                 float* part) {
    const int mask_mode = MM >= 0 ? MM : mask_mode_rt;
    __shared__ float red[2 * VN * kThreads];
    const int rl = kThreads / cqb;
    const int cq = blockIdx.x * cqb + threadIdx.x % cqb, lane_r = threadIdx.x / cqb;
    const bool ok = cq * VN < C;
    const int c = cq * VN;
    float mu[VN], rs[VN], sc[VN], sh[VN], s1[VN], s2[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) mu[j] = rs[j] = sc[j] = sh[j] = s1[j] = s2[j] = 0.f;
    if (ok) {
        ldp<VN>(mean, c, mu);
        ldp<VN>(invstd, c, rs);
        if (mask_mode >= 2) {
            ldp<VN>(scale, c, sc);
            ldp<VN>(shift, c, sh);
        }
    }
    const long r0 = (long)blockIdx.y * rows, r1 = min(M, r0 + rows);
    if (ok) {
        auto body = [&](long r, float (&gv)[VN], const float (&zv)[VN], const float (&yv)[VN], unsigned mb) {
            bn_mask<ET, VN>(gv, zv, yv, sc, sh, mask_mode, mb);
            if (gm_out) stv<ET, VN>(gm_out + r * C + c, gv);
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                s1[j] += gv[j];
                s2[j] += gv[j] * ((zv[j] - mu[j]) * rs[j]);
            }
        };
        long r = r0 + lane_r;
        for (; r + rl < r1; r += 2 * rl) {          // two rows per trip: 4-6 independent loads in flight per lane
            float g0[VN], g1[VN], z0[VN], z1[VN], y0[VN], y1[VN];
            ldv<ET, VN>(g + r * g_pitch + c, g0); ldv<ET, VN>(g + (r + rl) * g_pitch + c, g1);
            ldv<ET, VN>(z + r * C + c, z0); ldv<ET, VN>(z + (r + rl) * C + c, z1);
            unsigned m0 = 0u, m1 = 0u;
            if (mask_mode == 1) { ldv<ET, VN>(ymask + r * C + c, y0); ldv<ET, VN>(ymask + (r + rl) * C + c, y1); }
            if (mask_mode == 4) { m0 = ld_maskbits<VN>(ymask, r, C, c); m1 = ld_maskbits<VN>(ymask, r + rl, C, c); }
            body(r, g0, z0, y0, m0);
            body(r + rl, g1, z1, y1, m1);
        }
        for (; r < r1; r += rl) {
            float g0[VN], z0[VN], y0[VN];
            ldv<ET, VN>(g + r * g_pitch + c, g0);
            ldv<ET, VN>(z + r * C + c, z0);
            if (mask_mode == 1) ldv<ET, VN>(ymask + r * C + c, y0);
            body(r, g0, z0, y0, mask_mode == 4 ? ld_maskbits<VN>(ymask, r, C, c) : 0u);
        }
    }
    rowlane_reduce<VN>(s1, s2, cqb, rl, red);
    if (ok && lane_r == 0) {
        float* p = part + ((long)blockIdx.y * C + c) * 2;
#pragma unroll
        for (int j = 0; j < VN; ++j) { p[2 * j] = s1[j]; p[2 * j + 1] = s2[j]; }
    }
}

template <int MODE = 0>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(int C, int nblk, const float* part, float* dgamma, float* dbeta) {
    int c;
    double s1, s2;
    if constexpr (MODE == 0) {
        if (!reduce_partials(part, C, nblk, c, s1, s2)) return;
    } else {
        if (!reduce_partials_cm<MODE>(part, C, nblk, c, s1, s2)) return;
    }
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
}

// dz = gamma*invstd * (gm - dbeta/M - xhat*dgamma/M); gm = g*mask recomputed as above (mask_mode 0: g is already masked)
// Column-tiled like bn_apply_kernel: the six per-channel coefficients are folded once per lane into registers.
template <typename ET, int VN, int MM = -1>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const ET* g, int g_pitch, const ET* z, const ET* ymask, long M, int C, const float* gamma,
                                                                const float* mean, const float* invstd, const float* scale,
                                                                const float* shift, const float* dgamma, const float* dbeta,
                                                                int mask_mode_rt, ET* dz, int cqb, int rows) {
    const int mask_mode = MM >= 0 ? MM : mask_mode_rt;
    const int rl = kThreads / cqb;
    const int cq = blockIdx.x * cqb + threadIdx.x % cqb, lane_r = threadIdx.x / cqb;
    if (cq * VN >= C) return;
    const int c = cq * VN;
    const float inv_m = 1.0f / (float)M;
    float a[VN], d0[VN], kx[VN], mu[VN], sc[VN], sh[VN];
    {
        float ga[VN], rs[VN], dg[VN], db[VN];
        ldp<VN>(gamma, c, ga); ldp<VN>(invstd, c, rs); ldp<VN>(dgamma, c, dg); ldp<VN>(dbeta, c, db);
        ldp<VN>(mean, c, mu);
        ldp<VN>(mask_mode >= 2 ? scale : nullptr, c, sc);
        ldp<VN>(mask_mode >= 2 ? shift : nullptr, c, sh);
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            a[j] = ga[j] * rs[j];
            d0[j] = db[j] * inv_m;
            kx[j] = rs[j] * dg[j] * inv_m;
        }
    }
    const long r0 = (long)blockIdx.y * rows, r1 = min(M, r0 + rows);
    auto one = [&](long row, float (&gv)[VN], const float (&zv)[VN]) {
        if (mask_mode == 1) {
            float yv[VN];
            ldv<ET, VN>(ymask + row * C + c, yv);
            bn_mask<ET, VN>(gv, zv, yv, sc, sh, 1);
        } else {
            bn_mask<ET, VN>(gv, zv, zv, sc, sh, mask_mode, mask_mode == 4 ? ld_maskbits<VN>(ymask, row, C, c) : 0u);
        }
#pragma unroll
        for (int j = 0; j < VN; ++j) gv[j] = a[j] * (gv[j] - d0[j] - (zv[j] - mu[j]) * kx[j]);
        stv<ET, VN>(dz + row * C + c, gv);
    };
    long row = r0 + lane_r;
    for (; row + rl < r1; row += 2 * rl) {
        float g0[VN], g1[VN], z0[VN], z1[VN];
        ldv<ET, VN>(g + row * g_pitch + c, g0);
        ldv<ET, VN>(g + (row + rl) * g_pitch + c, g1);
        ldv<ET, VN>(z + row * C + c, z0);
        ldv<ET, VN>(z + (row + rl) * C + c, z1);
        one(row, g0, z0);
        one(row + rl, g1, z1);
    }
    for (; row < r1; row += rl) {
        float g0[VN], z0[VN];
        ldv<ET, VN>(g + row * g_pitch + c, g0);
        ldv<ET, VN>(z + row * C + c, z0);
        one(row, g0, z0);
    }
}

// ---- PAIRED BatchNorm backward for a block with a downsample branch: out = relu(bn3(z3) + bnd(zd)), so both BatchNorms receive the
// ---- SAME masked gradient gm = g * [out > 0] (mask = the sign bits bn_apply wrote).  One kernel reads g and the bits once for both:
// ---- the two separate backward passes move 2 x (g + z + bits [+ dz]) = 10 tensor passes, the pair 8.  Same column plan, same row
// ---- order and the same per-BatchNorm arithmetic as bn_bwd_reduce_kernel<.., 4> / bn_bwd_apply_kernel<.., 4>: bit-identical results.
template <typename ET, int VN>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce2_kernel(const ET* g, int g_pitch, const ET* za, const ET* zb, const void* bits, long M, int C,
                                                                  const float* mean_a, const float* invstd_a, const float* mean_b, const float* invstd_b,
                                                                  int cqb, int rows, float* part_a, float* part_b) {
    __shared__ float red[2 * VN * kThreads];
    const int rl = kThreads / cqb;
    const int cq = blockIdx.x * cqb + threadIdx.x % cqb, lane_r = threadIdx.x / cqb;
    const bool ok = cq * VN < C;
    const int c = cq * VN;
    float mua[VN], rsa[VN], mub[VN], rsb[VN], a1[VN], a2[VN], b1[VN], b2[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) mua[j] = rsa[j] = mub[j] = rsb[j] = a1[j] = a2[j] = b1[j] = b2[j] = 0.f;
    if (ok) {
        ldp<VN>(mean_a, c, mua); ldp<VN>(invstd_a, c, rsa);
        ldp<VN>(mean_b, c, mub); ldp<VN>(invstd_b, c, rsb);
    }
    const long r0 = (long)blockIdx.y * rows, r1 = min(M, r0 + rows);
    if (ok) {
        auto body = [&](float (&gv)[VN], const float (&zav)[VN], const float (&zbv)[VN], unsigned mb) {
            bn_mask<ET, VN>(gv, zav, zav, mua, mua, 4, mb);
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                a1[j] += gv[j];
                a2[j] += gv[j] * ((zav[j] - mua[j]) * rsa[j]);
                b1[j] += gv[j];
                b2[j] += gv[j] * ((zbv[j] - mub[j]) * rsb[j]);
            }
        };
        long r = r0 + lane_r;
        for (; r + rl < r1; r += 2 * rl) {
            float g0[VN], g1[VN], x0[VN], x1[VN], y0[VN], y1[VN];
            ldv<ET, VN>(g + r * g_pitch + c, g0); ldv<ET, VN>(g + (r + rl) * g_pitch + c, g1);
            ldv<ET, VN>(za + r * C + c, x0); ldv<ET, VN>(za + (r + rl) * C + c, x1);
            ldv<ET, VN>(zb + r * C + c, y0); ldv<ET, VN>(zb + (r + rl) * C + c, y1);
            const unsigned m0 = ld_maskbits<VN>(bits, r, C, c), m1 = ld_maskbits<VN>(bits, r + rl, C, c);
            body(g0, x0, y0, m0);
            body(g1, x1, y1, m1);
        }
        for (; r < r1; r += rl) {
            float g0[VN], x0[VN], y0[VN];
            ldv<ET, VN>(g + r * g_pitch + c, g0);
            ldv<ET, VN>(za + r * C + c, x0);
            ldv<ET, VN>(zb + r * C + c, y0);
            body(g0, x0, y0, ld_maskbits<VN>(bits, r, C, c));
        }
    }
    rowlane_reduce<VN>(a1, a2, cqb, rl, red);
    if (ok && lane_r == 0) {
        float* p = part_a + ((long)blockIdx.y * C + c) * 2;
#pragma unroll
        for (int j = 0; j < VN; ++j) { p[2 * j] = a1[j]; p[2 * j + 1] = a2[j]; }
    }
    __syncthreads();
    rowlane_reduce<VN>(b1, b2, cqb, rl, red);
    if (ok && lane_r == 0) {
        float* p = part_b + ((long)blockIdx.y * C + c) * 2;
#pragma unroll
        for (int j = 0; j < VN; ++j) { p[2 * j] = b1[j]; p[2 * j + 1] = b2[j]; }
    }
}

// both BatchNorms' partials in one launch: workgroups [0, gA) finalize a, [gA, 2 gA) finalize b
__global__ __launch_bounds__(256) void bn_bwd_finalize2_kernel(int C, int nblk, const float* part_a, const float* part_b, float* dgamma_a, float* dbeta_a,
                                                               float* dgamma_b, float* dbeta_b) {
    const int gA = (C + kFinCh - 1) / kFinCh;
    const bool second = (int)blockIdx.x >= gA;
    int c;
    double s1, s2;
    if (!reduce_partials_at(second ? part_b : part_a, C, nblk, (int)blockIdx.x - (second ? gA : 0), c, s1, s2)) return;
    (second ? dbeta_b : dbeta_a)[c] = (float)s1;
    (second ? dgamma_b : dgamma_a)[c] = (float)s2;
}

template <typename ET, int VN>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply2_kernel(const ET* g, int g_pitch, const ET* za, const ET* zb, const void* bits, long M, int C,
                                                                 const float* gamma_a, const float* mean_a, const float* invstd_a, const float* dgamma_a, const float* dbeta_a,
                                                                 const float* gamma_b, const float* mean_b, const float* invstd_b, const float* dgamma_b, const float* dbeta_b,
                                                                 ET* dza, ET* dzb, int cqb, int rows) {
    const int rl = kThreads / cqb;
    const int cq = blockIdx.x * cqb + threadIdx.x % cqb, lane_r = threadIdx.x / cqb;
    if (cq * VN >= C) return;
    const int c = cq * VN;
    const float inv_m = 1.0f / (float)M;
    float aa[VN], da[VN], ka[VN], mua[VN], ab[VN], db_[VN], kb[VN], mub[VN];
    {
        float ga[VN], rs[VN], dg[VN], db[VN];
        ldp<VN>(gamma_a, c, ga); ldp<VN>(invstd_a, c, rs); ldp<VN>(dgamma_a, c, dg); ldp<VN>(dbeta_a, c, db); ldp<VN>(mean_a, c, mua);
#pragma unroll
        for (int j = 0; j < VN; ++j) { aa[j] = ga[j] * rs[j]; da[j] = db[j] * inv_m; ka[j] = rs[j] * dg[j] * inv_m; }
        ldp<VN>(gamma_b, c, ga); ldp<VN>(invstd_b, c, rs); ldp<VN>(dgamma_b, c, dg); ldp<VN>(dbeta_b, c, db); ldp<VN>(mean_b, c, mub);
#pragma unroll
        for (int j = 0; j < VN; ++j) { ab[j] = ga[j] * rs[j]; db_[j] = db[j] * inv_m; kb[j] = rs[j] * dg[j] * inv_m; }
    }
    const long r0 = (long)blockIdx.y * rows, r1 = min(M, r0 + rows);
    auto one = [&](long row, float (&gv)[VN], const float (&zav)[VN], const float (&zbv)[VN]) {
        bn_mask<ET, VN>(gv, zav, zav, mua, mua, 4, ld_maskbits<VN>(bits, row, C, c));
        float oa[VN], ob[VN];
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            oa[j] = aa[j] * (gv[j] - da[j] - (zav[j] - mua[j]) * ka[j]);
            ob[j] = ab[j] * (gv[j] - db_[j] - (zbv[j] - mub[j]) * kb[j]);
        }
        stv<ET, VN>(dza + row * C + c, oa);
        stv<ET, VN>(dzb + row * C + c, ob);
    };
    long row = r0 + lane_r;
    for (; row + rl < r1; row += 2 * rl) {
        float g0[VN], g1[VN], x0[VN], x1[VN], y0[VN], y1[VN];
        ldv<ET, VN>(g + row * g_pitch + c, g0);
        ldv<ET, VN>(g + (row + rl) * g_pitch + c, g1);
        ldv<ET, VN>(za + row * C + c, x0);
        ldv<ET, VN>(za + (row + rl) * C + c, x1);
        ldv<ET, VN>(zb + row * C + c, y0);
        ldv<ET, VN>(zb + (row + rl) * C + c, y1);
        one(row, g0, x0, y0);
        one(row + rl, g1, x1, y1);
    }
    for (; row < r1; row += rl) {
        float g0[VN], x0[VN], y0[VN];
        ldv<ET, VN>(g + row * g_pitch + c, g0);
        ldv<ET, VN>(za + row * C + c, x0);
        ldv<ET, VN>(zb + row * C + c, y0);
        one(row, g0, x0, y0);
    }
}

// ---- max-pool 3x3/2 pad 1 over relu(bn(z)) (stem), forward and backward (argmax recomputed: first max in scan order) ----
template <typename ET>
__global__ void maxpool_bn_fwd_kernel(const ET* z, int n, int h, int w, int c, int ho, int wo, const float* scale,
                                      const float* shift, ET* y, unsigned char* amax) {
    const int c4 = c >> 2;
    const long total = (long)n * ho * wo * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, ow, oh, img;
        if (total <= 0x7fffffffL) {
            unsigned t = (unsigned)i;
            cq = (int)(t % (unsigned)c4); t /= (unsigned)c4;
            ow = (int)(t % (unsigned)wo); t /= (unsigned)wo;
            oh = (int)(t % (unsigned)ho);
            img = (int)(t / (unsigned)ho);
        } else {
            cq = (int)(i % c4);
            long t = i / c4;
            ow = (int)(t % wo); t /= wo;
            oh = (int)(t % ho);
            img = (int)(t / ho);
        }
        const float4 s = *reinterpret_cast<const float4*>(scale + cq * 4), b = *reinterpret_cast<const float4*>(shift + cq * 4);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        unsigned am[4] = {0, 0, 0, 0};                    // window position (dy*3+dx) of the FIRST max, as ATen picks it
        // the nine window loads are unconditional (out-of-image taps re-read a clamped pixel and are skipped in the compare):
        // with a branch per tap the compiler waits for every load before issuing the next (nine serial round trips)
        float4 win[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int ih = oh * 2 - 1 + k / 3, iw = ow * 2 - 1 + k % 3;
            const int ihc = min(max(ih, 0), h - 1), iwc = min(max(iw, 0), w - 1);
            win[k] = ld4(z + (((long)img * h + ihc) * w + iwc) * c + cq * 4);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int ih = oh * 2 - 1 + k / 3, iw = ow * 2 - 1 + k % 3;
            const bool in = (unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w;
            float4 v = win[k];
            v.x = fmaxf(v.x * s.x + b.x, 0.f); v.y = fmaxf(v.y * s.y + b.y, 0.f);
            v.z = fmaxf(v.z * s.z + b.z, 0.f); v.w = fmaxf(v.w * s.w + b.w, 0.f);
            if (in && v.x > m.x) { m.x = v.x; am[0] = k; }
            if (in && v.y > m.y) { m.y = v.y; am[1] = k; }
            if (in && v.z > m.z) { m.z = v.z; am[2] = k; }
            if (in && v.w > m.w) { m.w = v.w; am[3] = k; }
        }
        const long oidx = (((long)img * ho + oh) * wo + ow) * c + cq * 4;
        st4(y + oidx, m);
        if (amax) *reinterpret_cast<unsigned*>(amax + oidx) = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
    }
}

// [r3] BLOCK form of the forward (even h and w): a thread owns a 2 x 2 block of POOLED outputs x 16 bytes of channels; their four 3 x 3
// windows share a 5 x 5 patch of z, loaded once as 25 16-byte accesses for 32 outputs where the element form issues nine 8-byte loads per
// 4.  Each output scans its window in the same order with the same strict compare: identical y and argmax.
template <typename ET>
__global__ __launch_bounds__(256) void maxpool_bn_fwd_blk_kernel(const ET* z, int n, int h, int w, int c, int ho, int wo, const float* scale,
                                                                 const float* shift, ET* y, unsigned char* amax) {
    constexpr int VC = 16 / (int)sizeof(ET);
    const int ncg = c / VC, bwn = wo >> 1, bhn = ho >> 1;
    const long total = (long)n * bhn * bwn * ncg;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % ncg);
        long t = i / ncg;
        const int bw = (int)(t % bwn); t /= bwn;
        const int bh = (int)(t % bhn);
        const int img = (int)(t / bhn);
        float sc[VC], sh[VC];
#pragma unroll
        for (int j = 0; j < VC; ++j) { sc[j] = scale[cq * VC + j]; sh[j] = shift[cq * VC + j]; }
        // patch rows 4 bh - 1 .. 4 bh + 3, columns 4 bw - 1 .. 4 bw + 3 (clamped loads; out-of-image taps are skipped in the compare)
        float a[5][5][VC];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int ih = 4 * bh - 1 + r, iw = 4 * bw - 1 + q;
                const int ihc = min(max(ih, 0), h - 1), iwc = min(max(iw, 0), w - 1);
                const uint4 u = *reinterpret_cast<const uint4*>(z + (((long)img * h + ihc) * w + iwc) * c + cq * VC);
                float f[VC];
                if constexpr (sizeof(ET) == 4) {
                    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
                } else {
                    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u); f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
                    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u); f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
                }
#pragma unroll
                for (int j = 0; j < VC; ++j) a[r][q][j] = fmaxf(f[j] * sc[j] + sh[j], 0.f);
            }
#pragma unroll
        for (int p = 0; p < 4; ++p) {                     // pooled output (2 bh + (p >> 1), 2 bw + (p & 1))
            const int oh = 2 * bh + (p >> 1), ow = 2 * bw + (p & 1);
            float m[VC];
            unsigned am[VC];
#pragma unroll
            for (int j = 0; j < VC; ++j) { m[j] = -INFINITY; am[j] = 0; }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int r = 2 * (p >> 1) + k / 3, q = 2 * (p & 1) + k % 3;
                const int ih = oh * 2 - 1 + k / 3, iw = ow * 2 - 1 + k % 3;
                const bool in = (unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w;
#pragma unroll
                for (int j = 0; j < VC; ++j)
                    if (in && a[r][q][j] > m[j]) { m[j] = a[r][q][j]; am[j] = k; }
            }
            const long oidx = (((long)img * ho + oh) * wo + ow) * c + cq * VC;
            uint4 u;
            if constexpr (sizeof(ET) == 4) u = make_uint4(__float_as_uint(m[0]), __float_as_uint(m[1]), __float_as_uint(m[2]), __float_as_uint(m[3]));
            else u = make_uint4(pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]), pack_bf16x2(m[6], m[7]));
            *reinterpret_cast<uint4*>(y + oidx) = u;
            if (amax) {
                if constexpr (VC == 8)
                    *reinterpret_cast<uint2*>(amax + oidx) = make_uint2(am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24), am[4] | (am[5] << 8) | (am[6] << 16) | (am[7] << 24));
                else
                    *reinterpret_cast<unsigned*>(amax + oidx) = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
            }
        }
    }
}

// ga[input pixel] = sum over the (<=4) windows containing it of g[window] * [this pixel is the window's arg-max].
// Gather form (no atomics); the forward stored one byte per pooled element (window position of the first max), so the
// backward reads 4 bytes + one 4-channel gradient per covering window instead of re-evaluating 9 activations.
// Row form: blockIdx.x = (image, input row), threads over (column, 4-channel group) of that row: the only division left is the
// per-thread split of the row position (by c4, a shift for the 64-channel stem), instead of three runtime divisions per element.
template <typename ET>
__global__ void maxpool_bn_bwd_rows_kernel(const unsigned char* amax, const ET* g, int n, int h, int w, int c, int ho, int wo, ET* ga) {
    const int c4 = c >> 2;
    const int row = blockIdx.x, img = row / h, ih = row - img * h;
    const int per = w * c4;
    for (int i = threadIdx.x; i < per; i += blockDim.x) {
        const int iw = i / c4, cq = i - iw * c4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned amv[4];
        float4 gvv[4];
        bool okv[4];
        unsigned mev[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oh = (ih + (k >> 1)) >> 1, ow = (iw + (k & 1)) >> 1;
            okv[k] = oh < ho && ow < wo && ((k >> 1) == 0 || (ih & 1)) && ((k & 1) == 0 || (iw & 1));
            const int ohc = min(oh, ho - 1), owc = min(ow, wo - 1);
            mev[k] = (unsigned)(ih - (oh * 2 - 1)) * 3u + (unsigned)(iw - (ow * 2 - 1));
            const long oidx = (((long)img * ho + ohc) * wo + owc) * c + cq * 4;
            amv[k] = *reinterpret_cast<const unsigned*>(amax + oidx);
            gvv[k] = ld4(g + oidx);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned am = amv[k], me = mev[k];
            if (okv[k] && (am & 255u) == me) acc[0] += gvv[k].x;
            if (okv[k] && ((am >> 8) & 255u) == me) acc[1] += gvv[k].y;
            if (okv[k] && ((am >> 16) & 255u) == me) acc[2] += gvv[k].z;
            if (okv[k] && (am >> 24) == me) acc[3] += gvv[k].w;
        }
        st4(ga + ((long)row * w + iw) * c + cq * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// The row form that ALSO accumulates the backward sums of the BatchNorm under the pool (stem: pool(relu(bn(z)))): a workgroup walks
// kPoolRB input rows; a thread's channel quad is the same for every position it visits (256 % (c/4) == 0), so it keeps the quad's
// scale / shift / mean / invstd and two running sums in registers: gm = ga * [scale*z + shift > 0] (ga as STORED, i.e. rounded),
// sum gm and sum gm * xhat.  One channel-major partial row per workgroup ([c][nblk][2], mvf_bn_bwd_finalize): the separate
// reduce pass over ga and z (2 x 411 MB for the R50 stem at 32 x 8 frames) becomes one extra read of z here.
constexpr int kPoolRB = 8;
template <typename ET>
__global__ __launch_bounds__(256) void maxpool_bn_bwd_rows_sums_kernel(const unsigned char* amax, const ET* g, int n, int h, int w, int c, int ho, int wo, ET* ga,
                                                                       const ET* z, const float* mean, const float* invstd, const float* scale, const float* shift,
                                                                       float* part, int nblk) {
    __shared__ float red[256 * 8];
    const int c4 = c >> 2;
    const int per = w * c4;
    const int cq = threadIdx.x % c4;                   // fixed per thread: 256 % c4 == 0 (host-checked)
    const float4 sc = *reinterpret_cast<const float4*>(scale + cq * 4), sh = *reinterpret_cast<const float4*>(shift + cq * 4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + cq * 4), rs = *reinterpret_cast<const float4*>(invstd + cq * 4);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int row_end = min(n * h, ((int)blockIdx.x + 1) * kPoolRB);
    for (int row = blockIdx.x * kPoolRB; row < row_end; ++row) {
        const int img = row / h, ih = row - img * h;
        for (int i = threadIdx.x; i < per; i += 256) {
            const int iw = i / c4;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            unsigned amv[4];
            float4 gvv[4];
            bool okv[4];
            unsigned mev[4];
            const long zi = ((long)row * w + iw) * c + cq * 4;
            const float4 zv = ld4(z + zi);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int oh = (ih + (k >> 1)) >> 1, ow = (iw + (k & 1)) >> 1;
                okv[k] = oh < ho && ow < wo && ((k >> 1) == 0 || (ih & 1)) && ((k & 1) == 0 || (iw & 1));
                const int ohc = min(oh, ho - 1), owc = min(ow, wo - 1);
                mev[k] = (unsigned)(ih - (oh * 2 - 1)) * 3u + (unsigned)(iw - (ow * 2 - 1));
                const long oidx = (((long)img * ho + ohc) * wo + owc) * c + cq * 4;
                amv[k] = *reinterpret_cast<const unsigned*>(amax + oidx);
                gvv[k] = ld4(g + oidx);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned am = amv[k], me = mev[k];
                if (okv[k] && (am & 255u) == me) acc[0] += gvv[k].x;
                if (okv[k] && ((am >> 8) & 255u) == me) acc[1] += gvv[k].y;
                if (okv[k] && ((am >> 16) & 255u) == me) acc[2] += gvv[k].z;
                if (okv[k] && (am >> 24) == me) acc[3] += gvv[k].w;
            }
            if (ga) st4(ga + zi, make_float4(acc[0], acc[1], acc[2], acc[3]));
            if constexpr (sizeof(ET) == 2) {             // the sums see what the apply pass will read back / re-gather (rounded to storage)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = bf16_to_f32(f32_to_bf16(acc[j]));
            }
            const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
            const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gm = (zz[j] * scv[j] + shv[j]) > 0.f ? acc[j] : 0.f;
                s1[j] += gm;
                s2[j] += gm * ((zz[j] - muv[j]) * rsv[j]);
            }
        }
    }
    // threads t, t + c4, t + 2*c4, ... share the channel quad: fixed-order sum through LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[(2 * j) * 256 + threadIdx.x] = s1[j];
        red[(2 * j + 1) * 256 + threadIdx.x] = s2[j];
    }
    __syncthreads();
    if ((int)threadIdx.x < c4 * 4) {
        const int q = threadIdx.x >> 2, j = threadIdx.x & 3;           // channel q*4 + j
        float a = 0.f, b = 0.f;
        for (int t = q; t < 256; t += c4) {
            a += red[(2 * j) * 256 + t];
            b += red[(2 * j + 1) * 256 + t];
        }
        reinterpret_cast<float2*>(part)[(long)(q * 4 + j) * nblk + blockIdx.x] = make_float2(a, b);
    }
}

// Second half of the stem backward without a materialised ga: dz = gamma*invstd * (gm - dbeta/M - xhat*dgamma/M) with gm = the SAME
// gather as above (rounded to the storage type, as the sums saw it) masked by the recomputed ReLU.  Reads g / argmax (pooled map,
// 1/4 of the pixels) and z, writes dz: the 2 x 411 MB round trip of ga through HBM (R50 stem, 32 x 8 frames) is gone.
template <typename ET>
__global__ __launch_bounds__(256) void maxpool_bn_bwd_rows_apply_kernel(const unsigned char* amax, const ET* g, int n, int h, int w, int c, int ho, int wo,
                                                                        const ET* z, const float* gamma, const float* mean, const float* invstd,
                                                                        const float* scale, const float* shift, const float* dgamma, const float* dbeta,
                                                                        float inv_m, ET* dz) {
    const int c4 = c >> 2;
    const int per = w * c4;
    const int cq = threadIdx.x % c4;
    float av[4], d0[4], kx[4], muv[4], scv[4], shv[4];
    {
        const float4 ga_ = *reinterpret_cast<const float4*>(gamma + cq * 4), rs = *reinterpret_cast<const float4*>(invstd + cq * 4);
        const float4 dg = *reinterpret_cast<const float4*>(dgamma + cq * 4), db = *reinterpret_cast<const float4*>(dbeta + cq * 4);
        const float4 mu = *reinterpret_cast<const float4*>(mean + cq * 4);
        const float4 sc = *reinterpret_cast<const float4*>(scale + cq * 4), sh = *reinterpret_cast<const float4*>(shift + cq * 4);
        const float gav[4] = {ga_.x, ga_.y, ga_.z, ga_.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w}, dgv[4] = {dg.x, dg.y, dg.z, dg.w}, dbv[4] = {db.x, db.y, db.z, db.w};
        const float mu_[4] = {mu.x, mu.y, mu.z, mu.w}, sc_[4] = {sc.x, sc.y, sc.z, sc.w}, sh_[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            av[j] = gav[j] * rsv[j];
            d0[j] = dbv[j] * inv_m;
            kx[j] = rsv[j] * dgv[j] * inv_m;
            muv[j] = mu_[j]; scv[j] = sc_[j]; shv[j] = sh_[j];
        }
    }
    const int row_end = min(n * h, ((int)blockIdx.x + 1) * kPoolRB);
    for (int row = blockIdx.x * kPoolRB; row < row_end; ++row) {
        const int img = row / h, ih = row - img * h;
        for (int i = threadIdx.x; i < per; i += 256) {
            const int iw = i / c4;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            unsigned amv[4];
            float4 gvv[4];
            bool okv[4];
            unsigned mev[4];
            const long zi = ((long)row * w + iw) * c + cq * 4;
            const float4 zv = ld4(z + zi);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int oh = (ih + (k >> 1)) >> 1, ow = (iw + (k & 1)) >> 1;
                okv[k] = oh < ho && ow < wo && ((k >> 1) == 0 || (ih & 1)) && ((k & 1) == 0 || (iw & 1));
                const int ohc = min(oh, ho - 1), owc = min(ow, wo - 1);
                mev[k] = (unsigned)(ih - (oh * 2 - 1)) * 3u + (unsigned)(iw - (ow * 2 - 1));
                const long oidx = (((long)img * ho + ohc) * wo + owc) * c + cq * 4;
                amv[k] = *reinterpret_cast<const unsigned*>(amax + oidx);
                gvv[k] = ld4(g + oidx);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned am = amv[k], me = mev[k];
                if (okv[k] && (am & 255u) == me) acc[0] += gvv[k].x;
                if (okv[k] && ((am >> 8) & 255u) == me) acc[1] += gvv[k].y;
                if (okv[k] && ((am >> 16) & 255u) == me) acc[2] += gvv[k].z;
                if (okv[k] && (am >> 24) == me) acc[3] += gvv[k].w;
            }
            if constexpr (sizeof(ET) == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = bf16_to_f32(f32_to_bf16(acc[j]));
            }
            const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gm = (zz[j] * scv[j] + shv[j]) > 0.f ? acc[j] : 0.f;
                o[j] = av[j] * (gm - d0[j] - (zz[j] - muv[j]) * kx[j]);
            }
            st4(dz + zi, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}

// [r3] BLOCK form of the two kernels above (even h and w): a thread owns a 2 x 2 block of input pixels x 16 bytes of channels.  The four
// pool windows that cover the block -- (bh, bh + 1) x (bw, bw + 1) -- are the SAME for its four pixels, so their argmax bytes and gradients
// are loaded once per block instead of once per pixel, as 16-byte (g, z) / 8-byte (argmax) accesses instead of 8 / 4: 12 loads per 32 outputs
// where the row form issues 9 per 4 (it ran at 3.3 TB/s of HBM bytes with 7 x that in L1 / L2 traffic).  Per pixel the windows are visited in
// the row form's order (row offset major) with its conditions, the gather is rounded to the storage type before the gate: identical dz / ga;
// the sums differ from the row form's in summation order only (another thread -> pixel map).  MODE 0: sums (+ optional ga), MODE 1: apply.
template <typename ET, int MODE>
__global__ __launch_bounds__(256) void maxpool_bn_bwd_blk_kernel(const unsigned char* amax, const ET* g, int n, int h, int w, int c, int ho, int wo, ET* ga,
                                                                 const ET* z, const float* gamma, const float* mean, const float* invstd, const float* scale,
                                                                 const float* shift, const float* dgamma, const float* dbeta, float inv_m, float* part, int nblk,
                                                                 ET* dz) {
    constexpr int VC = 16 / (int)sizeof(ET);           // channels per thread
    __shared__ float red[MODE == 0 ? 256 * 2 * VC : 1];
    const int ncg = c / VC, hw2 = w >> 1, hb = h >> 1;
    const int cq = threadIdx.x % ncg;                  // fixed per thread: 256 % ncg == 0 (host-checked)
    float scv[VC], shv[VC], muv[VC], k1[VC], k2[VC], k3[VC];      // MODE 0: k1 = invstd; MODE 1: k1 = gamma * invstd, k2 = dbeta / M, k3 = invstd * dgamma / M
#pragma unroll
    for (int j = 0; j < VC; ++j) {
        const int ch = cq * VC + j;
        scv[j] = scale[ch]; shv[j] = shift[ch]; muv[j] = mean[ch];
        if constexpr (MODE == 0) { k1[j] = invstd[ch]; k2[j] = 0.f; k3[j] = 0.f; }
        else { k1[j] = gamma[ch] * invstd[ch]; k2[j] = dbeta[ch] * inv_m; k3[j] = invstd[ch] * dgamma[ch] * inv_m; }
    }
    float s1[VC], s2[VC];
#pragma unroll
    for (int j = 0; j < VC; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    auto ldv = [](const ET* p, float (&f)[VC]) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        if constexpr (sizeof(ET) == 4) {
            f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
        } else {
            f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u); f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
            f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u); f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
        }
    };
    auto stv = [](ET* p, const float (&f)[VC]) {
        uint4 u;
        if constexpr (sizeof(ET) == 4) {
            u = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
        } else {
            u = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
        }
        *reinterpret_cast<uint4*>(p) = u;
    };
    const int items = (kPoolRB / 2) * hw2 * ncg;         // 4 block rows per workgroup
    const long brows = (long)n * hb;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int bi = i / ncg;                           // (block row, block column) of the item; i % ncg == cq
        const int br = bi / hw2, bw = bi - br * hw2;
        const long R = (long)blockIdx.x * (kPoolRB / 2) + br;
        if (R >= brows) break;
        const int img = (int)(R / hb), bh = (int)(R - (long)img * hb);
        float gv[4][VC], zv[4][VC];
        unsigned char am[4][VC];
        bool okw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                     // window k = (bh + (k >> 1), bw + (k & 1))
            const int oh = bh + (k >> 1), ow = bw + (k & 1);
            okw[k] = oh < ho && ow < wo;
            const long oidx = (((long)img * ho + min(oh, ho - 1)) * wo + min(ow, wo - 1)) * c + cq * VC;
            if constexpr (VC == 8) {
                const uint2 a8 = *reinterpret_cast<const uint2*>(amax + oidx);
#pragma unroll
                for (int j = 0; j < 4; ++j) { am[k][j] = (unsigned char)(a8.x >> (8 * j)); am[k][4 + j] = (unsigned char)(a8.y >> (8 * j)); }
            } else {
                const unsigned a4 = *reinterpret_cast<const unsigned*>(amax + oidx);
#pragma unroll
                for (int j = 0; j < 4; ++j) am[k][j] = (unsigned char)(a4 >> (8 * j));
            }
            ldv(g + oidx, gv[k]);
        }
        long zi[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {                     // pixel p = (2 bh + (p >> 1), 2 bw + (p & 1))
            zi[p] = (((long)img * h + 2 * bh + (p >> 1)) * w + 2 * bw + (p & 1)) * c + cq * VC;
            ldv(z + zi[p], zv[p]);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int a = p >> 1, b = p & 1;
            float acc[VC];
#pragma unroll
            for (int j = 0; j < VC; ++j) acc[j] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int wi = k >> 1, wj = k & 1;
                if ((wi == 1 && a == 0) || (wj == 1 && b == 0)) continue;      // an even row / column lies in one window only
                const unsigned me = (unsigned)(a + 1 - 2 * wi) * 3u + (unsigned)(b + 1 - 2 * wj);
#pragma unroll
                for (int j = 0; j < VC; ++j)
                    if (okw[k] && am[k][j] == me) acc[j] += gv[k][j];
            }
            if constexpr (MODE == 0) {
                if (ga) stv(ga + zi[p], acc);
            }
            if constexpr (sizeof(ET) == 2) {             // what the apply pass reads back / re-gathers: rounded to storage
#pragma unroll
                for (int j = 0; j < VC; ++j) acc[j] = bf16_to_f32(f32_to_bf16(acc[j]));
            }
            if constexpr (MODE == 0) {
#pragma unroll
                for (int j = 0; j < VC; ++j) {
                    const float gm = (zv[p][j] * scv[j] + shv[j]) > 0.f ? acc[j] : 0.f;
                    s1[j] += gm;
                    s2[j] += gm * ((zv[p][j] - muv[j]) * k1[j]);
                }
            } else {
                float o[VC];
#pragma unroll
                for (int j = 0; j < VC; ++j) {
                    const float gm = (zv[p][j] * scv[j] + shv[j]) > 0.f ? acc[j] : 0.f;
                    o[j] = k1[j] * (gm - k2[j] - (zv[p][j] - muv[j]) * k3[j]);
                }
                stv(dz + zi[p], o);
            }
        }
    }
    if constexpr (MODE == 0) {
        // threads t, t + ncg, t + 2 ncg, ... share the channel group: fixed-order sum through LDS, one channel-major partial row per workgroup
#pragma unroll
        for (int j = 0; j < VC; ++j) {
            red[(2 * j) * 256 + threadIdx.x] = s1[j];
            red[(2 * j + 1) * 256 + threadIdx.x] = s2[j];
        }
        __syncthreads();
        if ((int)threadIdx.x < c) {
            const int q = threadIdx.x / VC, j = threadIdx.x - q * VC;          // channel q * VC + j
            float a = 0.f, b = 0.f;
            for (int t = q; t < 256; t += ncg) {
                a += red[(2 * j) * 256 + t];
                b += red[(2 * j + 1) * 256 + t];
            }
            reinterpret_cast<float2*>(part)[(long)threadIdx.x * nblk + blockIdx.x] = make_float2(a, b);
        }
    }
}

template <typename ET>
__global__ void maxpool_bn_bwd_kernel(const unsigned char* amax, const ET* g, int n, int h, int w, int c, int ho, int wo, ET* ga) {
    const int c4 = c >> 2;
    const long total = (long)n * h * w * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, iw, ih, img;
        if (total <= 0x7fffffffL) {                       // 32-bit index math (64-bit division by a runtime value is ~100 instructions)
            unsigned t = (unsigned)i;
            cq = (int)(t % (unsigned)c4); t /= (unsigned)c4;
            iw = (int)(t % (unsigned)w); t /= (unsigned)w;
            ih = (int)(t % (unsigned)h);
            img = (int)(t / (unsigned)h);
        } else {
            cq = (int)(i % c4);
            long t = i / c4;
            iw = (int)(t % w); t /= w;
            ih = (int)(t % h);
            img = (int)(t / h);
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // the (<= 4) covering windows: oh in {ih/2, (ih+1)/2}, ow likewise (the second differs only for odd coordinates);
        // all eight loads are issued unconditionally on clamped indices, invalid candidates are masked in the compare
        unsigned amv[4];
        float4 gvv[4];
        bool okv[4];
        unsigned mev[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oh = (ih + (k >> 1)) >> 1, ow = (iw + (k & 1)) >> 1;
            okv[k] = oh < ho && ow < wo && ((k >> 1) == 0 || (ih & 1)) && ((k & 1) == 0 || (iw & 1));
            const int ohc = min(oh, ho - 1), owc = min(ow, wo - 1);
            mev[k] = (unsigned)(ih - (oh * 2 - 1)) * 3u + (unsigned)(iw - (ow * 2 - 1));
            const long oidx = (((long)img * ho + ohc) * wo + owc) * c + cq * 4;
            amv[k] = *reinterpret_cast<const unsigned*>(amax + oidx);
            gvv[k] = ld4(g + oidx);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned am = amv[k], me = mev[k];
            if (okv[k] && (am & 255u) == me) acc[0] += gvv[k].x;
            if (okv[k] && ((am >> 8) & 255u) == me) acc[1] += gvv[k].y;
            if (okv[k] && ((am >> 16) & 255u) == me) acc[2] += gvv[k].z;
            if (okv[k] && (am >> 24) == me) acc[3] += gvv[k].w;
        }
        st4(ga + i * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// ---- head: per-frame avg-pool -> fc -> mean over segments -> cross-entropy (mean over clips) ----
// pooled[frame][ch] = mean over hw ; lanes along channels
template <typename ET>
__global__ void frame_pool_kernel(const ET* feat, int hw, int c, const float* drop_mask, float* pooled) {
    const int f = blockIdx.y, ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const ET* p = feat + (long)f * hw * c + ch;
    float s = 0.f;
    for (int r = 0; r < hw; ++r) s += ldf(p + (long)r * c);
    s = s / (float)hw;
    if (drop_mask) s *= drop_mask[(long)f * c + ch];      // nn.Dropout: mask already scaled by 1/(1-p) (tsn_clshead.py:86-87)
    pooled[(long)f * c + ch] = s;
}

// one block per clip: softmax CE of scores[clip] against label; writes loss contribution and dscores = (p - onehot)/clips
__global__ void ce_loss_kernel(const float* scores, const long long* labels, int clips, int classes, float* loss_part, float* dscores) {
    __shared__ float red[256];
    const int cl = blockIdx.x, tid = threadIdx.x;
    const float* s = scores + (long)cl * classes;
    float mx = -INFINITY;
    for (int k = tid; k < classes; k += blockDim.x) mx = fmaxf(mx, s[k]);
    red[tid] = mx;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float e = 0.f;
    for (int k = tid; k < classes; k += blockDim.x) e += expf(s[k] - mx);
    red[tid] = e;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float den = red[0];
    const int lab = (int)labels[cl];
    if (tid == 0) loss_part[cl] = (logf(den) + mx) - s[lab];
    if (!dscores) return;
    for (int k = tid; k < classes; k += blockDim.x) {
        const float p = expf(s[k] - mx) / den;
        dscores[(long)cl * classes + k] = (p - (k == lab ? 1.f : 0.f)) / (float)clips;
    }
}

__global__ void mean_reduce_kernel(const float* v, int n, float* out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += v[i];
        out[0] = (float)(s / n);
    }
}

// dW[k][c] = sum_clip dscores[clip][k] * pooledclip[clip][c] ; db[k] = sum_clip dscores[clip][k]   (pooledclip = mean over T)
// thread = one channel, workgroup = (256 channels, kHeadKc classes): the channel's clip-mean features are formed once per thread
// (clips are walked in chunks of 16 registers) and reused for every class of the range, instead of re-reading the T frame rows for
// each of the 400 classes (69 -> ~15 us).  Fixed clip order: deterministic.
constexpr int kHeadKc = 4;
__global__ void head_clipmean_kernel(const float* pooled, int T, int c, float* pcm) {
    const int cl = blockIdx.y, ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += pooled[((long)cl * T + t) * c + ch];
    pcm[(long)cl * c + ch] = s / (float)T;
}
__global__ __launch_bounds__(256) void head_fc_bwd_w_kernel(const float* dscores, const float* pcm, int clips, int T, int c, int classes, float* dw, float* db) {
    const int k0 = blockIdx.y * kHeadKc, ch = blockIdx.x * blockDim.x + threadIdx.x;
    const int nk = min(kHeadKc, classes - k0);
    if (ch < c) {
        float acc[kHeadKc];
#pragma unroll
        for (int j = 0; j < kHeadKc; ++j) acc[j] = 0.f;
        for (int cl0 = 0; cl0 < clips; cl0 += 16) {
            float pcv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) pcv[q] = pcm[(long)min(cl0 + q, clips - 1) * c + ch];      // clamped: unconditional loads
#pragma unroll
            for (int j = 0; j < kHeadKc; ++j) {
                if (j < nk) {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (cl0 + q < clips) acc[j] += dscores[(long)(cl0 + q) * classes + k0 + j] * pcv[q];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kHeadKc; ++j)
            if (j < nk) dw[(long)(k0 + j) * c + ch] = acc[j];
    }
    if (blockIdx.x == 0 && threadIdx.x < nk) {
        float s = 0.f;
        for (int cl = 0; cl < clips; ++cl) s += dscores[(long)cl * classes + k0 + threadIdx.x];
        db[k0 + threadIdx.x] = s;
    }
}

// dfeat[frame][hw][ch] = (1/(T*hw)) * sum_k dscores[clip][k] * W[k][ch]
__global__ void head_dpool_kernel(const float* dscores, const float* w, int classes, int c, float scale, float* dpool) {
    const int cl = blockIdx.y, ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    // eight independent accumulators: the class loop is a chain of L2 round trips otherwise (81 -> ~15 us)
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* ds = dscores + (long)cl * classes;
    int k = 0;
    for (; k + 8 <= classes; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += ds[k + u] * w[(long)(k + u) * c + ch];
    }
    for (; k < classes; ++k) a[0] += ds[k] * w[(long)k * c + ch];
    dpool[(long)cl * c + ch] = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) * scale;
}
template <typename ET>
__global__ void head_dfeat_kernel(const float* dpool, const float* drop_mask, int T, int hw, int c, long total, ET* dfeat) {
    // blockIdx.y = frame, threads over (row, 4 channels): the frame's gradient row (dpool x mask) is the same for its hw pixels
    const int frame = blockIdx.y, c4 = c >> 2;
    const int per = hw * c4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
        const int q = i % c4;
        float4 v = *reinterpret_cast<const float4*>(dpool + (long)(frame / T) * c + q * 4);
        if (drop_mask) {
            const float4 mk = *reinterpret_cast<const float4*>(drop_mask + (long)frame * c + q * 4);
            v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
        }
        st4(dfeat + ((long)frame * hw * c + (long)i * 4), v);
    }
}

// ---- optimizer: global grad norm (two-stage), then clip + weight decay + SGD nesterov on a flat fp32 buffer ----
__global__ void sqsum_partial_kernel(const float* g, long n, float* part) {
    __shared__ float red[256];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void norm_finalize_kernel(const float* part, int nblk, float max_norm, float gscale, float* out /*[0]=norm,[1]=coef*/) {
    // 64 lanes sum strided slices of the partials (fixed order), lane 0 combines them in lane order: one wave, a handful of
    // memory round trips instead of nblk serial ones (47 -> ~6 us on the optimizer's critical path)
    __shared__ double red[64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) acc += part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < 64; ++i) s += red[i];
        const float nrm = fabsf(gscale) * (float)sqrt(s);
        out[0] = nrm;
        float coef = max_norm > 0.f ? max_norm / (nrm + 1e-6f) : 1.f;     // torch clip_grad_norm_
        out[1] = coef < 1.f ? coef : 1.f;
    }
}
__global__ void sgd_nesterov_kernel(float* p, const float* g, float* buf, long n, const float* coef_ptr, float gscale, float lr,
                                    float momentum, float wd, int first_step) {
    const float coef = (coef_ptr ? coef_ptr[1] : 1.f) * gscale;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        const float d = g[i] * coef + wd * pv;
        const float b = first_step ? d : momentum * buf[i] + d;
        buf[i] = b;
        p[i] = pv - lr * (d + momentum * b);
    }
}
// sum of squares over the elements of the segments that take part in training (lr_mult >= 0): the clip norm of
// torch.nn.utils.clip_grad_norm_(filter(requires_grad, params)) when parameters are excluded in scattered places
__device__ __forceinline__ int seg_of(const mvf_sgd_segment_t* seg, int nseg, long i) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg[mid].first <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ void sqsum_segments_kernel(const float* g, long n, const mvf_sgd_segment_t* seg, int nseg, float* part) {
    __shared__ float red[256];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        if (seg[seg_of(seg, nseg, i)].lr_mult >= 0.f) s += g[i] * g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
// the same update with per-SEGMENT learning-rate / weight-decay multipliers (build_optimizer's paramwise_options, reference
// codes/core/train.py:117-156) and an optional plain-momentum form: seg[k] = {first element, lr multiplier, decay multiplier},
// sorted by first element, seg[0].first == 0; a workgroup's 256-element run looks its segment up by binary search per element
__global__ void sgd_segments_kernel(float* p, const float* g, float* buf, long n, const float* coef_ptr, float gscale, float lr,
                                    float momentum, float wd, int first_step, int nesterov, const mvf_sgd_segment_t* seg, int nseg) {
    const float coef = (coef_ptr ? coef_ptr[1] : 1.f) * gscale;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int lo = seg_of(seg, nseg, i);
        if (seg[lo].lr_mult < 0.f) continue;                 // excluded from training (requires_grad False): parameter and momentum untouched
        const float lr_i = lr * seg[lo].lr_mult, wd_i = wd * seg[lo].decay_mult;
        const float pv = p[i];
        const float d = g[i] * coef + wd_i * pv;
        const float b = first_step ? d : momentum * buf[i] + d;
        buf[i] = b;
        p[i] = pv - lr_i * (nesterov ? d + momentum * b : b);
    }
}

inline int grid_for(long total, int cap = 256 * 16) { return (int)std::min<long>((total + 255) / 256, cap); }

}  // namespace

extern "C" {

size_t mvf_bn_workspace_bytes(long m, int c) {
    if (m <= 0 || c <= 0 || c % 4) return 0;
    int gy = col_plan(m, c, 4).gy;
    if (c % 8 == 0) gy = std::max(gy, col_plan(m, c, 8).gy);       // bf16 runs 8-wide lanes when it can
    return align_up((size_t)gy * c * 2 * sizeof(float), 256);
}

// DISPATCH(kernel, plan-blocks, args...): picks <float,4>, <bf16_t,8> (16-byte lanes: c, pitches % 8 == 0 and 16-B aligned
// pointers) or <bf16_t,4>, builds the column plan `p` for that width and launches on `st`.
#define MVF_BN_DISPATCH(KERNEL, WIDE_OK, BLOCKS, ...)                                                                        \
    do {                                                                                                                     \
        if (dtype == MVF_F32) {                                                                                              \
            using ET = float;                                                                                                \
            const ColPlan p = col_plan(m, c, 4, BLOCKS);                                                                     \
            hipLaunchKernelGGL((KERNEL<float, 4>), dim3(p.gx, p.gy), dim3(kThreads), 0, st, __VA_ARGS__);                    \
        } else if (WIDE_OK) {                                                                                                \
            using ET = bf16_t;                                                                                               \
            const ColPlan p = col_plan(m, c, 8, BLOCKS);                                                                     \
            hipLaunchKernelGGL((KERNEL<bf16_t, 8>), dim3(p.gx, p.gy), dim3(kThreads), 0, st, __VA_ARGS__);                   \
        } else {                                                                                                             \
            using ET = bf16_t;                                                                                               \
            const ColPlan p = col_plan(m, c, 4, BLOCKS);                                                                     \
            hipLaunchKernelGGL((KERNEL<bf16_t, 4>), dim3(p.gx, p.gy), dim3(kThreads), 0, st, __VA_ARGS__);                   \
        }                                                                                                                    \
    } while (0)

// the same with the mask mode (0-4) as a third template argument
#define MVF_BN_DISPATCH_MM1(KERNEL, MMV, WIDE_OK, BLOCKS, ...)                                                               \
    do {                                                                                                                     \
        if (dtype == MVF_F32) {                                                                                              \
            using ET = float;                                                                                                \
            const ColPlan p = col_plan(m, c, 4, BLOCKS);                                                                     \
            hipLaunchKernelGGL((KERNEL<float, 4, MMV>), dim3(p.gx, p.gy), dim3(kThreads), 0, st, __VA_ARGS__);               \
        } else if (WIDE_OK) {                                                                                                \
            using ET = bf16_t;                                                                                               \
            const ColPlan p = col_plan(m, c, 8, BLOCKS);                                                                     \
            hipLaunchKernelGGL((KERNEL<bf16_t, 8, MMV>), dim3(p.gx, p.gy), dim3(kThreads), 0, st, __VA_ARGS__);              \
        } else {                                                                                                             \
            using ET = bf16_t;                                                                                               \
            const ColPlan p = col_plan(m, c, 4, BLOCKS);                                                                     \
            hipLaunchKernelGGL((KERNEL<bf16_t, 4, MMV>), dim3(p.gx, p.gy), dim3(kThreads), 0, st, __VA_ARGS__);              \
        }                                                                                                                    \
    } while (0)
#define MVF_BN_DISPATCH_MM(KERNEL, MODE, WIDE_OK, BLOCKS, ...)                                \
    do {                                                                                      \
        switch (MODE) {                                                                       \
            case 0: MVF_BN_DISPATCH_MM1(KERNEL, 0, WIDE_OK, BLOCKS, __VA_ARGS__); break;      \
            case 1: MVF_BN_DISPATCH_MM1(KERNEL, 1, WIDE_OK, BLOCKS, __VA_ARGS__); break;      \
            case 2: MVF_BN_DISPATCH_MM1(KERNEL, 2, WIDE_OK, BLOCKS, __VA_ARGS__); break;      \
            case 3: MVF_BN_DISPATCH_MM1(KERNEL, 3, WIDE_OK, BLOCKS, __VA_ARGS__); break;      \
            default: MVF_BN_DISPATCH_MM1(KERNEL, 4, WIDE_OK, BLOCKS, __VA_ARGS__); break;     \
        }                                                                                     \
    } while (0)

constexpr int kFinLongRows = 1024;       // channel-major partials: from this many rows on, a whole workgroup per channel
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int mvf_bn_train_stats(const void* z, long m, int c, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* scale,
                       float* shift, void* ws, size_t ws_bytes, int dtype, void* stream) {
    MVF_REQUIRE(z && gamma && beta && save_mean && save_invstd && scale && shift && m > 0 && c > 0 && c % 4 == 0, MVF_EINVAL, "bn_train_stats: bad argument (c %% 4?)");
    MVF_REQUIRE(ws && ws_bytes >= mvf_bn_workspace_bytes(m, c), MVF_EWS, "bn_train_stats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    const bool wide = c % 8 == 0 && al16(z);
    const int gy = col_plan(m, c, (dtype != MVF_F32 && wide) ? 8 : 4).gy;
    MVF_BN_DISPATCH(bn_stats_kernel, wide, 2048, (const ET*)z, m, c, running_mean, p.cqb, p.rows, part);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_finalize_kernel<0>, dim3((c + kFinCh - 1) / kFinCh), dim3(256), 0, st, c, gy, m, part, gamma, beta, eps, momentum,
                       running_mean, running_var, save_mean, save_invstd, scale, shift);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

// finalize from partial sums produced elsewhere (the conv epilogue: mvf_conv2d_nhwc_fwd_stats), CHANNEL-MAJOR layout [c][nblk][2]: a channel's
// nblk partial pairs are contiguous (row-major [nblk][c][2] partials would be summed into the wrong channels -- silently)
int mvf_bn_train_finalize(const float* part, int nblk, long m, int c, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                          float* scale, float* shift, void* stream) {
    MVF_REQUIRE(part && gamma && beta && save_mean && save_invstd && scale && shift && nblk > 0 && m > 0 && c > 0, MVF_EINVAL, "bn_train_finalize: bad argument");
    if (nblk >= kFinLongRows)
        hipLaunchKernelGGL(bn_stats_finalize_kernel<1>, dim3(c), dim3(256), 0, (hipStream_t)stream, c, nblk, m, part, gamma, beta, eps, momentum,
                           running_mean, running_var, save_mean, save_invstd, scale, shift);
    else
        hipLaunchKernelGGL(bn_stats_finalize_kernel<2>, dim3((c + 3) / 4), dim3(256), 0, (hipStream_t)stream, c, nblk, m, part, gamma, beta, eps, momentum,
                           running_mean, running_var, save_mean, save_invstd, scale, shift);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_bn_apply_bits(const void* z, long m, int c, const float* scale, const float* shift, const void* residual,
                      const float* rscale, const float* rshift, int relu, void* out, unsigned char* sign_bits, int dtype,
                      void* stream) {
    MVF_REQUIRE(z && scale && shift && out && m > 0 && c > 0 && c % 4 == 0, MVF_EINVAL, "bn_apply: bad argument");
    MVF_REQUIRE((rscale == nullptr) == (rshift == nullptr), MVF_EINVAL, "bn_apply: rscale/rshift must come together");
    hipStream_t st = (hipStream_t)stream;
    const bool wide = c % 8 == 0 && al16(z) && al16(out) && al16(residual) && ((uintptr_t)sign_bits & 1) == 0;
    MVF_BN_DISPATCH(bn_apply_kernel, wide, 4096, (const ET*)z, m, c, scale, shift, (const ET*)residual, rscale, rshift, relu, (ET*)out, sign_bits,
                    p.cqb, p.rows);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

// [r5] out = act(z * scale + shift) AND the column means of what is stored (mean_out [c]; partial rows in ws, >= mvf_bn_workspace_bytes(m, c)): the column
// means of conv3's input for mvf_bn_train_stats_gram without a pass over it
int mvf_bn_apply_colmeans(const void* z, long m, int c, const float* scale, const float* shift, int act, void* out, float* mean_out, void* ws, size_t ws_bytes,
                          int dtype, void* stream) {
    MVF_REQUIRE(z && scale && shift && out && mean_out && ws && m > 0 && c > 0 && c % 8 == 0, MVF_EINVAL, "bn_apply_colmeans: bad argument (c %% 8?)");
    MVF_REQUIRE(al16(z) && al16(out), MVF_EINVAL, "bn_apply_colmeans: z / out must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int gy = col_plan(m, c, dtype == MVF_F32 ? 4 : 8, 4096).gy;
    MVF_REQUIRE(ws_bytes >= (size_t)gy * c * 2 * sizeof(float), MVF_EWS, "bn_apply_colmeans: workspace too small (%d partial rows)", gy);
    float* part = (float*)ws;
    const bool wide = true;
    MVF_BN_DISPATCH(bn_apply_cs_kernel, wide, 4096, (const ET*)z, m, c, scale, shift, act, (ET*)out, p.cqb, p.rows, part);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(colmean_finalize_kernel, dim3((c + kFinCh - 1) / kFinCh), dim3(256), 0, st, c, gy, m, part, mean_out);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_bn_apply(const void* z, long m, int c, const float* scale, const float* shift, const void* residual,
                 const float* rscale, const float* rshift, int relu, void* out, int dtype, void* stream) {
    return mvf_bn_apply_bits(z, m, c, scale, shift, residual, rscale, rshift, relu, out, nullptr, dtype, stream);
}

int mvf_bn_bwd_reduce(const void* g, int g_pitch, const void* z, const void* ymask, long m, int c, const float* mean, const float* invstd,
                      const float* scale, const float* shift, int mask_mode, void* gm_out, float* dgamma, float* dbeta,
                      void* ws, size_t ws_bytes, int dtype, void* stream) {
    MVF_REQUIRE(g && z && mean && invstd && dgamma && dbeta && m > 0 && c > 0 && c % 4 == 0, MVF_EINVAL, "bn_bwd_reduce: bad argument");
    MVF_REQUIRE(mask_mode >= 0 && mask_mode <= 4 && ((mask_mode != 1 && mask_mode != 4) || ymask) &&
                    ((mask_mode != 2 && mask_mode != 3) || (scale && shift)) && g_pitch >= c && g_pitch % 4 == 0,
                MVF_EINVAL, "bn_bwd_reduce: bad mask_mode / pitch");
    MVF_REQUIRE(ws && ws_bytes >= mvf_bn_workspace_bytes(m, c), MVF_EWS, "bn_bwd_reduce: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    const bool wide = false;                 // measured: the 8-byte-lane variant is ~7% faster for this kernel (more rows in flight per wave)
    const int gy = col_plan(m, c, (dtype != MVF_F32 && wide) ? 8 : 4).gy;
    static const bool bn_spec = (mvf_policy_int("bn_spec", 1) != 0);      // A/B switch
    if (bn_spec)
        MVF_BN_DISPATCH_MM(bn_bwd_reduce_kernel, mask_mode, wide, 2048, (const ET*)g, g_pitch, (const ET*)z, (const ET*)ymask, m, c, mean, invstd, scale, shift, mask_mode, (ET*)gm_out, p.cqb, p.rows, part);
    else
        MVF_BN_DISPATCH(bn_bwd_reduce_kernel, wide, 2048, (const ET*)g, g_pitch, (const ET*)z, (const ET*)ymask, m, c, mean, invstd, scale, shift, mask_mode, (ET*)gm_out, p.cqb, p.rows, part);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<0>, dim3((c + kFinCh - 1) / kFinCh), dim3(256), 0, st, c, gy, part, dgamma, dbeta);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_bn_bwd_finalize(const float* sums_part, int nblk, int c, float* dgamma, float* dbeta, void* stream) {
    MVF_REQUIRE(sums_part && dgamma && dbeta && nblk > 0 && c > 0, MVF_EINVAL, "bn_bwd_finalize: bad argument");
    if (nblk >= kFinLongRows)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<1>, dim3(c), dim3(256), 0, (hipStream_t)stream, c, nblk, sums_part, dgamma, dbeta);
    else
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<2>, dim3((c + 3) / 4), dim3(256), 0, (hipStream_t)stream, c, nblk, sums_part, dgamma, dbeta);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_bn_bwd_apply_masked(const void* g, int g_pitch, const void* z, const void* ymask, long m, int c, const float* gamma,
                            const float* mean, const float* invstd, const float* scale, const float* shift, const float* dgamma,
                            const float* dbeta, int mask_mode, void* dz, int dtype, void* stream) {
    MVF_REQUIRE(g && z && gamma && mean && invstd && dgamma && dbeta && dz && m > 0 && c > 0 && c % 4 == 0, MVF_EINVAL, "bn_bwd_apply: bad argument");
    MVF_REQUIRE(mask_mode >= 0 && mask_mode <= 4 && ((mask_mode != 1 && mask_mode != 4) || ymask) &&
                    ((mask_mode != 2 && mask_mode != 3) || (scale && shift)) && g_pitch >= c && g_pitch % 4 == 0,
                MVF_EINVAL, "bn_bwd_apply: bad mask_mode / pitch");
    hipStream_t st = (hipStream_t)stream;
    const bool wide = c % 8 == 0 && g_pitch % 8 == 0 && al16(g) && al16(z) && al16(dz) && (mask_mode != 1 || al16(ymask)) &&
                      (mask_mode != 4 || ((uintptr_t)ymask & 1) == 0);
    static const bool bn_spec = (mvf_policy_int("bn_spec", 1) != 0);      // A/B switch
    if (bn_spec)
        MVF_BN_DISPATCH_MM(bn_bwd_apply_kernel, mask_mode, wide, 4096, (const ET*)g, g_pitch, (const ET*)z, (const ET*)ymask, m, c, gamma, mean, invstd, scale, shift,
                    dgamma, dbeta, mask_mode, (ET*)dz, p.cqb, p.rows);
    else
        MVF_BN_DISPATCH(bn_bwd_apply_kernel, wide, 4096, (const ET*)g, g_pitch, (const ET*)z, (const ET*)ymask, m, c, gamma, mean, invstd, scale, shift,
                    dgamma, dbeta, mask_mode, (ET*)dz, p.cqb, p.rows);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_bn_bwd_apply(const void* g, int g_pitch, const void* z, long m, int c, const float* gamma, const float* mean, const float* invstd,
                     const float* scale, const float* shift, const float* dgamma, const float* dbeta, int mask_mode, void* dz,
                     int dtype, void* stream) {
    MVF_REQUIRE(mask_mode == 0 || mask_mode == 2 || mask_mode == 3, MVF_EINVAL, "bn_bwd_apply: mask_mode must be 0, 2 or 3 (use mvf_bn_bwd_apply_masked)");
    return mvf_bn_bwd_apply_masked(g, g_pitch, z, nullptr, m, c, gamma, mean, invstd, scale, shift, dgamma, dbeta, mask_mode, dz, dtype, stream);
}

int mvf_bn_bwd_pair(const void* g, int g_pitch, const void* z_a, const void* z_b, const unsigned char* sign_bits, long m, int c,
                    const float* gamma_a, const float* mean_a, const float* invstd_a, float* dgamma_a, float* dbeta_a,
                    const float* gamma_b, const float* mean_b, const float* invstd_b, float* dgamma_b, float* dbeta_b,
                    void* dz_a, void* dz_b, void* ws, size_t ws_bytes, int dtype, void* stream) {
    MVF_REQUIRE(g && z_a && z_b && sign_bits && gamma_a && mean_a && invstd_a && dgamma_a && dbeta_a && gamma_b && mean_b && invstd_b && dgamma_b && dbeta_b &&
                    dz_a && dz_b && m > 0 && c > 0 && c % 4 == 0 && g_pitch >= c && g_pitch % 4 == 0, MVF_EINVAL, "bn_bwd_pair: bad argument");
    MVF_REQUIRE(ws && ws_bytes >= 2 * mvf_bn_workspace_bytes(m, c), MVF_EWS, "bn_bwd_pair: workspace too small (2 x mvf_bn_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    float* part_a = (float*)ws;
    float* part_b = (float*)((char*)ws + mvf_bn_workspace_bytes(m, c));
    {   // reductions: the 8-byte-lane plan of mvf_bn_bwd_reduce (wide = false there)
        const bool wide = false;
        const int gy = col_plan(m, c, 4).gy;
        MVF_BN_DISPATCH(bn_bwd_reduce2_kernel, wide, 2048, (const ET*)g, g_pitch, (const ET*)z_a, (const ET*)z_b, (const void*)sign_bits, m, c, mean_a, invstd_a,
                        mean_b, invstd_b, p.cqb, p.rows, part_a, part_b);
        MVF_LAUNCH_CHECK();
        const int gA = (c + kFinCh - 1) / kFinCh;
        hipLaunchKernelGGL(bn_bwd_finalize2_kernel, dim3(2 * gA), dim3(256), 0, st, c, gy, part_a, part_b, dgamma_a, dbeta_a, dgamma_b, dbeta_b);
        MVF_LAUNCH_CHECK();
    }
    const bool wide = c % 8 == 0 && g_pitch % 8 == 0 && al16(g) && al16(z_a) && al16(z_b) && al16(dz_a) && al16(dz_b) && ((uintptr_t)sign_bits & 1) == 0;
    MVF_BN_DISPATCH(bn_bwd_apply2_kernel, wide, 4096, (const ET*)g, g_pitch, (const ET*)z_a, (const ET*)z_b, (const void*)sign_bits, m, c, gamma_a, mean_a, invstd_a,
                    dgamma_a, dbeta_a, gamma_b, mean_b, invstd_b, dgamma_b, dbeta_b, (ET*)dz_a, (ET*)dz_b, p.cqb, p.rows);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

// ---- BatchNorm backward apply fused with the pointwise conv's weight gradient (csrc/bnbwd_wgrad.hip) ----
int mvf_bn_bwd_wgrad_splits(long m, int c, int k, int nbn, int mask_mode) {
    int rows, ct, kt;
    if (m <= 0 || c <= 0 || k <= 0 || m * (long)std::max(c, k) * 2 >= 0x7ffffff0L) return 0;
    return mvf_internal::bnbwd_wgrad_plan(m, c, k, nbn, mask_mode, &rows, &ct, &kt);
}

size_t mvf_bn_bwd_wgrad_slab_bytes(long m, int c, int k, int nbn, int mask_mode) {
    return align_up((size_t)mvf_bn_bwd_wgrad_splits(m, c, k, nbn, mask_mode) * c * k * sizeof(float), 256);
}

static int bnwg_fill(mvf_internal::BnBwdWgradArgs& a, long m, int c, int k, int nbn, int mask_mode, size_t slab_bytes) {
    a.M = (int)m; a.C = c; a.K = k;
    a.nsplit = mvf_internal::bnbwd_wgrad_plan(m, c, k, nbn, mask_mode, &a.rows_per_split, &a.ctiles, &a.ktiles);
    MVF_REQUIRE(a.nsplit > 0, MVF_EUNSUPPORTED, "bn backward + weight gradient: shape c=%d k=%d (BatchNorms %d, mask mode %d) is not built; use the separate calls", c, k, nbn, mask_mode);
    MVF_REQUIRE(slab_bytes >= (size_t)a.nsplit * c * k * sizeof(float), MVF_EWS, "bn backward + weight gradient: slab buffer too small (mvf_bn_bwd_wgrad_slab_bytes)");
    return MVF_OK;
}

int mvf_bn_bwd_apply_wgrad(const void* g, int g_pitch, const void* z, const void* ymask, long m, int c, const float* gamma, const float* mean,
                           const float* invstd, const float* scale, const float* shift, const float* dgamma, const float* dbeta, int mask_mode,
                           void* dz, const void* x, int x_pitch, int k, float* slabs, size_t slab_bytes, int dtype, void* stream) {
    MVF_REQUIRE(g && z && gamma && mean && invstd && dgamma && dbeta && dz && x && slabs && m > 0 && c > 0 && k > 0 && g_pitch >= c && x_pitch >= k,
                MVF_EINVAL, "bn_bwd_apply_wgrad: bad argument");
    MVF_REQUIRE(dtype == MVF_BF16, MVF_EUNSUPPORTED, "bn_bwd_apply_wgrad: bf16 storage only");
    MVF_REQUIRE(mask_mode == 2 || mask_mode == 4, MVF_EUNSUPPORTED, "bn_bwd_apply_wgrad: mask mode 2 (scale * z + shift > 0) or 4 (sign bits)");
    MVF_REQUIRE(mask_mode == 2 ? (scale && shift) : (ymask != nullptr), MVF_EINVAL, "bn_bwd_apply_wgrad: the mask operand of mode %d is missing", mask_mode);
    MVF_REQUIRE(g_pitch % 8 == 0 && x_pitch % 8 == 0 && al16(g) && al16(z) && al16(dz) && al16(x) && ((uintptr_t)ymask & 1) == 0 &&
                    m * (long)std::max(g_pitch, x_pitch) * 2 < 0x7ffffff0L, MVF_ESHAPE, "bn_bwd_apply_wgrad: alignment / 2 GB addressing");
    mvf_internal::BnBwdWgradArgs a = {};
    if (int rc = bnwg_fill(a, m, c, k, 1, mask_mode, slab_bytes)) return rc;
    a.g = g; a.g_pitch = g_pitch; a.z[0] = z; a.bits = ymask;
    a.gamma[0] = gamma; a.mean[0] = mean; a.invstd[0] = invstd; a.dgamma[0] = dgamma; a.dbeta[0] = dbeta;
    a.scale = scale; a.shift = shift;
    a.dz[0] = dz; a.x[0] = x; a.xps[0] = x_pitch; a.part[0] = slabs;
    return mvf_internal::bnbwd_wgrad_launch(a, 1, mask_mode, (hipStream_t)stream);
}

int mvf_bn_bwd_pair_wgrad(const void* g, int g_pitch, const void* z_a, const void* z_b, const unsigned char* sign_bits, long m, int c,
                          const float* gamma_a, const float* mean_a, const float* invstd_a, float* dgamma_a, float* dbeta_a,
                          const float* gamma_b, const float* mean_b, const float* invstd_b, float* dgamma_b, float* dbeta_b,
                          void* dz_a, void* dz_b, const void* x_a, int xa_pitch, const void* x_b, int xb_pitch, int k,
                          float* slabs_a, float* slabs_b, size_t slab_bytes, void* ws, size_t ws_bytes, int dtype, void* stream) {
    MVF_REQUIRE(g && z_a && z_b && sign_bits && gamma_a && mean_a && invstd_a && dgamma_a && dbeta_a && gamma_b && mean_b && invstd_b && dgamma_b && dbeta_b &&
                    dz_a && dz_b && x_a && slabs_a && (!x_b || slabs_b) && m > 0 && c > 0 && k > 0 && g_pitch >= c && xa_pitch >= k && (!x_b || xb_pitch >= k),
                MVF_EINVAL, "bn_bwd_pair_wgrad: bad argument");
    MVF_REQUIRE(dtype == MVF_BF16, MVF_EUNSUPPORTED, "bn_bwd_pair_wgrad: bf16 storage only");
    MVF_REQUIRE(ws && ws_bytes >= 2 * mvf_bn_workspace_bytes(m, c), MVF_EWS, "bn_bwd_pair_wgrad: workspace too small (2 x mvf_bn_workspace_bytes)");
    MVF_REQUIRE(g_pitch % 8 == 0 && xa_pitch % 8 == 0 && (!x_b || xb_pitch % 8 == 0) && al16(g) && al16(z_a) && al16(z_b) && al16(dz_a) && al16(dz_b) &&
                    al16(x_a) && (!x_b || al16(x_b)) && ((uintptr_t)sign_bits & 1) == 0 &&
                    m * (long)std::max(g_pitch, std::max(xa_pitch, x_b ? xb_pitch : 0)) * 2 < 0x7ffffff0L, MVF_ESHAPE, "bn_bwd_pair_wgrad: alignment / 2 GB addressing");
    mvf_internal::BnBwdWgradArgs a = {};
    if (int rc = bnwg_fill(a, m, c, k, 2, 4, slab_bytes)) return rc;
    MVF_REQUIRE(x_b || a.ktiles == 1, MVF_EUNSUPPORTED, "bn_bwd_pair_wgrad: k tiles");
    int kt_, ct_;
    MVF_REQUIRE(mvf_internal::bnbwd_wgrad_tile(c, k, 2, 4, &ct_, &kt_) && (!x_b || kt_ == 64), MVF_EUNSUPPORTED,
                "bn_bwd_pair_wgrad: both convs are contracted for k = 64 only (k = 128: pass x_b = NULL)");
    hipStream_t st = (hipStream_t)stream;
    float* part_a = (float*)ws;
    float* part_b = (float*)((char*)ws + mvf_bn_workspace_bytes(m, c));
    {   // the sums: exactly mvf_bn_bwd_pair's reduce + finalize
        const bool wide = false;
        const int gy = col_plan(m, c, 4).gy;
        MVF_BN_DISPATCH(bn_bwd_reduce2_kernel, wide, 2048, (const ET*)g, g_pitch, (const ET*)z_a, (const ET*)z_b, (const void*)sign_bits, m, c, mean_a, invstd_a,
                        mean_b, invstd_b, p.cqb, p.rows, part_a, part_b);
        MVF_LAUNCH_CHECK();
        const int gA = (c + kFinCh - 1) / kFinCh;
        hipLaunchKernelGGL(bn_bwd_finalize2_kernel, dim3(2 * gA), dim3(256), 0, st, c, gy, part_a, part_b, dgamma_a, dbeta_a, dgamma_b, dbeta_b);
        MVF_LAUNCH_CHECK();
    }
    a.g = g; a.g_pitch = g_pitch; a.z[0] = z_a; a.z[1] = z_b; a.bits = sign_bits;
    a.gamma[0] = gamma_a; a.mean[0] = mean_a; a.invstd[0] = invstd_a; a.dgamma[0] = dgamma_a; a.dbeta[0] = dbeta_a;
    a.gamma[1] = gamma_b; a.mean[1] = mean_b; a.invstd[1] = invstd_b; a.dgamma[1] = dgamma_b; a.dbeta[1] = dbeta_b;
    a.dz[0] = dz_a; a.dz[1] = dz_b; a.x[0] = x_a; a.x[1] = x_b; a.xps[0] = xa_pitch; a.xps[1] = xb_pitch;
    a.part[0] = slabs_a; a.part[1] = slabs_b;
    return mvf_internal::bnbwd_wgrad_launch(a, 2, 4, st);
}

// ---- the whole backward of a z3-free bottleneck's last conv in one pass (csrc/pw_bwd_fused.hip) ----
int mvf_conv1x1_bwd_fused_splits(long m, int c, int k) { return mvf_internal::pw_bwd_fused_plan(m, c, k, nullptr); }

int mvf_conv1x1_bwd_fused(const void* a_in, int a_pitch, const void* w_packed, const void* g, int g_pitch, const unsigned char* sign_bits, long m, int c,
                          int k, const float* gamma, const float* mean, const float* invstd, const float* dgamma, const float* dbeta, const void* z_in,
                          const float* in_mean, const float* in_invstd, const float* in_scale, const float* in_shift, void* dx, float* sums_part,
                          int sums_rows, float* slabs, size_t slab_bytes, int dtype, void* stream) {
    MVF_REQUIRE(a_in && w_packed && g && sign_bits && gamma && mean && invstd && dgamma && dbeta && dx && slabs && m > 0 && a_pitch >= k && g_pitch >= c &&
                    (!z_in || (in_mean && in_invstd && in_scale && in_shift && sums_part)), MVF_EINVAL, "conv1x1_bwd_fused: bad argument");
    MVF_REQUIRE(dtype == MVF_BF16, MVF_EUNSUPPORTED, "conv1x1_bwd_fused: bf16 storage only");
    mvf_internal::PwBwdFusedArgs a = {};
    a.nsplit = mvf_internal::pw_bwd_fused_plan(m, c, k, &a.rows_per_split);
    MVF_REQUIRE(a.nsplit > 0, MVF_EUNSUPPORTED, "conv1x1_bwd_fused: built for 64 -> 256 channels (c=%d k=%d); use the separate calls", c, k);
    MVF_REQUIRE(!z_in || sums_rows == 2 * a.nsplit, MVF_EINVAL, "conv1x1_bwd_fused: sums_rows must be 2 x mvf_conv1x1_bwd_fused_splits (%d), got %d", 2 * a.nsplit, sums_rows);
    MVF_REQUIRE(slab_bytes >= (size_t)a.nsplit * c * k * sizeof(float), MVF_EWS, "conv1x1_bwd_fused: slab buffer too small (splits x c x k floats)");
    MVF_REQUIRE(a_pitch % 8 == 0 && g_pitch % 8 == 0 && al16(a_in) && al16(w_packed) && al16(g) && (!z_in || al16(z_in)) && al16(dx) && ((uintptr_t)sign_bits & 15) == 0 &&
                    m * (long)std::max(g_pitch, a_pitch) * 2 < 0x7ffffff0L, MVF_ESHAPE, "conv1x1_bwd_fused: alignment / 2 GB addressing");
    a.a = a_in; a.aps = a_pitch; a.w = w_packed; a.g = g; a.g_pitch = g_pitch; a.bits = sign_bits; a.M = (int)m;
    a.gamma = gamma; a.mean = mean; a.invstd = invstd; a.dgamma = dgamma; a.dbeta = dbeta;
    a.z_in = z_in; a.in_mean = in_mean; a.in_invstd = in_invstd; a.in_scale = in_scale; a.in_shift = in_shift;
    a.dx = dx; a.sums_part = sums_part; a.sums_rows = sums_rows; a.part = slabs;
    return mvf_internal::pw_bwd_fused_launch(a, (hipStream_t)stream);
}

int mvf_conv1x1_bnbwd_sums_pair(const void* a_in, int a_pitch, const void* w_a, const void* x_in, int x_pitch, const void* w_b, const void* g, int g_pitch,
                                const unsigned char* sign_bits, long m, int c, int k, const float* mean_a, const float* invstd_a, const float* mean_b,
                                const float* invstd_b, float* part_a, float* part_b, int rows, int dtype, void* stream) {
    MVF_REQUIRE(a_in && w_a && g && sign_bits && mean_a && invstd_a && part_a && m > 0 && a_pitch >= k && g_pitch >= c &&
                    (!x_in || (w_b && mean_b && invstd_b && part_b && x_pitch >= k)), MVF_EINVAL, "conv1x1_bnbwd_sums_pair: bad argument");
    MVF_REQUIRE(dtype == MVF_BF16, MVF_EUNSUPPORTED, "conv1x1_bnbwd_sums_pair: bf16 storage only");
    mvf_internal::PwSumsPairArgs a = {};
    a.nsplit = mvf_internal::pw_bwd_fused_plan(m, c, k, &a.rows_per_split);
    MVF_REQUIRE(a.nsplit > 0, MVF_EUNSUPPORTED, "conv1x1_bnbwd_sums_pair: built for 64 -> 256 channels (c=%d k=%d); use the separate calls", c, k);
    MVF_REQUIRE(rows == 2 * a.nsplit, MVF_EINVAL, "conv1x1_bnbwd_sums_pair: rows must be 2 x mvf_conv1x1_bwd_fused_splits (%d), got %d", 2 * a.nsplit, rows);
    MVF_REQUIRE(a_pitch % 8 == 0 && g_pitch % 8 == 0 && al16(a_in) && al16(w_a) && al16(g) && (!x_in || (x_pitch % 8 == 0 && al16(x_in) && al16(w_b))) &&
                    ((uintptr_t)sign_bits & 15) == 0 && m * (long)std::max(g_pitch, std::max(a_pitch, x_in ? x_pitch : 0)) * 2 < 0x7ffffff0L, MVF_ESHAPE,
                "conv1x1_bnbwd_sums_pair: alignment / 2 GB addressing");
    a.a = a_in; a.aps = a_pitch; a.x = x_in; a.xps = x_pitch; a.w_a = w_a; a.w_b = w_b; a.g = g; a.g_pitch = g_pitch; a.bits = sign_bits; a.M = (int)m;
    a.mean_a = mean_a; a.invstd_a = invstd_a; a.mean_b = mean_b; a.invstd_b = invstd_b; a.part_a = part_a; a.part_b = part_b; a.rows = rows;
    return mvf_internal::pw_sums_pair_launch(a, (hipStream_t)stream);
}

int mvf_wgrad_slab_reduce(const float* slabs, int nsplit, int cout, int k, float* dw_oihw, void* stream) {
    MVF_REQUIRE(slabs && dw_oihw && nsplit > 0 && cout > 0 && k > 0, MVF_EINVAL, "wgrad_slab_reduce: bad argument");
    return mvf_internal::wgrad_slab_reduce_launch(slabs, nsplit, cout, k, dw_oihw, (hipStream_t)stream);
}

int mvf_maxpool_bn_relu_fwd(const void* z, int n, int h, int w, int c, const float* scale, const float* shift, void* y,
                            unsigned char* argmax, int dtype, void* stream) {
    MVF_REQUIRE(z && y && scale && shift && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, MVF_EINVAL, "maxpool_bn_relu_fwd: bad argument");
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    const long total = (long)n * ho * wo * (c / 4);
    {   // [r3] 2 x 2 pooled outputs per thread, 16-byte channel groups (policy pool_fwd_block=0: the element form)
        static const int blk_on = mvf_policy_int("pool_fwd_block", 1);
        const int vc = dtype == MVF_F32 ? 4 : 8;
        if (blk_on && h % 4 == 0 && w % 4 == 0 && c % vc == 0 && ((uintptr_t)z | (uintptr_t)y) % 16 == 0 && (uintptr_t)argmax % 8 == 0) {
            const long tb = (long)n * (ho / 2) * (wo / 2) * (c / vc);
            if (dtype == MVF_F32)
                hipLaunchKernelGGL(maxpool_bn_fwd_blk_kernel<float>, dim3(grid_for(tb, 256 * 8)), dim3(256), 0, (hipStream_t)stream, (const float*)z, n, h, w, c, ho, wo, scale, shift, (float*)y, argmax);
            else
                hipLaunchKernelGGL(maxpool_bn_fwd_blk_kernel<bf16_t>, dim3(grid_for(tb, 256 * 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, n, h, w, c, ho, wo, scale, shift, (bf16_t*)y, argmax);
            MVF_LAUNCH_CHECK();
            return MVF_OK;
        }
    }
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(maxpool_bn_fwd_kernel<float>, dim3(grid_for(total, 256 * 32)), dim3(256), 0, (hipStream_t)stream, (const float*)z, n, h, w, c, ho, wo, scale, shift, (float*)y, argmax);
    else
        hipLaunchKernelGGL(maxpool_bn_fwd_kernel<bf16_t>, dim3(grid_for(total, 256 * 32)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, n, h, w, c, ho, wo, scale, shift, (bf16_t*)y, argmax);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_maxpool_bn_relu_bwd(const unsigned char* argmax, const void* g, int n, int h, int w, int c, void* ga, int dtype, void* stream) {
    MVF_REQUIRE(argmax && g && ga && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, MVF_EINVAL, "maxpool_bn_relu_bwd: bad argument");
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    const long total = (long)n * h * w * (c / 4);
    if ((long)n * h < (1L << 31) && (long)w * (c / 4) >= 256) {           // row form (see maxpool_bn_bwd_rows_kernel)
        if (dtype == MVF_F32)
            hipLaunchKernelGGL(maxpool_bn_bwd_rows_kernel<float>, dim3(n * h), dim3(256), 0, (hipStream_t)stream, argmax, (const float*)g, n, h, w, c, ho, wo, (float*)ga);
        else
            hipLaunchKernelGGL(maxpool_bn_bwd_rows_kernel<bf16_t>, dim3(n * h), dim3(256), 0, (hipStream_t)stream, argmax, (const bf16_t*)g, n, h, w, c, ho, wo, (bf16_t*)ga);
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(maxpool_bn_bwd_kernel<float>, dim3(grid_for(total, 256 * 64)), dim3(256), 0, (hipStream_t)stream, argmax, (const float*)g, n, h, w, c, ho, wo, (float*)ga);
    else
        hipLaunchKernelGGL(maxpool_bn_bwd_kernel<bf16_t>, dim3(grid_for(total, 256 * 64)), dim3(256), 0, (hipStream_t)stream, argmax, (const bf16_t*)g, n, h, w, c, ho, wo, (bf16_t*)ga);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

// the block form (maxpool_bn_bwd_blk_kernel): even image sides, 16-byte channel groups that divide 256 threads, 16-byte aligned tensors
static bool pool_blk_ok(const void* amax, const void* g, const void* ga, const void* z, const void* dz, int h, int w, int c, int dtype) {
    static const int on = mvf_policy_int("pool_bwd_block", 1);
    const int vc = dtype == MVF_F32 ? 4 : 8;
    return on && h % 2 == 0 && w % 2 == 0 && c % vc == 0 && 256 % (c / vc) == 0 && c <= 256 && (uintptr_t)amax % 8 == 0 &&
           ((uintptr_t)g | (uintptr_t)(ga ? ga : g) | (uintptr_t)z | (uintptr_t)(dz ? dz : z)) % 16 == 0;
}

int mvf_maxpool_bwd_sums_rows(int n, int h) { return n > 0 && h > 0 ? (int)(((long)n * h + kPoolRB - 1) / kPoolRB) : 0; }

int mvf_maxpool_bn_relu_bwd_sums(const unsigned char* argmax, const void* g, int n, int h, int w, int c, void* ga, const void* z, const float* mean,
                                 const float* invstd, const float* scale, const float* shift, float* sums_part, int dtype, void* stream) {
    MVF_REQUIRE(argmax && g && z && mean && invstd && scale && shift && sums_part && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, MVF_EINVAL,
                "maxpool_bn_relu_bwd_sums: bad argument");             // ga may be NULL (mvf_maxpool_bn_relu_bwd_apply re-gathers it)
    MVF_REQUIRE(256 % (c / 4) == 0 && (long)n * h < (1L << 31), MVF_EUNSUPPORTED, "maxpool_bn_relu_bwd_sums: needs c/4 to divide 256 (use the two-pass form)");
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1, nblk = mvf_maxpool_bwd_sums_rows(n, h);
    if (pool_blk_ok(argmax, g, ga, z, nullptr, h, w, c, dtype)) {          // [r3] 2 x 2 pixel blocks per thread
        if (dtype == MVF_F32)
            hipLaunchKernelGGL((maxpool_bn_bwd_blk_kernel<float, 0>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, argmax, (const float*)g, n, h, w, c, ho, wo, (float*)ga,
                               (const float*)z, nullptr, mean, invstd, scale, shift, nullptr, nullptr, 0.f, sums_part, nblk, nullptr);
        else
            hipLaunchKernelGGL((maxpool_bn_bwd_blk_kernel<bf16_t, 0>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, argmax, (const bf16_t*)g, n, h, w, c, ho, wo, (bf16_t*)ga,
                               (const bf16_t*)z, nullptr, mean, invstd, scale, shift, nullptr, nullptr, 0.f, sums_part, nblk, nullptr);
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(maxpool_bn_bwd_rows_sums_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, argmax, (const float*)g, n, h, w, c, ho, wo, (float*)ga,
                           (const float*)z, mean, invstd, scale, shift, sums_part, nblk);
    else
        hipLaunchKernelGGL(maxpool_bn_bwd_rows_sums_kernel<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, argmax, (const bf16_t*)g, n, h, w, c, ho, wo, (bf16_t*)ga,
                           (const bf16_t*)z, mean, invstd, scale, shift, sums_part, nblk);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_maxpool_bn_relu_bwd_apply(const unsigned char* argmax, const void* g, int n, int h, int w, int c, const void* z, const float* gamma,
                                  const float* mean, const float* invstd, const float* scale, const float* shift, const float* dgamma,
                                  const float* dbeta, void* dz, int dtype, void* stream) {
    MVF_REQUIRE(argmax && g && z && gamma && mean && invstd && scale && shift && dgamma && dbeta && dz && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0,
                MVF_EINVAL, "maxpool_bn_relu_bwd_apply: bad argument");
    MVF_REQUIRE(256 % (c / 4) == 0 && (long)n * h < (1L << 31), MVF_EUNSUPPORTED, "maxpool_bn_relu_bwd_apply: needs c/4 to divide 256 (use the two-pass form)");
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1, nblk = mvf_maxpool_bwd_sums_rows(n, h);
    const float inv_m = 1.0f / (float)((long)n * h * w);
    if (pool_blk_ok(argmax, g, nullptr, z, dz, h, w, c, dtype)) {
        if (dtype == MVF_F32)
            hipLaunchKernelGGL((maxpool_bn_bwd_blk_kernel<float, 1>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, argmax, (const float*)g, n, h, w, c, ho, wo, (float*)nullptr,
                               (const float*)z, gamma, mean, invstd, scale, shift, dgamma, dbeta, inv_m, nullptr, nblk, (float*)dz);
        else
            hipLaunchKernelGGL((maxpool_bn_bwd_blk_kernel<bf16_t, 1>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, argmax, (const bf16_t*)g, n, h, w, c, ho, wo, (bf16_t*)nullptr,
                               (const bf16_t*)z, gamma, mean, invstd, scale, shift, dgamma, dbeta, inv_m, nullptr, nblk, (bf16_t*)dz);
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(maxpool_bn_bwd_rows_apply_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, argmax, (const float*)g, n, h, w, c, ho, wo,
                           (const float*)z, gamma, mean, invstd, scale, shift, dgamma, dbeta, inv_m, (float*)dz);
    else
        hipLaunchKernelGGL(maxpool_bn_bwd_rows_apply_kernel<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, argmax, (const bf16_t*)g, n, h, w, c, ho, wo,
                           (const bf16_t*)z, gamma, mean, invstd, scale, shift, dgamma, dbeta, inv_m, (bf16_t*)dz);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

// head, training: feat (clips*T, hw, c) -> pooled (clips*T, c) fp32 [kept for backward] -> scores (clips, classes) -> loss
int mvf_head_train_fwd(const void* feat, int clips, int t, int hw, int c, const float* fc_w, const float* fc_b, int classes,
                       const long long* labels, const float* drop_mask, float* pooled, float* scores, float* dscores, float* loss_part, float* loss,
                       int dtype, void* stream);
int mvf_head_train_bwd(const float* dscores, const float* pooled, const float* fc_w, const float* drop_mask, int clips, int t, int hw, int c, int classes,
                       float* dfc_w, float* dfc_b, float* dpool_ws, void* dfeat, int dtype, void* stream);

}  // extern "C"

// frame-level fc + segment mean (training keeps per-frame pooled features for the weight gradient)
namespace {
// fc per frame, then consensus mean over the clip's T frames (tsn_clshead.py:92-96) -- both linear, so the clip's mean feature is
// formed once in LDS (fixed t order) and every class is one dot product with it: T x fewer multiply-adds and, with a workgroup per
// (clip, class range) instead of a wave per (clip, class), 8 x less L2 traffic than re-reading the T frame rows for every class
// (137 -> ~20 us at 32 clips x 400 classes x 2048 channels).
constexpr int kHeadSplit = 25;      // class ranges per clip (16 classes per workgroup at 400 classes: 4 per wave, all in flight)
__global__ __launch_bounds__(256) void head_fc_seg_kernel(const float* pooled, const float* w, const float* b, int clips, int T, int c, int classes, float* scores) {
    extern __shared__ float pc[];                   // [c] the clip's mean pooled feature
    const int clip = blockIdx.x, part = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float inv = 1.f / (float)T;
    for (int i = tid; i < c; i += 256) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += pooled[((long)clip * T + t) * c + i];
        pc[i] = s * inv;
    }
    __syncthreads();
    const int per = (classes + kHeadSplit - 1) / kHeadSplit;
    const int k_end = min(classes, (part + 1) * per);
    // a wave takes classes k, k+4, k+8, k+12 of the range together: four independent dot products keep four weight rows in flight
    const int kb = part * per + wave;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < c; i += 64) {
        const float f = pc[i];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = kb + 4 * u;
            s[u] += f * w[(long)(k < k_end ? k : kb < k_end ? kb : 0) * c + i];       // clamped row: unconditional load
        }
    }
    for (int u0 = 0; u0 * 16 < per; ++u0) {            // ranges longer than 16 classes: further rounds of four
        if (u0 > 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] = 0.f;
            for (int i = lane; i < c; i += 64) {
                const float f = pc[i];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb + 16 * u0 + 4 * u;
                    s[u] += f * w[(long)(k < k_end ? k : 0) * c + i];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = s[u];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            const int k = kb + 16 * u0 + 4 * u;
            if (lane == 0 && k < k_end) scores[(long)clip * classes + k] = v + (b ? b[k] : 0.f);
        }
    }
}
}  // namespace

extern "C" {

int mvf_head_train_fwd(const void* feat, int clips, int t, int hw, int c, const float* fc_w, const float* fc_b, int classes,
                       const long long* labels, const float* drop_mask, float* pooled, float* scores, float* dscores, float* loss_part, float* loss,
                       int dtype, void* stream) {
    MVF_REQUIRE(feat && fc_w && labels && pooled && scores && dscores && loss_part && loss && clips > 0 && t > 0 && hw > 0 && c > 0 && classes > 0, MVF_EINVAL, "head_train_fwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    dim3 g((c + 255) / 256, clips * t);
    if (dtype == MVF_F32) hipLaunchKernelGGL(frame_pool_kernel<float>, g, dim3(256), 0, st, (const float*)feat, hw, c, drop_mask, pooled);
    else hipLaunchKernelGGL(frame_pool_kernel<bf16_t>, g, dim3(256), 0, st, (const bf16_t*)feat, hw, c, drop_mask, pooled);
    MVF_LAUNCH_CHECK();
    MVF_REQUIRE((size_t)c * sizeof(float) <= 160 * 1024, MVF_EUNSUPPORTED, "head_train_fwd: c=%d does not fit the LDS feature buffer", c);
    hipLaunchKernelGGL(head_fc_seg_kernel, dim3(clips, kHeadSplit), dim3(256), (size_t)c * sizeof(float), st, pooled, fc_w, fc_b, clips, t, c, classes, scores);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(ce_loss_kernel, dim3(clips), dim3(256), 0, st, scores, labels, clips, classes, loss_part, dscores);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(mean_reduce_kernel, dim3(1), dim3(64), 0, st, loss_part, clips, loss);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_ce_loss(const float* scores, const long long* labels, int clips, int classes, float* dscores, float* loss_part, float* loss, void* stream) {
    MVF_REQUIRE(scores && labels && loss_part && loss && clips > 0 && classes > 0, MVF_EINVAL, "ce_loss: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_loss_kernel, dim3(clips), dim3(256), 0, st, scores, labels, clips, classes, loss_part, dscores);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(mean_reduce_kernel, dim3(1), dim3(64), 0, st, loss_part, clips, loss);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_head_train_bwd(const float* dscores, const float* pooled, const float* fc_w, const float* drop_mask, int clips, int t, int hw, int c, int classes,
                       float* dfc_w, float* dfc_b, float* dpool_ws, void* dfeat, int dtype, void* stream) {
    MVF_REQUIRE(dscores && pooled && fc_w && dfc_w && dfc_b && dpool_ws && dfeat, MVF_EINVAL, "head_train_bwd: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    MVF_REQUIRE(c % 4 == 0, MVF_ESHAPE, "head_train_bwd: c=%d must be a multiple of 4", c);
    // the clips' mean features borrow dpool_ws ([clips][c] floats), which the NEXT kernel of this call overwrites with its real content
    float* pcm = dpool_ws;
    hipLaunchKernelGGL(head_clipmean_kernel, dim3((c + 255) / 256, clips), dim3(256), 0, st, pooled, t, c, pcm);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_fc_bwd_w_kernel, dim3((c + 255) / 256, (classes + kHeadKc - 1) / kHeadKc), dim3(256), 0, st, dscores, pcm, clips, t, c, classes, dfc_w, dfc_b);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_dpool_kernel, dim3((c + 255) / 256, clips), dim3(256), 0, st, dscores, fc_w, classes, c, 1.0f / ((float)t * hw), dpool_ws);
    MVF_LAUNCH_CHECK();
    const long total = (long)clips * t * hw * c;
    const dim3 gdf((unsigned)std::min<long>(((long)hw * (c / 4) + 255) / 256, 64), (unsigned)(clips * t));
    if (dtype == MVF_F32) hipLaunchKernelGGL(head_dfeat_kernel<float>, gdf, dim3(256), 0, st, dpool_ws, drop_mask, t, hw, c, total, (float*)dfeat);
    else hipLaunchKernelGGL(head_dfeat_kernel<bf16_t>, gdf, dim3(256), 0, st, dpool_ws, drop_mask, t, hw, c, total, (bf16_t*)dfeat);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

// clip_grad_norm_(max_norm, L2) over the flat gradient, then p -= lr * (d + mom*buf), d = g*coef*gscale + wd*p, buf = mom*buf + d
// (torch.optim.SGD nesterov; first_step: buf = d).  norm_out[0] = total norm (after gscale), norm_out[1] = clip coefficient.
size_t mvf_sgd_workspace_bytes(long n) { return 1024 * sizeof(float); }

int mvf_sgd_nesterov_step(float* params, const float* grads, float* momentum_buf, long n, float grad_scale, float max_norm,
                          float lr, float momentum, float weight_decay, int first_step, float* norm_out, void* ws,
                          size_t ws_bytes, void* stream) {
    MVF_REQUIRE(params && grads && momentum_buf && norm_out && n > 0, MVF_EINVAL, "sgd_nesterov_step: bad argument");
    MVF_REQUIRE(ws && ws_bytes >= mvf_sgd_workspace_bytes(n), MVF_EWS, "sgd_nesterov_step: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    const int nb = (int)std::min<long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(sqsum_partial_kernel, dim3(nb), dim3(256), 0, st, grads, n, part);
    MVF_LAUNCH_CHECK();
    // the norm is of the scaled gradient (all-reduce sum / world happens before clipping, dist_utils.py:63-66)
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(64), 0, st, part, nb, max_norm, grad_scale, norm_out);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(grid_for(n)), dim3(256), 0, st, params, grads, momentum_buf, n, norm_out, grad_scale, lr, momentum, weight_decay, first_step);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_sgd_step_segments(float* params, const float* grads, float* momentum_buf, long n, float grad_scale, float max_norm, float lr,
                          float momentum, float weight_decay, int first_step, int nesterov, const mvf_sgd_segment_t* segments, int nseg,
                          float* norm_out, void* ws, size_t ws_bytes, void* stream) {
    MVF_REQUIRE(params && grads && momentum_buf && norm_out && segments && nseg > 0 && n > 0, MVF_EINVAL, "sgd_step_segments: bad argument");
    MVF_REQUIRE(ws && ws_bytes >= mvf_sgd_workspace_bytes(n), MVF_EWS, "sgd_step_segments: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    const int nb = (int)std::min<long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(sqsum_segments_kernel, dim3(nb), dim3(256), 0, st, grads, n, segments, nseg, part);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(64), 0, st, part, nb, max_norm, grad_scale, norm_out);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(sgd_segments_kernel, dim3(grid_for(n)), dim3(256), 0, st, params, grads, momentum_buf, n, norm_out, grad_scale, lr, momentum, weight_decay,
                       first_step, nesterov, segments, nseg);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // extern "C"
