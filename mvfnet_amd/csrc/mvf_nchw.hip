// MVF-proper on the NCHW (N*T, C, H, W) tensor: LDS-staged clip-channel volumes (gfx950).
//
// Replaces codes/models/modules/MVF.py:104-137 (view/transpose/split -> three depthwise Conv3d -> add ->
// BatchNorm3d -> HardSwish -> cat -> transpose -> contiguous) by a slice-only stencil: a workgroup owns the
// (clip n, channels c0..c0+KC) volume  T x KC x H x W  -- in NCHW that is T contiguous runs of KC*H*W
// elements, one per frame, C*H*W apart -- stages it once in LDS (coalesced, 16 B/lane when aligned), and
// every view's 3-tap window (t+-1 = +-KC*H*W, h+-1 = +-W, w+-1 = +-1 in LDS) is served from there, so HBM
// sees each slice element exactly once per pass.  Channels >= cs are never touched.
//
// Thread mapping in the compute phase: KC channels x TPC threads (KC*TPC = 256, both powers of two);
// a thread walks hw = j, j+TPC, ... and slides a (prev,cur,next) register window along t.
//
// Phases (one kernel template):
//   APPLY      out = hswish(scale*y + shift)                       (inference, and pass 2 of training)
//   STATS      per-(n,c) mean and centred M2 of y                   (pass 1 of training BN)
//   BWD_SUMS   per-(n,c) sum(du), sum(du*xhat), du = g*hswish'(u)   (BN backward reductions)
//   BWD_MAIN   dy -> LDS; dx slice = transposed stencil of dy; per-(n,c) 7 tap-gradient sums
// Cross-clip reductions are deterministic: partials [c][n][k] are summed in n order by tiny finalize kernels.
#include <algorithm>

#include "common.h"

namespace {

constexpr int kThreads = 256;
enum Phase { APPLY = 0, STATS = 1, BWD_SUMS = 2, BWD_MAIN = 3 };

struct MvfArgs {
    const void* x;
    void* out;
    const void* g;
    void* dx;
    const float* wt;
    const float* wh;
    const float* ww;
    const float* scale;   // folded BN: u = scale*y + shift ; nullptr <=> use_hs = False
    const float* shift;
    const float* mean;    // BWD: batch (training) or running (eval) mean
    const float* invstd;
    const float* gamma;
    const float* dsum;    // BWD_MAIN, training: [cs][2] = (dbeta, dgamma)
    float* part;          // partial sums out
    int nt, c, h, w, T, cs, mode, n_clips, kc, tpc, training;
    float inv_m;          // 1 / (n_clips*T*H*W)
};

template <int NV>
__device__ __forceinline__ void group_reduce(float (&v)[NV], int tpc, float* red) {
    const int lane_span = tpc < 64 ? tpc : 64;
    for (int off = 1; off < lane_span; off <<= 1) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += __shfl_xor(v[i], off, 64);
    }
    if (tpc > 64) {  // group spans tpc/64 whole waves: combine through LDS in fixed wave order
        const int wave = threadIdx.x >> 6;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) red[i * 4 + wave] = v[i];
        }
        __syncthreads();
        const int w0 = (threadIdx.x / tpc) * (tpc >> 6);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float s = 0.f;
            for (int k = 0; k < (tpc >> 6); ++k) s += red[i * 4 + w0 + k];
            v[i] = s;
        }
    }
}

template <typename ET>
__device__ __forceinline__ void stage_tile(float* tile, const ET* src, int T, long frame_stride, int p_eff, int P,
                                           bool vec_ok) {
    // tile[t][p] = src[t*frame_stride + p], p < p_eff ; consecutive lanes -> consecutive addresses
    if (vec_ok) {
        const int nv = p_eff >> 2;
        for (int t = 0; t < T; ++t) {
            const ET* s = src + (long)t * frame_stride;
            float* d = tile + t * P;
            for (int v = threadIdx.x; v < nv; v += kThreads) {
                float4 q = ld4(s + 4 * v);
                *reinterpret_cast<float4*>(d + 4 * v) = q;
            }
        }
    } else {
        for (int t = 0; t < T; ++t) {
            const ET* s = src + (long)t * frame_stride;
            float* d = tile + t * P;
            for (int p = threadIdx.x; p < p_eff; p += kThreads) d[p] = ldf(s + p);
        }
    }
}

template <typename ET, int PHASE>
__global__ __launch_bounds__(kThreads) void mvf_nchw_kernel(MvfArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HW = a.h * a.w, W = a.w, H = a.h, T = a.T;
    const int groups = (a.cs + a.kc - 1) / a.kc;
    const int n = blockIdx.x / groups;
    const int c0 = (blockIdx.x % groups) * a.kc;
    const int kc_eff = min(a.kc, a.cs - c0);
    const int P = ((a.kc * HW + 3) & ~3);          // LDS frame pitch (floats), 16 B aligned
    const int p_eff = kc_eff * HW;
    float* tile = smem;                              // [T][P]   x slice
    float* dyt = smem + (size_t)T * P;               // [T][P]   dy (BWD_MAIN only)
    float* red = (PHASE == BWD_MAIN) ? dyt + (size_t)T * P : dyt;   // [8][4] reduction scratch

    const long frame_stride = (long)a.c * HW;
    const long base0 = ((long)n * T * a.c + c0) * HW;   // element offset of (n, t=0, c0, 0, 0)
    const bool vec_ok = ((frame_stride & 3) == 0) && ((base0 & 3) == 0) && ((p_eff & 3) == 0);
    const ET* x = reinterpret_cast<const ET*>(a.x);
    stage_tile<ET>(tile, x + base0, T, frame_stride, p_eff, P, vec_ok);
    __syncthreads();

    const int k = threadIdx.x / a.tpc, j = threadIdx.x % a.tpc;
    const bool kvalid = k < kc_eff;
    const int ch = c0 + (kvalid ? k : 0);
    const bool vh = a.mode & MVF_VIEW_H, vw = a.mode & MVF_VIEW_W;
    const float wt0 = a.wt[ch * 3 + 0], wt1 = a.wt[ch * 3 + 1], wt2 = a.wt[ch * 3 + 2];
    const float wh0 = vh ? a.wh[ch * 3 + 0] : 0.f, wh1 = vh ? a.wh[ch * 3 + 1] : 0.f, wh2 = vh ? a.wh[ch * 3 + 2] : 0.f;
    const float ww0 = vw ? a.ww[ch * 3 + 0] : 0.f, ww1 = vw ? a.ww[ch * 3 + 1] : 0.f, ww2 = vw ? a.ww[ch * 3 + 2] : 0.f;
    const bool hs = a.scale != nullptr;
    const float sc = hs ? a.scale[ch] : 1.f, sh = hs ? a.shift[ch] : 0.f;

    // y at (t, k, hw) from the LDS tile; the three views are summed in the reference's order
    // shift_conv(x) + h_conv(x) + w_conv(x)  (MVF.py:120)
    auto stencil = [&](const float* tl, int t, int idx, int hh, int wv, float prev, float cur, float next) {
        const float* f = tl + t * P + idx;
        float yt = wt0 * prev + wt1 * cur + wt2 * next;
        float up = hh > 0 ? f[-W] : 0.f, dn = hh < H - 1 ? f[W] : 0.f;
        float lf = wv > 0 ? f[-1] : 0.f, rt = wv < W - 1 ? f[1] : 0.f;
        float yh = wh0 * up + wh1 * cur + wh2 * dn;
        float yw = ww0 * lf + ww1 * cur + ww2 * rt;
        return (yt + yh) + yw;
    };

    if constexpr (PHASE == APPLY) {
        ET* out = reinterpret_cast<ET*>(a.out);
        if (kvalid) {
            for (int hw = j; hw < HW; hw += a.tpc) {
                const int hh = hw / W, wv = hw - hh * W, idx = k * HW + hw;
                float prev = 0.f, cur = tile[idx];
                for (int t = 0; t < T; ++t) {
                    float next = (t + 1 < T) ? tile[(t + 1) * P + idx] : 0.f;
                    float y = stencil(tile, t, idx, hh, wv, prev, cur, next);
                    if (hs) {
                        float u = sc * y + sh;
                        y = u * (fminf(fmaxf(u + 3.0f, 0.0f), 6.0f) / 6.0f);
                    }
                    stf(out + base0 + (long)t * frame_stride + idx, y);
                    prev = cur;
                    cur = next;
                }
            }
        }
    } else if constexpr (PHASE == STATS) {
        float v[1] = {0.f};
        if (kvalid) {
            for (int hw = j; hw < HW; hw += a.tpc) {
                const int hh = hw / W, wv = hw - hh * W, idx = k * HW + hw;
                float prev = 0.f, cur = tile[idx];
                for (int t = 0; t < T; ++t) {
                    float next = (t + 1 < T) ? tile[(t + 1) * P + idx] : 0.f;
                    v[0] += stencil(tile, t, idx, hh, wv, prev, cur, next);
                    prev = cur;
                    cur = next;
                }
            }
        }
        group_reduce<1>(v, a.tpc, red);
        const float mloc = v[0] / (float)(T * HW);
        float q[1] = {0.f};
        if (kvalid) {
            for (int hw = j; hw < HW; hw += a.tpc) {
                const int hh = hw / W, wv = hw - hh * W, idx = k * HW + hw;
                float prev = 0.f, cur = tile[idx];
                for (int t = 0; t < T; ++t) {
                    float next = (t + 1 < T) ? tile[(t + 1) * P + idx] : 0.f;
                    float d = stencil(tile, t, idx, hh, wv, prev, cur, next) - mloc;
                    q[0] += d * d;
                    prev = cur;
                    cur = next;
                }
            }
        }
        group_reduce<1>(q, a.tpc, red);
        if (kvalid && j == 0) {
            float* p = a.part + ((long)ch * a.n_clips + n) * 2;
            p[0] = mloc;
            p[1] = q[0];
        }
    } else if constexpr (PHASE == BWD_SUMS) {
        const ET* g = reinterpret_cast<const ET*>(a.g);
        const float mu = a.mean[ch], rs = a.invstd[ch];
        float v[2] = {0.f, 0.f};
        if (kvalid) {
            for (int hw = j; hw < HW; hw += a.tpc) {
                const int hh = hw / W, wv = hw - hh * W, idx = k * HW + hw;
                float prev = 0.f, cur = tile[idx];
                for (int t = 0; t < T; ++t) {
                    float next = (t + 1 < T) ? tile[(t + 1) * P + idx] : 0.f;
                    float y = stencil(tile, t, idx, hh, wv, prev, cur, next);
                    float gv = ldf(g + base0 + (long)t * frame_stride + idx);
                    float du = gv * hswish_grad_f(sc * y + sh);
                    v[0] += du;
                    v[1] += du * ((y - mu) * rs);
                    prev = cur;
                    cur = next;
                }
            }
        }
        group_reduce<2>(v, a.tpc, red);
        if (kvalid && j == 0) {
            float* p = a.part + ((long)ch * a.n_clips + n) * 2;
            p[0] = v[0];
            p[1] = v[1];
        }
    } else {  // BWD_MAIN
        const ET* g = reinterpret_cast<const ET*>(a.g);
        ET* dx = reinterpret_cast<ET*>(a.dx);
        float mu = 0.f, rs = 1.f, gr = 1.f, db = 0.f, dg = 0.f;
        if (hs) {
            mu = a.mean[ch];
            rs = a.invstd[ch];
            gr = a.gamma[ch] * rs;
            if (a.training) {
                db = a.dsum[ch * 2 + 0] * a.inv_m;
                dg = a.dsum[ch * 2 + 1] * a.inv_m;
            }
        }
        // pass A: dy = dL/dy for every element of the volume -> LDS
        if (kvalid) {
            for (int hw = j; hw < HW; hw += a.tpc) {
                const int hh = hw / W, wv = hw - hh * W, idx = k * HW + hw;
                float prev = 0.f, cur = tile[idx];
                for (int t = 0; t < T; ++t) {
                    float next = (t + 1 < T) ? tile[(t + 1) * P + idx] : 0.f;
                    float gv = ldf(g + base0 + (long)t * frame_stride + idx);
                    float dyv = gv;
                    if (hs) {
                        float y = stencil(tile, t, idx, hh, wv, prev, cur, next);
                        float du = gv * hswish_grad_f(sc * y + sh);
                        float xh = (y - mu) * rs;
                        dyv = gr * (du - db - xh * dg);          // db = dg = 0 in eval mode
                    }
                    dyt[t * P + idx] = dyv;
                    prev = cur;
                    cur = next;
                }
            }
        }
        __syncthreads();
        // pass B: dx slice (transposed stencil: taps flipped) + the 7 distinct tap-gradient sums
        float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // t0, t2, h0, h2, w0, w2, centre
        if (kvalid) {
            for (int hw = j; hw < HW; hw += a.tpc) {
                const int hh = hw / W, wv = hw - hh * W, idx = k * HW + hw;
                float dprev = 0.f, dcur = dyt[idx];
                float xprev = 0.f, xcur = tile[idx];
                for (int t = 0; t < T; ++t) {
                    const bool has_next = t + 1 < T;
                    float dnext = has_next ? dyt[(t + 1) * P + idx] : 0.f;
                    float xnext = has_next ? tile[(t + 1) * P + idx] : 0.f;
                    const float* fd = dyt + t * P + idx;
                    const float* fx = tile + t * P + idx;
                    float dup = hh > 0 ? fd[-W] : 0.f, ddn = hh < H - 1 ? fd[W] : 0.f;
                    float dlf = wv > 0 ? fd[-1] : 0.f, drt = wv < W - 1 ? fd[1] : 0.f;
                    float xup = hh > 0 ? fx[-W] : 0.f, xdn = hh < H - 1 ? fx[W] : 0.f;
                    float xlf = wv > 0 ? fx[-1] : 0.f, xrt = wv < W - 1 ? fx[1] : 0.f;
                    // ds[p] = sum_j w[j] * dy[p - (j-1)]
                    float dst = wt0 * dnext + wt1 * dcur + wt2 * dprev;
                    float dsh = wh0 * ddn + wh1 * dcur + wh2 * dup;
                    float dsw = ww0 * drt + ww1 * dcur + ww2 * dlf;
                    stf(dx + base0 + (long)t * frame_stride + idx, (dst + dsh) + dsw);
                    // dw[j] += dy[p] * s[p + (j-1)]
                    s[0] += dcur * xprev;
                    s[1] += dcur * xnext;
                    s[2] += dcur * xup;
                    s[3] += dcur * xdn;
                    s[4] += dcur * xlf;
                    s[5] += dcur * xrt;
                    s[6] += dcur * xcur;
                    dprev = dcur;
                    dcur = dnext;
                    xprev = xcur;
                    xcur = xnext;
                }
            }
        }
        group_reduce<7>(s, a.tpc, red);
        if (kvalid && j == 0) {
            float* p = a.part + ((long)ch * a.n_clips + n) * 7;
#pragma unroll
            for (int i = 0; i < 7; ++i) p[i] = s[i];
        }
    }
}

// ---- tiny per-channel kernels ----------------------------------------------------------------
__global__ void mvf_fold_bn(int cs, const float* gamma, const float* beta, const float* mean, const float* invstd,
                            float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cs) return;
    float s = gamma[c] * invstd[c];
    scale[c] = s;
    shift[c] = beta[c] - mean[c] * s;
}

// merge per-clip (mean_i, M2_i) (equal counts) -> batch statistics; Chan's parallel formula in fixed order
__global__ void mvf_stats_finalize(int cs, int n_clips, int cnt, const float* part, const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean,
                                   float* running_var, float* save_mean, float* save_invstd, float* scale,
                                   float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cs) return;
    const float* p = part + (long)c * n_clips * 2;
    double msum = 0.0;
    for (int i = 0; i < n_clips; ++i) msum += p[2 * i];
    const double mean = msum / n_clips;
    double m2 = 0.0;
    for (int i = 0; i < n_clips; ++i) {
        double d = p[2 * i] - mean;
        m2 += p[2 * i + 1] + d * d * cnt;
    }
    const double m = (double)n_clips * cnt;
    const float var = (float)(m2 / m);
    const float invstd = 1.0f / sqrtf(var + eps);
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    const float s = gamma[c] * invstd;
    scale[c] = s;
    shift[c] = beta[c] - (float)mean * s;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const float unb = (float)(m2 / (m > 1.0 ? m - 1.0 : 1.0));
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
}

__global__ void mvf_sum_partials(int cs, int n_clips, int nv, const float* part, float* out) {
    // out[c][v] = sum_n part[c][n][v]  (fixed order)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cs * nv) return;
    int c = i / nv, v = i - c * nv;
    const float* p = part + (long)c * n_clips * nv + v;
    float s = 0.f;
    for (int k = 0; k < n_clips; ++k) s += p[(long)k * nv];
    out[i] = s;
}

__global__ void mvf_scatter_dw(int cs, int mode, const float* s7, const float* dsum, float* dwt, float* dwh,
                               float* dww, float* dgamma, float* dbeta) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cs) return;
    const float* s = s7 + c * 7;
    dwt[c * 3 + 0] = s[0]; dwt[c * 3 + 1] = s[6]; dwt[c * 3 + 2] = s[1];
    if (dwh) {
        const bool on = mode & MVF_VIEW_H;
        dwh[c * 3 + 0] = on ? s[2] : 0.f; dwh[c * 3 + 1] = on ? s[6] : 0.f; dwh[c * 3 + 2] = on ? s[3] : 0.f;
    }
    if (dww) {
        const bool on = mode & MVF_VIEW_W;
        dww[c * 3 + 0] = on ? s[4] : 0.f; dww[c * 3 + 1] = on ? s[6] : 0.f; dww[c * 3 + 2] = on ? s[5] : 0.f;
    }
    if (dsum) {
        if (dbeta) dbeta[c] = dsum[c * 2 + 0];
        if (dgamma) dgamma[c] = dsum[c * 2 + 1];
    }
}

// copy channels [cs, c) of every image: rows of (c-cs)*hw contiguous elements, c*hw apart (NCHW)
template <typename ET>
__global__ void copy_tail_rows(const ET* src, ET* dst, long rows, long row_len, long row_stride, long row_off) {
    const long total = rows * row_len;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / row_len, k = i - r * row_len;
        dst[r * row_stride + row_off + k] = src[r * row_stride + row_off + k];
    }
}

struct Plan {
    int kc, tpc, P, groups;
    size_t lds_fwd, lds_bwd;
};

Plan make_plan(const mvf_desc_t* d) {
    Plan p;
    const int HW = d->h * d->w;
    const long vol = (long)d->n_segment * HW;        // floats per channel volume
    int kc = 1;
    while (kc < 16 && kc * 2 <= d->cs && (long)(kc * 2) * vol <= 6144) kc *= 2;   // ~24 KB tiles
    p.kc = kc;
    p.tpc = kThreads / kc;
    p.P = ((kc * HW + 3) & ~3);
    p.groups = (d->cs + kc - 1) / kc;
    p.lds_fwd = ((size_t)d->n_segment * p.P + 32) * sizeof(float);
    p.lds_bwd = ((size_t)2 * d->n_segment * p.P + 32) * sizeof(float);
    return p;
}

constexpr size_t kMaxLds = 160 * 1024;

template <typename ET, int PHASE>
int launch_phase(const MvfArgs& a, int blocks, size_t lds, hipStream_t st) {
    auto kern = mvf_nchw_kernel<ET, PHASE>;
    if (lds > 64 * 1024) MVF_HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kThreads), lds, st, a);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

template <int PHASE>
int launch_phase_dt(int dtype, const MvfArgs& a, int blocks, size_t lds, hipStream_t st) {
    if (dtype == MVF_F32) return launch_phase<float, PHASE>(a, blocks, lds, st);
    return launch_phase<bf16_t, PHASE>(a, blocks, lds, st);
}

int copy_tail(const mvf_desc_t* d, const void* src, void* dst, hipStream_t st) {
    if (src == dst || d->cs >= d->c) return MVF_OK;
    const long HW = (long)d->h * d->w;
    const long rows = d->nt, row_len = (long)(d->c - d->cs) * HW, stride = (long)d->c * HW, off = (long)d->cs * HW;
    const int blocks = (int)std::min<long>((rows * row_len + 255) / 256, 256L * 16);
    if (d->dtype == MVF_F32)
        hipLaunchKernelGGL(copy_tail_rows<float>, dim3(blocks), dim3(256), 0, st, (const float*)src, (float*)dst, rows, row_len, stride, off);
    else
        hipLaunchKernelGGL(copy_tail_rows<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, rows, row_len, stride, off);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

MvfArgs base_args(const mvf_desc_t* d, const Plan& p) {
    MvfArgs a = {};
    a.nt = d->nt; a.c = d->c; a.h = d->h; a.w = d->w; a.T = d->n_segment; a.cs = d->cs; a.mode = d->mode;
    a.n_clips = d->nt / d->n_segment; a.kc = p.kc; a.tpc = p.tpc;
    a.inv_m = 1.0f / ((float)a.n_clips * d->n_segment * d->h * d->w);
    return a;
}

}  // namespace

// ---- entry points used by mvf_abi.cpp (NCHW) ---------------------------------------------------
int mvf_nchw_check(const mvf_desc_t* d, bool bwd) {
    Plan p = make_plan(d);
    size_t need = bwd ? p.lds_bwd : p.lds_fwd;
    MVF_REQUIRE(need <= kMaxLds, MVF_EUNSUPPORTED,
                "mvf(NCHW): clip-channel volume T*H*W = %d*%d*%d needs %zu B of LDS (> 160 KiB)", d->n_segment, d->h,
                d->w, need);
    return MVF_OK;
}

size_t mvf_nchw_ws_fwd_train(const mvf_desc_t* d) {
    const size_t n = d->nt / d->n_segment;
    return align_up(((size_t)d->cs * n * 2 + 2 * (size_t)d->cs) * sizeof(float), 256);
}

size_t mvf_nchw_ws_bwd(const mvf_desc_t* d) {
    const size_t n = d->nt / d->n_segment;
    return align_up(((size_t)d->cs * n * 9 + (size_t)d->cs * (2 + 2 + 7)) * sizeof(float), 256);
}

int mvf_nchw_fwd_infer(const mvf_desc_t* d, const void* x, void* out, const float* wt, const float* wh,
                       const float* ww, const float* scale, const float* shift, hipStream_t st) {
    int rc = mvf_nchw_check(d, false);
    if (rc) return rc;
    Plan p = make_plan(d);
    MvfArgs a = base_args(d, p);
    a.x = x; a.out = out; a.wt = wt; a.wh = wh; a.ww = ww; a.scale = scale; a.shift = shift;
    rc = launch_phase_dt<APPLY>(d->dtype, a, a.n_clips * p.groups, p.lds_fwd, st);
    if (rc) return rc;
    return copy_tail(d, x, out, st);
}

int mvf_nchw_fwd_train(const mvf_desc_t* d, const void* x, void* out, const float* wt, const float* wh,
                       const float* ww, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* save_mean, float* save_invstd, void* ws,
                       hipStream_t st) {
    int rc = mvf_nchw_check(d, false);
    if (rc) return rc;
    Plan p = make_plan(d);
    MvfArgs a = base_args(d, p);
    float* part = (float*)ws;
    float* scale = part + (size_t)d->cs * a.n_clips * 2;
    float* shift = scale + d->cs;
    a.x = x; a.out = out; a.wt = wt; a.wh = wh; a.ww = ww; a.part = part;
    rc = launch_phase_dt<STATS>(d->dtype, a, a.n_clips * p.groups, p.lds_fwd, st);
    if (rc) return rc;
    hipLaunchKernelGGL(mvf_stats_finalize, dim3((d->cs + 63) / 64), dim3(64), 0, st, d->cs, a.n_clips,
                       d->n_segment * d->h * d->w, part, gamma, beta, eps, momentum, running_mean, running_var,
                       save_mean, save_invstd, scale, shift);
    MVF_LAUNCH_CHECK();
    a.scale = scale; a.shift = shift;
    rc = launch_phase_dt<APPLY>(d->dtype, a, a.n_clips * p.groups, p.lds_fwd, st);
    if (rc) return rc;
    return copy_tail(d, x, out, st);
}

int mvf_nchw_bwd(const mvf_desc_t* d, const void* g, const void* x, const float* wt, const float* wh,
                 const float* ww, const float* gamma, const float* beta, const float* mean, const float* invstd,
                 int training, void* dx, float* dwt, float* dwh, float* dww, float* dgamma, float* dbeta, void* ws,
                 hipStream_t st) {
    int rc = mvf_nchw_check(d, true);
    if (rc) return rc;
    Plan p = make_plan(d);
    MvfArgs a = base_args(d, p);
    const size_t n = a.n_clips, cs = d->cs;
    float* part2 = (float*)ws;              // [cs][n][2]
    float* part7 = part2 + cs * n * 2;      // [cs][n][7]
    float* scale = part7 + cs * n * 7;      // [cs]
    float* shift = scale + cs;
    float* dsum = shift + cs;               // [cs][2]
    float* s7 = dsum + cs * 2;              // [cs][7]
    const bool hs = gamma != nullptr;
    a.x = x; a.g = g; a.dx = dx; a.wt = wt; a.wh = wh; a.ww = ww; a.training = training;
    const int blocks = a.n_clips * p.groups;
    if (hs) {
        hipLaunchKernelGGL(mvf_fold_bn, dim3((d->cs + 63) / 64), dim3(64), 0, st, d->cs, gamma, beta, mean, invstd, scale, shift);
        MVF_LAUNCH_CHECK();
        a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.gamma = gamma;
        a.part = part2;
        rc = launch_phase_dt<BWD_SUMS>(d->dtype, a, blocks, p.lds_fwd, st);
        if (rc) return rc;
        hipLaunchKernelGGL(mvf_sum_partials, dim3((d->cs * 2 + 63) / 64), dim3(64), 0, st, d->cs, a.n_clips, 2, part2, dsum);
        MVF_LAUNCH_CHECK();
        a.dsum = dsum;
    }
    a.part = part7;
    rc = launch_phase_dt<BWD_MAIN>(d->dtype, a, blocks, p.lds_bwd, st);
    if (rc) return rc;
    hipLaunchKernelGGL(mvf_sum_partials, dim3((d->cs * 7 + 63) / 64), dim3(64), 0, st, d->cs, a.n_clips, 7, part7, s7);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(mvf_scatter_dw, dim3((d->cs + 63) / 64), dim3(64), 0, st, d->cs, d->mode, s7, hs ? dsum : nullptr, dwt, dwh, dww, dgamma, dbeta);
    MVF_LAUNCH_CHECK();
    return copy_tail(d, g, dx, st);
}
