// BatchNorm backward apply FUSED with the weight gradient of the pointwise conv that produced the BatchNorm's input (bf16 storage).
//
// Reference arithmetic: autograd of Bottleneck.forward (codes/models/backbones/resnet.py:213-244): out = relu(bn3(conv3(a2)) + identity),
// a1 = relu(bn1(conv1(x))).  For z = conv1x1(x, W) and the BatchNorm behind it, backward needs
//     dz[m][co] = gamma * invstd * (gm - dbeta / M - xhat * dgamma / M)          (bn_bwd_apply_kernel, train_ops.hip)
//     dW[co][k] = sum_m dz[m][co] * x[m][k]                                         (wgrad_bf16_kernel, wgrad_nhwc.hip)
// The un-fused step writes dz, then reads it twice (data gradient, weight gradient).  Layer1 / layer2's pointwise convs are byte-bound
// on both passes (Cout x Cin <= 64 K: the whole dW accumulator fits one workgroup's registers), so here the pass that FORMS dz also
// contracts it: a persistent workgroup (8 waves, one per CU) walks 64-pixel chunks of its row range; per chunk every thread forms its
// dz units with bn_bwd_apply_kernel's arithmetic (expression by expression: dz is BIT-identical to the un-fused kernel), stores them to
// global memory for the data gradient AND into an LDS tile in the weight-gradient kernel's [pixel][channel] XOR-swizzled image; the
// conv input's chunk arrives by LDS-DMA (double-buffered, issued one chunk ahead, ordered BEFORE the chunk's g / z loads so that their
// arrival implies its arrival); then 32x32x16 bf16 MFMAs on transpose reads (ds_read_b64_tr_b16) accumulate the CT x KT tile of dW in
// registers.  One fp32 partial slab per workgroup, summed in fixed order by wgrad_reduce_kernel (deterministic, no atomics).
// dz is read ONCE afterwards (by the data gradient) and the side stream loses a byte-bound GEMM that re-read it.
//
// NBN = 2: the PAIRED backward of a downsample block (bn_bwd_apply2_kernel): out = relu(bn3(z3) + bn_d(z_d)), both BatchNorms receive
// the same gated gradient; one pass over g and the sign bits forms dz3 and dz_d and contracts them with a2 and the block input.
// MM = 4: ReLU gate from the block output's sign bits ([M][C/4] bytes); MM = 2: gate recomputed from scale * z + shift > 0 (bn1 / bn2).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kT = 512;          // threads per workgroup: 8 waves, two per SIMD -> up to 256 VGPRs each
constexpr int CH = 64;           // pixels per chunk = 4 MFMA k-steps
constexpr unsigned kOOB = 0x80000000u;

template <int U>
__device__ __forceinline__ int swz16(int row) {          // XOR swizzle of the 16-byte units of an LDS row (wgrad_nhwc.hip)
    static_assert(U == 8 || U == 16 || U == 32, "64-, 128- or 256-channel tile rows");
    return U >= 16 ? (row & 3) << 2 : ((row >> 1) & 1) << 2;
}

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
__device__ __forceinline__ void ldp8(const float* p, int c, float (&f)[8], float fill = 0.f) {
#pragma unroll
    for (int j = 0; j < 8; j += 4) {
        const float4 v = p ? *reinterpret_cast<const float4*>(p + c + j) : make_float4(fill, fill, fill, fill);
        f[j] = v.x; f[j + 1] = v.y; f[j + 2] = v.z; f[j + 3] = v.w;
    }
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CT, int KT, int NBN, int NX, int MM>
__global__ __launch_bounds__(kT) void bnbwd_wgrad_kernel(const mvf_internal::BnBwdWgradArgs a) {
    static_assert(MM == 2 || MM == 4, "gate from scale * z + shift (2) or from the sign bits (4)");
    static_assert(NX >= 1 && NX <= NBN, "NX = how many of the BatchNorms' convs are contracted here (the first NX)");
    constexpr int WM = CT >= 128 ? 4 : 2, WN = 8 / WM;              // wave grid over the CT x KT accumulator tile
    constexpr int TM = CT / 32 / WM, TN = KT / 32 / WN;
    static_assert(TM >= 1 && TN >= 1 && TM * WM * 32 == CT && TN * WN * 32 == KT, "tile does not divide over 8 waves");
    constexpr int PA = CT * 2, PB = KT * 2;                          // LDS row pitches (bytes)
    constexpr int UA = CT / 8, UB = KT / 8;                          // 16-byte units per row
    constexpr int RPA = kT / UA, NA = CH / RPA;                      // dz: rows per pass, passes per chunk
    constexpr int RPB = kT / UB, NB = CH / RPB;                      // x: the same for the DMA
    static_assert(NA >= 1 && NB >= 1, "chunk smaller than one pass");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ds = smem;                                                 // [NX][CH][PA]
    char* Xs = smem + NX * CH * PA;                                  // [NX][2][CH][PB]

    // workgroup -> (row split, column tile, k tile): the tiles of one split are neighbours in a contiguous per-XCD range (they re-read
    // the same rows of g / z / x through one L2)
    int split, ctile, ktile;
    {
        const int nwg = (int)gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int tiles = a.ctiles * a.ktiles;
        split = id / tiles;
        const int t = id - split * tiles;
        ctile = t / a.ktiles;
        ktile = t - ctile * a.ktiles;
    }
    const int c0 = ctile * CT, k0 = ktile * KT;
    const int m_begin = split * a.rows_per_split, m_end = min(a.M, m_begin + a.rows_per_split);
    const int nchunks = (m_end - m_begin + CH - 1) / CH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- this thread's dz units: channel group q (8 channels) of rows r0 + RPA * i ----
    const int q = tid % UA, r0 = tid / UA;
    const int c = c0 + q * 8;
    const float inv_m = 1.0f / (float)a.M;
    float ca[NBN][8], cd[NBN][8], ck[NBN][8], cmu[NBN][8], sc[8], sh[8];
#pragma unroll
    for (int b = 0; b < NBN; ++b) {
        float ga[8], rs[8], dg[8], db[8];
        ldp8(a.gamma[b], c, ga); ldp8(a.invstd[b], c, rs); ldp8(a.dgamma[b], c, dg); ldp8(a.dbeta[b], c, db); ldp8(a.mean[b], c, cmu[b]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { ca[b][j] = ga[j] * rs[j]; cd[b][j] = db[j] * inv_m; ck[b][j] = rs[j] * dg[j] * inv_m; }
    }
    ldp8(MM == 2 ? a.scale : nullptr, c, sc);
    ldp8(MM == 2 ? a.shift : nullptr, c, sh);

    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)a.g, 0, (unsigned)min((long)a.M * a.g_pitch * 2, 0x7ffffff0L), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_z[NBN], rs_dz[NBN];
#pragma unroll
    for (int b = 0; b < NBN; ++b) {
        rs_z[b] = __builtin_amdgcn_make_buffer_rsrc((void*)a.z[b], 0, (unsigned)min((long)a.M * a.C * 2, 0x7ffffff0L), 0x00020000);
        rs_dz[b] = __builtin_amdgcn_make_buffer_rsrc(a.dz[b], 0, (unsigned)min((long)a.M * a.C * 2, 0x7ffffff0L), 0x00020000);
    }
    const bool store_dz = ktile == 0;                                // every k tile forms dz, one of them stores it

    // ---- the conv input's chunk by LDS-DMA: lane at LDS position (row, p) fetches source unit p ^ swz16(row) ----
    const int rb0 = tid / UB;
    const int qb = (tid % UB) ^ swz16<UB>(rb0);                      // constant per thread: every pass advances the row by a multiple of 4
    i32x4 gs_x[NX];
    unsigned lds_x[NX];
#pragma unroll
    for (int b = 0; b < NX; ++b) {
        gs_x[b] = rsrc_words(a.x[b], (unsigned)min((long)a.M * a.xps[b] * 2, 0x7ffffff0L));
        lds_x[b] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)(Xs + b * 2 * CH * PB) +
                                                            (wave * 64 / UB) * PB);
    }
    auto dma_x = [&](int cc, int buf) {
        const int mc = m_begin + cc * CH;
#pragma unroll
        for (int b = 0; b < NX; ++b) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int m = mc + rb0 + RPB * i;
                const unsigned off = m < m_end ? (unsigned)(m * a.xps[b] + k0 + qb * 8) * 2u : kOOB;
                glds16(gs_x[b], lds_x[b] + (unsigned)((buf * CH + RPB * i) * PB), off);
            }
        }
    };

    struct Stage {
        u32x4 g[NA], z[NBN][NA];
        unsigned mb[NA];
    } st;
    auto load_chunk = [&](int cc) {
        const int mc = m_begin + cc * CH;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = mc + r0 + RPA * i;
            const bool ok = m < m_end;
            st.g[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, ok ? (unsigned)(m * a.g_pitch + c) * 2u : kOOB, 0, 0);
#pragma unroll
            for (int b = 0; b < NBN; ++b) st.z[b][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_z[b], ok ? (unsigned)(m * a.C + c) * 2u : kOOB, 0, 0);
            if constexpr (MM == 4) {
                const long mr = ok ? m : (long)a.M - 1;              // unconditional load on a clamped row
                st.mb[i] = *reinterpret_cast<const unsigned short*>(reinterpret_cast<const unsigned char*>(a.bits) + mr * (a.C / 4) + c / 4);
            }
        }
    };
    auto form_dz = [&](int cc) {                                     // stage -> dz: global memory + the LDS tile
        const int mc = m_begin + cc * CH;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = r0 + RPA * i, m = mc + row;
            float gv[8];
            unpack8(st.g[i], gv);
            float zv[NBN][8];
#pragma unroll
            for (int b = 0; b < NBN; ++b) unpack8(st.z[b][i], zv[b]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (MM == 4) gv[j] = ((st.mb[i] >> (j + (j >= 4 ? 4 : 0))) & 1u) ? gv[j] : 0.f;
                else gv[j] = (zv[0][j] * sc[j] + sh[j]) > 0.f ? gv[j] : 0.f;
            }
#pragma unroll
            for (int b = 0; b < NBN; ++b) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = ca[b][j] * (gv[j] - cd[b][j] - (zv[b][j] - cmu[b][j]) * ck[b][j]);
                const u32x4 p = pack8(o);
                if (store_dz) __builtin_amdgcn_raw_buffer_store_b128(p, rs_dz[b], m < m_end ? (unsigned)(m * a.C + c) * 2u : kOOB, 0, 0);
                if (b < NX) *reinterpret_cast<u32x4*>(Ds + (b * CH + row) * PA + ((q ^ swz16<UA>(row)) * 16)) = p;
            }
        }
    };

    // ---- transpose-read offsets (wgrad_bf16_kernel): group g = lane >> 4 supplies pixel rows (i >> 2) + 8 * (g >> 1) and channel quad
    // 16 * (g & 1) + 4 * (i & 3) of a 32-channel block ----
    const int tg = lane >> 4, ti = lane & 15;
    const int trow = (ti >> 2) + 8 * (tg >> 1);
    const int tunit = 2 * (tg & 1) + ((ti & 3) >> 1), thalf = ti & 1;
    int offA[TM], offB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) offA[i] = trow * PA + ((((wm * TM + i) * 4 + tunit) ^ swz16<UA>(trow)) * 16) + thalf * 8;
#pragma unroll
    for (int j = 0; j < TN; ++j) offB[j] = trow * PB + ((((wn * TN + j) * 4 + tunit) ^ swz16<UB>(trow)) * 16) + thalf * 8;
    auto gather = [&](const char* p, int pitch) {
        typedef short v4s __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4s* lds_v4s;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p + 4 * pitch));
        bf16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };
    f32x16 acc[NX][TM][TN];
#pragma unroll
    for (int b = 0; b < NX; ++b)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][i][j][r] = 0.f;
    auto contract = [&](int buf) {
#pragma unroll
        for (int b = 0; b < NX; ++b) {
            const char* dsb = Ds + b * CH * PA;
            const char* xsb = Xs + (b * 2 + buf) * CH * PB;
#pragma unroll
            for (int ks = 0; ks < CH / 16; ++ks) {
                bf16x8_t fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = gather(dsb + ks * 16 * PA + offA[i], PA);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = gather(xsb + ks * 16 * PB + offB[j], PB);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[b][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[b][i][j], 0, 0, 0);
            }
        }
    };

    // ---- the chunk loop.  Issue order inside an iteration: DMA of x(c + 1), loads of g / z(c + 1), [next iteration] stores of dz(c + 1):
    // VMEM reads return in order, so when the stage's last load has arrived the chunk's DMA has too -- no vmcnt(0) that would also drain
    // the dz stores.  Two LDS-only barriers per chunk (A: the MFMAs of chunk c - 1 have read the dz tile; B: dz(c) and x(c) are in LDS).
    if (nchunks > 0) {
        dma_x(0, 0);
        load_chunk(0);
    }
    for (int cc = 0; cc < nchunks; ++cc) {
        if (cc > 0) lds_barrier();                                   // A
        form_dz(cc);
        if (cc + 1 < nchunks) {
            dma_x(cc + 1, (cc + 1) & 1);
            load_chunk(cc + 1);
        }
        lds_barrier();                                               // B
        contract(cc & 1);
    }

    // ---- one fp32 partial slab per workgroup: part[split][C][K] (wgrad_reduce_kernel's layout) ----
    const int lr = lane >> 5, lc = lane & 31;
#pragma unroll
    for (int b = 0; b < NX; ++b) {
        float* out = a.part[b] + (long)split * a.C * a.K;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = k0 + (wn * TN + j) * 32 + lc;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = c0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lr;
                    out[(long)row * a.K + col] = acc[b][i][j][r];
                }
        }
    }
}

template <int CT, int KT, int NBN, int NX, int MM>
int launch(const mvf_internal::BnBwdWgradArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)NX * CH * (CT * 2) + (size_t)NX * 2 * CH * (KT * 2);
    auto k = bnbwd_wgrad_kernel<CT, KT, NBN, NX, MM>;
    static bool attr = false;
    if (!attr) {
        MVF_HIP_OK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL(k, dim3(a.nsplit * a.ctiles * a.ktiles), dim3(kT), lds, st, a);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // namespace

namespace mvf_internal {

// (CT, KT) of the accumulator tile for a [M][c] gradient and a k-channel conv input; 0 = not built for this shape
bool bnbwd_wgrad_tile(int c, int k, int nbn, int mask_mode, int* ct, int* kt) {
    int CT = 0, KT = 0;
    if (mask_mode == 4 && c % 256 == 0 && (k == 64 || (k == 128 && nbn == 1))) { CT = 256; KT = k; }        // layer1 / layer2 conv3 (+ downsample)
    else if (mask_mode == 4 && nbn == 2 && c % 256 == 0 && k == 128) { CT = 256; KT = 128; }                 // layer2.0: conv3 only (x_b = NULL: its stride-2 downsample conv keeps the GEMM)
    else if (mask_mode == 2 && nbn == 1 && c == 64 && k % 256 == 0) { CT = 64; KT = 256; }                   // layer1 conv1
    else if (mask_mode == 2 && nbn == 1 && c % 128 == 0 && c <= 256 && k % 256 == 0) { CT = 128; KT = 256; } // layer2 conv1
    if (!CT) return false;
    *ct = CT; *kt = KT;
    return true;
}

int bnbwd_wgrad_plan(long m, int c, int k, int nbn, int mask_mode, int* rows_per_split, int* ctiles, int* ktiles) {
    int ct, kt;
    if (!bnbwd_wgrad_tile(c, k, nbn, mask_mode, &ct, &kt)) return 0;
    *ctiles = c / ct;
    *ktiles = k / kt;
    static const int wgs_env = mvf_policy_int("bnwg_wgs", 256);                 // one persistent workgroup per CU
    const int want = wgs_env / (*ctiles * *ktiles) > 0 ? wgs_env / (*ctiles * *ktiles) : 1;
    long rows = (m + want - 1) / want;
    rows = (rows + CH - 1) / CH * CH;
    if (rows < 4 * CH) rows = 4 * CH;
    *rows_per_split = (int)rows;
    return (int)((m + rows - 1) / rows);
}

int bnbwd_wgrad_launch(const BnBwdWgradArgs& a, int nbn, int mask_mode, hipStream_t st) {
    int ct, kt;
    if (!bnbwd_wgrad_tile(a.C, a.K, nbn, mask_mode, &ct, &kt)) {
        mvf_set_error("bn backward + weight gradient: no kernel for c=%d k=%d nbn=%d mask_mode=%d", a.C, a.K, nbn, mask_mode);
        return MVF_EUNSUPPORTED;
    }
    const int nx = (nbn == 2 && a.x[1]) ? 2 : 1;
    if (nbn == 2 && nx == 2 && ct == 256 && kt == 64) return launch<256, 64, 2, 2, 4>(a, st);
    if (nbn == 2 && nx == 1 && ct == 256 && kt == 64) return launch<256, 64, 2, 1, 4>(a, st);
    if (nbn == 2 && nx == 1 && ct == 256 && kt == 128) return launch<256, 128, 2, 1, 4>(a, st);
    if (nbn == 1 && ct == 256 && kt == 64) return launch<256, 64, 1, 1, 4>(a, st);
    if (nbn == 1 && ct == 256 && kt == 128) return launch<256, 128, 1, 1, 4>(a, st);
    if (nbn == 1 && ct == 64 && kt == 256) return launch<64, 256, 1, 1, 2>(a, st);
    if (nbn == 1 && ct == 128 && kt == 256) return launch<128, 256, 1, 1, 2>(a, st);
    mvf_set_error("bn backward + weight gradient: tile %d x %d not instantiated", ct, kt);
    return MVF_EUNSUPPORTED;
}

}  // namespace mvf_internal
