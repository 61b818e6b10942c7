// Weight gradient of the channels-last conv as an MFMA GEMM with the contraction over pixels (fp32).
//
//   dW[co][kh][kw][ci] = sum_m dZ[m][co] * X[img(m)][oh(m)*s+kh-p][ow(m)*s+kw-p][ci]        (zero outside the image)
//
// GEMM view: C[co][kcol] += A^T B with A = dZ [M][Cout], B = im2col(X) [M][K], K = KH*KW*Cin, contraction over m.
// Both operands are "m-major" in memory (rows of Cout / Cin contiguous channels), which is exactly what the fp32
// MFMA wants here: for v_mfma_f32_32x32x2_f32 lane l supplies A[i = l&31][k = l>>5] -- 32 consecutive output channels
// of ONE pixel -- so the LDS tiles stay row-major [m][channels], are filled with plain 16-B stores and read with
// conflict-free ds_read_b32 (2 reads per MFMA; the 64-cycle fp32 MFMA leaves the LDS idle anyway).
// The pixel range is split over gridDim.y workgroups; each writes its partial tile, and a second kernel sums the
// partials in fixed order (deterministic, no atomics) while transposing to the parameter's OIHW layout.
// Reference: autograd of nn.Conv2d in Bottleneck.forward (codes/models/backbones/resnet.py:208-244).
#include <algorithm>

#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int BMR = 32;                             // pixel rows per chunk
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgArgs {
    const void* dz;
    const void* x;
    const void* x2;
    float* part;
    int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, xps, split_c, x2ps;
    int M, K, rows_per_split, tiles_k;
    int tiles, nsplit;   // output tiles and pixel splits; the 1-D grid is nsplit * tiles (see wg_map)
    unsigned fd_hw_mul, fd_hw_shr, fd_w_mul, fd_w_shr;   // magic numbers for n / (Ho*Wo) and n / Wo, n < 2^31 (bf16 kernel)
    int xcd_rr;          // wg_map: 1 = splits round-robin over the XCDs (nsplit % 8 == 0), 0 = contiguous balanced ranges
    int abl;             // -DMVF_WGRAD_ABLATE builds only (timing ablation, wrong results): 1 no main loop, 2 no partial-slab stores
};

// n / d for 0 <= n < 2^31 with host-made magic: l = ceil(log2 d), mul = floor(2^32 (2^l - d) / d) + 1, q = (mulhi(n, mul) + n) >> l
__device__ __forceinline__ int wg_fd_div(int n, unsigned mul, unsigned shr) { return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> shr); }
// Workgroup -> (output tile, pixel split).  All tiles of one split read the SAME dz / x rows, so they should share an XCD (each has its own
// L2; consecutive workgroup ids round-robin over the 8 XCDs): with the plain (tile, split) grid every XCD re-fetched every row, 2-4x the
// algorithmic HBM reads (rocprofv3 FETCH_SIZE, profiles/r01_pmc_summary_bf16_train.json before/after).
// [r3] The split-major list of (split, tile) pairs is cut into 8 CONTIGUOUS, equally long ranges, one per XCD (the bijective remap of the
// conv kernels).  Rounds 1-2 pinned split s to XCD s % 8: with 28 splits x 9 tiles of the 256 x 256 plan that is 36 workgroups on XCDs 0-3
// (32 CUs each: a second, nearly empty round -- twice the launch's time) and 27 on XCDs 4-7; 252 workgroups are 31-32 per XCD in one round.
// A split that straddles two ranges is fetched by two L2s (at most 7 of them).
// When the number of splits is a multiple of 8 the round-robin form (split s on XCD s % 8, grid = nsplit * tiles as well) is just as
// balanced and keeps the 8 XCDs streaming one compact window of pixels instead of 8 distant ones -- measured 10 % faster on the
// HBM-bound layer1 launches (M = 802816, K = 64: 4.2 vs 3.8 TB/s) -- so the host picks it there (WgArgs::xcd_rr).
__device__ __forceinline__ bool wg_map(const int tiles, const int nsplit, int& tile, int& split, const int rr = 0) {
    const int nwg = (int)gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (rr) {                                          // nsplit % 8 == 0: XCD x owns splits x, x + 8, ...
        const int sl = idx / tiles;
        tile = idx - sl * tiles;
        split = sl * 8 + xcd;
        return split < nsplit;
    }
    const int q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    split = id / tiles;
    tile = id - split * tiles;
    return split < nsplit;
}
inline void wg_fd_make(unsigned d, unsigned& mul, unsigned& shr) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    shr = l;
}

// Output tile BCO x BK = (2*TM*32) x (2*TN*32): 128x128 (TM=TN=2), 64x128 for Cout <= 64 (TM=1), 128x64 for K <= 64 (TN=1)
// ET = storage type of dz / x (fp32 or bf16).  bf16 operands are widened to fp32 on the way into LDS and the contraction
// runs on the exact-fp32 MFMA: the pixel-major operand shape cannot feed the bf16 MFMA (8 consecutive k per lane) without a
// transposing read, and the weight gradient wants fp32 accumulation over ~10^6 pixels anyway.
// DMA (fp32 storage only): the two operand tiles arrive by LDS-DMA -- the row-major [pixel][channel] image is exactly what a wave
// instruction writes (thread t -> byte 16 t of an 8-row pass), so no swizzle is involved; rows past the split, padding taps and
// channel tails are out-of-range buffer offsets = zeros; row -> (image, oh, ow) by magic division.  The MVF split operand is taken
// when split_c is a multiple of the tile's K width (one source tensor per workgroup).
template <typename ET, int TM, int TN, bool DMA = false>
__global__ __launch_bounds__(kThreads) void wgrad_kernel(WgArgs a) {
    constexpr int BCO = 2 * TM * 32, BK = 2 * TN * 32;
    constexpr int UA = BCO / 4, UB = BK / 4;           // 16-B units per row of each operand tile
    constexpr int PA = BMR * UA / kThreads, PB = BMR * UB / kThreads;   // units per thread per chunk
    __shared__ __attribute__((aligned(16))) float Ds[2][BMR][BCO];
    __shared__ __attribute__((aligned(16))) float Xs[2][BMR][BK];
    int wg_tile, wg_split;
    if (!wg_map(a.tiles, a.nsplit, wg_tile, wg_split, a.xcd_rr)) return;
    const int tile_k = wg_tile % a.tiles_k, tile_co = wg_tile / a.tiles_k;
    const int co0 = tile_co * BCO, k0 = tile_k * BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // loaders: thread -> (16-B unit q within the row, first row); rows advance by kThreads/U per pass
    const int qa = tid % UA, ra0 = tid / UA, qb = tid % UB, rb0 = tid / UB;
    const int co = co0 + qa * 4;
    const bool co_ok = co < a.Cout;
    const int kcol = k0 + qb * 4;
    const bool k_ok = kcol < a.K;
    const int tap = k_ok ? kcol / a.Cin : 0;
    const int ci = k_ok ? kcol - tap * a.Cin : 0;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const bool from2 = a.split_c > 0 && ci < a.split_c;
    const ET* xb = reinterpret_cast<const ET*>(from2 ? a.x2 : a.x);
    const ET* dzp = reinterpret_cast<const ET*>(a.dz);
    const int ps = from2 ? a.x2ps : a.xps;

    const int m_begin = wg_split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nchunks = (m_end - m_begin + BMR - 1) / BMR;

    float4 rd[PA], rx[PB];
    static_assert(!DMA || sizeof(ET) == 4, "LDS-DMA loaders: fp32 storage");
    constexpr unsigned kOOBw = 0x80000000u;
    const int hw_o = a.Ho * a.Wo;
    const bool x2u = a.split_c > 0 && (k0 % a.Cin) < a.split_c;        // uniform per workgroup on this path (host guarantee)
    const i32x4 gs_dz = rsrc_words(dzp + (long)m_begin * a.Cout, (unsigned)min((long)(m_end - m_begin) * a.Cout * 4, 0x7ffffff0L));
    const i32x4 gs_x = rsrc_words(x2u ? a.x2 : a.x, (unsigned)min((long)a.N * a.H * a.W * (x2u ? a.x2ps : a.xps) * 4, 0x7ffffff0L));
    const unsigned lds_d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)&Ds[0][0][0] + wave * 1024);
    const unsigned lds_x0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)&Xs[0][0][0] + wave * 1024);
    auto dma_chunk = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = c * BMR + ra0 + (kThreads / UA) * i;                 // row within the split
            const unsigned off = co_ok ? (unsigned)(r * a.Cout + co) * 4u : kOOBw;
            glds16(gs_dz, lds_d0 + (unsigned)((buf * BMR * BCO + (kThreads / UA) * i * BCO) * 4), off);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int m = m_begin + c * BMR + rb0 + (kThreads / UB) * i;
            const int img = wg_fd_div(m, a.fd_hw_mul, a.fd_hw_shr), rem = m - img * hw_o;
            const int oh = wg_fd_div(rem, a.fd_w_mul, a.fd_w_shr), ow = rem - oh * a.Wo;
            const int ih = oh * a.stride - a.pad + kh, iw = ow * a.stride - a.pad + kw;
            const bool ok = m < m_end && k_ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)(((img * a.H + ih) * a.W + iw) * ps + ci) * 4u : kOOBw;
            glds16(gs_x, lds_x0 + (unsigned)((buf * BMR * BK + (kThreads / UB) * i * BK) * 4), off);
        }
    };
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = m_begin + c * BMR + ra0 + (kThreads / UA) * i;
            rd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end && co_ok) rd[i] = ld4(dzp + (long)m * a.Cout + co);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int m = m_begin + c * BMR + rb0 + (kThreads / UB) * i;
            rx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end && k_ok) {
                const int img = m / (a.Ho * a.Wo), rem = m - img * (a.Ho * a.Wo);
                const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
                const int ih = oh * a.stride - a.pad + kh, iw = ow * a.stride - a.pad + kw;
                if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
                    rx[i] = ld4(xb + ((long)(img * a.H + ih) * a.W + iw) * ps + ci);
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i) *reinterpret_cast<float4*>(&Ds[buf][ra0 + (kThreads / UA) * i][qa * 4]) = rd[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) *reinterpret_cast<float4*>(&Xs[buf][rb0 + (kThreads / UB) * i][qb * 4]) = rx[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (DMA) {
        if (nchunks > 0) dma_chunk(0, 0);
    } else {
        if (nchunks > 0) {
            load_chunk(0);
            store_chunk(0);
        }
        __syncthreads();
    }
    const int lr = lane >> 5, lc = lane & 31;
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < nchunks;
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                           // chunk c landed for every wave; the other buffer is free again
            if (more) dma_chunk(c + 1, buf ^ 1);
        } else {
            if (more) load_chunk(c + 1);
        }
#pragma unroll
        for (int kk = 0; kk < BMR / 2; ++kk) {
            float fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = Ds[buf][2 * kk + lr][(wm * TM + i) * 32 + lc];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = Xs[buf][2 * kk + lr][(wn * TN + j) * 32 + lc];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if constexpr (!DMA) {
            if (more) store_chunk(buf ^ 1);
            __syncthreads();
        }
    }
    // partial[split][co][kcol]
    float* out = a.part + (long)wg_split * a.Cout * a.K;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = k0 + (wn * TN + j) * 32 + lc;
        if (col >= a.K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lr;
                if (row < a.Cout) out[(long)row * a.K + col] = acc[i][j][r];
            }
    }
}

// ---- bf16 operands on the bf16 matrix cores --------------------------------------------------------------------------
// v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE contraction indices (= pixels) per lane, but both operands are pixel-major in
// memory.  The tiles are staged in their natural layout ([pixel][channel], 16-B coalesced global loads, 16-B LDS stores, rows
// XOR-swizzled) and the transposition happens in the operand fetch with the gfx950 transpose read ds_read_b64_tr_b16 (two per
// fragment; see wgrad_bf16_kernel).  Compared with widening to fp32 this is 4x fewer HBM bytes and 16x the matrix rate.
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
// XOR swizzle of the 16-byte units of an LDS row with U units (U = 16: 256-B rows, U = 8: 128-B rows), see wgrad_bf16_kernel
template <int U>
__device__ __forceinline__ int swz16(int row) {
    static_assert(U == 8 || U == 16 || U == 32, "64-, 128- or 256-channel tile rows");
    return U >= 16 ? (row & 3) << 2 : ((row >> 1) & 1) << 2;          // (U = 32: 512-B rows, the XOR stays inside a 256-B bank row)
}
constexpr int BMR16 = 64;                            // pixels per chunk = 4 MFMA k-steps

// DMA = true: both operand tiles go global -> LDS by LDS-DMA (see conv_nhwc.hip / common.h glds16): no staging registers, no
// ds_write pass.  The DMA writes lane-linearly, so the XOR image of the transpose reads is made on the SOURCE side: the lane at
// position q of row r fetches unit q ^ swz16(r) (constant per thread: every pass advances the row by a multiple of 4).  Padding /
// out-of-range taps, rows past the split and channel tails are out-of-range buffer offsets = DMA'd zeros.  Not for the MVF
// split operand (two source tensors inside one wave instruction).
// WMG = wave rows (2: four waves, 4: eight waves = the 256 x 256 tile with TM = 2, TN = 4: half the L2 -> LDS bytes per flop of the
// 128 x 128 tile, one workgroup per CU)
template <int TM, int TN, bool DMA = false, int WMG = 2>
__global__ __launch_bounds__(WMG * 128) void wgrad_bf16_kernel(WgArgs a) {
    constexpr int kThreads = WMG * 128;
    constexpr int BCO = WMG * TM * 32, BK = 2 * TN * 32;
    // LDS images stay in the natural [pixel][channel] layout (16-byte coalesced staging writes, no padding); the MFMA
    // operands (8 consecutive PIXELS of one channel per lane) come out of it with the gfx950 transpose read
    // ds_read_b64_tr_b16: every 16-lane group hands in a [4 pixel][16 channel] block (lane i: pixel i>>2, channels
    // 4*(i&3)..+3, 8 bytes) and lane i gets channel i of the 4 pixels (mapping probed in tools/probes/tr_b16_probe.hip).
    // Bank conflicts: a 32-lane half reads 4 pixel rows x 64 B; the 16-byte units of a row are XOR-swizzled by the row
    // (swz16) so that those four 64-B pieces fall into four different 16-bank ranges.
    constexpr int PA = BCO * 2, PB = BK * 2;                        // LDS row pitches in bytes
    constexpr int UA = BCO / 8, UB = BK / 8;                        // 16-B units (8 bf16) per row
    constexpr int NA = BMR16 * UA / kThreads, NB = BMR16 * UB / kThreads;
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    char* Ds = smem16;                                              // [2][BMR16][PA]
    char* Xs = smem16 + 2 * BMR16 * PA;                             // [2][BMR16][PB]
    int wg_tile, wg_split;
    if (!wg_map(a.tiles, a.nsplit, wg_tile, wg_split, a.xcd_rr)) return;
    const int tile_k = wg_tile % a.tiles_k, tile_co = wg_tile / a.tiles_k;
    const int co0 = tile_co * BCO, k0 = tile_k * BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ra0 = tid / UA, rb0 = tid / UB;
    const int qa = DMA ? ((tid % UA) ^ swz16<UA>(ra0)) : tid % UA;       // DMA: the source unit that belongs at LDS position tid % U
    const int qb = DMA ? ((tid % UB) ^ swz16<UB>(rb0)) : tid % UB;
    const int co = co0 + qa * 8;
    const bool co_ok = co < a.Cout;
    const int kcol = k0 + qb * 8;
    const bool k_ok = kcol < a.K;
    const int tap = k_ok ? kcol / a.Cin : 0;
    const int ci = k_ok ? kcol - tap * a.Cin : 0;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const bool from2 = a.split_c > 0 && ci < a.split_c;
    const bf16_t* xb = reinterpret_cast<const bf16_t*>(from2 ? a.x2 : a.x);
    const bf16_t* dzp = reinterpret_cast<const bf16_t*>(a.dz);
    const int ps = from2 ? a.x2ps : a.xps;

    const int m_begin = wg_split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nchunks = (m_end - m_begin + BMR16 - 1) / BMR16;

    struct Stage {                        // one chunk of this thread's loader rows, in registers
        uint4 d[NA], x[NB];
        unsigned okx;                     // bit i: x row i is a real tap (else it is stored as zeros)
    };
    Stage s0, s1;
    // dz: buffer loads relative to the split's first row (rows past the split / channel tails get an out-of-range offset ->
    // the hardware returns 0).  x: the source tensor (x or x2 for the MVF split) and the tap differ per lane, so these are
    // flat loads, kept branch-free by reading the tensor base for padding / out-of-range taps and zeroing at the LDS write.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned kOOB = 0x80000000u;
    const int hw_o = a.Ho * a.Wo;
    const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dzp + (long)m_begin * a.Cout), 0, (unsigned)min((long)(m_end - m_begin) * a.Cout * 2, 0x7ffffff0L), 0x00020000);
    // ---- DMA variant: descriptors as SGPR words, 32-bit byte offsets from the tensor base (host guarantees < 2 GB) ----
    const i32x4 gs_dz = rsrc_words(dzp + (long)m_begin * a.Cout, (unsigned)min((long)(m_end - m_begin) * a.Cout * 2, 0x7ffffff0L));
    // MVF split operand: channels < split_c of a tap come from x2 (pitch x2ps).  The host only takes this variant when split_c is
    // a multiple of the tile's K width, so a whole workgroup reads one of the two tensors (from2 / ps are then uniform)
    const bool x2u = a.split_c > 0 && (k0 % a.Cin) < a.split_c;
    const i32x4 gs_x = rsrc_words(x2u ? a.x2 : a.x, (unsigned)min((long)a.N * a.H * a.W * (x2u ? a.x2ps : a.xps) * 2, 0x7ffffff0L));
    const unsigned lds_d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Ds + (wave * 64 / UA) * PA);
    const unsigned lds_x0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Xs + (wave * 64 / UB) * PB);
    auto dma_chunk = [&](int c, int buf) {
        const int mc = m_begin + c * BMR16;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = c * BMR16 + ra0 + (kThreads / UA) * i;
            const unsigned off = co_ok ? (unsigned)(r * a.Cout + qa * 8 + co0) * 2u : kOOB;
            glds16(gs_dz, lds_d0 + (unsigned)((buf * BMR16 + (kThreads / UA) * i) * PA), off);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int m = mc + rb0 + (kThreads / UB) * i;
            const int img = wg_fd_div(m, a.fd_hw_mul, a.fd_hw_shr), rem = m - img * hw_o;
            const int oh = wg_fd_div(rem, a.fd_w_mul, a.fd_w_shr), ow = rem - oh * a.Wo;
            const int ih = oh * a.stride - a.pad + kh, iw = ow * a.stride - a.pad + kw;
            const bool ok = m < m_end && k_ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)(((img * a.H + ih) * a.W + iw) * ps + ci) * 2u : kOOB;
            glds16(gs_x, lds_x0 + (unsigned)((buf * BMR16 + (kThreads / UB) * i) * PB), off);
        }
    };
    auto load_chunk = [&](int c, Stage& st) {
        const int mc = m_begin + c * BMR16;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = c * BMR16 + ra0 + (kThreads / UA) * i;                       // row within the split
            const unsigned off = co_ok ? (unsigned)(r * a.Cout + qa * 8 + co0) * 2u : kOOB;   // past m_end -> past num_records
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_dz, off, 0, 0);
            st.d[i] = make_uint4(v.x, v.y, v.z, v.w);
        }
        st.okx = 0u;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int m = mc + rb0 + (kThreads / UB) * i;
            const int img = wg_fd_div(m, a.fd_hw_mul, a.fd_hw_shr), rem = m - img * hw_o;
            const int oh = wg_fd_div(rem, a.fd_w_mul, a.fd_w_shr), ow = rem - oh * a.Wo;
            const int ih = oh * a.stride - a.pad + kh, iw = ow * a.stride - a.pad + kw;
            const bool ok = m < m_end && k_ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const long off = ok ? ((long)((img * a.H + ih) * a.W + iw) * ps + ci) : 0L;
            st.x[i] = *reinterpret_cast<const uint4*>(xb + off);
            st.okx |= ok ? (1u << i) : 0u;
        }
    };
    auto store_chunk = [&](int buf, const Stage& st) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = ra0 + (kThreads / UA) * i;
            *reinterpret_cast<uint4*>(Ds + (buf * BMR16 + row) * PA + ((qa ^ swz16<UA>(row)) * 16)) = st.d[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = rb0 + (kThreads / UB) * i;
            const bool ok = (st.okx >> i) & 1u;
            const uint4 v = st.x[i];
            *reinterpret_cast<uint4*>(Xs + (buf * BMR16 + row) * PB + ((qb ^ swz16<UB>(row)) * 16)) =
                make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
        }
    };
    // per-lane byte offsets of the transpose reads: group g = lane>>4 supplies pixel rows (i>>2) + 8*(g>>1) and channel
    // quad 16*(g&1) + 4*(i&3) of a 32-channel block; k-steps / the second 4-pixel half / the buffer are immediates on top
    const int tg = lane >> 4, ti = lane & 15;
    const int trow = (ti >> 2) + 8 * (tg >> 1);
    const int tunit = 2 * (tg & 1) + ((ti & 3) >> 1), thalf = ti & 1;
    int offA[TM], offB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) offA[i] = trow * PA + ((((wm * TM + i) * 4 + tunit) ^ swz16<UA>(trow)) * 16) + thalf * 8;
#pragma unroll
    for (int j = 0; j < TN; ++j) offB[j] = trow * PB + ((((wn * TN + j) * 4 + tunit) ^ swz16<UB>(trow)) * 16) + thalf * 8;
    auto gather = [&](const char* p, int pitch) {          // rows +0..3 and +4..7 of the lane's channel
        typedef short v4s __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4s* lds_v4s;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p + 4 * pitch));
        bf16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane >> 5, lc = lane & 31;
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < BMR16 / 16; ++ks) {
            const int r0 = buf * BMR16 + ks * 16;
            bf16x8_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = gather(Ds + r0 * PA + offA[i], PA);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = gather(Xs + r0 * PB + offB[j], PB);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };
    // two chunks ahead in registers (s0: even chunks, s1: odd chunks), two LDS buffers: the loop is bound by the global-load
    // round trip per chunk (16 MFMAs = 0.25 us of matrix work against ~2 us of latency), so keep two round trips in flight
    // The prefetch loads are UNCONDITIONAL (also past the last chunk: rows >= m_end read zeros / the tensor base): a conditional
    // load forces the compiler to assume no younger load is in flight at each static s_waitcnt, i.e. vmcnt(0) everywhere.
    if constexpr (DMA) {
        // two LDS buffers: chunk c+1 in flight under the MFMAs of chunk c; one barrier per chunk (conv_tile's GLDS >= 2 loop)
        if (nchunks > 0) dma_chunk(0, 0);
        for (int cc = 0; cc < nchunks; ++cc) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (cc + 1 < nchunks) dma_chunk(cc + 1, (cc + 1) & 1);
            compute(cc & 1);
        }
        float* outp = a.part + (long)wg_split * a.Cout * a.K;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = k0 + (wn * TN + j) * 32 + lc;
            if (col >= a.K) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = co0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lr;
                    if (row < a.Cout) outp[(long)row * a.K + col] = acc[i][j][r];
                }
        }
        return;
    }
    load_chunk(0, s0);
    load_chunk(1, s1);
    if (nchunks > 0) store_chunk(0, s0);
    __syncthreads();
    load_chunk(2, s0);
    int c = 0;
    for (; c + 1 < nchunks; c += 2) {
        compute(0);                                    // chunk c
        store_chunk(1, s1);                            // chunk c+1
        load_chunk(c + 3, s1);
        __syncthreads();
        compute(1);                                    // chunk c+1
        if (c + 2 < nchunks) store_chunk(0, s0);       // chunk c+2
        load_chunk(c + 4, s0);
        __syncthreads();
    }
    if (c < nchunks) compute(0);
    float* out = a.part + (long)wg_split * a.Cout * a.K;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = k0 + (wn * TN + j) * 32 + lc;
        if (col >= a.K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lr;
                if (row < a.Cout) out[(long)row * a.K + col] = acc[i][j][r];
            }
    }
}

// ---- [r6] the four-wave bf16 tile on an NS-stage LDS-DMA ring -------------------------------------------------------------------------------------
// wgrad_bf16_kernel<.., DMA = true> keeps ONE chunk in flight (two buffers, vmcnt(0) + barrier per chunk): a chunk is 16 matrix instructions per wave = 0.2 us
// of matrix work against ~1 us of L2 / HBM latency, and the plans give these launches one workgroup per CU (the side stream's footprint rule), so nothing else
// on the CU hides it -- layer2's 3 x 3 (M = 200704, N = 128, K = 1152) ran 112 chunks in 126 us = 1.1 us per chunk, 19 % of the matrix rate.  Here NS - 1
// chunks are in flight: NS buffers of [64 pixels][BCO + BK channels], the wait is a COUNTED vmcnt((NS - 2) * instructions per chunk) (loads retire in order,
// so the oldest chunk has landed), one raw s_barrier per chunk (it also says every wave is done reading the buffer the next DMA overwrites); the look-ahead
// DMAs past the split's last chunk are issued unconditionally with out-of-range offsets (zeros into a buffer nobody reads) so that the count stays static.
// Same LDS images, transpose reads, accumulation order and slab layout as wgrad_bf16_kernel: results are bit-identical to it.
template <int N>
__device__ __forceinline__ void wg_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int TM, int TN, int NS>
__global__ __launch_bounds__(256) void wgrad_bf16_pipe_kernel(WgArgs a) {
    constexpr int kThreads = 256;
    constexpr int BCO = 2 * TM * 32, BK = 2 * TN * 32;
    constexpr int PA = BCO * 2, PB = BK * 2;                        // LDS row pitches in bytes
    constexpr int UA = BCO / 8, UB = BK / 8;                        // 16-B units (8 bf16) per row
    constexpr int NA = BMR16 * UA / kThreads, NB = BMR16 * UB / kThreads;
    constexpr int IPC = NA + NB;                                    // LDS-DMA instructions per thread per chunk
    static_assert(NS >= 3 && (NS - 2) * IPC < 64, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    char* Ds = smem16;                                              // [NS][BMR16][PA]
    char* Xs = smem16 + NS * BMR16 * PA;                            // [NS][BMR16][PB]
    int wg_tile, wg_split;
    if (!wg_map(a.tiles, a.nsplit, wg_tile, wg_split, a.xcd_rr)) return;
    const int tile_k = wg_tile % a.tiles_k, tile_co = wg_tile / a.tiles_k;
    const int co0 = tile_co * BCO, k0 = tile_k * BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ra0 = tid / UA, rb0 = tid / UB;
    const int qa = (tid % UA) ^ swz16<UA>(ra0);                     // the source unit that belongs at LDS position tid % U (rows advance by multiples of 4)
    const int qb = (tid % UB) ^ swz16<UB>(rb0);
    const bool co_ok = co0 + qa * 8 < a.Cout;
    const int kcol = k0 + qb * 8;
    const bool k_ok = kcol < a.K;
    const int tap = k_ok ? kcol / a.Cin : 0;
    const int ci = k_ok ? kcol - tap * a.Cin : 0;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const bf16_t* dzp = reinterpret_cast<const bf16_t*>(a.dz);
    const int m_begin = wg_split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nchunks = (m_end - m_begin + BMR16 - 1) / BMR16;
    constexpr unsigned kOOB = 0x80000000u;
    const int hw_o = a.Ho * a.Wo;
    const i32x4 gs_dz = rsrc_words(dzp + (long)m_begin * a.Cout, (unsigned)min((long)(m_end - m_begin) * a.Cout * 2, 0x7ffffff0L));
    const bool x2u = a.split_c > 0 && (k0 % a.Cin) < a.split_c;     // uniform per workgroup (host: split_c % BK == 0, cin % BK == 0)
    const int ps = x2u ? a.x2ps : a.xps;
    const i32x4 gs_x = rsrc_words(x2u ? a.x2 : a.x, (unsigned)min((long)a.N * a.H * a.W * ps * 2, 0x7ffffff0L));
    const unsigned lds_d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Ds + (wave * 64 / UA) * PA);
    const unsigned lds_x0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Xs + (wave * 64 / UB) * PB);
    const unsigned dz_col = (unsigned)(qa * 8 + co0) * 2u;
    const unsigned x_tap = (unsigned)((kh * a.W + kw) * ps + ci) * 2u;
    auto dma_chunk = [&](int c, int buf) {
        const int mc = m_begin + c * BMR16;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = c * BMR16 + ra0 + (kThreads / UA) * i;                       // row within the split; past its end = past the descriptor's range
            const unsigned off = co_ok ? (unsigned)(r * a.Cout) * 2u + dz_col : kOOB;
            glds16(gs_dz, lds_d0 + (unsigned)((buf * BMR16 + (kThreads / UA) * i) * PA), off);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int m = mc + rb0 + (kThreads / UB) * i;
            const int img = wg_fd_div(m, a.fd_hw_mul, a.fd_hw_shr), rem = m - img * hw_o;
            const int oh = wg_fd_div(rem, a.fd_w_mul, a.fd_w_shr), ow = rem - oh * a.Wo;
            const int ih0 = oh * a.stride - a.pad, iw0 = ow * a.stride - a.pad;
            const bool ok = m < m_end && k_ok && (unsigned)(ih0 + kh) < (unsigned)a.H && (unsigned)(iw0 + kw) < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)(((img * a.H + ih0) * a.W + iw0) * ps) * 2u + x_tap : kOOB;
            glds16(gs_x, lds_x0 + (unsigned)((buf * BMR16 + (kThreads / UB) * i) * PB), off);
        }
    };
    const int tg = lane >> 4, ti = lane & 15;
    const int trow = (ti >> 2) + 8 * (tg >> 1);
    const int tunit = 2 * (tg & 1) + ((ti & 3) >> 1), thalf = ti & 1;
    int offA[TM], offB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) offA[i] = trow * PA + ((((wm * TM + i) * 4 + tunit) ^ swz16<UA>(trow)) * 16) + thalf * 8;
#pragma unroll
    for (int j = 0; j < TN; ++j) offB[j] = trow * PB + ((((wn * TN + j) * 4 + tunit) ^ swz16<UB>(trow)) * 16) + thalf * 8;
    auto gather = [&](const char* p, int pitch) {          // rows +0..3 and +4..7 of the lane's channel
        typedef short v4s __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4s* lds_v4s;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p + 4 * pitch));
        bf16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < BMR16 / 16; ++ks) {
            const int r0 = buf * BMR16 + ks * 16;
            bf16x8_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = gather(Ds + r0 * PA + offA[i], PA);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = gather(Xs + r0 * PB + offB[j], PB);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_) dma_chunk(s_, s_);
    for (int cc = 0; cc < nchunks; cc += NS) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            if (cc + s_ < nchunks) {                                 // (wave-uniform)
                wg_wait_vmcnt<(NS - 2) * IPC>();                     // chunk cc + s_ has landed (this thread's pieces) ...
                __builtin_amdgcn_s_barrier();                        // ... everybody's; and the buffer of chunk cc + s_ - 1 is free
                dma_chunk(cc + s_ + NS - 1, (s_ + NS - 1) % NS);
                compute(s_);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the look-ahead DMAs target this workgroup's LDS: let them land before it is released
    float* outp = a.part + (long)wg_split * a.Cout * a.K;
    const int lr = lane >> 5, lc = lane & 31;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = k0 + (wn * TN + j) * 32 + lc;
        if (col >= a.K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lr;
                if (row < a.Cout) outp[(long)row * a.K + col] = acc[i][j][r];
            }
    }
}

// ---- [r4] fp32 operands on the BF16 matrix cores with fp32 accuracy (the weight-gradient side of conv_nhwc.hip's X3) -------------------
// Every fp32 value of dz and x is split exactly into three bf16 terms (hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid): both
// differences are exactly representable in fp32) on its way from the staging registers into LDS, one [pixel][channel] bf16 image per term in
// wgrad_bf16_kernel's swizzled layout; the contraction is the six partial products of total order <= 2 (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi,
// mid.mid -- what is dropped is below 2^-24 of |dz||x|), smallest first, accumulated in fp32 by v_mfma_f32_32x32x16_bf16 through the same
// transpose reads.  6 matrix instructions at 16x the fp32 instruction's rate = 2.7x its ceiling; the loop carries ~22 VALU operations per
// float4 for the split, so it is a register-staged single-buffer kernel at three workgroups per CU (48 KB of planes each) whose neighbours
// fill each other's split / barrier phases, as in conv_tile.  A chunk = 32 pixels = 2 k-steps.
constexpr int BMRX = 32;
template <int TM, int TN>
__global__ __launch_bounds__(256, 3) void wgrad_x3_kernel(WgArgs a) {
    constexpr int kThreads = 256;
    constexpr int BCO = 2 * TM * 32, BK = 2 * TN * 32;
    constexpr int PA = BCO * 2, PB = BK * 2;                        // plane row pitches in bytes
    constexpr int UA = BCO / 8, UB = BK / 8;                        // 16-B units per plane row
    constexpr int QA = BCO / 4, QB = BK / 4;                        // float4 units per fp32 row
    constexpr int NA = BMRX * QA / kThreads, NB = BMRX * QB / kThreads;
    __shared__ __attribute__((aligned(16))) char Ds[3 * BMRX * PA];      // [plane][pixel][BCO bf16]
    __shared__ __attribute__((aligned(16))) char Xs[3 * BMRX * PB];
    int wg_tile, wg_split;
    if (!wg_map(a.tiles, a.nsplit, wg_tile, wg_split, a.xcd_rr)) return;
    const int tile_k = wg_tile % a.tiles_k, tile_co = wg_tile / a.tiles_k;
    const int co0 = tile_co * BCO, k0 = tile_k * BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int qa = tid % QA, ra0 = tid / QA, qb = tid % QB, rb0 = tid / QB;
    const int co = co0 + qa * 4;
    const bool co_ok = co < a.Cout;
    const int kcol = k0 + qb * 4;
    const bool k_ok = kcol < a.K;
    const int tap = k_ok ? kcol / a.Cin : 0;
    const int ci = k_ok ? kcol - tap * a.Cin : 0;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const bool from2 = a.split_c > 0 && ci < a.split_c;
    const float* xb = reinterpret_cast<const float*>(from2 ? a.x2 : a.x);
    const float* dzp = reinterpret_cast<const float*>(a.dz);
    const int ps = from2 ? a.x2ps : a.xps;

    const int m_begin = wg_split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nchunks = (m_end - m_begin + BMRX - 1) / BMRX;

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned kOOB = 0x80000000u;
    const int hw_o = a.Ho * a.Wo;
    // dz: buffer loads relative to the split's first row (rows past the split / channel tails are out of range = zeros); x: flat loads,
    // branch-free (padding / out-of-range taps read the tensor base and are zeroed before the split), as in wgrad_bf16_kernel
    const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dzp + (long)m_begin * a.Cout), 0, (unsigned)min((long)(m_end - m_begin) * a.Cout * 4, 0x7ffffff0L), 0x00020000);
    uint4 sd[NA], sx[NB];
    unsigned okx = 0u;
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = c * BMRX + ra0 + (kThreads / QA) * i;
            const unsigned off = co_ok ? (unsigned)(r * a.Cout + co) * 4u : kOOB;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_dz, off, 0, 0);
            sd[i] = make_uint4(v.x, v.y, v.z, v.w);
        }
        okx = 0u;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int m = m_begin + c * BMRX + rb0 + (kThreads / QB) * i;
            const int img = wg_fd_div(m, a.fd_hw_mul, a.fd_hw_shr), rem = m - img * hw_o;
            const int oh = wg_fd_div(rem, a.fd_w_mul, a.fd_w_shr), ow = rem - oh * a.Wo;
            const int ih = oh * a.stride - a.pad + kh, iw = ow * a.stride - a.pad + kw;
            const bool ok = m < m_end && k_ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const long off = ok ? ((long)((img * a.H + ih) * a.W + iw) * ps + ci) : 0L;
            sx[i] = *reinterpret_cast<const uint4*>(xb + off);
            okx |= ok ? (1u << i) : 0u;
        }
    };
    // hi / mid / lo bf16 terms of four fp32 values, 8 bytes per plane (conv_tile's split3: scalar v_sub_f32 through asm -- packed fp32 VALU
    // beside matrix instructions is an anti-lever on this chip)
    auto split3 = [&](const uint4& v, uint2 (&pl)[3]) {
        float r[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned lo2 = pack_bf16x2(r[0], r[1]), hi2 = pack_bf16x2(r[2], r[3]);
            pl[p] = make_uint2(lo2, hi2);
            if (p < 2) {
                auto sub1 = [](float x, float y) { float d; asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
                r[0] = sub1(r[0], __uint_as_float(lo2 << 16)); r[1] = sub1(r[1], __uint_as_float(lo2 & 0xffff0000u));
                r[2] = sub1(r[2], __uint_as_float(hi2 << 16)); r[3] = sub1(r[3], __uint_as_float(hi2 & 0xffff0000u));
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = ra0 + (kThreads / QA) * i;
            uint2 pl[3];
            split3(sd[i], pl);
            char* d = Ds + row * PA + ((((qa >> 1) ^ swz16<UA>(row)) << 4) | ((qa & 1) << 3));
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(d + p * BMRX * PA) = pl[p];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = rb0 + (kThreads / QB) * i;
            const bool ok = (okx >> i) & 1u;
            const uint4 v = sx[i];
            uint2 pl[3];
            split3(make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u), pl);
            char* d = Xs + row * PB + ((((qb >> 1) ^ swz16<UB>(row)) << 4) | ((qb & 1) << 3));
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(d + p * BMRX * PB) = pl[p];
        }
    };
    // transpose reads exactly as in wgrad_bf16_kernel (same images, one per plane)
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

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < BMRX / 16; ++ks) {
            bf16x8_t fa[3][TM], fb[3][TN];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[p][i] = gather(Ds + (p * BMRX + ks * 16) * PA + offA[i], PA);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[p][j] = gather(Xs + (p * BMRX + ks * 16) * PB + offB[j], PB);
            }
            // the product term is the OUTER loop (consecutive matrix instructions on different accumulators), smallest terms first
            constexpr int PA_[6] = {0, 2, 1, 0, 1, 0}, PB_[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA_[t]][i], fb[PB_[t]][j], acc[i][j], 0, 0, 0);
        }
    };
    // one chunk ahead in registers; the loads are unconditional (rows >= m_end read zeros / the tensor base) so that the compiler's static
    // s_waitcnt counts stay exact
    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        store_chunk();
        __syncthreads();
        load_chunk(c + 1);
        compute();
        __syncthreads();
    }
    const int lr = lane >> 5, lc = lane & 31;
    float* out = a.part + (long)wg_split * a.Cout * a.K;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = k0 + (wn * TN + j) * 32 + lc;
        if (col >= a.K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lr;
                if (row < a.Cout) out[(long)row * a.K + col] = acc[i][j][r];
            }
    }
}

// ---- the 256 x 256 weight-gradient tile on the four-phase ping-pong loop (conv_nhwc.hip, conv_tile GLDS = 4; probe: tools/probes/gemm8p_probe.hip) ----
// Same machine as the conv kernel with the roles A := im2col(X) columns (4 fragments of 32 per wave), B := dZ columns (2 fragments per wave):
// 8 waves = 4 (co) x 2 (k), two groups (waves 0-3 / 4-7: one wave of each per SIMD) one barrier apart; a chunk = 64 pixels = 4 MFMA k-steps.
//   X(c): read x fragments {0,1} + dz fragment 0 of chunk c       -> acc[0..1][0..1]     stage dz-half 0 (c+1), then x-half 1 (c+1)
//   Y(c): read x fragments {2,3} of chunk c + dz fragment 1 of c+1 -> acc[0..1][2..3]     stage dz-half 1 (c+2), then x-half 0 (c+2)
// every phase: 24 transpose reads, 4 LDS-DMA pieces, vmcnt(6) (three half tiles in flight), the 16 x reads retired before the first barrier.
// LDS image: the contraction index (pixel) is the ROW here, so a half tile is a COLUMN range; to keep a wave's DMA footprint (1 KB)
// contiguous the operand tiles are stored as column blocks [block of 64 channels][64 pixels][128 B] (8 KB each, 4 per operand per buffer),
// rows XOR-swizzled on the source side as in wgrad_bf16_kernel (swz16<8>).  Fragment -> column mapping (any bijection works, the
// partial-slab store applies the same one): x fragment j of wave column wn = columns (j >> 1) * 128 + wn * 64 + (j & 1) * 32 (x-half h =
// columns [128 h, 128 h + 128)); dz fragment i of wave row wm = channels i * 128 + wm * 32 (dz-half i = channels [128 i, 128 i + 128)).
__global__ __launch_bounds__(512) void wgrad_bf16_p4_kernel(WgArgs a) {
    constexpr int kRows = BMR16;                       // 64 pixels per chunk
    constexpr int kBlk = kRows * 128;                  // one column block: [64 pixels][64 channels] = 8 KB
    constexpr int kBuf = 8 * kBlk;                     // x blocks 0-3, dz blocks 0-3
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    int wg_tile, wg_split;
    if (!wg_map(a.tiles, a.nsplit, wg_tile, wg_split, a.xcd_rr)) return;
    const int tile_k = wg_tile % a.tiles_k, tile_co = wg_tile / a.tiles_k;
    const int co0 = tile_co * 256, k0 = tile_k * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    const int m_begin = wg_split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nchunks = (m_end - m_begin + kRows - 1) / kRows;
    constexpr unsigned kOOB = 0x80000000u;
    const int hw_o = a.Ho * a.Wo;
    const bf16_t* dzp = reinterpret_cast<const bf16_t*>(a.dz);
    const i32x4 gs_dz = rsrc_words(dzp + (long)m_begin * a.Cout, (unsigned)min((long)(m_end - m_begin) * a.Cout * 2, 0x7ffffff0L));
    const bool x2u = a.split_c > 0 && (k0 % a.Cin) < a.split_c;            // uniform per workgroup (host: split_c % 256 == 0, cin % 256 == 0)
    const int ps = x2u ? a.x2ps : a.xps;
    const i32x4 gs_x = rsrc_words(x2u ? a.x2 : a.x, (unsigned)min((long)a.N * a.H * a.W * ps * 2, 0x7ffffff0L));
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)smem16);
    // DMA piece n (0 / 1) of a half: instruction q = 2 wave + n of 16: column block (q >> 3) of the half, pixel rows (q & 7) * 8 .. + 8
    const int drow = lane >> 3;                        // row within the 8-row piece
    int prow[2], prow0[2], pblk[2], punit[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int q = wave * 2 + n;
        pblk[n] = q >> 3;
        prow0[n] = (q & 7) * 8;                         // wave-uniform: the LDS-DMA destination base must be a scalar
        prow[n] = prow0[n] + drow;
        punit[n] = (lane & 7) ^ swz16<8>(prow[n]);      // the source unit that belongs at LDS position lane & 7 of this row
    }
    // dz piece n of half h, chunk c: byte offset = (row within the split) * Cout * 2 + column; rows past the split's end are past the
    // descriptor's range (zeros), so are all rows of a chunk past the last one
    unsigned dz_col[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) dz_col[n] = (unsigned)(prow[n] * a.Cout + co0 + pblk[n] * 64 + punit[n] * 8) * 2u;
    const unsigned dz_chunk = (unsigned)(kRows * a.Cout) * 2u;
    auto stage_dz = [&](int h, int buf, int c) {       // channels [co0 + 128 h, + 128) of the chunk's 64 pixels
#ifdef MVF_WGRAD_ABLATE
        if ((a.abl & 32) && c >= 2) return;
#endif
#pragma unroll
        for (int n = 0; n < 2; ++n)
            glds16(gs_dz, lds0 + (unsigned)(buf * kBuf + (4 + h * 2 + pblk[n]) * kBlk + prow0[n] * 128),
                   c < nchunks ? dz_col[n] + (unsigned)c * dz_chunk + (unsigned)h * 256u : kOOB);
    };
    // x piece n of half h: its 64-column block lies inside ONE tap (host: cin % 64 == 0) -> tap and channel offset are per-thread constants;
    // the pixel decomposition (two magic divisions per row) is done ONCE per chunk for the thread's two rows, by x_offsets(), which the
    // loop calls inside the MFMA block of a phase: beside the wave's own matrix instructions VALU issue is nearly free, while in the
    // load segment (the partner wave holds the matrix pipe at raised priority) every VALU instruction is on the critical path.
    int x_kh[2][2], x_kw[2][2];
    unsigned x_cofs[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int kcol = k0 + h * 128 + pblk[n] * 64;
            const int tap = kcol / a.Cin, ci = kcol - tap * a.Cin + punit[n] * 8;
            x_kh[h][n] = tap / a.KW;
            x_kw[h][n] = tap - x_kh[h][n] * a.KW;
            x_cofs[h][n] = (unsigned)((x_kh[h][n] * a.W + x_kw[h][n]) * ps + ci) * 2u;
        }
    unsigned xo[2][2];                                 // [half][piece] byte offsets of the chunk prepared last (out of range = zeros)
    // x_step(q, c), q = 0..15: the q-th slice (about five VALU instructions) of chunk c's offset arithmetic -- rows n = q >> 3, stages
    // q & 7 -- so that phase X can issue one slice behind each of its 16 matrix instructions (left to itself the scheduler puts all
    // ~75 instructions behind the last MFMA, where they delay the barrier; sched_group_barrier pipelines were not honoured here)
    int t_m[2], t_img[2], t_rem[2], t_oh[2], t_ih0[2], t_iw0[2], t_live[2];
    unsigned t_q[2], t_base[2];
    auto x_step = [&](int q, int c) {
        const int n = q >> 3;
        // (every slice ends by making its result opaque: otherwise the sinking passes move the whole chain down to its use, the DMA
        // statements of the NEXT phase's load segment -- exactly where it must not be)
#define MVF_KEEP(v) asm volatile("" : "+v"(v))
        switch (q & 7) {
        case 0: t_m[n] = m_begin + c * kRows + prow[n]; t_q[n] = __umulhi((unsigned)t_m[n], a.fd_hw_mul); MVF_KEEP(t_q[n]); break;
        case 1: t_img[n] = (int)((t_q[n] + (unsigned)t_m[n]) >> a.fd_hw_shr); t_rem[n] = t_m[n] - t_img[n] * hw_o; MVF_KEEP(t_rem[n]); break;
        case 2: t_oh[n] = (int)((__umulhi((unsigned)t_rem[n], a.fd_w_mul) + (unsigned)t_rem[n]) >> a.fd_w_shr); MVF_KEEP(t_oh[n]); break;
        case 3: t_ih0[n] = t_oh[n] * a.stride - a.pad; t_iw0[n] = (t_rem[n] - t_oh[n] * a.Wo) * a.stride - a.pad; MVF_KEEP(t_ih0[n]); MVF_KEEP(t_iw0[n]); break;
        case 4: t_base[n] = (unsigned)(((t_img[n] * a.H + t_ih0[n]) * a.W + t_iw0[n]) * ps) * 2u; MVF_KEEP(t_base[n]); break;       // (may wrap for a padded corner; valid taps land in range)
        case 5: t_live[n] = (c < nchunks && t_m[n] < m_end) ? 1 : 0; MVF_KEEP(t_live[n]); break;
        case 6: xo[0][n] = (t_live[n] && (unsigned)(t_ih0[n] + x_kh[0][n]) < (unsigned)a.H && (unsigned)(t_iw0[n] + x_kw[0][n]) < (unsigned)a.W) ? t_base[n] + x_cofs[0][n] : kOOB; MVF_KEEP(xo[0][n]); break;
        default: xo[1][n] = (t_live[n] && (unsigned)(t_ih0[n] + x_kh[1][n]) < (unsigned)a.H && (unsigned)(t_iw0[n] + x_kw[1][n]) < (unsigned)a.W) ? t_base[n] + x_cofs[1][n] : kOOB; MVF_KEEP(xo[1][n]); break;
        }
#undef MVF_KEEP
    };
    auto x_offsets = [&](int c) {
#pragma unroll
        for (int q = 0; q < 16; ++q) x_step(q, c);
    };
    auto stage_x = [&](int h, int buf) {               // im2col columns [k0 + 128 h, + 128) of the chunk x_offsets() prepared
#ifdef MVF_WGRAD_ABLATE
        if (a.abl & 32) return;
#endif
#pragma unroll
        for (int n = 0; n < 2; ++n) glds16(gs_x, lds0 + (unsigned)(buf * kBuf + (h * 2 + pblk[n]) * kBlk + prow0[n] * 128), xo[h][n]);
    };
    // transpose reads (wgrad_bf16_kernel): 16-lane group g supplies pixel rows (i >> 2) + 8 (g >> 1), channel quad 16 (g & 1) + 4 (i & 3)
    const int tg = lane >> 4, ti = lane & 15;
    const int trow = (ti >> 2) + 8 * (tg >> 1);
    const int tunit = 2 * (tg & 1) + ((ti & 3) >> 1), thalf = ti & 1;
    // fragment = 32 channels = units [4 f, 4 f + 4) of a 128-byte row, f = 0 / 1
    const int tofs0 = trow * 128 + (((0 * 4 + tunit) ^ swz16<8>(trow)) * 16) + thalf * 8;
    const int tofs1 = trow * 128 + (((1 * 4 + tunit) ^ swz16<8>(trow)) * 16) + thalf * 8;
    auto gather = [&](const char* p) {
        typedef short v4s __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4s* lds_v4s;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p + 4 * 128));
        bf16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };
    // x fragment j: half j >> 1, block wn of the half, row half j & 1;  dz fragment i: half i, block wm >> 1 of the half, row half wm & 1
    auto fetch_x = [&](int buf, int j, bf16x8_t (&f)[4]) {
#ifdef MVF_WGRAD_ABLATE
        if (a.abl & 16) { for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(f[ks])); return; }
#endif
        const char* p = smem16 + buf * kBuf + ((j >> 1) * 2 + wn) * kBlk + ((j & 1) ? tofs1 : tofs0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f[ks] = gather(p + ks * 16 * 128);
    };
    const int dofs = (wm & 1) ? tofs1 : tofs0;
    auto fetch_dz = [&](int buf, int i, bf16x8_t (&f)[4]) {
#ifdef MVF_WGRAD_ABLATE
        if (a.abl & 16) { for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(f[ks])); return; }
#endif
        const char* p = smem16 + buf * kBuf + (4 + i * 2 + (wm >> 1)) * kBlk + dofs;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f[ks] = gather(p + ks * 16 * 128);
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#define MVF_PIN2(x, y) asm volatile("" : "+v"(x), "+v"(y))
#ifdef MVF_WGRAD_ABLATE
#define MVF_WG_PRIO(v) do { if (!(a.abl & 4)) __builtin_amdgcn_s_setprio(v); } while (0)
#else
#define MVF_WG_PRIO(v) __builtin_amdgcn_s_setprio(v)
#endif
    // prologue in the steady-state issue order: chunk 0 whole, then dz-half 1 and x-half 0 of chunk 1
    x_offsets(0);
    stage_dz(1, 0, 0); stage_x(0, 0); stage_dz(0, 0, 0); stage_x(1, 0);
    x_offsets(1);                                      // its half 1 is staged by X(0)
    stage_dz(1, 1, 1); stage_x(0, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8_t fx[2][4] = {}, fd0[4] = {}, fd1a[4] = {}, fd1b[4] = {};
    fetch_dz(0, 1, fd1a);
    if (grp == 1) __builtin_amdgcn_s_barrier();        // the second group runs one barrier behind the first
    auto phase_x = [&](int c, int buf, bf16x8_t (&d1)[4]) {
        fetch_x(buf, 0, fx[0]);
        fetch_x(buf, 1, fx[1]);
        __builtin_amdgcn_sched_barrier(0);
        fetch_dz(buf, 0, fd0);
        __builtin_amdgcn_sched_barrier(0);
        stage_dz(0, buf ^ 1, c + 1);
        stage_x(1, buf ^ 1);                               // x-half 1 of chunk c + 1 (offsets from the previous X phase's MFMA block)
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(8)" ::: "memory");        // x reads (16, issued first) retired: their slot is restaged next phase
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        MVF_PIN2(acc[0][0], acc[0][1]); MVF_PIN2(acc[1][0], acc[1][1]);
        MVF_WG_PRIO(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                   // one slice of chunk c + 2's offset arithmetic behind every matrix instruction
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd0[ks], fx[0][ks], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0); x_step(4 * ks + 0, c + 2); __builtin_amdgcn_sched_barrier(0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd0[ks], fx[1][ks], acc[0][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0); x_step(4 * ks + 1, c + 2); __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1[ks], fx[0][ks], acc[1][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0); x_step(4 * ks + 2, c + 2); __builtin_amdgcn_sched_barrier(0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1[ks], fx[1][ks], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0); x_step(4 * ks + 3, c + 2); __builtin_amdgcn_sched_barrier(0);
        }
        MVF_WG_PRIO(0);
        MVF_PIN2(acc[0][0], acc[0][1]); MVF_PIN2(acc[1][0], acc[1][1]);
        __builtin_amdgcn_s_barrier();
    };
    auto phase_y = [&](int c, int buf, bf16x8_t (&d1)[4], bf16x8_t (&d1n)[4]) {
        fetch_x(buf, 2, fx[0]);
        fetch_x(buf, 3, fx[1]);
        __builtin_amdgcn_sched_barrier(0);
        fetch_dz(buf ^ 1, 1, d1n);
        __builtin_amdgcn_sched_barrier(0);
        stage_dz(1, buf, c + 2);
        stage_x(0, buf);                                   // x-half 0 of chunk c + 2 (offsets from this chunk's X phase)
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        MVF_PIN2(acc[0][2], acc[0][3]); MVF_PIN2(acc[1][2], acc[1][3]);
        MVF_WG_PRIO(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd0[ks], fx[0][ks], acc[0][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd0[ks], fx[1][ks], acc[0][3], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1[ks], fx[0][ks], acc[1][2], 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1[ks], fx[1][ks], acc[1][3], 0, 0, 0);
        }
        MVF_WG_PRIO(0);
        MVF_PIN2(acc[0][2], acc[0][3]); MVF_PIN2(acc[1][2], acc[1][3]);
        __builtin_amdgcn_s_barrier();
    };
#ifdef MVF_WGRAD_ABLATE
    const int nloop = (a.abl & 1) ? 0 : nchunks;
#else
    const int nloop = nchunks;
#endif
    for (int c = 0; c < nloop; c += 2) {
        phase_x(c, 0, fd1a);
        phase_y(c, 0, fd1a, fd1b);
        if (c + 1 < nchunks) {
            phase_x(c + 1, 1, fd1b);
            phase_y(c + 1, 1, fd1b, fd1a);
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef MVF_PIN2
#ifdef MVF_WGRAD_ABLATE
    if (a.abl & 2) {
        if (acc[0][0][0] == 123.456f) a.part[0] = acc[1][3][5];      // keep the accumulators live
        return;
    }
#endif
    // partial[split][co][kcol] with the fragment -> column maps above
    float* outp = a.part + (long)wg_split * a.Cout * a.K;
    const int lr = lane >> 5, lc = lane & 31;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = k0 + (j >> 1) * 128 + wn * 64 + (j & 1) * 32 + lc;
        if (col >= a.K) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + i * 128 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lr;
                if (row < a.Cout) outp[(long)row * a.K + col] = acc[i][j][r];
            }
    }
}

template <int TM, int TN>
int launch_wgrad_bf16(const WgArgs& a, int tiles, int nsplit, hipStream_t st) {
    constexpr int BCO = 2 * TM * 32, BK = 2 * TN * 32;
    const size_t lds = (size_t)2 * BMR16 * (BCO * 2 + BK * 2);
    // LDS-DMA loaders by default where they measured faster (R50 bf16 train step, per layer): +6-15 % on the 3x3 and the wide
    // pointwise layers, -3-7 % on layer1's K = 64 pointwise convs and the stem (kept on the register-staged kernel); the MVF
    // split operand cannot use them.  policy wgrad_dma=0 / 2 = never / wherever possible.  Weight gradients alone 5.23 -> 4.78 ms.
    static const int dma_env = mvf_policy_int("wgrad_dma", 1);
    const bool dma = dma_env && (a.split_c == 0 || (dma_env != 3 && a.split_c % BK == 0 && a.Cin % BK == 0)) &&
                     (long)a.N * a.H * a.W * std::max(a.xps, a.x2ps) * 2 < 0x7ffffff0L && (dma_env == 2 || (a.K >= 128 && a.Cin >= 32));
    // [r6] policy wgrad_stages: LDS-DMA ring depth of the four-wave tile (2 = the two-buffer kernel; default 3 = two chunks in flight; 4 measured the same)
    static const int stages_env = mvf_policy_int("wgrad_stages", 3);
    if (dma && stages_env >= 3) {
        const size_t lds_p = (size_t)(stages_env >= 4 ? 4 : 3) * BMR16 * (BCO * 2 + BK * 2);
        auto k3 = wgrad_bf16_pipe_kernel<TM, TN, 3>;
        auto k4 = wgrad_bf16_pipe_kernel<TM, TN, 4>;
        static bool attr_p = false;
        if (!attr_p) {
            MVF_HIP_OK(hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(3 * BMR16 * (BCO * 2 + BK * 2))));
            MVF_HIP_OK(hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * BMR16 * (BCO * 2 + BK * 2))));
            attr_p = true;
        }
        if (stages_env >= 4) hipLaunchKernelGGL(k4, dim3(nsplit * tiles), dim3(kThreads), lds_p, st, a);
        else hipLaunchKernelGGL(k3, dim3(nsplit * tiles), dim3(kThreads), lds_p, st, a);
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    if (dma) {
        auto kd = wgrad_bf16_kernel<TM, TN, true>;
        static bool attr_d = false;
        if (!attr_d) {
            MVF_HIP_OK(hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_d = true;
        }
        hipLaunchKernelGGL(kd, dim3(nsplit * tiles), dim3(kThreads), lds, st, a);
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    auto kern = wgrad_bf16_kernel<TM, TN>;
    static bool attr_done = false;
    if (!attr_done) {
        MVF_HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(nsplit * tiles), dim3(kThreads), lds, st, a);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int launch_wgrad_bf16_big(const WgArgs& a, int tiles, int nsplit, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * BMR16 * (256 * 2 + 256 * 2);
    auto kd = wgrad_bf16_kernel<2, 4, true, 4>;
    auto kp = wgrad_bf16_p4_kernel;
    static bool attr_d = false;
    if (!attr_d) {
        MVF_HIP_OK(hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MVF_HIP_OK(hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_d = true;
    }
    // the four-phase ping-pong loop carries the tile when a 64-column block lies inside one tap (policy wgrad_p4=0: the two-barrier loop)
    static const int p4_on = mvf_policy_int("wgrad_p4", 1);
    if (p4_on && a.Cin % 64 == 0) {
        hipLaunchKernelGGL(kp, dim3(nsplit * tiles), dim3(512), lds, st, a);
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    hipLaunchKernelGGL(kd, dim3(nsplit * tiles), dim3(512), lds, st, a);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

// dW_oihw[co][ci][kh][kw] = sum_s part[s][co][(kh*KWP + kw)*CINP + ci]   (KWP/CINP = packed extents; stem: 8 / 4): one element per thread, four split-lanes
// combined by a fixed LDS tree (deterministic, no atomics).  ([r4] a 16-byte-load form measured the same alone -- the slabs come from L2 / MALL -- and
// slower in the step, where its wider loads took more from the launch stream's kernels than they gave back: removed in round 6.)
__global__ __launch_bounds__(256) void wgrad_reduce1_kernel(const float* part, int nsplit, int cout, int cin, int kh, int kw, int kwp, int cinp,
                                    float* dw) {
    __shared__ float red[4][64];
    const long K = (long)kh * kwp * cinp;
    const long total = (long)cout * K;
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < total) {
        int k = sl;
        for (; k + 12 < nsplit; k += 16) {
            s0 += part[(long)k * total + i];
            s1 += part[(long)(k + 4) * total + i];
            s2 += part[(long)(k + 8) * total + i];
            s3 += part[(long)(k + 12) * total + i];
        }
        for (; k < nsplit; k += 4) s0 += part[(long)k * total + i];
    }
    red[sl][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && i < total) {
        long t = i;
        const int ci = (int)(t % cinp); t /= cinp;
        const int x = (int)(t % kwp); t /= kwp;
        const int y = (int)(t % kh);
        const int co = (int)(t / kh);
        if (ci < cin && x < kw) dw[(((long)co * cin + ci) * kh + y) * kw + x] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    }
}
int launch_wgrad_reduce(const float* part, int nsplit, int cout, int cin, int kh, int kw, int kwp, int cinp, float* dw, hipStream_t st) {
    const long total = (long)cout * kh * kwp * cinp;
    hipLaunchKernelGGL(wgrad_reduce1_kernel, dim3((int)((total + 63) / 64)), dim3(256), 0, st, part, nsplit, cout, cin, kh, kw, kwp, cinp, dw);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

template <typename ET>
__global__ void pack_dgrad_weight_kernel(const float* w, int cout, int cin, int kh, int kw, ET* out) {
    // out[ci][kh'][kw'][co] = w[co][ci][KH-1-kh'][KW-1-kw']
    const long total = (long)cin * kh * kw * cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int co = (int)(t % cout); t /= cout;
        const int x = (int)(t % kw); t /= kw;
        const int y = (int)(t % kh);
        const int ci = (int)(t / kh);
        stf(out + i, w[(((long)co * cin + ci) * kh + (kh - 1 - y)) * kw + (kw - 1 - x)]);
    }
}

// All weight packs of a step in ONE launch (53 forward + 52 data-gradient packs are ~5 us of dispatch each when launched one
// by one): workgroup b serves the job whose [first_block, next first_block) range holds b, 2048 elements per workgroup.
// kind 0: out[co][kh][kw_pad][cin_pad] = w[co][ci][kh][kw] (zero padded)      (= pack_weight_kernel, no scale)
// kind 1: out[ci][kh'][kw'][co] = w[co][ci][KH-1-kh'][KW-1-kw']               (= pack_dgrad_weight_kernel)
// kind 2 / 3: the same two packs as kind 0 / 1 for cout % 32 == 0, cin % 32 == 0, kh * kw <= 9, no padding, done as an LDS-tiled
// transpose: a workgroup owns 32 output x 32 input channels x all taps, reads 32 contiguous runs of 32 * taps floats (whole
// cache lines; the gather of kind 0 / 1 touches one line per LANE for the data-gradient layout) and writes 64-byte runs.
// LDS image [co][ci * taps + tap] with a row pitch of 32 * taps + 1 floats: lanes along ci step by `taps` (odd -> conflict-free),
// lanes along co by the odd pitch.  The two batched packs took 0.36 ms of every training step (0.8 TB/s); blocks per job =
// (cout / 32) * (cin / 32).
constexpr int kPackT = 32, kPackTapsMax = 9;
template <typename ET>
__global__ __launch_bounds__(256) void pack_batched_kernel(const mvf_pack_job_t* jobs, int njobs) {
    __shared__ float tile[kPackT * (kPackT * kPackTapsMax + 1)];
    int lo = 0, hi = njobs - 1;                          // last job with first_block <= blockIdx.x (wave-uniform binary search)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const mvf_pack_job_t j = jobs[lo];
    const float* w = j.w;
    ET* out = reinterpret_cast<ET*>(j.out);
    if (j.kind >= 2) {
        const int taps = j.kh * j.kw, run = kPackT * taps, pitch = run + 1;
        const int b = (int)blockIdx.x - j.first_block, tiles_ci = j.cin / kPackT;
        const int co0 = (b / tiles_ci) * kPackT, ci0 = (b % tiles_ci) * kPackT;
        for (int e = threadIdx.x; e < kPackT * run; e += 256) {            // row co0 + r: floats [ci0 * taps, (ci0 + 32) * taps)
            const int r = e / run, q = e - r * run;
            tile[r * pitch + q] = w[((long)(co0 + r) * j.cin + ci0) * taps + q];
        }
        __syncthreads();
        const int l = threadIdx.x & 31, g = threadIdx.x >> 5;               // 8 groups of 32 lanes
        if (j.kind == 2) {                                                  // out[co][tap][ci]: lanes along ci
            for (int p = g; p < kPackT * taps; p += 8) {
                const int r = p / taps, t = p - r * taps;
                stf(out + ((long)(co0 + r) * taps + t) * j.cin + ci0 + l, tile[r * pitch + l * taps + t]);
            }
        } else {                                                            // out[ci][flipped tap][co]: lanes along co
            for (int p = g; p < kPackT * taps; p += 8) {
                const int c = p / taps, t = p - c * taps;
                stf(out + ((long)(ci0 + c) * taps + t) * j.cout + co0 + l, tile[l * pitch + c * taps + (taps - 1 - t)]);
            }
        }
        return;
    }
    const long base = (long)((int)blockIdx.x - j.first_block) * 2048;
    const long total = j.kind == 0 ? (long)j.cout * j.kh * j.kw_pad * j.cin_pad : (long)j.cin * j.kh * j.kw * j.cout;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long i = base + u * 256 + threadIdx.x;
        if (i >= total) break;
        long t = i;
        float v = 0.f;
        if (j.kind == 0) {
            const int ci = (int)(t % j.cin_pad); t /= j.cin_pad;
            const int x = (int)(t % j.kw_pad); t /= j.kw_pad;
            const int yk = (int)(t % j.kh);
            const int co = (int)(t / j.kh);
            if (ci < j.cin && x < j.kw) v = w[(((long)co * j.cin + ci) * j.kh + yk) * j.kw + x];
        } else {
            const int co = (int)(t % j.cout); t /= j.cout;
            const int x = (int)(t % j.kw); t /= j.kw;
            const int y = (int)(t % j.kh);
            const int ci = (int)(t / j.kh);
            v = w[(((long)co * j.cin + ci) * j.kh + (j.kh - 1 - y)) * j.kw + (j.kw - 1 - x)];
        }
        stf(out + i, v);
    }
}

int plan_split(int M, int tiles, int target_override = 0) {
    static const int target_env = std::max(64, mvf_policy_int("wgrad_wgs", 1024));     // workgroups aimed at per launch (A/B switch)
    const int target = target_override > 0 ? target_override : target_env;
    int want = std::max(1, target / std::max(tiles, 1));
    int rows = std::max((M + want - 1) / want, 256);
    rows = (rows + 63) / 64 * 64;          // multiple of both chunk heights (32 fp32 / 64 bf16)
    return rows;
}

struct WgTile { int bco, bk; };
WgTile pick_tile(int cout, int K) {
    if (cout <= 64) return {64, 128};
    if (K <= 64) return {128, 64};
    return {128, 128};
}

}  // namespace

namespace mvf_internal {
// the fixed-order sum of [nsplit][cout][k] fp32 partial slabs of a pointwise conv's weight gradient into dw (cout, k, 1, 1): the tail of
// mvf_conv2d_nhwc_wgrad, also behind the fused BatchNorm-backward + weight-gradient kernels (bnbwd_wgrad.hip)
int wgrad_slab_reduce_launch(const float* part, int nsplit, int cout, int k, float* dw_oihw, hipStream_t st) {
    return launch_wgrad_reduce(part, nsplit, cout, k, 1, 1, 1, k, dw_oihw, st);
}
}  // namespace mvf_internal

extern "C" {

// the 256 x 256 eight-wave tile (bf16, LDS-DMA): shape eligibility (pointer alignment is checked at launch) and its pixel split --
// ~256 workgroups (one per CU), i.e. the same number of splits as the 128 x 128 plan's ~1024
static bool wg_big_shape(const mvf_conv_desc_t* d) {
    // policy wgrad_big: 0 never, 1 (default) every eligible shape, 2 pointwise convs only, 3 pointwise convs with cout >= 512.
    // Measured on the R50 bf16 train step, two alternations: weight gradients ALONE 5.23 (0) / 5.17 (1) / 5.13 (2) / 5.13 (3) ms -- the
    // tile wins -10...-18 % on the wide pointwise layers and loses 5-10 % on the 3x3 ones --, but the STEP is 22.24 / 22.16 / 22.37 /
    // 22.39 ms: one 8-wave workgroup per CU (252 of them for a 3x3 layer) leaves the launch stream's kernels more of the chip than
    // 1008 four-wave workgroups do.
    static const int big_env = mvf_policy_int("wgrad_big", 1);
    const long K = (long)d->kh * d->kw * d->cin;
    if (big_env >= 2 && (d->kh != 1 || d->kw != 1)) return false;
    if (big_env >= 3 && d->cout < 512) return false;
    return big_env && d->dtype == MVF_BF16 && d->cout % 256 == 0 && K % 256 == 0 && d->cin % 8 == 0 && d->x_pix_stride % 4 == 0 &&
           (d->split_c == 0 || (d->split_c % 256 == 0 && d->cin % 256 == 0));
}
static int wg_big_rows(const mvf_conv_desc_t* d, int wgs_override = 0) {
    const int M = d->n * d->ho * d->wo, K = d->kh * d->kw * d->cin;
    // [r5] 128 workgroups (half the CUs), not one per CU: these GEMMs run on the side stream beside the launch stream's kernels, and the two queues share the chip
    // work-conservingly -- what the weight gradients cost the step is their FOOTPRINT (a 256 x 256 workgroup owns its CU's whole register file), not their own
    // length.  Measured in the step (alternating runs; ms, C3 / C4): 256 workgroups 19.20 / 33.08, 192: 18.95 / 32.72, 128: 18.88 / 32.36 on one box; 128: 18.34 / 31.45,
    // 96: 18.35 / 31.91, 64: 18.50 / 32.38 on another.  (Round 2 measured 128 = 256 on a step whose launch stream still carried 7 ms of BatchNorm passes.)
    static const int big_wgs = std::max(32, mvf_policy_int("wgrad_big_wgs", 128));      // A/B switch
    return plan_split(M, (d->cout / 256) * (K / 256), wgs_override > 0 ? wgs_override : big_wgs);
}

// [r4] layer1's 3x3 (64 -> 64 channels, stride 1, pad 1, bf16) on the direct kernel of wgrad3x3_c64.hip (policy wgrad3x3_direct=0: the implicit GEMM)
static bool wg_direct3x3(const mvf_conv_desc_t* d) {
    return d->dtype == MVF_BF16 && d->cin == 64 && d->cout == 64 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->split_c == 0 &&
           d->ho == d->h && d->wo == d->w && mvf_internal::wgrad3x3_c64_ok(d->n, d->h, d->w, d->x_pix_stride);
}

// [r4] the stem (7 x 1 taps over the 32-"channel" view of the padded NHWC4 operand, stride 2, bf16) on the direct kernel of wgrad_stem.hip (policy wgrad_stem_direct=0: the implicit GEMM)
static bool wg_direct_stem(const mvf_conv_desc_t* d) {
    return d->dtype == MVF_BF16 && d->cin == 32 && d->cout == 64 && d->kh == 7 && d->kw == 1 && d->stride == 2 && d->pad == 0 && d->split_c == 0 &&
           d->x_pix_stride == 4 && mvf_internal::wgrad_stem_ok(d->n, d->h, d->w, d->ho, d->wo);
}

size_t mvf_conv2d_wgrad_workspace_bytes(const mvf_conv_desc_t* d) {
    if (!d || d->cout <= 0 || d->cin <= 0) return 0;
    const int M = d->n * d->ho * d->wo, K = d->kh * d->kw * d->cin;
    const WgTile t = pick_tile(d->cout, K);
    const int tiles = ((d->cout + t.bco - 1) / t.bco) * ((K + t.bk - 1) / t.bk);
    const int rows = plan_split(M, tiles);
    int nsplit = (M + rows - 1) / rows;
    if (wg_big_shape(d)) nsplit = std::max(nsplit, (M + wg_big_rows(d) - 1) / wg_big_rows(d));
    if (wg_direct3x3(d)) nsplit = std::max(nsplit, mvf_internal::wgrad3x3_c64_wgs(d->n, d->h));
    if (wg_direct_stem(d)) nsplit = std::max(nsplit, mvf_internal::wgrad_stem_wgs(d->n, d->ho));
    return align_up((size_t)nsplit * d->cout * K * sizeof(float), 256);
}

// dw_oihw (cout, cin_real, kh, kw_real) fp32 <- dz (n,ho,wo,cout), x as in the forward descriptor.
// kw_real/cin_real < packed extents only for the stem view (kh x 1 x 32 over the padded NHWC4 input = 7 x 8 x 4).
}  // extern "C"
static int wgrad_impl(const mvf_conv_desc_t* d, const void* dz, const void* x, const void* x2, int kw_real, int cin_real,
                      int kw_packed, int cin_packed, float* dw_oihw, void* ws, size_t ws_bytes, void* stream, int wgs_target) {
    MVF_REQUIRE(d && dz && x && dw_oihw, MVF_EINVAL, "wgrad: NULL argument");
    MVF_REQUIRE(d->dtype == MVF_F32 || d->dtype == MVF_BF16, MVF_EINVAL, "wgrad: bad dtype");
    MVF_REQUIRE(d->cin % 4 == 0 && d->cout % 4 == 0 && d->x_pix_stride > 0, MVF_ESHAPE, "wgrad: cin/cout must be multiples of 4");
    MVF_REQUIRE(kw_packed * cin_packed == d->kw * d->cin && kw_real <= kw_packed && cin_real <= cin_packed, MVF_EINVAL, "wgrad: packed extents inconsistent");
    MVF_REQUIRE(ws && ws_bytes >= mvf_conv2d_wgrad_workspace_bytes(d), MVF_EWS, "wgrad: workspace too small");
    if (d->split_c) MVF_REQUIRE(x2 && d->kh == 1 && d->kw == 1 && d->split_c % 4 == 0, MVF_EINVAL, "wgrad: bad split_c");
    hipStream_t st0 = (hipStream_t)stream;
    if (wg_direct3x3(d) && kw_real == 3 && cin_real == 64 && kw_packed == 3 && cin_packed == 64 && ((uintptr_t)dz | (uintptr_t)x) % 16 == 0) {
        Wgrad3x3C64Args w = {dz, x, (float*)ws, d->n, d->h, d->w, d->x_pix_stride, mvf_internal::wgrad3x3_c64_wgs(d->n, d->h)};
        const int rc = mvf_internal::wgrad3x3_c64_launch(w, st0);
        if (rc != MVF_OK) return rc;
        return launch_wgrad_reduce((const float*)ws, w.nwg, 64, 64, 3, 3, 3, 64, dw_oihw, st0);
    }
    if (wg_direct_stem(d) && kw_real == 7 && cin_real == 3 && kw_packed == 8 && cin_packed == 4 && ((uintptr_t)dz | (uintptr_t)x) % 16 == 0) {
        WgradStemArgs w = {dz, x, (float*)ws, d->n, d->h, d->w, d->ho, d->wo, mvf_internal::wgrad_stem_wgs(d->n, d->ho)};
        const int rc = mvf_internal::wgrad_stem_launch(w, st0);
        if (rc != MVF_OK) return rc;
        return launch_wgrad_reduce((const float*)ws, w.nwg, 64, 3, 7, 7, 8, 4, dw_oihw, st0);
    }
    WgArgs a = {};
    a.dz = dz; a.x = x; a.x2 = x2; a.part = (float*)ws;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.KH = d->kh; a.KW = d->kw;
    a.stride = d->stride; a.pad = d->pad; a.Ho = d->ho; a.Wo = d->wo; a.xps = d->x_pix_stride;
    a.split_c = d->split_c; a.x2ps = d->x2_pix_stride;
    a.M = d->n * d->ho * d->wo; a.K = d->kh * d->kw * d->cin;
    WgTile t = pick_tile(d->cout, a.K);
    // [r5] a Gram matrix (dz == x: a^T a of ONE tensor, the dz3-free BatchNorm backward's A2, bn_dzfree.hip): its output is a single small tile, so the usual
    // plan -- one workgroup per CU -- spends 256 x 256 KB of fp32 slabs (written, then read by the reduce) on a 256 KB result: 128 MB of traffic per call, 2.4 GB per
    // C3 step.  It runs on the side stream with slack, so it takes policy gram_wgs (default 32) workgroups instead: an eighth of the slabs and of the CUs.  Measured in the
    // step (alternating runs; ms, C3 / C4): 256 workgroups 18.64 / 32.2, 128: 18.68 / 32.45, 64: 18.75 / 32.6, 32: 18.59 / 31.8 on one box; 32: 19.10 / 33.08, 24: 19.08 /
    // 33.1, 16: 19.16 / 33.3, 8: 19.30 / 33.6 on another.
    static const int gram_env = std::max(8, mvf_policy_int("gram_wgs", 32));
    // (wgs_target > 0, mvf_conv2d_nhwc_wgrad_wgs: the caller names the workgroup count to aim at -- a GEMM the LAUNCH stream waits for wants the whole chip)
    const int gram_wgs = wgs_target > 0 ? wgs_target : (dz == x && !x2 && d->kh == 1 && d->kw == 1 && d->cin == d->cout) ? gram_env : 0;
    // 256 x 256 tile: the shapes of wg_big_shape() when the LDS-DMA address ranges and alignments hold and every split has >= 4 chunks
    // (a caller-named workgroup count takes the 128 x 128 plans)
    bool big = wg_big_shape(d) && wgs_target <= 0 && ((uintptr_t)dz | (uintptr_t)x | (uintptr_t)(x2 ? x2 : x)) % 16 == 0 &&
               (long)a.N * a.H * a.W * std::max(a.xps, a.x2ps) * 2 < 0x7ffffff0L;
    if (big) {
        const int rows = wg_big_rows(d, gram_wgs);
        big = rows >= 4 * 64 && ((long)rows + 6 * 64) * d->cout * 2 < 0x7ffffff0L;
        if (big) t = WgTile{256, 256};
    }
    a.tiles_k = (a.K + t.bk - 1) / t.bk;
    const int tiles = ((d->cout + t.bco - 1) / t.bco) * a.tiles_k;
    // [r4] fp32 storage on the bf16 matrix cores (wgrad_x3_kernel): on with the conv kernels' switch (policy f32_x3 != 0) unless policy wgrad_x3=0.
    // Three workgroups per CU = 768 slots: the pixel split aims at one full round of them (1024 / 1536 / 2304: 3185 / 3218 / 3308 us over the
    // twelve C3 shapes against 3115), unless policy wgrad_wgs says otherwise
    static const int x3_env = (mvf_policy_int("f32_x3", 1) != 0) && (mvf_policy_int("wgrad_x3", 1) != 0);
    bool x3 = d->dtype == MVF_F32 && x3_env && d->cin % 4 == 0 && d->cout % 4 == 0 && d->x_pix_stride % 4 == 0 && d->split_c % 4 == 0 &&
              (d->x2_pix_stride % 4 == 0 || !d->split_c) && ((uintptr_t)dz | (uintptr_t)x | (uintptr_t)(x2 ? x2 : x)) % 16 == 0;
    static const bool wgs_forced = mvf_policy_has("wgrad_wgs");
    // [r5] bf16 128 x 128 plans aim at 256 workgroups instead of 1024 for the same reason (with the big tile at 128; ms, C3 / C4: 1024: 18.34 / 31.45, 512: 18.20 / 31.29,
    // 384: 18.21 / 31.23, 256: 18.15 / 31.30)
    const int bf16_wgs = (d->dtype == MVF_BF16 && !wgs_forced) ? 256 : 0;
    a.rows_per_split = big ? wg_big_rows(d, gram_wgs) : plan_split(a.M, tiles, gram_wgs ? gram_wgs : (x3 && !wgs_forced ? 768 : bf16_wgs));
    x3 = x3 && ((long)a.rows_per_split + 64) * d->cout * 4 < 0x7ffffff0L;
    if (d->dtype == MVF_F32 && !x3 && !wgs_forced && !gram_wgs) a.rows_per_split = plan_split(a.M, tiles);
    const int nsplit = (a.M + a.rows_per_split - 1) / a.rows_per_split;
    MVF_REQUIRE(ws_bytes >= (size_t)nsplit * d->cout * a.K * sizeof(float), MVF_EWS, "wgrad: workspace too small for %d slabs", nsplit);
    a.tiles = tiles;
    a.nsplit = nsplit;
    a.xcd_rr = nsplit % 8 == 0;
#ifdef MVF_WGRAD_ABLATE
    a.abl = mvf_policy_int("wgrad_abl", 0);
#endif
    wg_fd_make((unsigned)(a.Ho * a.Wo), a.fd_hw_mul, a.fd_hw_shr);
    wg_fd_make((unsigned)a.Wo, a.fd_w_mul, a.fd_w_shr);
    hipStream_t st = (hipStream_t)stream;
    // fp32 storage: LDS-DMA loaders by default (weight gradients 25.3 -> 21.8 ms per R50 step = 82 -> 96 TF/s, fp32 step 75.9 ->
    // 73.8 ms); policy wgrad_dma_f32=0 restores the register-staged loaders
    static const int dma32_env = mvf_policy_int("wgrad_dma_f32", 1);
    const bool dma32 = d->dtype == MVF_F32 && dma32_env && d->cin % 4 == 0 && d->cout % 4 == 0 && d->x_pix_stride % 4 == 0 &&
                       (a.split_c == 0 || (a.split_c % t.bk == 0 && a.Cin % t.bk == 0)) &&
                       ((uintptr_t)dz | (uintptr_t)x | (uintptr_t)(x2 ? x2 : x)) % 16 == 0 &&
                       (long)a.N * a.H * a.W * std::max(a.xps, a.x2ps) * 4 < 0x7ffffff0L && ((long)a.rows_per_split + 64) * d->cout * 4 < 0x7ffffff0L;
    if (x3) {
        if (t.bco == 64) hipLaunchKernelGGL((wgrad_x3_kernel<1, 2>), dim3(nsplit * tiles), dim3(256), 0, st, a);
        else if (t.bk == 64) hipLaunchKernelGGL((wgrad_x3_kernel<2, 1>), dim3(nsplit * tiles), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wgrad_x3_kernel<2, 2>), dim3(nsplit * tiles), dim3(256), 0, st, a);
    } else if (dma32) {
        if (t.bco == 64) hipLaunchKernelGGL((wgrad_kernel<float, 1, 2, true>), dim3(nsplit * tiles), dim3(kThreads), 0, st, a);
        else if (t.bk == 64) hipLaunchKernelGGL((wgrad_kernel<float, 2, 1, true>), dim3(nsplit * tiles), dim3(kThreads), 0, st, a);
        else hipLaunchKernelGGL((wgrad_kernel<float, 2, 2, true>), dim3(nsplit * tiles), dim3(kThreads), 0, st, a);
    } else if (d->dtype == MVF_F32) {
        if (t.bco == 64) hipLaunchKernelGGL((wgrad_kernel<float, 1, 2>), dim3(nsplit * tiles), dim3(kThreads), 0, st, a);
        else if (t.bk == 64) hipLaunchKernelGGL((wgrad_kernel<float, 2, 1>), dim3(nsplit * tiles), dim3(kThreads), 0, st, a);
        else hipLaunchKernelGGL((wgrad_kernel<float, 2, 2>), dim3(nsplit * tiles), dim3(kThreads), 0, st, a);
    } else if (d->cin % 8 == 0 && d->cout % 8 == 0 && d->x_pix_stride % 4 == 0 && (d->split_c % 8) == 0 &&
               ((uintptr_t)dz | (uintptr_t)x | (uintptr_t)(x2 ? x2 : x)) % 16 == 0 &&
               ((long)a.rows_per_split + 6 * 64) * d->cout * 2 < 0x7ffffff0L) {      // 32-bit split-relative dz offsets (incl. look-ahead chunks)
        int rc;
        if (big) rc = launch_wgrad_bf16_big(a, tiles, nsplit, st);
        else if (t.bco == 64) rc = launch_wgrad_bf16<1, 2>(a, tiles, nsplit, st);
        else if (t.bk == 64) rc = launch_wgrad_bf16<2, 1>(a, tiles, nsplit, st);
        else rc = launch_wgrad_bf16<2, 2>(a, tiles, nsplit, st);
        if (rc) return rc;
    } else {     // odd channel counts: widen to fp32 on the way into LDS
        if (t.bco == 64) hipLaunchKernelGGL((wgrad_kernel<bf16_t, 1, 2>), dim3(nsplit * tiles), dim3(kThreads), 0, st, a);
        else if (t.bk == 64) hipLaunchKernelGGL((wgrad_kernel<bf16_t, 2, 1>), dim3(nsplit * tiles), dim3(kThreads), 0, st, a);
        else hipLaunchKernelGGL((wgrad_kernel<bf16_t, 2, 2>), dim3(nsplit * tiles), dim3(kThreads), 0, st, a);
    }
    MVF_LAUNCH_CHECK();
    const int kh_p = d->kh * d->kw * d->cin / (kw_packed * cin_packed);
    return launch_wgrad_reduce(a.part, nsplit, d->cout, cin_real, kh_p, kw_real, kw_packed, cin_packed, dw_oihw, st);
}
extern "C" {

int mvf_conv2d_nhwc_wgrad(const mvf_conv_desc_t* d, const void* dz, const void* x, const void* x2, int kw_real, int cin_real,
                          int kw_packed, int cin_packed, float* dw_oihw, void* ws, size_t ws_bytes, void* stream) {
    return wgrad_impl(d, dz, x, x2, kw_real, cin_real, kw_packed, cin_packed, dw_oihw, ws, ws_bytes, stream, 0);
}

// [r5] the same GEMM with the workgroup count to aim at named by the caller (the library's own policy sizes weight gradients for the SIDE stream: half the chip);
// same results up to the fp32 summation order of the pixel split
int mvf_conv2d_nhwc_wgrad_wgs(const mvf_conv_desc_t* d, const void* dz, const void* x, const void* x2, int kw_real, int cin_real,
                              int kw_packed, int cin_packed, float* dw_oihw, void* ws, size_t ws_bytes, int wgs, void* stream) {
    MVF_REQUIRE(wgs >= 8 && wgs <= 4096, MVF_EINVAL, "wgrad_wgs: wgs=%d outside 8 .. 4096", wgs);
    return wgrad_impl(d, dz, x, x2, kw_real, cin_real, kw_packed, cin_packed, dw_oihw, ws, ws_bytes, stream, wgs);
}

int mvf_pack_conv_weights_batched(const mvf_pack_job_t* jobs_dev, int njobs, int total_blocks, int dtype, void* stream) {
    MVF_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0 && (dtype == MVF_F32 || dtype == MVF_BF16), MVF_EINVAL, "pack_conv_weights_batched: bad argument");
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(pack_batched_kernel<float>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    else
        hipLaunchKernelGGL(pack_batched_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_pack_conv_weight_dgrad(const float* w_oihw, int cout, int cin, int kh, int kw, void* w_packed, int dtype, void* stream) {
    MVF_REQUIRE(w_oihw && w_packed && cout > 0 && cin > 0 && kh > 0 && kw > 0, MVF_EINVAL, "pack_conv_weight_dgrad: bad argument");
    const long total = (long)cout * cin * kh * kw;
    const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(pack_dgrad_weight_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, kh, kw, (float*)w_packed);
    else
        hipLaunchKernelGGL(pack_dgrad_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, kh, kw, (bf16_t*)w_packed);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // extern "C"
