// Implicit-GEMM convolution on the CDNA4 matrix cores, channels-last, fused bias/residual/ReLU epilogue.
//
//   Y[m][co] = act( sum_{kh,kw,ci} X[img][oh*s+kh-p][ow*s+kw-p][ci] * Wp[co][kh][kw][ci] + bias[co] (+ R[m][co]) )
//   m = (img, oh, ow) flattened, GEMM M = N*Ho*Wo, N = Cout, K = KH*KW*Cin.
//
// Replaces nn.Conv2d -> BatchNorm2d(eval, folded into Wp/bias) -> (+= identity) -> ReLU of the reference's
// Bottleneck (codes/models/backbones/resnet.py:208-244) and stem (:481-483).
//
// Design (gfx950, wave64):
//  * K is walked in chunks of 128 BYTES per row (32 fp32 / 64 bf16 channels of ONE tap), so every A row of a
//    chunk is one contiguous, 128-B aligned run of the NHWC tensor and every B row one run of the packed
//    weights: global loads are 16 B per lane, 8 lanes per row -> whole cache lines.
//  * LDS tiles are row-major [rows][128 B + 16 B pad] (144 B pitch): the global->LDS store is a plain
//    ds_write_b128 (no transpose) and the MFMA operand fetch is ONE ds_read_b128 per lane at
//    (row = lane&31, 16-B unit = 2*kstep + lane>>5): with the 144-B pitch each 16-lane service group of
//    ds_read_b128 hits 16 distinct 16-B slots of the 256-B bank row -> conflict-free (checked by hand
//    against the lane groups in MI355X_MICROARCH.md, LDS table).
//  * fp32: v_mfma_f32_32x32x2_f32 -- one 16-B unit per lane feeds FOUR MFMAs (k-pairs (j, j+4) of the 8
//    channels the two half-waves hold); exact fp32 (an fmaf chain), 157 TF/s peak.
//    bf16: v_mfma_f32_32x32x16_bf16 -- one 16-B unit (8 bf16) per lane feeds ONE MFMA; fp32 accumulate.
//    The byte-level data path is identical for both types.
//  * 256 threads = 4 waves; block tile 128 x 128 (2x2 waves, each 2x2 MFMA tiles of 32x32) or 128 x 64
//    (4x1 waves, each 1x2) for Cout = 64.  Register-staged double buffering: chunk k+1 is loaded from
//    global into VGPRs before the MFMAs of chunk k and written to the other LDS buffer after them; one
//    barrier per chunk.
//  * XCD-aware tile order: the 8 XCDs get contiguous ranges of (m-tile, n-tile) pairs with n fastest, so the
//    n-tiles that re-read one A row panel run on the same XCD and hit its L2.
#include <algorithm>
#include <type_traits>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPitch = 144;          // LDS row pitch in bytes (128 B data + 16 B pad)
constexpr int kBM = 128;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct ConvArgs {
    const char* x;
    const char* x2;
    const char* w;
    const char* res;
    const float* bias;
    char* y;
    int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int xps, split_c, x2ps, relu;
    int res_c0;    // residual only for output channels >= res_c0
    const unsigned char* res_mask;   // optional [M][Cout/4] bytes: bit j of byte k gates residual channel 4k+j (ReLU sign bits)
    // [r5] optional [M][Cout/4] bytes in the same layout: the OUTPUT (accumulator + residual) of channels >= res_c0 is gated by them before the
    // store -- the data gradient then hands the block below gm = g * [out > 0] instead of g + sign bits (mvf_conv2d_nhwc_fwd_resmask_gate)
    const unsigned char* out_gate;
    int x_c0;      // [r5] split operand: x holds channels [split_c, Cin) of the contraction at column (channel - x_c0) of its rows (0: at their own offset)
    int mask_lds;  // stage the gate bytes in LDS (experiment switch policy mask_lds=0)
    int prio;      // experiment switch policy conv_prio=1: raise the wave priority around the MFMA phase of the LDS-DMA loops; in
                   // -DMVF_CONV_ABLATE builds bits 1-5 additionally switch parts of the kernel OFF (timing ablation, wrong results)
    int dil;       // input dilation (generic fallback; the strided data-gradient is normally decomposed into parity classes)
    int pad_w;     // horizontal padding (pad = vertical)
    int w_kh0, w_kw0, w_ts, w_kwfull;   // weight tap (kh,kw) of this launch = full-pack tap (kh0 + kh*ts, kw0 + kw*ts)
    int o_s, o_ph, o_pw, o_hfull, o_wfull;   // o_s > 0: output pixel (oh,ow) lands at (oh*o_s+o_ph, ow*o_s+o_pw) of an o_hfull x o_wfull map
    float* stats_part;         // optional, CHANNEL-MAJOR [Cout][stats_rows][2]: per-128-row column sums of (y - K), (y - K)^2 (BatchNorm batch
                               // statistics); a channel's partial rows are contiguous, so the finalize kernel reads whole cache lines
    int stats_rows;            // row extent of stats_part = mvf_conv2d_stats_rows(d)
    const float* stats_shift;  // K per output channel (the BN's old running mean; any K is exact, a close one avoids cancellation)
    // EPI 6 (data gradient whose output feeds a ReLU(BN(z)) backward): per-tile column sums of gm and gm * xhat into stats_part,
    // gm = y * [bn_scale*z + bn_shift > 0], xhat = (z - bn_mean) * bn_invstd; z has the output's shape
    const char* bn_z;
    const float *bn_mean, *bn_invstd, *bn_scale, *bn_shift;
    // EPI 8 (training forward, second pass of a bottleneck's conv3: reference resnet.py:229-244 conv3 -> bn3 -> += identity -> relu): the
    // accumulators are rounded to the storage type (= the z3 the first pass stored), then out = relu(ap_scale * z3 + ap_shift + residual')
    // with residual' = residual, or ap_rscale * residual + ap_rshift for a downsample branch whose BatchNorm is applied here as well;
    // ap_bits ([M][Cout/4] bytes, bit j of byte k = [out channel 4k+j > 0]) are the sign bits the backward pass gates with.  The arithmetic
    // is bn_apply_kernel's (train_ops.hip), expression by expression.
    const float *ap_scale, *ap_shift, *ap_rscale, *ap_rshift;
    unsigned char* ap_bits;
    // EPI 9 / 10 (training backward of out = relu(bn3(conv3(a2)) + identity), BatchNorm backward WITHOUT a stored z3): the conv is recomputed,
    // its accumulators rounded to the storage type are z3; `res` is the block-output gradient g, `res_mask` the sign bits of out, gm = g * bit.
    //   bw_mode 10: per-128-row column sums of gm and gm * (z3 - bn_mean) * bn_invstd into stats_part (-> mvf_bn_bwd_finalize); nothing is stored
    //   bw_mode  9: y = dz3 = bw_gamma * bn_invstd * (gm - bw_dbeta / M - (z3 - bn_mean) * bn_invstd * bw_dgamma / M)
    // The arithmetic is bn_bwd_reduce_kernel's / bn_bwd_apply_kernel's (train_ops.hip, mask mode 4), expression by expression.
    const float *bw_gamma, *bw_dgamma, *bw_dbeta;
    int bw_mode;
    // MVFL (inference, MVF fused into the wrapped 1x1 conv's A-operand load, reference MVF.py:104-138): channels [0, mvf_cs) of the
    // input are replaced ON THE FLY by hswish(scale * (9-tap T/H/W stencil) + shift); mvf_coef = [mvf_cs][12] floats per channel:
    // {wt[0..2], wh[0..2], ww[0..2], scale, shift, 0}; mvf_act = 1: affine + hard-swish, 0: the bare tap sum (use_hs = False)
    const float* mvf_coef;
    int mvf_cs, mvf_T, mvf_act;
    unsigned fd_t_mul, fd_t_shr;     // magic number of n / mvf_T
    int M;
    int cpt;       // chunks per tap = ceil(Cin*esz / 128)
    int nchunks;   // KH*KW*cpt
    long wK;       // packed weight row length in elements
    int tiles_m, tiles_n;
    unsigned fd_hw_mul, fd_hw_shr, fd_w_mul, fd_w_shr;   // magic numbers: n / (Ho*Wo) and n / Wo for n < 2^31 (set by launch_conv)
};

template <typename ET>
struct TT;
template <>
struct TT<float> {
    static constexpr int ESZ = 4, UE = 4, CE = 32;   // element size, elements per 16-B unit, per 128-B chunk
};
template <>
struct TT<bf16_t> {
    static constexpr int ESZ = 2, UE = 8, CE = 64;
};

// n / d for 0 <= n < 2^31 with host-made magic: l = ceil(log2 d), mul = floor(2^32 (2^l - d) / d) + 1, q = (mulhi(n, mul) + n) >> l
__device__ __forceinline__ int fd_div(int n, unsigned mul, unsigned shr) { return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> shr); }
inline void fd_make(unsigned d, unsigned& mul, unsigned& shr) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    shr = l;
}

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    // bijective remap: XCD x (= bid % 8) owns the contiguous logical range starting at base(x)
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// dynamic LDS of the single-buffer variant: the A/B tiles, or the half C tile + the residual gate bytes of the epilogue
template <int BM, int BN>
__host__ __device__ constexpr int kLowkLds() {
    return (BM + BN) * kPitch > (BM / 2) * (BN * 4 + 16) + BM * (BN / 4) ? (BM + BN) * kPitch : (BM / 2) * (BN * 4 + 16) + BM * (BN / 4);
}

// the single-buffer variant with three bf16 planes per fp32 tile (conv_tile X3): 3 x 64-byte rows, or the epilogue's needs
template <int BM, int BN>
__host__ __device__ constexpr int kLowkLdsX3() {
    return (BM + BN) * 192 > (BM / 2) * (BN * 4 + 16) + BM * (BN / 4) ? (BM + BN) * 192 : (BM / 2) * (BN * 4 + 16) + BM * (BN / 4);
}

// dynamic LDS of the LDS-DMA variant: NB unpadded A/B tile buffers (128-B rows), or the half C tile + gate bytes
template <int BM, int BN, int NB>
__host__ __device__ constexpr int kGldsLds() {
    return NB * (BM + BN) * 128 > (BM / 2) * (BN * 4 + 16) + BM * (BN / 4) ? NB * (BM + BN) * 128 : (BM / 2) * (BN * 4 + 16) + BM * (BN / 4);
}

// What a workgroup does with the accumulators of one (tile, chunk range) segment
enum SegMode { SEG_FULL = 0, SEG_PRODUCE = 1, SEG_FINISH = 2 };

struct SkArgs {            // stream-K tail (see launch_conv): G workgroups share `tail` tiles x nchunks chunk-units
    float* ws;             // [G][tile elements] partial accumulators, one slot per producing workgroup
    unsigned* flags;       // [G] 0 -> 1 when slot g is published
    unsigned* err;         // set to 1 if a bounded spin gives up
    int tile0;             // first tile of the tail set
    int units_base, units_rem, G;
};

// LOWK = single-buffered A/B tiles and a two-pass (half-tile) epilogue: 36 KB of LDS instead of 72 KB -> 3-4 workgroups per CU.
// Used for the small-K pointwise convs (1-2 K chunks), which are HBM-bound and need memory-level parallelism, not MFMA overlap.
// EPI specialises the epilogue at compile time for the three shapes the training step launches all the time (the generic code
// keeps a uniform branch per feature per row): 0 generic, 1 forward + BatchNorm statistics (no bias / residual / ReLU),
// 2 plain (data gradient), 3 data gradient + [gated] residual, 4 bias + ReLU, 5 bias + residual + ReLU (inference),
// 6 plain data gradient + the BatchNorm-backward sums of the BN it feeds.  1-6 imply contiguous output rows (no scatter).
// PW specialises the loader for pointwise launches (1x1 taps, no padding, no split operand): a row is either valid for every
// chunk or never, so the tap masks and the second operand's offsets disappear.
// GLDS = N > 0: the A/B tiles go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds; no staging VGPRs, no ds_write pass, which
// at ~79 B/clk/CU is what bounds the register-staged loop on the long-K layers) into N unpadded buffers.  The DMA writes
// lane-linearly (M0 base + 16 * lane), so a wave instruction lands 8 rows x 128 B; the bank-conflict-free image is made on the
// SOURCE side: position p of row r holds the 16-byte unit p ^ ((r >> 1) & 7), and the operand fetch applies the same XOR.
// Out-of-range buffer offsets DMA zeros (tools/probes/glds_probe.hip), so padding taps stay branch-free.
// X3 (fp32 storage, register-staged loaders only): the products run on the BF16 matrix cores with fp32 accuracy.  Every fp32 operand is
// split EXACTLY into three bf16 terms, x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 3 x 8 mantissa bits),
// when it is written to LDS (three bf16 planes per tile instead of one fp32 tile), and a product a * b is the six partial products of
// order <= 2^-16 -- hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid -- each EXACT in the fp32 accumulator's input (8 x 8 bits), summed by
// v_mfma_f32_32x32x16_bf16.  The dropped terms (mid*lo, lo*mid, lo*lo) are <= 3 x 2^-24 |a b|: the same order as ONE fp32 rounding of the
// product, and below the fp32 accumulation error of a K >= 64 dot product (measured: 1.4e-7 relative L2 against fp64 where the fp32 MFMA
// gives 3.1e-7).  gfx950's bf16 matrix rate is 16 x its fp32 matrix rate (2.5 PF/s vs 157 TF/s), so six bf16 instructions per k-step cost
// 6 / 16 of the fp32 instruction they replace: 2.67 x the matrix throughput at fp32 accuracy.
template <typename ET, int WM, int WN, int TM, int TN, bool LOWK = false, bool PF2 = false, bool GEN = false, int EPI = 0, bool PW = false, int GLDS = 0,
          bool MVFL = false, bool ILV = false, bool HALFK = false, bool X3 = false>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, char* smem, const int tile, const int c_begin, const int c_end,
                                          const int mode, const SkArgs& sk, const int g_first, const int g_self) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(BM == kBM || (BM == 2 * kBM && (GLDS >= 2 || X3)), "BM is 128 (256 for the 8-wave LDS-DMA tiles and the 8-wave X3 tile)");
    static_assert(WM * WN == 4 || (WM * WN == 8 && (GLDS >= 2 || X3)), "4 waves (8 for the LDS-DMA experiments and the X3 tile)");
    // P4 = the four-phase ping-pong main loop (see the loop itself): 2 x 4 waves of 128 x 64 outputs on a 256 x 256 tile
    constexpr bool P4 = GLDS == 4;
    static_assert(!P4 || (WM == 2 && WN == 4 && TM == 4 && TN == 2 && sizeof(ET) == 2 && !MVFL && !ILV && !HALFK), "the four-phase loop is written for the bf16 256 x 256 tile");
    constexpr int NT = WM * WN * 64;        // threads per workgroup
    constexpr int RP = NT / 8;              // rows per loader pass (8 lanes x 16 B per row)
    constexpr int ESZ = TT<ET>::ESZ, UE = TT<ET>::UE, CE = TT<ET>::CE;
    constexpr int A_ROWS_PT = BM / RP;      // rows per thread in the A loader (256 thr = 32 rows x 8 units)
    constexpr int B_ROWS_PT = BN / RP;
#ifdef MVF_CONV_ABLATE
    if (a.prio & 8) return;                            // ablation: launch + dispatch floor
#endif
    __syncthreads();                                   // LDS hand-over from a previous segment of this workgroup
    static_assert(!GLDS || (!GEN && !PF2 && !LOWK), "the LDS-DMA loop is its own variant");
    static_assert(!MVFL || (((LOWK && PW) || GLDS == 1 || GLDS == 2) && !GEN), "the fused MVF loader: single-buffer register-staged pointwise kernel or the 4-wave LDS-DMA kernels");
    static_assert(!ILV || (GLDS == 2 && !MVFL), "interleaved DMA issue: the two-buffer LDS-DMA loop");
    constexpr int NBUF = P4 ? 2 : (GLDS ? GLDS : (LOWK ? 1 : 2));
    static_assert(!X3 || (sizeof(ET) == 4 && !GLDS && !MVFL && !PF2), "X3: fp32 storage, register-staged kernels (single- or double-buffered)");
    // X3: a chunk row is 32 channels = 64 bytes per bf16 plane, three planes per tile; unpadded rows, 16-byte units XOR-swizzled by the row
    // (unit u of row r at u ^ ((r >> 2) & 3): the 16 rows a ds_read_b128 service group reads fall on 16 distinct 16-byte bank slots)
    constexpr int PITCH = X3 ? 64 : (GLDS ? 128 : kPitch);         // LDS-DMA rows are unpadded (lane-linear destination)
    constexpr int kSmem = X3 ? (LOWK ? kLowkLdsX3<BM, BN>() : 2 * (BM + BN) * 192) : (GLDS ? kGldsLds<BM, BN, GLDS ? NBUF : 1>() : (LOWK ? kLowkLds<BM, BN>() : 2 * (BM + BN) * kPitch));
    constexpr int NPL = X3 ? 3 : 1;                    // operand planes per tile
    char* As = smem;                                   // [NBUF][NPL][BM][PITCH]
    char* Bs = smem + NBUF * NPL * BM * PITCH;         // [NBUF][NPL][BN][PITCH]

    const int tn_i = tile % a.tiles_n, tm_i = tile / a.tiles_n;
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid >> 3;                         // loader: row-in-32, 16-B unit within the 128-B chunk
    const int q = GLDS ? ((tid & 7) ^ ((lrow >> 1) & 7)) : (tid & 7);     // LDS-DMA: the unit that belongs at position tid & 7
    // loader row i of this thread.  P4 stages HALF tiles (128 rows = the rows one phase's operand reads touch, in every wave's sub-tile):
    // i = 2 * half + n, wave instruction n of a half covers 8 consecutive rows -- A half h = rows {wm' * 128 + h * 64 + r, r < 64}
    // (fragment rows 2h, 2h + 1 of both wave rows), B half h = rows {wn' * 64 + h * 32 + r, r < 32} (fragment h of the four wave columns).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto a_row0 = [&](int i) { return P4 ? (wave_u >> 2) * 128 + (i >> 1) * 64 + (wave_u & 3) * 16 + (i & 1) * 8 : wave_u * 8 + RP * i; };
    auto b_row0 = [&](int i) { return P4 ? (wave_u >> 1) * 64 + (i >> 1) * 32 + (wave_u & 1) * 16 + (i & 1) * 8 : wave_u * 8 + RP * i; };
    auto a_row = [&](int i) { return P4 ? a_row0(i) + (lane >> 3) : lrow + RP * i; };
    auto b_row = [&](int i) { return P4 ? b_row0(i) + (lane >> 3) : lrow + RP * i; };
    auto q_of = [&](int row) { return P4 ? ((tid & 7) ^ ((row >> 1) & 7)) : q; };     // (the other variants' rows differ by multiples of 16)

    // ---- per-thread loader state (rows are fixed for the whole K loop) ----
    struct Stage {                       // one K chunk of this thread's loader rows, in registers
        uint4 a[A_ROWS_PT], b[B_ROWS_PT];
    };
    Stage s0, s1;
    int cc = c_begin % a.cpt, kw, kh;    // position of the NEXT chunk to load
    {
        const int tap = c_begin / a.cpt;
        kh = tap / (a.KW > 0 ? a.KW : 1);
        kw = tap - kh * a.KW;
    }
    auto advance = [&]() {
        if (++cc == a.cpt) {
            cc = 0;
            if (++kw == a.KW) {
                kw = 0;
                ++kh;
            }
        }
    };

    // GEN (generic fallback: dilated input, huge images): 64-bit flat addressing with per-row bounds tests.
    int a_pix0[A_ROWS_PT], a_ih0[A_ROWS_PT], a_iw0[A_ROWS_PT];
    const char* b_ptr[B_ROWS_PT];
    bool b_ok[B_ROWS_PT];
    // fast path: buffer loads (hardware zero-fill for an out-of-range offset -> branch-free padding / tails), 32-bit byte
    // offsets relative to the first image of the tile, and per-row bit masks of the valid kh / kw taps.
    constexpr unsigned kOOB = 0x80000000u;
    unsigned a_off[A_ROWS_PT], a_off2[A_ROWS_PT], hmask[A_ROWS_PT], wmask[A_ROWS_PT], b_off[B_ROWS_PT];
    __amdgpu_buffer_rsrc_t rs_x, rs_x2, rs_w;
    i32x4 gs_x = {0, 0, 0, 0}, gs_x2 = gs_x, gs_w = gs_x;      // the same descriptors as plain SGPR words (LDS-DMA asm operands)
    if constexpr (GEN) {
#pragma unroll
        for (int i = 0; i < A_ROWS_PT; ++i) {
            const int m = m0 + lrow + RP * i;
            if (m < a.M) {
                const int img = m / (a.Ho * a.Wo), rem = m - img * (a.Ho * a.Wo);
                const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
                a_ih0[i] = oh * a.stride - a.pad;
                a_iw0[i] = ow * a.stride - a.pad_w;
                a_pix0[i] = a.dil > 1 ? img * a.H * a.W : (img * a.H + a_ih0[i]) * a.W + a_iw0[i];
            } else {
                a_ih0[i] = -100000;   // never in bounds
                a_iw0[i] = -100000;
                a_pix0[i] = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < B_ROWS_PT; ++i) {
            const int co = n0 + lrow + RP * i;
            b_ok[i] = co < a.Cout;
            b_ptr[i] = a.w + ((long)(b_ok[i] ? co : 0) * a.wK + q * UE) * ESZ;
        }
    } else {
        const int hw_o = a.Ho * a.Wo;
        const int img0 = fd_div(m0, a.fd_hw_mul, a.fd_hw_shr);         // wave-uniform (tile index comes from blockIdx)
        const long img_px = (long)a.H * a.W;
        auto span = [&](int ps) {                                      // bytes from the tile's first image to the tensor's end
            const long bytes = (long)(a.N - img0) * img_px * ps * ESZ;
            return (unsigned)(bytes < 0x7ffffff0L ? bytes : 0x7ffffff0L);
        };
        rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long)img0 * img_px * a.xps * ESZ), 0, span(a.xps), 0x00020000);
        rs_x2 = rs_x;
        if (a.split_c > 0) rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 + (long)img0 * img_px * a.x2ps * ESZ), 0, span(a.x2ps), 0x00020000);
        rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)((long)a.Cout * a.wK * ESZ), 0x00020000);
        if constexpr (GLDS > 0) {
            gs_x = rsrc_words(a.x + (long)img0 * img_px * a.xps * ESZ, span(a.xps));
            gs_x2 = gs_x;
            if (a.split_c > 0) gs_x2 = rsrc_words(a.x2 + (long)img0 * img_px * a.x2ps * ESZ, span(a.x2ps));
            gs_w = rsrc_words(a.w, (unsigned)((long)a.Cout * a.wK * ESZ));
        }
#pragma unroll
        for (int i = 0; i < A_ROWS_PT; ++i) {
            const int m = m0 + a_row(i);
            const int q = q_of(a_row(i));
            if constexpr (PW) {
                if (a.stride == 1) {             // pointwise stride 1: input pixel == output pixel, no division at all
                    a_off[i] = m < a.M ? (unsigned)((m - img0 * hw_o) * a.xps + q * UE) * ESZ : kOOB;
                    a_off2[i] = 0u; hmask[i] = 0u; wmask[i] = 0u;
                    continue;
                }
            }
            const int img = fd_div(m, a.fd_hw_mul, a.fd_hw_shr), rem = m - img * hw_o;
            const int oh = fd_div(rem, a.fd_w_mul, a.fd_w_shr), ow = rem - oh * a.Wo;
            const int ih0 = oh * a.stride - a.pad, iw0 = ow * a.stride - a.pad_w;
            const int pix = ((img - img0) * a.H + ih0) * a.W + iw0;    // may be negative for a padded corner; valid taps land >= 0
            if constexpr (PW) {
                a_off[i] = m < a.M ? (unsigned)(pix * a.xps + q * UE) * ESZ : kOOB;     // valid for every chunk, or never
                a_off2[i] = 0u; hmask[i] = 0u; wmask[i] = 0u;
                continue;
            }
            a_off[i] = (unsigned)(pix * a.xps + q * UE) * ESZ;
            a_off2[i] = (unsigned)(pix * a.x2ps + q * UE) * ESZ;
            // bit k of hmask: 0 <= ih0 + k < H  (k in [lo, hi)); all zero for rows past M
            auto range_mask = [](int x0, int lim) {
                const int lo = x0 < 0 ? -x0 : 0, hi = lim - x0;          // valid k in [lo, hi)
                const unsigned up = hi >= 32 ? 0xffffffffu : (hi <= 0 ? 0u : ((1u << hi) - 1u));
                const unsigned dn = lo >= 32 ? 0xffffffffu : ((1u << lo) - 1u);
                return up & ~dn;
            };
            hmask[i] = m < a.M ? range_mask(ih0, a.H) : 0u;
            wmask[i] = range_mask(iw0, a.W);
        }
#pragma unroll
        for (int i = 0; i < B_ROWS_PT; ++i) {
            const int co = n0 + b_row(i);
            b_off[i] = co < a.Cout ? (unsigned)((long)co * a.wK + q_of(b_row(i)) * UE) * ESZ : kOOB;
        }
    }

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned lds_a0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)As + wave * 8 * PITCH);
    const unsigned lds_b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Bs + wave * 8 * PITCH);
    // MVFL: the A rows of a chunk below mvf_cs = hswish(bn(stencil)) of the block input, built straight into the LDS tile.
    // A thread owns ONE channel quad (its 9 taps + scale + shift stay in 11 x 4 registers) and walks BM * CE / (4 * NT) rows;
    // per row the seven neighbours (t-1, t+1: +- one frame; h-1, h+1: +- W pixels; w-1, w+1: +- 1 pixel; centre) are loaded
    // unconditionally -- a neighbour outside the clip / image gets an out-of-range buffer offset and reads zeros, which IS the
    // zero padding of the reference's Conv3d views (MVF.py:65-81).  The arithmetic order is mvf_nhwc_apply's.
    auto mvf_rows = [&](int buf, int scc) {
        constexpr int QPR = CE / 4, RST = NT / QPR, RPT = BM / RST;     // quads per row, row step, rows per thread
        const int cq4 = tid % QPR, r00 = tid / QPR;
        const int c0 = scc * CE + cq4 * 4;
        float wt_[4][3], wh_[4][3], ww_[4][3], sc_[4], sh_[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 c_a = *reinterpret_cast<const float4*>(a.mvf_coef + (long)(c0 + i) * 12);
            const float4 c_b = *reinterpret_cast<const float4*>(a.mvf_coef + (long)(c0 + i) * 12 + 4);
            const float4 c_c = *reinterpret_cast<const float4*>(a.mvf_coef + (long)(c0 + i) * 12 + 8);
            wt_[i][0] = c_a.x; wt_[i][1] = c_a.y; wt_[i][2] = c_a.z; wh_[i][0] = c_a.w;
            wh_[i][1] = c_b.x; wh_[i][2] = c_b.y; ww_[i][0] = c_b.z; ww_[i][1] = c_b.w;
            ww_[i][2] = c_c.x; sc_[i] = c_c.y; sh_[i] = c_c.z;
        }
        const int hw = a.H * a.W;
        const int img0 = fd_div(m0, a.fd_hw_mul, a.fd_hw_shr);
        const int imgb = img0 > 0 ? img0 - 1 : 0;                      // the tile's first rows may need the frame before img0
        const long img_bytes = (long)hw * a.xps * ESZ;
        const long left = (long)(a.N - imgb) * img_bytes;
        const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long)imgb * img_bytes), 0,
                                                                                (unsigned)(left < 0x7ffffff0L ? left : 0x7ffffff0L), 0x00020000);
        const unsigned d_t = (unsigned)img_bytes, d_h = (unsigned)(a.W * a.xps * ESZ), d_w = (unsigned)(a.xps * ESZ);
        typedef typename std::conditional<sizeof(ET) == 2, unsigned __attribute__((ext_vector_type(2))), u32x4>::type q_t;
        auto ldq = [&](unsigned off, float (&f)[4]) {
            if constexpr (sizeof(ET) == 2) {
                const q_t v = __builtin_amdgcn_raw_buffer_load_b64(rsm, off, 0, 0);
                f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
                f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
            } else {
                const q_t v = __builtin_amdgcn_raw_buffer_load_b128(rsm, off, 0, 0);
                f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
            }
        };
        char* dst = As + buf * BM * PITCH + (GLDS ? 0 : cq4 * (4 * ESZ));
        // LDS-DMA image: 16-byte unit u of row r sits at position u ^ ((r >> 1) & 7) (the source-side swizzle of load_chunk)
        auto dst_off = [&](int r) {
            if constexpr (GLDS > 0) {
                const int u = (cq4 * 4 * ESZ) >> 4, sub = (cq4 * 4 * ESZ) & 15;
                return r * PITCH + ((u ^ ((r >> 1) & 7)) << 4) + sub;
            } else {
                return r * PITCH;
            }
        };
#pragma unroll 2
        for (int k = 0; k < RPT; ++k) {
            const int r = r00 + RST * k, m = m0 + r;
            const int img = fd_div(m, a.fd_hw_mul, a.fd_hw_shr), rem = m - img * hw;
            const int oh = fd_div(rem, a.fd_w_mul, a.fd_w_shr), ow = rem - oh * a.W;
            const int t = img - fd_div(img, a.fd_t_mul, a.fd_t_shr) * a.mvf_T;
            const bool ok = m < a.M;
            const unsigned oc = (unsigned)(((img - imgb) * hw + rem) * a.xps + c0) * ESZ;
            float prev[4], cur[4], next[4], up[4], dn[4], lf[4], rt[4];
            ldq(ok ? oc : kOOB, cur);
            ldq(ok && t > 0 ? oc - d_t : kOOB, prev);
            ldq(ok && t + 1 < a.mvf_T ? oc + d_t : kOOB, next);
            ldq(ok && oh > 0 ? oc - d_h : kOOB, up);
            ldq(ok && oh + 1 < a.H ? oc + d_h : kOOB, dn);
            ldq(ok && ow > 0 ? oc - d_w : kOOB, lf);
            ldq(ok && ow + 1 < a.W ? oc + d_w : kOOB, rt);
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float yt = wt_[i][0] * prev[i] + wt_[i][1] * cur[i] + wt_[i][2] * next[i];
                const float yh = wh_[i][0] * up[i] + wh_[i][1] * cur[i] + wh_[i][2] * dn[i];
                const float yw = ww_[i][0] * lf[i] + ww_[i][1] * cur[i] + ww_[i][2] * rt[i];
                float v = (yt + yh) + yw;
                if (a.mvf_act) {
                    const float u = sc_[i] * v + sh_[i];
                    v = u * (fminf(fmaxf(u + 3.0f, 0.0f), 6.0f) / 6.0f);
                }
                y[i] = ok ? v : 0.f;
            }
            if constexpr (sizeof(ET) == 2) {
                uint2 pk;
                pk.x = pack_bf16x2(y[0], y[1]);
                pk.y = pack_bf16x2(y[2], y[3]);
                *reinterpret_cast<uint2*>(dst + dst_off(r)) = pk;
            } else {
                *reinterpret_cast<float4*>(dst + dst_off(r)) = make_float4(y[0], y[1], y[2], y[3]);
            }
        }
    };
    int staged_cc = 0;                   // K chunk (within the tap) of the Stage registers: MVFL builds its A rows at store time
    auto load_chunk = [&](Stage& st, int buf = 0) {
        (void)buf;
        staged_cc = cc;
        const bool mvf_chunk = MVFL && cc * CE < a.mvf_cs;
        const int ci = cc * CE + q * UE;                       // first channel of this thread's unit
        const bool cok = ci < a.Cin;
        const bool from2 = (a.split_c > 0) && (cc * CE < a.split_c);
        if constexpr (GEN) {
            const char* xb = from2 ? a.x2 : a.x;
            const int ps = from2 ? a.x2ps : a.xps;
#pragma unroll
            for (int i = 0; i < A_ROWS_PT; ++i) {
                int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                st.a[i] = make_uint4(0, 0, 0, 0);
                if (a.dil > 1) {
                    // zero-upsampled view: only positions that are multiples of dil carry data
                    bool ok = cok && ih >= 0 && iw >= 0 && (ih % a.dil == 0) && (iw % a.dil == 0);
                    ih /= a.dil;
                    iw /= a.dil;
                    ok = ok && ih < a.H && iw < a.W;
                    if (ok) {
                        const long pix = (long)a_pix0[i] + ih * a.W + iw;
                        st.a[i] = *reinterpret_cast<const uint4*>(xb + (pix * ps + ci - (from2 ? 0 : a.x_c0)) * ESZ);
                    }
                } else {
                    const bool ok = cok && ((unsigned)ih < (unsigned)a.H) && ((unsigned)iw < (unsigned)a.W);
                    if (ok) {
                        const long pix = (long)a_pix0[i] + kh * a.W + kw;
                        st.a[i] = *reinterpret_cast<const uint4*>(xb + (pix * ps + ci - (from2 ? 0 : a.x_c0)) * ESZ);
                    }
                }
            }
            const long koff = ((long)((a.w_kh0 + kh * a.w_ts) * a.w_kwfull + (a.w_kw0 + kw * a.w_ts)) * a.Cin + cc * CE) * ESZ;
#pragma unroll
            for (int i = 0; i < B_ROWS_PT; ++i) {
                st.b[i] = make_uint4(0, 0, 0, 0);
                if (b_ok[i] && cok) st.b[i] = *reinterpret_cast<const uint4*>(b_ptr[i] + koff);
            }
        } else {
            const unsigned cbad = cok ? 0u : kOOB;                          // channel tail of the last chunk of a tap
            const int ps = from2 ? a.x2ps : a.xps;
            const unsigned toff = (unsigned)((kh * a.W + kw) * ps + cc * CE - (from2 ? 0 : a.x_c0)) * ESZ;     // wave-uniform
            const __amdgpu_buffer_rsrc_t rs = from2 ? rs_x2 : rs_x;
#pragma unroll
            for (int i = 0; i < A_ROWS_PT; ++i) {
                if constexpr (MVFL) {
                    if (mvf_chunk) break;                      // wave-uniform: the A rows of this chunk are computed in store_chunk
                }
                unsigned voff;
                if constexpr (PW) {
                    voff = (a_off[i] + (unsigned)(cc * CE) * ESZ) | cbad;              // kOOB + small stays out of range
                } else {
                    const bool ok = ((hmask[i] >> kh) & (wmask[i] >> kw) & 1u) != 0u;
                    voff = ok ? (((from2 ? a_off2[i] : a_off[i]) + toff) | cbad) : kOOB;
                }
                if constexpr (GLDS > 0) {
                    glds16((PW || !from2) ? gs_x : gs_x2, lds_a0 + (unsigned)((buf * BM + RP * i) * PITCH), voff);
                } else {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(PW ? rs_x : rs, voff, 0, 0);
                    st.a[i] = make_uint4(v.x, v.y, v.z, v.w);
                }
            }
            const unsigned koff = (unsigned)(((a.w_kh0 + kh * a.w_ts) * a.w_kwfull + (a.w_kw0 + kw * a.w_ts)) * a.Cin + cc * CE) * ESZ;
#pragma unroll
            for (int i = 0; i < B_ROWS_PT; ++i) {
                if constexpr (GLDS > 0) {
                    glds16(gs_w, lds_b0 + (unsigned)((buf * BN + RP * i) * PITCH), (b_off[i] + koff) | cbad);
                } else {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (b_off[i] + koff) | cbad, 0, 0);
                    st.b[i] = make_uint4(v.x, v.y, v.z, v.w);
                }
            }
        }
        if constexpr (MVFL && GLDS > 0) {
            if (mvf_chunk) mvf_rows(buf, cc);              // the chunk's A rows: computed and written into the DMA image by the waves themselves
        }
        advance();
    };
    // ILV: the next chunk's DMA pieces are issued BETWEEN the MFMAs of the current chunk's first k-step instead of in a block
    // before them: a wave's VMEM issue (60-185 cycles per LDS-DMA piece in a loaded phase) then hides under the 32-cycle matrix
    // instructions already in the pipe -- with one workgroup per CU (the 256 x 256 tile) the two waves of a SIMD run in lockstep,
    // so a DMA block ahead of the MFMAs leaves the matrix pipe idle for its whole length.  prep_chunk() does the offset
    // arithmetic of load_chunk() for the chunk at (cc, kh, kw) and advances; issue_piece(n) issues piece n of it.
    unsigned pv[A_ROWS_PT + B_ROWS_PT];
    unsigned p_lds_a = 0, p_lds_b = 0;
    bool p_from2 = false;
    auto prep_chunk = [&](int buf) {
        const int ci = cc * CE + q * UE;
        const unsigned cbad = ci < a.Cin ? 0u : kOOB;
        p_from2 = (a.split_c > 0) && (cc * CE < a.split_c);
        const int ps = p_from2 ? a.x2ps : a.xps;
        const unsigned toff = (unsigned)((kh * a.W + kw) * ps + cc * CE - (p_from2 ? 0 : a.x_c0)) * ESZ;
#pragma unroll
        for (int i = 0; i < A_ROWS_PT; ++i) {
            const bool ok = ((hmask[i] >> kh) & (wmask[i] >> kw) & 1u) != 0u;
            pv[i] = ok ? (((p_from2 ? a_off2[i] : a_off[i]) + toff) | cbad) : kOOB;
        }
        const unsigned koff = (unsigned)(((a.w_kh0 + kh * a.w_ts) * a.w_kwfull + (a.w_kw0 + kw * a.w_ts)) * a.Cin + cc * CE) * ESZ;
#pragma unroll
        for (int i = 0; i < B_ROWS_PT; ++i) pv[A_ROWS_PT + i] = (b_off[i] + koff) | cbad;
        p_lds_a = lds_a0 + (unsigned)(buf * BM * PITCH);
        p_lds_b = lds_b0 + (unsigned)(buf * BN * PITCH);
        advance();
    };
    auto issue_piece = [&](int n) {
        if constexpr (GLDS > 0) {
            if (n < A_ROWS_PT) glds16(p_from2 ? gs_x2 : gs_x, p_lds_a + (unsigned)(RP * n * PITCH), pv[n]);
            else glds16(gs_w, p_lds_b + (unsigned)(RP * (n - A_ROWS_PT) * PITCH), pv[n]);
        }
    };
    // X3: hi / mid / lo bf16 terms of four fp32 values (exact: x - bf16(x) is representable, twice), packed 4 x bf16 = 8 bytes per plane
    auto split3 = [&](const uint4& v, uint2 (&pl)[3]) {
        float r[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned lo2 = pack_bf16x2(r[0], r[1]), hi2 = pack_bf16x2(r[2], r[3]);
            pl[p] = make_uint2(lo2, hi2);
#ifdef MVF_X3_ABLATE_SPLIT
            if (false) {                     // timing ablation (wrong results): the three planes all hold bf16(x), no subtraction
#else
            if (p < 2) {
#endif
                // scalar v_sub_f32 through asm: left alone hipcc SLP-packs the four into v_pk_add_f32, and packed fp32 VALU beside matrix instructions
                // is an anti-lever on this chip (MI355X_MICROARCH.md; tools/probes/x3_gemm_probe.hip: +5 %)
                auto sub1 = [](float x, float y) { float d; asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
                r[0] = sub1(r[0], __uint_as_float(lo2 << 16)); r[1] = sub1(r[1], __uint_as_float(lo2 & 0xffff0000u));
                r[2] = sub1(r[2], __uint_as_float(hi2 << 16)); r[3] = sub1(r[3], __uint_as_float(hi2 & 0xffff0000u));
            }
        }
    };
    auto store_chunk = [&](int buf, const Stage& st) {
        if constexpr (X3) {
            // thread (lrow, q): channels 4q .. 4q+3 of rows lrow + RP * i -> 8 bytes at unit (q >> 1) ^ ((row >> 2) & 3), half q & 1 of each plane
#pragma unroll
            for (int i = 0; i < A_ROWS_PT; ++i) {
                const int row = lrow + RP * i;
                uint2 pl[3];
                split3(st.a[i], pl);
                char* d = As + (buf * 3 * BM + row) * PITCH + ((((q >> 1) ^ ((row >> 2) & 3)) << 4) | ((q & 1) << 3));
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(d + p * BM * PITCH) = pl[p];
            }
#pragma unroll
            for (int i = 0; i < B_ROWS_PT; ++i) {
                const int row = lrow + RP * i;
                uint2 pl[3];
                split3(st.b[i], pl);
                char* d = Bs + (buf * 3 * BN + row) * PITCH + ((((q >> 1) ^ ((row >> 2) & 3)) << 4) | ((q & 1) << 3));
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(d + p * BN * PITCH) = pl[p];
            }
            return;
        }
        char* ad = As + buf * BM * PITCH + lrow * PITCH + q * 16;
        bool a_done = false;
        if constexpr (MVFL) {
            if (staged_cc * CE < a.mvf_cs) {
                mvf_rows(buf, staged_cc);
                a_done = true;
            }
        }
        if (!a_done) {
#pragma unroll
            for (int i = 0; i < A_ROWS_PT; ++i) *reinterpret_cast<uint4*>(ad + RP * i * PITCH) = st.a[i];
        }
        char* bd = Bs + buf * BN * PITCH + lrow * PITCH + q * 16;
#pragma unroll
        for (int i = 0; i < B_ROWS_PT; ++i) *reinterpret_cast<uint4*>(bd + RP * i * PITCH) = st.b[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nseg = c_end - c_begin;
    const int frag_off = (lane & 31) * PITCH + (GLDS ? 0 : (lane >> 5) * 16);
    const int fsw = (lane >> 1) & 7;                   // LDS-DMA image: XOR of the fragment row (row & 31 == lane & 31)
    // FPIPE: the operand fragments of k-step ks+2 are fetched from LDS behind the MFMAs of k-step ks (two fragment register
    // sets), so the ds_read round trip is exposed once per chunk instead of once per k-step
    constexpr bool FPIPE = GLDS != 0;
    auto compute = [&](int buf, bool ilv_more = false) {
        (void)ilv_more;
        if constexpr (X3) {
            // two k-steps of 16 channels; lane (row = lane & 31, half = lane >> 5) reads unit 2 * ks + half of its row in each plane
            const int fr = lane & 31, fx = (fr >> 2) & 3;          // (fragment rows are multiples of 32 apart: the XOR only depends on lane & 31)
            const char* Ab = As + (buf * 3 * BM + wm * TM * 32 + fr) * PITCH;
            const char* Bb = Bs + (buf * 3 * BN + wn * TN * 32 + fr) * PITCH;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ko = ((2 * ks + (lane >> 5)) ^ fx) << 4;
                bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const uint4 v = *reinterpret_cast<const uint4*>(Ab + (p * BM + i * 32) * PITCH + ko);
                        __builtin_memcpy(&fa[p][i], &v, 16);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const uint4 v = *reinterpret_cast<const uint4*>(Bb + (p * BN + j * 32) * PITCH + ko);
                        __builtin_memcpy(&fb[p][j], &v, 16);
                    }
                }
                // smallest terms first (planes 0 / 1 / 2 = hi / mid / lo); operands swapped as everywhere (D^T: see the epilogue).  The
                // product term is the OUTER loop: consecutive matrix instructions go to different accumulators (a dependent one would wait
                // for its predecessor's 8 passes + write-back -- the tile-by-tile order ran the loop at a third of the matrix rate)
                constexpr int PA_[6] = {0, 2, 1, 0, 1, 0}, PB_[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PB_[t]][j], fa[PA_[t]][i], acc[i][j], 0, 0, 0);
            }
            return;
        }
        const char* Ab = As + buf * BM * PITCH + (wm * TM * 32) * PITCH + frag_off;
        const char* Bb = Bs + buf * BN * PITCH + (wn * TN * 32) * PITCH + frag_off;
        auto fetch = [&](int ks, uint4 (&fa)[TM], uint4 (&fb)[TN]) {
            const int ko = GLDS ? (((2 * ks + (lane >> 5)) ^ fsw) * 16) : ks * 32;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(Ab + i * 32 * PITCH + ko);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bb + j * 32 * PITCH + ko);
        };
        auto mma = [&](const uint4 (&fa)[TM], const uint4 (&fb)[TN], int piece0 = -1) {
            (void)piece0;
            if constexpr (ESZ == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float av = __uint_as_float(e == 0 ? fa[i].x : e == 1 ? fa[i].y : e == 2 ? fa[i].z : fa[i].w);
                            const float bv = __uint_as_float(e == 0 ? fb[j].x : e == 1 ? fb[j].y : e == 2 ? fb[j].z : fb[j].w);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[i][j], 0, 0, 0);   // D^T: see the epilogue
                        }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bf16x8 av, bv;
                        __builtin_memcpy(&av, &fa[i], 16);
                        __builtin_memcpy(&bv, &fb[j], 16);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc[i][j], 0, 0, 0);
                        if constexpr (ILV) {                   // one DMA piece of the next chunk behind each matrix instruction
                            if (piece0 >= 0 && piece0 + i * TN + j < A_ROWS_PT + B_ROWS_PT) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_piece(piece0 + i * TN + j);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
            }
        };
        if constexpr (FPIPE) {
            if (a.prio & 1) __builtin_amdgcn_s_setprio(1);
            uint4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
            // (the order is pinned: left alone, the machine scheduler folds the two sets back into one and waits per k-step)
            fetch(0, fa0, fb0);
            fetch(1, fa1, fb1);
            if constexpr (HALFK) {                 // the chunk's upper 64 bytes are zero padding: two k-steps instead of four
                mma(fa0, fb0);
                mma(fa1, fb1);
                if (a.prio & 1) __builtin_amdgcn_s_setprio(0);
                return;
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(fa0, fb0, ilv_more ? 0 : -1);
            __builtin_amdgcn_sched_barrier(0);
            fetch(2, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mma(fa1, fb1, ilv_more ? TM * TN : -1);
            __builtin_amdgcn_sched_barrier(0);
            fetch(3, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mma(fa0, fb0);
            mma(fa1, fb1);
            if (a.prio & 1) __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                uint4 fa[TM], fb[TN];
                fetch(ks, fa, fb);
                mma(fa, fb);
            }
        }
    };
    if constexpr (GLDS == 1) {
        // one buffer: DMA, wait, compute; the overlap comes from the other workgroups of the CU (32 KB of LDS each)
        for (int kc = 0; kc < nseg; ++kc) {
#ifdef MVF_CONV_ABLATE
            if (!(a.prio & 4)) load_chunk(s0, 0);          // ablation: bit 2 = no loads, bit 1 = no MFMAs
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (!(a.prio & 2)) compute(0);
#else
            load_chunk(s0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
#endif
            __syncthreads();
        }
    } else if constexpr (GLDS == 3) {
        // three buffers, two chunks in flight: chunk kc+2 is issued as soon as every wave is past chunk kc-1 (whose buffer it
        // takes), and the wait at the top leaves the DMAs of chunk kc+1 outstanding (a counted vmcnt: the statements are asm,
        // so the count is this loop's own bookkeeping -- A_ROWS_PT + B_ROWS_PT DMAs per chunk per thread)
        constexpr int kPerChunk = A_ROWS_PT + B_ROWS_PT;
        static_assert(kPerChunk == 6, "the counted wait below is written for 6 DMAs per chunk");
        if (nseg > 0) load_chunk(s0, 0);
        if (nseg > 1) load_chunk(s0, 1);
        int bcur = 0, bnext = 2;
        for (int kc = 0; kc < nseg; ++kc) {
            if (kc + 1 < nseg) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                           // (hipcc does not know of the DMAs: this is a bare barrier + lgkmcnt)
            if (kc + 2 < nseg) load_chunk(s0, bnext);
            compute(bcur);
            bcur = bcur == 2 ? 0 : bcur + 1;
            bnext = bnext == 2 ? 0 : bnext + 1;
        }
        __syncthreads();
    } else if constexpr (GLDS == 4) {
        // ---- four-phase ping-pong loop (tools/probes/gemm8p_probe.hip VAR 3: the matrix pipe issues 99 % of the loop's cycles) ----------------
        // The 8 waves are two groups (wave rows wm = 0 / 1: one wave of each on every SIMD) running ONE BARRIER APART: while a group issues the
        // 16 MFMAs of a phase, the other issues its 12 operand reads and 4 LDS-DMA pieces for its next phase, then they swap -- every s_barrier
        // is a role switch.  A chunk is two phases per group:
        //   X(k): read A fragments {0,1} + B fragment 0 of chunk k      -> C[0..1][0..1] += ...   stage B0(k+1), then A1(k+1)
        //   Y(k): read A fragments {2,3} of chunk k + B fragment 1 of k+1 -> C[2..3][0..1] += ... stage B1(k+2), then A0(k+2)
        // (B fragment 1 of chunk k was read one phase early, in Y(k-1): 12 reads in every phase).  Half tiles are staged in the order they die;
        // every phase waits vmcnt(6), i.e. three half tiles stay in flight across the barrier: the activations get two phases of lead, the
        // weights (L2 resident) one.  Hazards (cdna_hip_programming.md section 5, "Read a staged buffer one phase AFTER the wait ..."):
        //  RAW  a half tile is read one phase after the counted wait that retired it, with that phase's first barrier in between -- by both groups;
        //  WAR  the A slots are restaged ONE phase after their last read: those 8 reads are issued first and retired (lgkmcnt(4)) before the
        //       reading phase's first barrier; the B slots are restaged two phases after their last read.
        // The MFMA builtin touches no memory, so neither s_barrier nor a "memory" clobber orders it and sched_barrier(0) does not bind the
        // sinking passes: the accumulators are made opaque (empty asm, "+v") on both sides of every MFMA block.
        // Chunks past the segment's end are staged with out-of-range offsets (the DMA writes zeros into dead slots), so the counted waits
        // need no tail variants.
        struct Cur { int cc, kw, kh, left; };              // position of a chunk in the K walk; left = chunks of the segment from it on
        auto cur_next = [&](Cur c) {
            if (++c.cc == a.cpt) {
                c.cc = 0;
                if (++c.kw == a.KW) { c.kw = 0; ++c.kh; }
            }
            --c.left;
            return c;
        };
        const unsigned lds_a = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)As);
        const unsigned lds_b = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Bs);
        auto stage_a = [&](int h, int buf, const Cur& c) {
            const bool live = c.left > 0;
            const bool from2 = (a.split_c > 0) && (c.cc * CE < a.split_c);
            const int ps = from2 ? a.x2ps : a.xps;
            const unsigned toff = (unsigned)((c.kh * a.W + c.kw) * ps + c.cc * CE - (from2 ? 0 : a.x_c0)) * ESZ;       // wave-uniform
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int i = 2 * h + n;
                unsigned voff;
                if constexpr (PW) {
                    voff = live ? a_off[i] + (unsigned)(c.cc * CE) * ESZ : kOOB;          // kOOB + small stays out of range
                } else {
                    const bool ok = live && ((hmask[i] >> c.kh) & (wmask[i] >> c.kw) & 1u) != 0u;
                    voff = ok ? (from2 ? a_off2[i] : a_off[i]) + toff : kOOB;
                }
                glds16((PW || !from2) ? gs_x : gs_x2, lds_a + (unsigned)((buf * BM + a_row0(i)) * PITCH), voff);
            }
        };
        auto stage_b = [&](int h, int buf, const Cur& c) {
            const unsigned koff = (unsigned)(((a.w_kh0 + c.kh * a.w_ts) * a.w_kwfull + (a.w_kw0 + c.kw * a.w_ts)) * a.Cin + c.cc * CE) * ESZ;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int i = 2 * h + n;
                glds16(gs_w, lds_b + (unsigned)((buf * BN + b_row0(i)) * PITCH), c.left > 0 ? b_off[i] + koff : kOOB);
            }
        };
        const char* a_base = As + (wm * 128 + (lane & 31)) * PITCH;
        const char* b_base = Bs + (wn * 64 + (lane & 31)) * PITCH;
        int ko[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ko[ks] = ((2 * ks + (lane >> 5)) ^ fsw) << 4;
        auto fetch_a = [&](int buf, int i, uint4 (&f)[4]) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const uint4*>(a_base + (buf * BM + i * 32) * PITCH + ko[ks]);
        };
        auto fetch_b = [&](int buf, int j, uint4 (&f)[4]) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const uint4*>(b_base + (buf * BN + j * 32) * PITCH + ko[ks]);
        };
        auto mma1 = [&](f32x16& d, const uint4& fa, const uint4& fb) {
            bf16x8 av, bv;
            __builtin_memcpy(&av, &fa, 16);
            __builtin_memcpy(&bv, &fb, 16);
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, d, 0, 0, 0);          // D^T: see the epilogue
        };
#define MVF_PIN2(x, y) asm volatile("" : "+v"(x), "+v"(y))
        // prologue: chunk 0 whole + B1 / A0 of chunk 1 in the steady-state issue order; chunk 0's A0, B0, B1 have landed after the wait
        Cur c1 = {cc, kw, kh, nseg};                       // the chunk X stages for (k + 1) ...
        stage_b(1, 0, c1); stage_a(0, 0, c1); stage_b(0, 0, c1); stage_a(1, 0, c1);
        c1 = cur_next(c1);
        stage_b(1, 1, c1); stage_a(0, 1, c1);
        Cur c2 = cur_next(c1);                             // ... and the one Y stages for (k + 2)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        uint4 fa[2][4], fb0[4], fb1a[4], fb1b[4];
        fetch_b(0, 1, fb1a);
        if (wm == 1) __builtin_amdgcn_s_barrier();         // the second group runs one barrier behind the first
        auto phase_x = [&](int buf, uint4 (&cur)[4]) {
            fetch_a(buf, 0, fa[0]);
            fetch_a(buf, 1, fa[1]);
            __builtin_amdgcn_sched_barrier(0);
            fetch_b(buf, 0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            stage_b(0, buf ^ 1, c1);
            stage_a(1, buf ^ 1, c1);
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MVF_PIN2(acc[0][0], acc[1][0]); MVF_PIN2(acc[0][1], acc[1][1]);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mma1(acc[0][0], fa[0][ks], fb0[ks]); mma1(acc[1][0], fa[1][ks], fb0[ks]);
                mma1(acc[0][1], fa[0][ks], cur[ks]); mma1(acc[1][1], fa[1][ks], cur[ks]);
            }
            __builtin_amdgcn_s_setprio(0);
            MVF_PIN2(acc[0][0], acc[1][0]); MVF_PIN2(acc[0][1], acc[1][1]);
            __builtin_amdgcn_s_barrier();
        };
        auto phase_y = [&](int buf, uint4 (&cur)[4], uint4 (&nxt)[4]) {
            fetch_a(buf, 2, fa[0]);
            fetch_a(buf, 3, fa[1]);
            __builtin_amdgcn_sched_barrier(0);
            fetch_b(buf ^ 1, 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            stage_b(1, buf, c2);
            stage_a(0, buf, c2);
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MVF_PIN2(acc[2][0], acc[3][0]); MVF_PIN2(acc[2][1], acc[3][1]);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mma1(acc[2][0], fa[0][ks], fb0[ks]); mma1(acc[3][0], fa[1][ks], fb0[ks]);
                mma1(acc[2][1], fa[0][ks], cur[ks]); mma1(acc[3][1], fa[1][ks], cur[ks]);
            }
            __builtin_amdgcn_s_setprio(0);
            MVF_PIN2(acc[2][0], acc[3][0]); MVF_PIN2(acc[2][1], acc[3][1]);
            __builtin_amdgcn_s_barrier();
            c1 = c2;
            c2 = cur_next(c2);
        };
        for (int kc = 0; kc < nseg; kc += 2) {
            phase_x(0, fb1a);
            phase_y(0, fb1a, fb1b);
            if (kc + 1 < nseg) {
                phase_x(1, fb1b);
                phase_y(1, fb1b, fb1a);
            }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();         // the groups meet again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill DMAs of the tail must not land in the epilogue's C tile
        __syncthreads();
#undef MVF_PIN2
    } else if constexpr (GLDS >= 2) {
        // GLDS buffers: chunk kc+GLDS-1 is in flight while chunk kc is multiplied; one barrier per chunk
        if (nseg > 0) load_chunk(s0, 0);
        for (int kc = 0; kc < nseg; ++kc) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                           // chunk kc has landed for every wave; the buffer of chunk kc-1 is free
#ifdef MVF_CONV_ABLATE
            if (kc + 1 < nseg && !(a.prio & 4)) load_chunk(s0, (kc + 1) & 1);      // ablation: bit 2 = no loads, bit 1 = no MFMAs
            if (!(a.prio & 2)) compute(kc & 1);
#else
            if constexpr (ILV) {
                const bool more = kc + 1 < nseg;
                if (more) prep_chunk((kc + 1) & 1);
                compute(kc & 1, more);
            } else {
                if (kc + 1 < nseg) load_chunk(s0, (kc + 1) & 1);
                compute(kc & 1);
            }
#endif
        }
        __syncthreads();
    } else if constexpr (LOWK) {
        // single LDS buffer, one chunk ahead in registers
        if (nseg > 0) {
            load_chunk(s0);
            store_chunk(0, s0);
        }
        __syncthreads();
        for (int kc = 0; kc < nseg; ++kc) {
            const bool more = kc + 1 < nseg;
            if (more) load_chunk(s0);
            compute(0);
            if (more) {
                __syncthreads();                       // every wave is done reading the single buffer
                store_chunk(0, s0);
            }
            __syncthreads();
        }
    } else if constexpr (PF2) {
        // two LDS buffers and TWO chunks ahead in registers (s0: even chunks, s1: odd chunks): the short-K pointwise layers are
        // bound by the global-load round trip per chunk, not by MFMA, so keep two round trips in flight per workgroup
        // The prefetch loads are issued UNCONDITIONALLY, also past the last chunk: a conditional load would force the compiler
        // to assume "no younger load in flight" at every s_waitcnt (a static count), i.e. vmcnt(0), which drains the prefetch.
        // Past-the-end loads are harmless: buffer loads are range-checked (zero fill) and their data is never stored.
        load_chunk(s0);                                // chunk 0
        load_chunk(s1);                                // chunk 1
        if (nseg > 0) store_chunk(0, s0);
        __syncthreads();
        load_chunk(s0);                                // chunk 2
        int kc = 0;
        for (; kc + 1 < nseg; kc += 2) {
            compute(0);                                // chunk kc
            store_chunk(1, s1);                        // chunk kc+1
            load_chunk(s1);                            // chunk kc+3
            __syncthreads();
            compute(1);                                // chunk kc+1
            if (kc + 2 < nseg) store_chunk(0, s0);     // chunk kc+2
            load_chunk(s0);                            // chunk kc+4
            __syncthreads();
        }
        if (kc < nseg) {
            compute(0);
            __syncthreads();
        }
    } else {
        if (nseg > 0) {
            load_chunk(s0);
            store_chunk(0, s0);
        }
        __syncthreads();
        for (int kc = 0; kc < nseg; ++kc) {
            const int buf = kc & 1;
            const bool more = kc + 1 < nseg;
            if (more) load_chunk(s0);
            compute(buf);
            if (more) store_chunk(buf ^ 1, s0);
            __syncthreads();
        }
    }

#ifdef MVF_CONV_ABLATE
    if (a.prio & 16) return;                           // ablation: no epilogue
#endif
    // ---- stream-K hand-over of partial accumulators (inter-workgroup, placement independent: agent-scope release on
    // the producer, ONE relaxed poll + agent-scope acquire on the consumer; cdna_hip_programming.md Guideline 16) -----
    if (mode == SEG_PRODUCE) {
        float* slot = sk.ws + (long)g_self * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) slot[((i * TN + j) * 16 + r) * NT + tid] = acc[i][j][r];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(sk.flags + g_self, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (mode == SEG_FINISH) {
        for (int g = g_first; g < g_self; ++g) {
            if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(sk.flags + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1u << 26)) {                 // bounded: never hang the device
                        __hip_atomic_store(sk.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            const float* slot = sk.ws + (long)g * (BM * BN);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += slot[((i * TN + j) * 16 + r) * NT + tid];
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------------------
    // The MFMAs are issued with the operands swapped (weights as the row operand), so the 32x32 C/D layout gives each lane
    // FOUR CONSECUTIVE OUTPUT CHANNELS of one pixel per register quad: pixel = lane&31, channel = 8*(r>>2) + 4*(lane>>5) + (r&3).
    // The block tile is staged through LDS (the A/B buffers are dead after the last barrier) with 16-byte writes and read
    // back row-wise, 16 B per lane -> whole 512-B (f32) / 256-B (bf16) row segments per 32 lanes for the stores and the
    // residual loads.
    const bool e_bias = EPI == 0 ? a.bias != nullptr : (EPI == 4 || EPI == 5 || (EPI == 6 && a.bias != nullptr));   // ([r5] EPI 6 + bias: mvf_conv2d_nhwc_dgrad_bnsums_split)
    const bool e_res = EPI == 0 ? a.res != nullptr : (EPI == 3 || EPI == 5 || EPI == 8 || EPI == 9 || EPI == 10 || EPI == 12);
    constexpr bool e_bw = EPI == 9 || EPI == 10;         // BatchNorm backward on the recomputed conv output (g and its sign-bit gate arrive as the residual operand)
    const bool e_relu = EPI == 0 ? a.relu != 0 : (EPI == 4 || EPI == 5);
    constexpr bool e_apply = EPI == 8;                   // BatchNorm apply + residual + ReLU + sign bits on the rounded accumulators
    const bool e_stats = EPI == 0 ? a.stats_part != nullptr : (EPI == 1 || EPI == 12);      // ([r5] EPI 12: residual + gate + the column sums of what is stored)
    constexpr bool e_bnb = EPI == 6;                     // BatchNorm-backward sums instead of forward statistics
    const bool e_scatter = (EPI == 0 || EPI == 6) && a.o_s > 0;      // a strided data gradient's parity class (plain or + BN sums)
    constexpr int CP = BN * 4 + 16;                      // C-tile pitch in bytes
    constexpr int NH = (LOWK || GLDS) ? 2 : 1;           // epilogue passes (row halves of the block tile)
    constexpr int HR = BM / NH;                          // rows per pass
    static_assert(HR * CP <= kSmem, "C tile (or half) must fit in the A/B LDS buffers");
    static_assert(HR % 32 == 0, "half split follows the MFMA row tiles");
    constexpr int PPT = BM / kBM;                        // statistics partials per tile: one per 128 rows, whatever the tile height
    constexpr int HPP = NH / PPT;                        // epilogue passes per partial
    static_assert(NH % PPT == 0 && HPP >= 1, "a statistics partial covers whole epilogue passes");
    constexpr int TPR = BN / 4, RPP = NT / TPR;   // threads per row, rows per pass
    const int cq = tid % TPR, r0 = tid / TPR;
    const int col = n0 + cq * 4;
    ET* y = reinterpret_cast<ET*>(a.y);
    const ET* res = reinterpret_cast<const ET*>(a.res);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e_bias && col < a.Cout) bv = *reinterpret_cast<const float4*>(a.bias + col);
    // fused BatchNorm statistics of the tensor being written (training): this thread's rows of its 4 columns
    float4 st1 = make_float4(0.f, 0.f, 0.f, 0.f), st2 = st1, kk = st1;
    if (e_stats && a.stats_shift && col < a.Cout) kk = *reinterpret_cast<const float4*>(a.stats_shift + col);
    float4 ap_s = st1, ap_b = st1, ap_rs = make_float4(1.f, 1.f, 1.f, 1.f), ap_rb = st1;
    if (e_apply && col < a.Cout) {
        ap_s = *reinterpret_cast<const float4*>(a.ap_scale + col); ap_b = *reinterpret_cast<const float4*>(a.ap_shift + col);
        if (a.ap_rscale) { ap_rs = *reinterpret_cast<const float4*>(a.ap_rscale + col); ap_rb = *reinterpret_cast<const float4*>(a.ap_rshift + col); }
    }
    float4 b_mu = st1, b_rs = st1, b_sc = st1, b_sh = st1;
    if (e_bnb && col < a.Cout) {
        b_mu = *reinterpret_cast<const float4*>(a.bn_mean + col); b_rs = *reinterpret_cast<const float4*>(a.bn_invstd + col);
        b_sc = *reinterpret_cast<const float4*>(a.bn_scale + col); b_sh = *reinterpret_cast<const float4*>(a.bn_shift + col);
    }
    float4 w_a = st1, w_d0 = st1, w_kx = st1;            // EPI 9: gamma * invstd, dbeta / M, invstd * dgamma / M (bn_bwd_apply_kernel's folding)
    if (e_bw && col < a.Cout) {
        b_mu = *reinterpret_cast<const float4*>(a.bn_mean + col); b_rs = *reinterpret_cast<const float4*>(a.bn_invstd + col);
        if constexpr (EPI == 9) {
            const float inv_m = 1.0f / (float)a.M;
            const float4 ga = *reinterpret_cast<const float4*>(a.bw_gamma + col), dg = *reinterpret_cast<const float4*>(a.bw_dgamma + col),
                         db = *reinterpret_cast<const float4*>(a.bw_dbeta + col);
            w_a = make_float4(ga.x * b_rs.x, ga.y * b_rs.y, ga.z * b_rs.z, ga.w * b_rs.w);
            w_d0 = make_float4(db.x * inv_m, db.y * inv_m, db.z * inv_m, db.w * inv_m);
            w_kx = make_float4(b_rs.x * dg.x * inv_m, b_rs.y * dg.y * inv_m, b_rs.z * dg.z * inv_m, b_rs.w * dg.w * inv_m);
        }
    }
    // output / residual descriptors: tile-relative 32-bit offsets (contiguous rows from m0, or the full-resolution images
    // from the tile's first image for a scattered data-gradient class)
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int eimg0 = fd_div(m0, a.fd_hw_mul, a.fd_hw_shr);
    __amdgpu_buffer_rsrc_t rs_y, rs_res;
    {
        const long total = (e_scatter ? (long)a.N * a.o_hfull * a.o_wfull : (long)a.M) * a.Cout * ESZ;
        const long base = (e_scatter ? (long)eimg0 * a.o_hfull * a.o_wfull : (long)m0) * a.Cout * ESZ;
        const long left = total - base;
        const unsigned nrec = (unsigned)(left < 0x7ffffff0L ? left : 0x7ffffff0L);
        rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + base), 0, a.y ? nrec : 0u, 0x00020000);      // no output tensor (a statistics-only pass): every store is out of range = dropped
        rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(e_res ? a.res + base : (e_bnb ? a.bn_z + base : a.y + base)), 0, nrec, 0x00020000);
    }
    // residual gate bits (contiguous outputs only): one byte per 4 channels, addressed like the output / (4 * ESZ)
    const __amdgpu_buffer_rsrc_t rs_mask = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.res_mask ? a.res_mask + (long)m0 * (a.Cout / 4) : (const unsigned char*)a.y), 0,
        (unsigned)min((long)(a.M - m0) * (a.Cout / 4), 0x7ffffff0L), 0x00020000);
    // [r5] the OUTPUT gate (same layout): staged into the HIGH nibble of the same LDS bytes (the sign-bit bytes only use their low nibble), so
    // the gated-output epilogue needs no LDS beyond the residual gate's and keeps its workgroups per CU
    const __amdgpu_buffer_rsrc_t rs_gate = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.out_gate ? a.out_gate + (long)m0 * (a.Cout / 4) : (const unsigned char*)a.y), 0,
        (unsigned)min((long)(a.M - m0) * (a.Cout / 4), 0x7ffffff0L), 0x00020000);
    const bool e_gate = e_res && !e_apply && !e_bw && a.out_gate != nullptr;
    (void)y; (void)res;
    // The tile's gate bytes (BM rows x BN/4) are staged in LDS behind the C tile with ONE 16-byte load per thread (instead of a
    // byte load per thread per row, which made the gated data gradient 35 % slower than the ungated one); visible after the
    // C-tile barrier below.  Needs 16-byte aligned rows: Cout % 64 == 0, else the rows read their byte from memory.
    constexpr int kMaskOff = HR * CP;
    constexpr int MSEG = BN / 64;                        // 16-byte segments per mask row
    static_assert(kMaskOff + BM * (BN / 4) <= kSmem, "mask tile must fit behind the C tile");
    const bool mask_lds = e_res && (a.res_mask || e_gate) && (a.Cout % 64 == 0) && a.mask_lds;
    if (mask_lds) {
#pragma unroll
        for (int it = tid; it < BM * MSEG; it += NT) {       // (one trip for the 128 x 128 / 128 x 64 tiles, two for 256 x 256)
            const int row = it / MSEG, seg = it - row * MSEG;
            const unsigned moff = (unsigned)(row * (a.Cout / 4) + n0 / 4 + seg * 16);
            u32x4 mv = {0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu};      // no residual gate: every residual passes
            if (a.res_mask) mv = __builtin_amdgcn_raw_buffer_load_b128(rs_mask, moff, 0, 0);
            if (e_gate) {
                const u32x4 gv = __builtin_amdgcn_raw_buffer_load_b128(rs_gate, moff, 0, 0);
                mv.x = (mv.x & 0x0f0f0f0fu) | ((gv.x & 0x0f0f0f0fu) << 4); mv.y = (mv.y & 0x0f0f0f0fu) | ((gv.y & 0x0f0f0f0fu) << 4);
                mv.z = (mv.z & 0x0f0f0f0fu) | ((gv.z & 0x0f0f0f0fu) << 4); mv.w = (mv.w & 0x0f0f0f0fu) | ((gv.w & 0x0f0f0f0fu) << 4);
            }
            *reinterpret_cast<uint4*>(smem + kMaskOff + row * (BN / 4) + seg * 16) = make_uint4(mv.x, mv.y, mv.z, mv.w);
        }
    }
    // ---- bf16 training epilogues (EPI 1 forward + BN statistics, 2 plain data gradient; written for 6 = data gradient + BN-backward sums too):
    // nothing is added to the accumulators before the store, so they are rounded to bf16 IN REGISTERS and the C tile is staged
    // as bf16 -- half the LDS bytes (8-byte instead of 16-byte staging writes), the whole 128-row tile in ONE pass instead of two
    // halves (one barrier less), and each thread handles 8 channels of a row: 16-byte loads of z, 16-byte stores, half the
    // per-row offset arithmetic.  Identical numerics: the statistics were always taken from the ROUNDED values.
    if constexpr (sizeof(ET) == 2 && (EPI == 1 || EPI == 2) && BM == kBM) {      // (EPI 6 measured 0...+12 % slower this way: its z loads already fill the registers)
        if ((a.Cout & 7) == 0) {
            constexpr int CPB = BN * 2 + 8;                  // pitch: 2 dwords past a multiple of 32 banks -> conflict-free 8-byte writes
            static_assert(BM * CPB <= kSmem, "bf16 C tile must fit in the A/B LDS buffers");
            constexpr int TPR8 = BN / 8, RPP8 = NT / TPR8, NPS8 = BM / RPP8;
            static_assert(2 * 2 * RPP8 * TPR8 * 16 <= kSmem, "statistics scratch");
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    char* cp = smem + ((wm * TM + i) * 32 + (lane & 31)) * CPB + ((wn * TN + j) * 32 + 4 * (lane >> 5)) * 2;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 pk;
                        pk.x = pack_bf16x2(acc[i][j][4 * g], acc[i][j][4 * g + 1]);
                        pk.y = pack_bf16x2(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                        *reinterpret_cast<uint2*>(cp + g * 16) = pk;
                    }
                }
            const int cq8 = tid % TPR8, r08 = tid / TPR8;
            const int col8 = n0 + cq8 * 8;
            const bool cok8 = col8 < a.Cout;
            float kk8[8], mu8[8], rs8[8], sc8[8], sh8[8], s1[8], s2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { kk8[k] = mu8[k] = rs8[k] = sc8[k] = sh8[k] = s1[k] = s2[k] = 0.f; }
            auto ld8 = [&](const float* p, float (&f)[8]) {
                const float4 u = *reinterpret_cast<const float4*>(p + col8), v = *reinterpret_cast<const float4*>(p + col8 + 4);
                f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w; f[4] = v.x; f[5] = v.y; f[6] = v.z; f[7] = v.w;
            };
            if (EPI == 1 && a.stats_shift && cok8) ld8(a.stats_shift, kk8);
            if (e_bnb && cok8) { ld8(a.bn_mean, mu8); ld8(a.bn_invstd, rs8); ld8(a.bn_scale, sc8); ld8(a.bn_shift, sh8); }
            unsigned offs8[NPS8];
            u32x4 zraw[NPS8];
            {
                const int mrow0 = m0 + r08;
#pragma unroll
                for (int ps = 0; ps < NPS8; ++ps) {
                    const int m = mrow0 + ps * RPP8;
                    unsigned off;
                    if (e_scatter) {
                        const int img = fd_div(m, a.fd_hw_mul, a.fd_hw_shr), rem = m - img * (a.Ho * a.Wo);
                        const int oh = fd_div(rem, a.fd_w_mul, a.fd_w_shr), ow = rem - oh * a.Wo;
                        off = (unsigned)((((img - eimg0) * a.o_hfull + oh * a.o_s + a.o_ph) * a.o_wfull + ow * a.o_s + a.o_pw) * a.Cout + col8) * 2u;
                    } else {
                        off = (unsigned)((m - m0) * a.Cout + col8) * 2u;
                    }
                    offs8[ps] = (cok8 && m < a.M) ? off : kOOB;
                    if constexpr (e_bnb) zraw[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, offs8[ps], 0, 0);
                }
            }
            __syncthreads();
            auto unpack8 = [](const u32x4& t, float (&f)[8]) {
                f[0] = __uint_as_float(t.x << 16); f[1] = __uint_as_float(t.x & 0xffff0000u);
                f[2] = __uint_as_float(t.y << 16); f[3] = __uint_as_float(t.y & 0xffff0000u);
                f[4] = __uint_as_float(t.z << 16); f[5] = __uint_as_float(t.z & 0xffff0000u);
                f[6] = __uint_as_float(t.w << 16); f[7] = __uint_as_float(t.w & 0xffff0000u);
            };
            // FULL: the tile lies inside M x Cout (a uniform test, true for every tile but the ragged last ones): the per-element
            // "row in range" selects of the statistics disappear (1 of 3.5 VALU instructions per element)
            auto rows_pass = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
            for (int ps = 0; ps < NPS8; ++ps) {
                const unsigned off = offs8[ps];
                const bool ok = FULL || off != kOOB;
                const char* src = smem + (r08 + ps * RPP8) * CPB + cq8 * 16;
                const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
                u32x4 pk;
                pk.x = lo.x; pk.y = lo.y; pk.z = hi.x; pk.w = hi.y;
#ifdef MVF_CONV_ABLATE
                if (!(a.prio & 32))                        // ablation: bit 5 = no output stores, bit 6 = no statistics
#endif
                __builtin_amdgcn_raw_buffer_store_b128(pk, rs_y, off, 0, 0);
#ifdef MVF_CONV_ABLATE
                if (a.prio & 64) continue;
#endif
                if constexpr (EPI == 1 || e_bnb) {
                    float v[8];
                    unpack8(pk, v);
                    if constexpr (e_bnb) {
                        float zv[8];
                        unpack8(zraw[ps], zv);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float gm = (ok && (zv[k] * sc8[k] + sh8[k]) > 0.f) ? v[k] : 0.f;
                            s1[k] += gm;
                            s2[k] += gm * ((zv[k] - mu8[k]) * rs8[k]);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float d = ok ? v[k] - kk8[k] : 0.f;        // rows past M contribute nothing
                            s1[k] += d;
                            s2[k] += d * d;
                        }
                    }
                }
            }
            };
            if (m0 + BM <= a.M && n0 + BN <= a.Cout) rows_pass(std::true_type());
            else rows_pass(std::false_type());
#ifdef MVF_CONV_ABLATE
            if (a.prio & 64) return;
#endif
            if constexpr (EPI == 1 || e_bnb) {               // column sums over the RPP8 row-threads, fixed order, one writer per column
                __syncthreads();
                float4* red = reinterpret_cast<float4*>(smem);
                red[((r08 * 2 + 0) * 2 + 0) * TPR8 + cq8] = make_float4(s1[0], s1[1], s1[2], s1[3]);
                red[((r08 * 2 + 0) * 2 + 1) * TPR8 + cq8] = make_float4(s1[4], s1[5], s1[6], s1[7]);
                red[((r08 * 2 + 1) * 2 + 0) * TPR8 + cq8] = make_float4(s2[0], s2[1], s2[2], s2[3]);
                red[((r08 * 2 + 1) * 2 + 1) * TPR8 + cq8] = make_float4(s2[4], s2[5], s2[6], s2[7]);
                __syncthreads();
                // 2 * TPR8 threads finish: thread (h, cq8) sums 4 of the 8 columns of group cq8 over the RPP8 row-threads
                if (tid < 2 * TPR8) {
                    const int h = tid / TPR8, cq = tid - h * TPR8, c4 = n0 + cq * 8 + 4 * h;
                    if (c4 < a.Cout && tm_i * kBM < a.M) {
                        float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
                        for (int r = 0; r < RPP8; ++r) {
                            const float4 p1 = red[((r * 2 + 0) * 2 + h) * TPR8 + cq], p2 = red[((r * 2 + 1) * 2 + h) * TPR8 + cq];
                            t1.x += p1.x; t1.y += p1.y; t1.z += p1.z; t1.w += p1.w;
                            t2.x += p2.x; t2.y += p2.y; t2.z += p2.z; t2.w += p2.w;
                        }
                        float2* p = reinterpret_cast<float2*>(a.stats_part) + (long)c4 * a.stats_rows + tm_i;
                        p[0] = make_float2(t1.x, t2.x);
                        p[a.stats_rows] = make_float2(t1.y, t2.y);
                        p[2 * (long)a.stats_rows] = make_float2(t1.z, t2.z);
                        p[3 * (long)a.stats_rows] = make_float2(t1.w, t2.w);
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
        if (hf > 0) __syncthreads();                     // previous half fully read out
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rbase = (wm * TM + i) * 32;
                if (rbase / HR != hf) continue;          // this MFMA row-tile belongs to the other half
                char* cp = smem + (rbase - hf * HR + (lane & 31)) * CP + ((wn * TN + j) * 32 + 4 * (lane >> 5)) * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(cp + g * 32) =
                        make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            }
        // Phase A, before the C-tile barrier: every row's output offset and its residual (or, for the BatchNorm-sum epilogue,
        // z) load are issued here, all at once -- rows past M / columns past Cout get an out-of-range buffer offset (loads
        // return 0, stores are dropped), so nothing below branches on them and one memory round trip covers the whole half
        // tile instead of one per row.
        constexpr int NPS = HR / RPP;
        typedef typename std::conditional<sizeof(ET) == 2, u32x2, u32x4>::type raw_t;
        unsigned offs[NPS];
        raw_t rraw[NPS];
        {
            const int mrow0 = m0 + hf * HR + r0;
            const bool cok2 = col < a.Cout;              // Cout % 4 == 0 (checked on the host)
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int m = mrow0 + ps * RPP;
                unsigned off;
                if (e_scatter) {                         // strided data-gradient class: scatter into the full-resolution map
                    const int img = fd_div(m, a.fd_hw_mul, a.fd_hw_shr), rem = m - img * (a.Ho * a.Wo);
                    const int oh = fd_div(rem, a.fd_w_mul, a.fd_w_shr), ow = rem - oh * a.Wo;
                    off = (unsigned)((((img - eimg0) * a.o_hfull + oh * a.o_s + a.o_ph) * a.o_wfull + ow * a.o_s + a.o_pw) * a.Cout + col) * ESZ;
                } else {
                    off = (unsigned)((m - m0) * a.Cout + col) * ESZ;
                }
                offs[ps] = (cok2 && m < a.M) ? off : kOOB;
                if (e_res || e_bnb) {
                    const unsigned roff = (e_bnb || col >= a.res_c0) ? offs[ps] : kOOB;       // skipped columns read zeros
                    if constexpr (sizeof(ET) == 2) rraw[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, roff, 0, 0);
                    else rraw[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, roff, 0, 0);
                }
            }
        }
        __syncthreads();
        {
            auto unpack = [](const raw_t& t) {
                if constexpr (sizeof(ET) == 2)
                    return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                                       __uint_as_float(t.y & 0xffff0000u));
                else
                    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
            };
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const unsigned off = offs[ps];
                const bool ok = off != kOOB;
                float4 v = *reinterpret_cast<const float4*>(smem + (r0 + ps * RPP) * CP + cq * 16);
                if constexpr (e_apply) {
                    if constexpr (sizeof(ET) == 2) {         // z3 as the first pass stored it
                        const unsigned p0 = pack_bf16x2(v.x, v.y), p1 = pack_bf16x2(v.z, v.w);
                        v = make_float4(__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u));
                    }
                    const float4 q = unpack(rraw[ps]);
                    float t0 = v.x * ap_s.x + ap_b.x, t1 = v.y * ap_s.y + ap_b.y, t2 = v.z * ap_s.z + ap_b.z, t3 = v.w * ap_s.w + ap_b.w;
                    t0 += q.x * ap_rs.x + ap_rb.x; t1 += q.y * ap_rs.y + ap_rb.y; t2 += q.z * ap_rs.z + ap_rb.z; t3 += q.w * ap_rs.w + ap_rb.w;
                    v = make_float4(fmaxf(t0, 0.f), fmaxf(t1, 0.f), fmaxf(t2, 0.f), fmaxf(t3, 0.f));
                    if (ok) a.ap_bits[(long)(m0 + hf * HR + r0 + ps * RPP) * (a.Cout / 4) + (col >> 2)] =
                        (unsigned char)((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u));
                }
                if constexpr (e_bw) {
                    if constexpr (sizeof(ET) == 2) {         // z3 as the forward pass rounded it
                        const unsigned p0 = pack_bf16x2(v.x, v.y), p1 = pack_bf16x2(v.z, v.w);
                        v = make_float4(__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u));
                    }
                    float4 gm = unpack(rraw[ps]);            // g, gated by the sign bits of the block output
                    const unsigned mb = mask_lds ? (unsigned)*reinterpret_cast<const unsigned char*>(smem + kMaskOff + (hf * HR + r0 + ps * RPP) * (BN / 4) + cq)
                                                 : (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_mask, ok ? off / (4 * ESZ) : kOOB, 0, 0);
                    gm.x = (mb & 1u) ? gm.x : 0.f; gm.y = (mb & 2u) ? gm.y : 0.f; gm.z = (mb & 4u) ? gm.z : 0.f; gm.w = (mb & 8u) ? gm.w : 0.f;
                    if constexpr (EPI == 10) {
                        if (!ok) gm = make_float4(0.f, 0.f, 0.f, 0.f);
                        st1.x += gm.x; st1.y += gm.y; st1.z += gm.z; st1.w += gm.w;
                        st2.x += gm.x * ((v.x - b_mu.x) * b_rs.x); st2.y += gm.y * ((v.y - b_mu.y) * b_rs.y);
                        st2.z += gm.z * ((v.z - b_mu.z) * b_rs.z); st2.w += gm.w * ((v.w - b_mu.w) * b_rs.w);
                        continue;                            // nothing is stored
                    } else {
                        v.x = w_a.x * (gm.x - w_d0.x - (v.x - b_mu.x) * w_kx.x); v.y = w_a.y * (gm.y - w_d0.y - (v.y - b_mu.y) * w_kx.y);
                        v.z = w_a.z * (gm.z - w_d0.z - (v.z - b_mu.z) * w_kx.z); v.w = w_a.w * (gm.w - w_d0.w - (v.w - b_mu.w) * w_kx.w);
                    }
                }
                if (e_bias) { v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
                if (e_res && !e_apply && !e_bw) {
                    float4 rv = unpack(rraw[ps]);
                    unsigned mb = 0xffu;
                    if (a.res_mask || e_gate) {
                        if (mask_lds) mb = (unsigned)*reinterpret_cast<const unsigned char*>(smem + kMaskOff + (hf * HR + r0 + ps * RPP) * (BN / 4) + cq);
                        else {
                            if (a.res_mask) mb = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_mask, ok ? off / (4 * ESZ) : kOOB, 0, 0) & 0xfu;
                            else mb = 0xfu;
                            mb |= e_gate ? ((unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_gate, ok ? off / (4 * ESZ) : kOOB, 0, 0) & 0xfu) << 4 : 0xf0u;
                        }
                        rv.x = (mb & 1u) ? rv.x : 0.f; rv.y = (mb & 2u) ? rv.y : 0.f;
                        rv.z = (mb & 4u) ? rv.z : 0.f; rv.w = (mb & 8u) ? rv.w : 0.f;
                    }
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                    if (e_gate && col >= a.res_c0) {         // [r5] gm = (data gradient + skip-connection gradient) * [block output below > 0]
                        v.x = (mb & 16u) ? v.x : 0.f; v.y = (mb & 32u) ? v.y : 0.f;
                        v.z = (mb & 64u) ? v.z : 0.f; v.w = (mb & 128u) ? v.w : 0.f;
                    }
                }
                if (e_relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                if constexpr (sizeof(ET) == 2) {
                    u32x2 pk;
                    pk.x = pack_bf16x2(v.x, v.y);
                    pk.y = pack_bf16x2(v.z, v.w);
#ifdef MVF_CONV_ABLATE
                    if (!(a.prio & 32))
#endif
                    __builtin_amdgcn_raw_buffer_store_b64(pk, rs_y, off, 0, 0);
                    // statistics of what is STORED (bf16-rounded), as the consumers will read it
                    v = make_float4(__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u),
                                    __uint_as_float(pk.y << 16), __uint_as_float(pk.y & 0xffff0000u));
                } else {
                    u32x4 pk;
                    pk.x = __float_as_uint(v.x); pk.y = __float_as_uint(v.y); pk.z = __float_as_uint(v.z); pk.w = __float_as_uint(v.w);
                    __builtin_amdgcn_raw_buffer_store_b128(pk, rs_y, off, 0, 0);
                }
                if constexpr (e_bnb) {               // v = the stored (rounded) gradient; z was fetched in phase A
                    const float4 zv = unpack(rraw[ps]);
                    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    v.x = (zv.x * b_sc.x + b_sh.x) > 0.f ? v.x : 0.f; v.y = (zv.y * b_sc.y + b_sh.y) > 0.f ? v.y : 0.f;
                    v.z = (zv.z * b_sc.z + b_sh.z) > 0.f ? v.z : 0.f; v.w = (zv.w * b_sc.w + b_sh.w) > 0.f ? v.w : 0.f;
                    st1.x += v.x; st1.y += v.y; st1.z += v.z; st1.w += v.w;
                    st2.x += v.x * ((zv.x - b_mu.x) * b_rs.x); st2.y += v.y * ((zv.y - b_mu.y) * b_rs.y);
                    st2.z += v.z * ((zv.z - b_mu.z) * b_rs.z); st2.w += v.w * ((zv.w - b_mu.w) * b_rs.w);
                }
                if (e_stats) {
                    if (!ok) v = kk;                     // rows past M contribute nothing
                    v.x -= kk.x; v.y -= kk.y; v.z -= kk.z; v.w -= kk.w;
                    st1.x += v.x; st1.y += v.y; st1.z += v.z; st1.w += v.w;
                    st2.x += v.x * v.x; st2.y += v.y * v.y; st2.z += v.z * v.z; st2.w += v.w * v.w;
                }
            }
        }
        if ((e_stats || e_bnb || EPI == 10) && (hf + 1) % HPP == 0) {   // column sums over the RPP row-threads, fixed order, one writer per column
            __syncthreads();
            float4* red = reinterpret_cast<float4*>(smem);
            red[(r0 * 2 + 0) * TPR + cq] = st1;
            red[(r0 * 2 + 1) * TPR + cq] = st2;
            __syncthreads();
            const int pidx = tm_i * PPT + hf / HPP;      // one partial per 128 rows
            if (r0 == 0 && col < a.Cout && pidx * kBM < a.M) {
                float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
                for (int r = 0; r < RPP; ++r) {
                    const float4 p1 = red[(r * 2 + 0) * TPR + cq], p2 = red[(r * 2 + 1) * TPR + cq];
                    s1.x += p1.x; s1.y += p1.y; s1.z += p1.z; s1.w += p1.w;
                    s2.x += p2.x; s2.y += p2.y; s2.z += p2.z; s2.w += p2.w;
                }
                float2* p = reinterpret_cast<float2*>(a.stats_part) + (long)col * a.stats_rows + pidx;
                p[0] = make_float2(s1.x, s2.x);
                p[a.stats_rows] = make_float2(s1.y, s2.y);
                p[2 * (long)a.stats_rows] = make_float2(s1.z, s2.z);
                p[3 * (long)a.stats_rows] = make_float2(s1.w, s2.w);
            }
            st1 = make_float4(0.f, 0.f, 0.f, 0.f);
            st2 = st1;
        }
    }
}

template <typename ET, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(kThreads) void conv_igemm_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkArgs sk = {};
    conv_tile<ET, WM, WN, TM, TN>(a, smem, xcd_swizzle(blockIdx.x, gridDim.x), 0, a.nchunks, SEG_FULL, sk, 0, 0);
}

// generic fallback (conv_tile GEN): input dilation, kernels wider than the 32-bit tap masks, or images too large for 32-bit
// tile-relative offsets
template <typename ET, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(kThreads) void conv_igemm_gen_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkArgs sk = {};
    conv_tile<ET, WM, WN, TM, TN, false, false, true>(a, smem, xcd_swizzle(blockIdx.x, gridDim.x), 0, a.nchunks, SEG_FULL, sk, 0, 0);
}

// two chunks ahead in registers (see conv_tile PF2): the latency-bound variant of conv_igemm_kernel
template <typename ET, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(kThreads) void conv_igemm_pf2_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkArgs sk = {};
    conv_tile<ET, WM, WN, TM, TN, false, true>(a, smem, xcd_swizzle(blockIdx.x, gridDim.x), 0, a.nchunks, SEG_FULL, sk, 0, 0);
}

template <typename ET, int WM, int WN, int TM, int TN, int EPI = 0, bool PW = false, bool MVFL = false>
__global__ __launch_bounds__(kThreads) void conv_igemm_lowk_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkArgs sk = {};
    conv_tile<ET, WM, WN, TM, TN, true, false, false, EPI, PW, 0, MVFL>(a, smem, xcd_swizzle(blockIdx.x, gridDim.x), 0, a.nchunks, SEG_FULL, sk, 0, 0);
}

// fp32 storage, products as six bf16 partial products on the bf16 matrix cores (conv_tile X3)
template <int WM, int WN, int TM, int TN, int EPI = 0, bool PW = false>
__global__ __launch_bounds__(kThreads, 3) void conv_igemm_x3_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkArgs sk = {};
    conv_tile<float, WM, WN, TM, TN, true, false, false, EPI, PW, 0, false, false, false, true>(a, smem, xcd_swizzle(blockIdx.x, gridDim.x), 0, a.nchunks, SEG_FULL, sk, 0, 0);
}

// LDS-DMA staged variant for the long-K (matrix-core bound) launches
// HALFK: a K chunk is 128 bytes per row; when the packed input channels of a tap fill only half of it (the bf16 stem: 8 pixels x 4
// channels = 64 bytes) the two upper k-steps would multiply zeros and are not issued: half the MFMAs and operand reads of that conv
// (a compile-time variant: the same test as a run-time branch inside the pinned MFMA / fetch schedule cost every conv launch 4x).
template <typename ET, int WM, int WN, int TM, int TN, int EPI, int NB, bool MVFL = false, bool PW = false, bool HALFK = false>
__global__ __launch_bounds__(kThreads, (NB == 1 && !MVFL) ? 4 : 1) void conv_igemm_glds_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkArgs sk = {};
    conv_tile<ET, WM, WN, TM, TN, false, false, false, EPI, PW, NB, MVFL, false, HALFK>(a, smem, xcd_swizzle(blockIdx.x, gridDim.x), 0, a.nchunks, SEG_FULL, sk, 0, 0);
}

// The long-K tile: 256 x 128 outputs per workgroup of 8 waves (4 x 2, 64 x 64 each), three 48 KB LDS-DMA buffers = one
// workgroup per CU with two chunks (96 KB) in flight.  A 128 x 128 tile moves 32 KB per 2.1 MFLOP (64 FLOP/B); at the ~1.5 us
// loaded latency the 64-96 KB a CU can keep in flight caps it near 700 TF/s whatever the staging (measured: register-staged
// 3 workgroups/CU 650, LDS-DMA 2 x 2 buffers 700-750) -- this tile needs 2/3 of the bytes per flop and keeps 1.5x in flight.
template <typename ET, int EPI>
__global__ __launch_bounds__(512) void conv_igemm_big_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkArgs sk = {};
    conv_tile<ET, 4, 2, 2, 2, false, false, false, EPI, false, 3>(a, smem, xcd_swizzle(blockIdx.x, gridDim.x), 0, a.nchunks, SEG_FULL, sk, 0, 0);
}

// experiment: the 128 x 128 two-buffer DMA tile computed by 8 waves (4 x 2, 32 x 64 each) -> 4 waves per SIMD at 2 workgroups/CU
template <typename ET, int EPI>
__global__ __launch_bounds__(512) void conv_igemm_glds8_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkArgs sk = {};
    conv_tile<ET, 4, 2, 1, 2, false, false, false, EPI, false, 2>(a, smem, xcd_swizzle(blockIdx.x, gridDim.x), 0, a.nchunks, SEG_FULL, sk, 0, 0);
}

// 256 x 256 outputs per workgroup of 8 waves (4 x 2, 64 x 128 each), two 64 KB LDS-DMA buffers, one workgroup per CU: per SIMD 16 DMA
// instructions feed 2048 cycles of MFMA work per chunk -- twice the ratio of the 128 x 128 and 256 x 128 tiles, which plateau on the DMA
// issue path.  Only for launches whose tile count suits 256 single-workgroup slots (policy conv_big2).
template <typename ET, int EPI, bool ILV>
__global__ __launch_bounds__(512) void conv_igemm_big2_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkArgs sk = {};
    conv_tile<ET, 4, 2, 2, 4, false, false, false, EPI, false, 2, false, ILV>(a, smem, xcd_swizzle(blockIdx.x, gridDim.x), 0, a.nchunks, SEG_FULL, sk, 0, 0);
}

template <typename ET, int EPI>
int launch_big2(hipStream_t st, const ConvArgs& a0) {
    ConvArgs a = a0;
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.Cout + 255) / 256;
    constexpr int lds = kGldsLds<256, 256, 2>();
    // experiment switch policy conv_ilv=1: the next chunk's DMA pieces issued one behind each MFMA of the first k-step instead of in a block
    // ahead of them.  Measured neutral on the K = 2304 launches and 6-8 % SLOWER on the K = 1024 ones (0.234 -> 0.25 ms): off.
    static const int ilv = mvf_policy_int("conv_ilv", 0);
    auto k0 = conv_igemm_big2_kernel<ET, EPI, false>;
    auto k1 = conv_igemm_big2_kernel<ET, EPI, true>;
    static bool attr = false;
    if (!attr) {
        MVF_HIP_OK(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        MVF_HIP_OK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    if (ilv) hipLaunchKernelGGL(k1, dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a);
    else hipLaunchKernelGGL(k0, dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a);
    return MVF_OK;
}

// The 256 x 256 tile on the four-phase ping-pong loop (conv_tile GLDS = 4): 8 waves as 2 x 4 (128 x 64 outputs each), two 64 KB LDS-DMA
// buffers staged by half tiles, three half tiles in flight across every barrier.  bf16 only.
template <typename ET, int EPI, bool PW>
__global__ __launch_bounds__(512) void conv_igemm_p4_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkArgs sk = {};
    conv_tile<ET, 2, 4, 4, 2, false, false, false, EPI, PW, 4>(a, smem, xcd_swizzle(blockIdx.x, gridDim.x), 0, a.nchunks, SEG_FULL, sk, 0, 0);
}

template <typename ET, int EPI>
int launch_p4(hipStream_t st, const ConvArgs& a0) {
    if constexpr (sizeof(ET) != 2) {
        return MVF_EUNSUPPORTED;
    } else {
        ConvArgs a = a0;
        a.tiles_m = (a.M + 255) / 256;
        a.tiles_n = (a.Cout + 255) / 256;
        constexpr int lds = kGldsLds<256, 256, 2>();
        auto k0 = conv_igemm_p4_kernel<ET, EPI, false>;
        auto k1 = conv_igemm_p4_kernel<ET, EPI, true>;
        static bool attr = false;
        if (!attr) {
            MVF_HIP_OK(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            MVF_HIP_OK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            attr = true;
        }
        const bool pw = a.KH == 1 && a.KW == 1 && a.pad == 0 && a.pad_w == 0 && a.split_c == 0 && a.dil <= 1;
        if (pw) hipLaunchKernelGGL(k1, dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a);
        else hipLaunchKernelGGL(k0, dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a);
        return MVF_OK;
    }
}

template <typename ET, int EPI>
int launch_big(hipStream_t st, const ConvArgs& a0) {
    ConvArgs a = a0;
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.Cout + 127) / 128;
    auto k = conv_igemm_big_kernel<ET, EPI>;
    constexpr int lds = kGldsLds<256, 128, 3>();
    static bool attr = false;
    if (!attr) {
        MVF_HIP_OK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    hipLaunchKernelGGL(k, dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a);
    return MVF_OK;
}

template <typename ET, int WM, int WN, int TM, int TN, int EPI>
int launch_glds(int nb, int tiles, hipStream_t st, const ConvArgs& a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    if constexpr (BM == 128 && BN == 128) {
        if (nb == 3) {
            auto k = conv_igemm_glds8_kernel<ET, EPI>;
            constexpr int lds = kGldsLds<128, 128, 2>();
            static bool attr8 = false;
            if (!attr8) {
                MVF_HIP_OK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                attr8 = true;
            }
            hipLaunchKernelGGL(k, dim3(tiles), dim3(512), lds, st, a);
            return MVF_OK;
        }
    }
    // pointwise launches (1x1 taps, no padding, no split operand) take the loader specialisation PW: no tap masks, no second-operand
    // offsets and, at stride 1, no division in the per-tile set-up (policy conv_epi bit 1, as for the register-staged kernel)
    static const int epi_spec = mvf_policy_int("conv_epi", 3);
    const bool pw = (epi_spec & 2) && a.KH == 1 && a.KW == 1 && a.pad == 0 && a.pad_w == 0 && a.split_c == 0 && a.dil <= 1;
    if (nb == 1) {
        constexpr int lds = kGldsLds<BM, BN, 1>();
        if constexpr (BN == 64 && sizeof(ET) == 2 && (EPI == 1 || EPI == 4)) {       // the stem (training: + statistics; inference: bias + ReLU)
            if (!pw && (size_t)a.Cin * sizeof(ET) <= 64 && a.split_c == 0) {
                hipLaunchKernelGGL((conv_igemm_glds_kernel<ET, WM, WN, TM, TN, EPI, 1, false, false, true>), dim3(tiles), dim3(kThreads), lds, st, a);
                return MVF_OK;
            }
        }
        if (pw) hipLaunchKernelGGL((conv_igemm_glds_kernel<ET, WM, WN, TM, TN, EPI, 1, false, true>), dim3(tiles), dim3(kThreads), lds, st, a);
        else hipLaunchKernelGGL((conv_igemm_glds_kernel<ET, WM, WN, TM, TN, EPI, 1>), dim3(tiles), dim3(kThreads), lds, st, a);
    } else {
        constexpr int lds = kGldsLds<BM, BN, 2>();
        static bool attr = false;
        if (!attr) {
            MVF_HIP_OK(hipFuncSetAttribute((const void*)(conv_igemm_glds_kernel<ET, WM, WN, TM, TN, EPI, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            MVF_HIP_OK(hipFuncSetAttribute((const void*)(conv_igemm_glds_kernel<ET, WM, WN, TM, TN, EPI, 2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            attr = true;
        }
        if (pw) hipLaunchKernelGGL((conv_igemm_glds_kernel<ET, WM, WN, TM, TN, EPI, 2, false, true>), dim3(tiles), dim3(kThreads), lds, st, a);
        else hipLaunchKernelGGL((conv_igemm_glds_kernel<ET, WM, WN, TM, TN, EPI, 2>), dim3(tiles), dim3(kThreads), lds, st, a);
    }
    return MVF_OK;
}

// X3 on a 256 x 128 tile computed by 8 waves (4 x 2, 64 x 64 each): the per-chunk fixed costs of the single-buffer loop (two barriers, the
// LDS round trips, the store phase) are paid once per 2 x the matrix work, and a workgroup keeps two waves on every SIMD
template <int EPI = 0, bool PW = false>
__global__ __launch_bounds__(512) void conv_igemm_x3w_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkArgs sk = {};
    conv_tile<float, 4, 2, 2, 2, true, false, false, EPI, PW, 0, false, false, false, true>(a, smem, xcd_swizzle(blockIdx.x, gridDim.x), 0, a.nchunks, SEG_FULL, sk, 0, 0);
}
int g_x3_wide = 0;               // policy x3_wide=n: the 8-wave 256 x 128 X3 tile for launches with >= n tiles of that size (0 = off)

int g_f32_x3 = 1;                // fp32 storage: products on the bf16 matrix cores as 3-term bf16 splits (policy f32_x3=0: the fp32 MFMA)
int g_x3_db_min = 1 << 30;       // X3: double-buffered planes (one workgroup per CU, one barrier per chunk) from this many K chunks on (policy x3_db)

// X3 with two LDS buffers: chunk k + 1 is split and written to the other buffer behind the MFMAs of chunk k, one barrier per chunk
template <int WM, int WN, int TM, int TN, int EPI = 0, bool PW = false>
__global__ __launch_bounds__(kThreads) void conv_igemm_x3db_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SkArgs sk = {};
    conv_tile<float, WM, WN, TM, TN, false, false, false, EPI, PW, 0, false, false, false, true>(a, smem, xcd_swizzle(blockIdx.x, gridDim.x), 0, a.nchunks, SEG_FULL, sk, 0, 0);
}

// policy f32_x3: 0 never, 1 always; diagnostics: 2 only launches with a forward epilogue (statistics / bias / BatchNorm apply), 3 only the others
// (data gradients, plain forwards), 4 only 3x3 taps, 5 only pointwise, 6 only the stem's 7 x 1 view, 7 everything but the stem
inline bool x3_on(const ConvArgs& a) {
    const bool fwd_like = a.stats_part || a.bias || a.ap_scale || a.bw_mode;
    switch (g_f32_x3) {
        case 1: return true;
        case 2: return fwd_like;
        case 3: return !fwd_like;
        case 4: return a.KH == 3;
        case 5: return a.KH == 1 && a.KW == 1;
        case 6: return a.KH == 7;
        case 7: return a.KH != 7;
        default: return false;
    }
}

template <typename ET, int WM, int WN, int TM, int TN, int EPI>
void launch_lowk(bool pw, int tiles, size_t lds, hipStream_t st, const ConvArgs& a) {
    if constexpr (sizeof(ET) == 4) {
        if (x3_on(a)) {
            constexpr size_t lds3 = (size_t)kLowkLdsX3<WM * TM * 32, WN * TN * 32>();
            if constexpr (WN * TN * 32 == 128 && EPI >= 1 && EPI <= 6) {
                const int t_wide = ((a.M + 255) / 256) * a.tiles_n;
                if (g_x3_wide > 0 && t_wide >= g_x3_wide) {
                    ConvArgs b = a;
                    b.tiles_m = (a.M + 255) / 256;
                    constexpr size_t ldsw = (size_t)kLowkLdsX3<256, 128>();
                    auto k0 = conv_igemm_x3w_kernel<EPI, false>;
                    auto k1 = conv_igemm_x3w_kernel<EPI, true>;
                    static bool attrw = false;
                    if (!attrw) {
                        (void)hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
                        (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
                        attrw = true;
                    }
                    if (pw) hipLaunchKernelGGL(k1, dim3(t_wide), dim3(512), ldsw, st, b);
                    else hipLaunchKernelGGL(k0, dim3(t_wide), dim3(512), ldsw, st, b);
                    return;
                }
            }
            if constexpr (EPI >= 1 && EPI <= 6) {
                if (a.nchunks >= g_x3_db_min) {
                    constexpr size_t ldsdb = (size_t)2 * (WM * TM * 32 + WN * TN * 32) * 192;
                    auto k0 = conv_igemm_x3db_kernel<WM, WN, TM, TN, EPI, false>;
                    auto k1 = conv_igemm_x3db_kernel<WM, WN, TM, TN, EPI, true>;
                    static bool attr = false;
                    if (!attr) {
                        (void)hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsdb);
                        (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsdb);
                        attr = true;
                    }
                    if (pw) hipLaunchKernelGGL(k1, dim3(tiles), dim3(kThreads), ldsdb, st, a);
                    else hipLaunchKernelGGL(k0, dim3(tiles), dim3(kThreads), ldsdb, st, a);
                    return;
                }
            }
            if (pw) hipLaunchKernelGGL((conv_igemm_x3_kernel<WM, WN, TM, TN, EPI, true>), dim3(tiles), dim3(kThreads), lds3, st, a);
            else hipLaunchKernelGGL((conv_igemm_x3_kernel<WM, WN, TM, TN, EPI, false>), dim3(tiles), dim3(kThreads), lds3, st, a);
            return;
        }
    }
    if (pw) {
        auto k = conv_igemm_lowk_kernel<ET, WM, WN, TM, TN, EPI, true>;
        hipLaunchKernelGGL(k, dim3(tiles), dim3(kThreads), lds, st, a);
    } else {
        auto k = conv_igemm_lowk_kernel<ET, WM, WN, TM, TN, EPI, false>;
        hipLaunchKernelGGL(k, dim3(tiles), dim3(kThreads), lds, st, a);
    }
}

// Stream-K tail: the last, partial wave of tiles is NOT run one tile per workgroup (which leaves e.g. 47 % of the CUs idle
// for 784 tiles on 512 slots); its tiles x chunks are cut into G equal contiguous unit ranges.  A range is
// [tail part of tile A][whole tiles][head part of tile B]; a workgroup first computes and PUBLISHES the head part (so its
// successor never waits long), then whole tiles, and last FINISHES tile A: it adds the partials published by the lower-
// numbered workgroups that own A's earlier chunks and runs the normal fused epilogue.  Waits only ever point at lower
// workgroup ids whose publication is the first thing they do -> no cyclic wait, no co-residency requirement.
template <typename ET, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(kThreads) void conv_streamk_kernel(ConvArgs a, SkArgs sk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int g = blockIdx.x, nc = a.nchunks;
    auto start_of = [&](int w) { return w < sk.units_rem ? w * (sk.units_base + 1) : sk.units_rem * (sk.units_base + 1) + (w - sk.units_rem) * sk.units_base; };
    auto wg_of = [&](int u) {
        const int bnd = sk.units_rem * (sk.units_base + 1);
        return u < bnd ? u / (sk.units_base + 1) : sk.units_rem + (u - bnd) / sk.units_base;
    };
    const int u0 = start_of(g), u1 = start_of(g + 1);
    const int t_first = u0 / nc, t_last = (u1 - 1) / nc;
    const int cb_first = u0 - t_first * nc, ce_last = u1 - t_last * nc;       // chunk range ends inside the first / last tile
    // 1) publish the trailing partial (head or middle part of tile t_last), if this range does not finish that tile
    const bool last_partial = ce_last < nc;
    if (last_partial) {
        const int cb = t_last == t_first ? cb_first : 0;
        conv_tile<ET, WM, WN, TM, TN>(a, smem, sk.tile0 + t_last, cb, ce_last, SEG_PRODUCE, sk, 0, g);
    }
    // 2) whole tiles, and 3) the leading tile last (it may have to gather partials from lower workgroups)
    const int t_hi = last_partial ? t_last - 1 : t_last;
    for (int t = t_first + 1; t <= t_hi; ++t) conv_tile<ET, WM, WN, TM, TN>(a, smem, sk.tile0 + t, 0, nc, SEG_FULL, sk, 0, g);
    if (t_first <= t_hi) {
        if (cb_first == 0) conv_tile<ET, WM, WN, TM, TN>(a, smem, sk.tile0 + t_first, 0, nc, SEG_FULL, sk, 0, g);
        else conv_tile<ET, WM, WN, TM, TN>(a, smem, sk.tile0 + t_first, cb_first, nc, SEG_FINISH, sk, wg_of(t_first * nc), g);
    }
}

template <typename ET>
__global__ void pack_weight_kernel(const float* w, int cout, int cin, int kh, int kw, int kwp, int cinp,
                                   const float* scale, ET* out) {
    const long total = (long)cout * kh * kwp * cinp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int ci = (int)(t % cinp); t /= cinp;
        const int x = (int)(t % kwp); t /= kwp;
        const int yk = (int)(t % kh); t /= kh;
        const int co = (int)t;
        float v = 0.f;
        if (ci < cin && x < kw) {
            v = w[(((long)co * cin + ci) * kh + yk) * kw + x];
            if (scale) v *= scale[co];
        }
        stf(out + i, v);
    }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                               int c, float* scale, float* shift) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const float s = gamma[i] / sqrtf(var[i] + eps);
    scale[i] = s;
    shift[i] = beta[i] - mean[i] * s;
}

static inline bool contiguous_ok(const ConvArgs& a) { return a.o_s <= 0; }
struct SkHost {
    void* ws;
    size_t ws_bytes;
};

// Launches with at most this many K chunks use the single-LDS-buffer variant (policy conv_lowk=n; 0 disables).  Measured on the
// R50 train step: the 36 KB variant (3-4 workgroups per CU) beats the double-buffered 72 KB one (2 per CU) at EVERY K, bf16 and
// fp32 -- occupancy hides more latency than the second buffer does (bf16 27.97 -> 26.96 ms, fp32 83.0 -> 79.6 ms per step).
int g_lowk_max_chunks = 1 << 30;
int g_pf2_mode = 1;              // two-chunk register prefetch: 0 off, 1 bf16 only, 2 both dtypes (policy conv_pf2)
int g_glds1_f32_infer = 1;       // fp32: only the inference epilogues (bias + ReLU [+ residual]) take it by default (policy conv_glds1_f32=0/1)
int g_glds1_max = -1;            // single-buffer LDS-DMA kernel (4 workgroups per CU) up to this many K chunks: -1 = default policy
                                 // (bf16: 8), 0 = off (policy conv_glds1)
int g_big_min = 0;               // 256 x 128 LDS-DMA tile from this many K chunks on (policy conv_big; 0 = off)
int g_big2_min = 16;             // 256 x 256 LDS-DMA tile (bf16) from this many K chunks on, when the tile count suits it (policy conv_big2; 0 = off).
                                 // Measured on the R50 bf16 train step: layer3's K = 1024 pointwise launches -16...-19 %, its 3x3 convs -5...-7 %
                                 // (conv family 9.59 -> 9.42 ms per step); from 8 chunks on the K = 512 launches lose (9.49)
int g_glds_min = -1, g_glds_nb = 2;  // LDS-DMA variant: -1 = the measured default policy (see launch_conv), 0 = off, n = from n K chunks on;
                                     // with 1 or 2 LDS buffers (policy conv_glds=min[,nb])

int sk_slots() {
    static int slots = 0;
    if (!slots) {
        g_lowk_max_chunks = mvf_policy_int("conv_lowk", g_lowk_max_chunks);
        g_pf2_mode = mvf_policy_int("conv_pf2", g_pf2_mode);
        g_glds1_f32_infer = mvf_policy_int("conv_glds1_f32", g_glds1_f32_infer);
        g_glds1_max = mvf_policy_int("conv_glds1", g_glds1_max);
        g_f32_x3 = mvf_policy_int("f32_x3", g_f32_x3);
        g_x3_wide = mvf_policy_int("x3_wide", g_x3_wide);
        g_x3_db_min = mvf_policy_int("x3_db", g_x3_db_min);
        g_big_min = mvf_policy_int("conv_big", g_big_min);
        g_big2_min = mvf_policy_int("conv_big2", g_big2_min);
        g_glds_min = mvf_policy_int("conv_glds", g_glds_min);
        g_glds_nb = mvf_policy_int("conv_glds_nb", g_glds_nb);      // 3 = two buffers, 8 waves (128 x 128 tile only)
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
        }
        slots = 2 * cus;                  // 2 workgroups per CU (72 KB LDS each)
    }
    return slots;
}

float sk_min_us() {          // stream-K when cutting the partial tile wave saves more than this (policy sk_min_us; the fix-up costs ~60 us)
    static const float v = (float)mvf_policy_int("sk_min_us", 60);
    return v;
}

size_t sk_ws_bytes() { return (size_t)sk_slots() * (128 * 128 * sizeof(float)) + 4096; }

template <typename ET, int WM, int WN, int TM, int TN>
int launch_conv(const ConvArgs& a0, const SkHost& skh, hipStream_t st) {
    ConvArgs a = a0;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    fd_make((unsigned)(a.Ho * a.Wo), a.fd_hw_mul, a.fd_hw_shr);
    fd_make((unsigned)a.Wo, a.fd_w_mul, a.fd_w_shr);
    const size_t lds = (size_t)2 * (BM + BN) * kPitch;
    auto kern = conv_igemm_kernel<ET, WM, WN, TM, TN>;
    auto kern_sk = conv_streamk_kernel<ET, WM, WN, TM, TN>;
    auto kern_pf = conv_igemm_pf2_kernel<ET, WM, WN, TM, TN>;
    static bool attr_done = false;   // per instantiation; idempotent, so a benign race at worst
    if (!attr_done) {
        MVF_HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MVF_HIP_OK(hipFuncSetAttribute((const void*)kern_pf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MVF_HIP_OK(hipFuncSetAttribute((const void*)kern_sk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const int tiles = a.tiles_m * a.tiles_n, slots = sk_slots();
    // the fast loader addresses a tile with 32-bit byte offsets relative to its first image (a tile spans <= 128 + 1 images)
    const long img_bytes = (long)a.H * a.W * std::max(a.xps, a.x2ps) * (long)sizeof(ET);
    const long span_imgs = std::min<long>(a.N, 128L / std::max(1, a.Ho * a.Wo) + 2);
    // the epilogue addresses the output the same way (only a scattered data-gradient class can span whole images)
    MVF_REQUIRE(a.o_s <= 0 || (long)a.o_hfull * a.o_wfull * a.Cout * (long)sizeof(ET) * span_imgs < 0x7ffffff0L, MVF_EUNSUPPORTED,
                "conv2d: output image too large for tile-relative 32-bit addressing");
    MVF_REQUIRE(!a.bn_z || !(a.dil > 1 || a.KH > 31 || a.KW > 31 || img_bytes * span_imgs >= 0x7ffffff0L), MVF_EUNSUPPORTED,
                "conv2d_dgrad_bnsums: shape needs the generic kernel, which has no BatchNorm-backward epilogue");
    MVF_REQUIRE(!a.bw_mode || !(a.dil > 1 || a.KH > 31 || a.KW > 31 || img_bytes * span_imgs >= 0x7ffffff0L || a.o_s > 0), MVF_EUNSUPPORTED,
                "conv2d_fwd_bnbwd: shape needs the generic kernel / a scattered output, which have no BatchNorm-backward-on-recompute epilogue");
    MVF_REQUIRE(!a.ap_scale || !(a.dil > 1 || a.KH > 31 || a.KW > 31 || img_bytes * span_imgs >= 0x7ffffff0L || a.o_s > 0), MVF_EUNSUPPORTED,
                "conv2d_fwd_bnapply: shape needs the generic kernel / a scattered output, which have no BatchNorm-apply epilogue");
    if (a.dil > 1 || a.KH > 31 || a.KW > 31 || img_bytes * span_imgs >= 0x7ffffff0L || (long)a.Cout * a.wK * (long)sizeof(ET) >= 0x7ffffff0L) {
        auto kern_gen = conv_igemm_gen_kernel<ET, WM, WN, TM, TN>;
        static bool gen_attr = false;
        if (!gen_attr) {
            MVF_HIP_OK(hipFuncSetAttribute((const void*)kern_gen, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            gen_attr = true;
        }
        hipLaunchKernelGGL(kern_gen, dim3(tiles), dim3(kThreads), lds, st, a);
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    // stream-K (double-buffered kernel + a cut last wave) keeps the launches where its cost model says it pays: long-K fp32
    // convs with a mostly empty last tile wave (+2 % on fp32 inference); everything else takes the single-buffer variant
    bool sk_wins = false;
    {
        const int full0 = tiles / slots * slots, tail0 = tiles - full0;
        const float wave_us0 = a.nchunks * (sizeof(ET) == 4 ? 3.9f : 1.0f);
        sk_wins = skh.ws && skh.ws_bytes >= sk_ws_bytes() && tail0 > 0 && (long)tail0 * a.nchunks >= slots &&
                  (1.0f - (float)tail0 / slots) * wave_us0 > sk_min_us();
    }
    if (a.mvf_coef) {                            // MVF fused into this pointwise conv's A loader (inference epilogue: bias + ReLU)
        MVF_REQUIRE(a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.pad_w == 0 && a.split_c == 0 && a.o_s <= 0 && a.bias && a.relu &&
                        !a.res && !a.stats_part && !a.bn_z, MVF_EINVAL, "conv2d_mvf: needs a plain 1x1 stride-1 conv with bias + ReLU");
        MVF_REQUIRE(img_bytes * (span_imgs + 1) < 0x7ffffff0L, MVF_EUNSUPPORTED, "conv2d_mvf: image too large for tile-relative 32-bit addressing");
        fd_make((unsigned)a.mvf_T, a.fd_t_mul, a.fd_t_shr);
        // which kernel carries the fused loader: policy fuse_kernel = 0 register-staged single buffer, 1 / 2 LDS-DMA with 1 / 2 buffers,
        // -1 (default) = by K: one DMA buffer up to 16 chunks, two beyond (the policy of the unfused inference launches)
        static const int fk_env = mvf_policy_int("fuse_kernel", -1);
        const int fk = fk_env >= 0 ? fk_env : (a.nchunks <= 16 ? 1 : 2);
        if (fk == 0) {
            auto k = conv_igemm_lowk_kernel<ET, WM, WN, TM, TN, 4, true, true>;
            constexpr size_t lds_mvf = (size_t)kLowkLds<BM, BN>();
            hipLaunchKernelGGL(k, dim3(tiles), dim3(kThreads), lds_mvf, st, a);
        } else if (fk == 1) {
            auto k = conv_igemm_glds_kernel<ET, WM, WN, TM, TN, 4, 1, true>;
            constexpr size_t lds_mvf = (size_t)kGldsLds<BM, BN, 1>();
            hipLaunchKernelGGL(k, dim3(tiles), dim3(kThreads), lds_mvf, st, a);
        } else {
            auto k = conv_igemm_glds_kernel<ET, WM, WN, TM, TN, 4, 2, true>;
            constexpr size_t lds_mvf = (size_t)kGldsLds<BM, BN, 2>();
            static bool attr_mvf = false;
            if (!attr_mvf) {
                MVF_HIP_OK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mvf));
                attr_mvf = true;
            }
            hipLaunchKernelGGL(k, dim3(tiles), dim3(kThreads), lds_mvf, st, a);
        }
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    if (a.bn_z || a.ap_scale || a.bw_mode) sk_wins = false;   // the BatchNorm-backward / BatchNorm-apply epilogues live in the single-buffer kernels only
    // fp32 on the bf16 matrix cores: the single-buffer register-staged kernel carries it (x3_on)
    const bool x3 = sizeof(ET) == 4 && x3_on(a);
    if (x3) sk_wins = false;
    if ((a.nchunks <= g_lowk_max_chunks && !sk_wins) || a.bn_z || a.ap_scale || a.bw_mode || x3) {      // single LDS buffer: half the LDS, 3-4 workgroups per CU
        constexpr size_t lds_lk = (size_t)kLowkLds<BM, BN>();
        static const int epi_spec = mvf_policy_int("conv_epi", 3);     // A/B switch: bit 0 epilogues, bit 1 pointwise loader
        const bool pw = (epi_spec & 2) && a.KH == 1 && a.KW == 1 && a.pad == 0 && a.pad_w == 0 && a.split_c == 0 && a.dil <= 1;
        const bool contiguous = (epi_spec & 1) && a.o_s <= 0;
        const bool train_like = contiguous && !a.bias && !a.relu;
        const bool infer_like = contiguous && a.bias && a.relu && !a.stats_part;
        const bool bnsum_epi = (epi_spec & 1) && a.bn_z && !a.relu && !a.res;      // contiguous or a scattered parity class ([r5] + an optional bias)
        // long K, wide output: the 256 x 128 LDS-DMA tile (policy conv_big = <min chunks>, 0 = off)
        if (BN == 128 && sizeof(ET) == 2 && g_big2_min > 0 && a.nchunks >= g_big2_min && a.Cout % 256 == 0 && a.o_s <= 0 && !a.ap_scale && !a.bw_mode) {
            const long t2 = (long)((a.M + 255) / 256) * (a.Cout / 256);
            const int cus = slots / 2;
            const long rounds = (t2 + cus - 1) / cus;
            static const bool force2 = mvf_policy_has("conv_big2_force");      // tests: every eligible shape, whatever its tile count
            if (force2 || (t2 >= cus / 2 && (double)t2 / (double)(rounds * cus) >= 0.75)) {       // the last round at least 3/4 full
                int rc;
                // the four-phase ping-pong loop carries the same tile (policy conv_p4=0 -> the two-barrier loop); whole K chunks only
                static const int p4_on = mvf_policy_int("conv_p4", 1);
                if (p4_on && a.Cin % 64 == 0 && (a.split_c % 64) == 0) {
                    if (train_like && a.stats_part && !a.res && !a.bn_z) rc = launch_p4<ET, 1>(st, a);
                    else if (bnsum_epi) rc = launch_p4<ET, 6>(st, a);
                    else if (train_like && !a.stats_part && !a.res) rc = launch_p4<ET, 2>(st, a);
                    else if (train_like && !a.stats_part && a.res) rc = launch_p4<ET, 3>(st, a);
                    else if (infer_like && !a.res) rc = launch_p4<ET, 4>(st, a);
                    else if (infer_like && a.res) rc = launch_p4<ET, 5>(st, a);
                    else rc = launch_p4<ET, 0>(st, a);
                    if (rc != MVF_OK) return rc;
                    MVF_LAUNCH_CHECK();
                    return MVF_OK;
                }
                if (train_like && a.stats_part && !a.res && !a.bn_z) rc = launch_big2<ET, 1>(st, a);
                else if (bnsum_epi) rc = launch_big2<ET, 6>(st, a);
                else if (train_like && !a.stats_part && !a.res) rc = launch_big2<ET, 2>(st, a);
                else if (train_like && !a.stats_part && a.res) rc = launch_big2<ET, 3>(st, a);
                else if (infer_like && !a.res) rc = launch_big2<ET, 4>(st, a);
                else if (infer_like && a.res) rc = launch_big2<ET, 5>(st, a);
                else rc = launch_big2<ET, 0>(st, a);
                if (rc != MVF_OK) return rc;
                MVF_LAUNCH_CHECK();
                return MVF_OK;
            }
        }
        if (BN == 128 && g_big_min > 0 && a.nchunks >= g_big_min && a.Cout >= 128 && a.M >= 256 && !a.ap_scale && !a.bw_mode) {
            int rc;
            if (train_like && a.stats_part && !a.res && !a.bn_z) rc = launch_big<ET, 1>(st, a);
            else if (bnsum_epi) rc = launch_big<ET, 6>(st, a);
            else if (train_like && !a.stats_part && !a.res) rc = launch_big<ET, 2>(st, a);
            else if (train_like && !a.stats_part && a.res) rc = launch_big<ET, 3>(st, a);
            else if (infer_like && !a.res) rc = launch_big<ET, 4>(st, a);
            else if (infer_like && a.res) rc = launch_big<ET, 5>(st, a);
            else rc = launch_big<ET, 0>(st, a);
            if (rc != MVF_OK) return rc;
            MVF_LAUNCH_CHECK();
            return MVF_OK;
        }
        // long K: LDS-DMA staging (policy conv_glds = "<min chunks>[,<buffers 1|2>]", 0 = off)
        // default policy (measured per layer on the R50 train step, bf16): the DMA variant wins 10-15 % from 32 chunks on
        // (K >= 2048: the 3x3 layers of layer3/4) and, for the 128 x 64 tile, from 9 chunks (layer1's 3x3); it loses 10-20 % on
        // the 8-18 chunk pointwise layers, where three register-staged workgroups per CU hide more latency than two DMA ones
        // short K: the single-buffer DMA kernel needs no staging registers -> 4 workgroups per CU instead of 3, which hides more
        // of the per-tile fixed latency chain (kernel arguments -> offsets -> first chunk -> epilogue -> store drain) that
        // dominates these launches (ablation: with loads AND MFMAs removed the conv launches still take 58 % of their time).
        // Not for the BatchNorm-sum data gradient (its epilogue spills at 128 registers) and not for fp32 unless forced.
        // (bf16 inference epilogues -- bias + ReLU [+ residual], half-batch launch chains -- keep winning up to 16 chunks: +1.9 %)
        const int glds1_max = x3 ? 0 : g_glds1_max >= 0 ? g_glds1_max
                              : (sizeof(ET) == 2 ? (infer_like ? 16 : 8) : ((infer_like && g_glds1_f32_infer) ? 8 : 0));
        if (glds1_max > 0 && a.nchunks <= glds1_max && !((bnsum_epi || a.bw_mode == 10) && g_glds1_max < 0)) {      // (the two sum epilogues spill at 128 registers)
            int rc;
            if (a.ap_scale) rc = launch_glds<ET, WM, WN, TM, TN, 8>(1, tiles, st, a);
            else if (a.bw_mode == 9) rc = launch_glds<ET, WM, WN, TM, TN, 9>(1, tiles, st, a);
            else if (a.bw_mode == 10) rc = launch_glds<ET, WM, WN, TM, TN, 10>(1, tiles, st, a);
            else if (train_like && a.stats_part && !a.res && !a.bn_z) rc = launch_glds<ET, WM, WN, TM, TN, 1>(1, tiles, st, a);
            else if (bnsum_epi) rc = launch_glds<ET, WM, WN, TM, TN, 6>(1, tiles, st, a);
            else if (train_like && !a.stats_part && !a.res) rc = launch_glds<ET, WM, WN, TM, TN, 2>(1, tiles, st, a);
            else if (train_like && !a.stats_part && a.res) rc = launch_glds<ET, WM, WN, TM, TN, 3>(1, tiles, st, a);
            else if (train_like && a.stats_part && a.res && a.out_gate && !a.bn_z) rc = launch_glds<ET, WM, WN, TM, TN, 12>(1, tiles, st, a);
            else if (infer_like && !a.res) rc = launch_glds<ET, WM, WN, TM, TN, 4>(1, tiles, st, a);
            else if (infer_like && a.res) rc = launch_glds<ET, WM, WN, TM, TN, 5>(1, tiles, st, a);
            else rc = launch_glds<ET, WM, WN, TM, TN, 0>(1, tiles, st, a);
            if (rc != MVF_OK) return rc;
            MVF_LAUNCH_CHECK();
            return MVF_OK;
        }
        const bool glds_auto = g_glds_min < 0 && sizeof(ET) == 2 && (a.nchunks >= 32 || (BN == 64 && a.nchunks >= 9));
        if (!x3 && ((g_glds_min > 0 && a.nchunks >= g_glds_min) || glds_auto)) {
            int rc;
            if (a.ap_scale) rc = launch_glds<ET, WM, WN, TM, TN, 8>(g_glds_nb, tiles, st, a);
            else if (a.bw_mode == 9) rc = launch_glds<ET, WM, WN, TM, TN, 9>(g_glds_nb, tiles, st, a);
            else if (a.bw_mode == 10) rc = launch_glds<ET, WM, WN, TM, TN, 10>(g_glds_nb, tiles, st, a);
            else if (train_like && a.stats_part && !a.res && !a.bn_z) rc = launch_glds<ET, WM, WN, TM, TN, 1>(g_glds_nb, tiles, st, a);
            else if (bnsum_epi) rc = launch_glds<ET, WM, WN, TM, TN, 6>(g_glds_nb, tiles, st, a);
            else if (train_like && !a.stats_part && !a.res) rc = launch_glds<ET, WM, WN, TM, TN, 2>(g_glds_nb, tiles, st, a);
            else if (train_like && !a.stats_part && a.res) rc = launch_glds<ET, WM, WN, TM, TN, 3>(g_glds_nb, tiles, st, a);
            else if (train_like && a.stats_part && a.res && a.out_gate && !a.bn_z) rc = launch_glds<ET, WM, WN, TM, TN, 12>(g_glds_nb, tiles, st, a);
            else if (infer_like && !a.res) rc = launch_glds<ET, WM, WN, TM, TN, 4>(g_glds_nb, tiles, st, a);
            else if (infer_like && a.res) rc = launch_glds<ET, WM, WN, TM, TN, 5>(g_glds_nb, tiles, st, a);
            else rc = launch_glds<ET, WM, WN, TM, TN, 0>(g_glds_nb, tiles, st, a);
            if (rc != MVF_OK) return rc;
            MVF_LAUNCH_CHECK();
            return MVF_OK;
        }
        if (a.ap_scale) launch_lowk<ET, WM, WN, TM, TN, 8>(pw, tiles, lds_lk, st, a);
        else if (a.bw_mode == 9) launch_lowk<ET, WM, WN, TM, TN, 9>(pw, tiles, lds_lk, st, a);
        else if (a.bw_mode == 10) launch_lowk<ET, WM, WN, TM, TN, 10>(pw, tiles, lds_lk, st, a);
        else if (train_like && a.stats_part && !a.res && !a.bn_z) launch_lowk<ET, WM, WN, TM, TN, 1>(pw, tiles, lds_lk, st, a);
        else if (bnsum_epi) launch_lowk<ET, WM, WN, TM, TN, 6>(pw, tiles, lds_lk, st, a);
        else if (train_like && !a.stats_part && !a.res) launch_lowk<ET, WM, WN, TM, TN, 2>(pw, tiles, lds_lk, st, a);
        else if (train_like && !a.stats_part && a.res) launch_lowk<ET, WM, WN, TM, TN, 3>(pw, tiles, lds_lk, st, a);
        else if (train_like && a.stats_part && a.res && a.out_gate && !a.bn_z) launch_lowk<ET, WM, WN, TM, TN, 12>(pw, tiles, lds_lk, st, a);
        else if (infer_like && !a.res) launch_lowk<ET, WM, WN, TM, TN, 4>(pw, tiles, lds_lk, st, a);
        else if (infer_like && a.res) launch_lowk<ET, WM, WN, TM, TN, 5>(pw, tiles, lds_lk, st, a);
        else launch_lowk<ET, WM, WN, TM, TN, 0>(pw, tiles, lds_lk, st, a);
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    const int full = tiles / slots * slots, tail = tiles - full;
    // stream-K pays when the last wave is substantially empty and there is enough K to cut (fix-up costs ~10 us)
    // one wave of tiles lasts ~nchunks x 3.9 us (fp32 MFMA, 2 workgroups per CU) / ~1 us (bf16); cutting the partial wave
    // saves (1 - tail/slots) of that and costs ~60 us (memset + launch + 64 KB partial round trip per workgroup)
    const float wave_us = a.nchunks * (sizeof(ET) == 4 ? 3.9f : 1.0f);
    const bool use_sk = skh.ws && skh.ws_bytes >= sk_ws_bytes() && tail > 0 && (long)tail * a.nchunks >= slots &&
                        (1.0f - (float)tail / slots) * wave_us > sk_min_us();
    if (!use_sk) {
        const bool pf2 = g_pf2_mode == 2 || (g_pf2_mode == 1 && sizeof(ET) == 2);
        hipLaunchKernelGGL(pf2 ? kern_pf : kern, dim3(tiles), dim3(kThreads), lds, st, a);
        MVF_LAUNCH_CHECK();
        return MVF_OK;
    }
    if (full > 0) {
        hipLaunchKernelGGL(kern, dim3(full), dim3(kThreads), lds, st, a);
        MVF_LAUNCH_CHECK();
    }
    SkArgs sk;
    const long units = (long)tail * a.nchunks;
    sk.G = slots;
    sk.units_base = (int)(units / sk.G);
    sk.units_rem = (int)(units % sk.G);
    sk.tile0 = full;
    sk.ws = (float*)skh.ws;
    sk.flags = (unsigned*)((char*)skh.ws + (size_t)slots * (128 * 128 * sizeof(float)));
    sk.err = sk.flags + slots;
    MVF_HIP_OK(hipMemsetAsync(sk.flags, 0, (size_t)(slots + 1) * sizeof(unsigned), st));
    hipLaunchKernelGGL(kern_sk, dim3(sk.G), dim3(kThreads), lds, st, a, sk);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // namespace

extern "C" {

size_t mvf_conv2d_workspace_bytes(const mvf_conv_desc_t*) { return sk_ws_bytes(); }

int mvf_conv2d_nhwc_fwd(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed,
                        const float* bias, const void* residual, void* y, void* stream) {
    return mvf_conv2d_nhwc_fwd_ws(d, x, x2, w_packed, bias, residual, y, nullptr, 0, stream);
}

struct BnBwdSums {            // optional: the data gradient also accumulates the BatchNorm-backward sums of the BN it feeds
    const void* z;
    const float *mean, *invstd, *scale, *shift;
};
struct MvfFuse {              // optional: MVF-proper applied to channels [0, cs) inside the A loader (see ConvArgs::mvf_coef)
    const float* coef;
    int cs, T, act;
};
struct BnBwdRecompute {       // optional: BatchNorm backward on the recomputed conv output (see ConvArgs::bw_mode); g / sign bits travel as residual / res_mask
    int mode;                 // 9 apply (y = dz), 10 sums (stats_part)
    const float *mean, *invstd, *gamma, *dgamma, *dbeta;
};
struct BnApply {              // optional: the epilogue applies a BatchNorm + residual + ReLU and writes the sign bits (see ConvArgs::ap_scale)
    const float *scale, *shift, *rscale, *rshift;
    unsigned char* bits;
};
static int conv_fwd_impl(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const float* bias,
                         const void* residual, void* y, float* stats_part, const float* stats_shift, void* ws, size_t ws_bytes,
                         void* stream, const unsigned char* res_mask = nullptr, const BnBwdSums* bnb = nullptr, const MvfFuse* mf = nullptr,
                         const BnApply* ap = nullptr, const BnBwdRecompute* bw = nullptr, const unsigned char* out_gate = nullptr);

int mvf_conv2d_nhwc_fwd_mvf(const mvf_conv_desc_t* d, const void* x, const void* w_packed, const float* bias, const float* mvf_coef,
                            int cs, int n_segment, int act, void* y, void* ws, size_t ws_bytes, void* stream) {
    MVF_REQUIRE(d && mvf_coef && bias && cs > 0 && n_segment > 0, MVF_EINVAL, "conv2d_fwd_mvf: bad argument");
    const int ce = d->dtype == MVF_F32 ? 32 : 64;
    MVF_REQUIRE(cs % ce == 0 && cs <= d->cin && d->n % n_segment == 0 && d->ho == d->h && d->wo == d->w && d->x_pix_stride == d->cin, MVF_ESHAPE,
                "conv2d_fwd_mvf: cs=%d must be a multiple of %d (one K chunk), n a multiple of n_segment, output size = input size", cs, ce);
    MVF_REQUIRE((uintptr_t)mvf_coef % 16 == 0, MVF_EINVAL, "conv2d_fwd_mvf: mvf_coef must be 16-byte aligned");
    const MvfFuse mf = {mvf_coef, cs, n_segment, act};
    return conv_fwd_impl(d, x, nullptr, w_packed, bias, nullptr, y, nullptr, nullptr, ws, ws_bytes, stream, nullptr, nullptr, &mf);
}

int mvf_conv2d_nhwc_dgrad_bnsums(const mvf_conv_desc_t* d, const void* dz, const void* w_packed_dgrad, void* y, const void* bn_z,
                                 const float* bn_mean, const float* bn_invstd, const float* bn_scale, const float* bn_shift,
                                 float* sums_part, void* ws, size_t ws_bytes, void* stream) {
    MVF_REQUIRE(d && bn_z && bn_mean && bn_invstd && bn_scale && bn_shift && sums_part, MVF_EINVAL, "conv2d_dgrad_bnsums: NULL argument");
    const BnBwdSums b = {bn_z, bn_mean, bn_invstd, bn_scale, bn_shift};
    return conv_fwd_impl(d, dz, nullptr, w_packed_dgrad, nullptr, nullptr, y, sums_part, nullptr, ws, ws_bytes, stream, nullptr, &b);
}

int mvf_conv2d_nhwc_fwd_bnapply(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const float* bn_scale,
                                const float* bn_shift, const void* residual, const float* res_scale, const float* res_shift, void* out,
                                unsigned char* sign_bits, void* ws, size_t ws_bytes, void* stream) {
    MVF_REQUIRE(d && bn_scale && bn_shift && residual && sign_bits, MVF_EINVAL, "conv2d_fwd_bnapply: NULL argument");
    MVF_REQUIRE((res_scale == nullptr) == (res_shift == nullptr), MVF_EINVAL, "conv2d_fwd_bnapply: res_scale / res_shift must come together");
    MVF_REQUIRE(d->in_dil <= 1 && !d->relu && d->cout % 4 == 0, MVF_EINVAL, "conv2d_fwd_bnapply: a forward launch (in_dil <= 1), relu = 0 in the descriptor");
    MVF_REQUIRE(((uintptr_t)bn_scale | (uintptr_t)bn_shift | (uintptr_t)(res_scale ? res_scale : bn_scale) | (uintptr_t)(res_shift ? res_shift : bn_shift)) % 16 == 0,
                MVF_EINVAL, "conv2d_fwd_bnapply: coefficient vectors must be 16-byte aligned");
    const BnApply ap = {bn_scale, bn_shift, res_scale, res_shift, sign_bits};
    return conv_fwd_impl(d, x, x2, w_packed, nullptr, residual, out, nullptr, nullptr, ws, ws_bytes, stream, nullptr, nullptr, nullptr, &ap);
}

int mvf_conv2d_nhwc_fwd_bnbwd_sums(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const void* g,
                                   const unsigned char* sign_bits, const float* bn_mean, const float* bn_invstd, float* sums_part, void* ws,
                                   size_t ws_bytes, void* stream) {
    MVF_REQUIRE(d && g && sign_bits && bn_mean && bn_invstd && sums_part, MVF_EINVAL, "conv2d_fwd_bnbwd_sums: NULL argument");
    MVF_REQUIRE(d->in_dil <= 1 && !d->relu, MVF_EINVAL, "conv2d_fwd_bnbwd_sums: a forward launch (in_dil <= 1), relu = 0");
    MVF_REQUIRE(((uintptr_t)bn_mean | (uintptr_t)bn_invstd) % 16 == 0, MVF_EINVAL, "conv2d_fwd_bnbwd_sums: coefficient vectors must be 16-byte aligned");
    const BnBwdRecompute bw = {10, bn_mean, bn_invstd, nullptr, nullptr, nullptr};
    return conv_fwd_impl(d, x, x2, w_packed, nullptr, g, nullptr, sums_part, nullptr, ws, ws_bytes, stream, sign_bits, nullptr, nullptr, nullptr, &bw);
}

int mvf_conv2d_nhwc_fwd_bnbwd_apply(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const void* g,
                                    const unsigned char* sign_bits, const float* bn_gamma, const float* bn_mean, const float* bn_invstd,
                                    const float* dgamma, const float* dbeta, void* dz, void* ws, size_t ws_bytes, void* stream) {
    MVF_REQUIRE(d && g && sign_bits && bn_gamma && bn_mean && bn_invstd && dgamma && dbeta && dz, MVF_EINVAL, "conv2d_fwd_bnbwd_apply: NULL argument");
    MVF_REQUIRE(d->in_dil <= 1 && !d->relu, MVF_EINVAL, "conv2d_fwd_bnbwd_apply: a forward launch (in_dil <= 1), relu = 0");
    MVF_REQUIRE(((uintptr_t)bn_gamma | (uintptr_t)bn_mean | (uintptr_t)bn_invstd | (uintptr_t)dgamma | (uintptr_t)dbeta) % 16 == 0, MVF_EINVAL,
                "conv2d_fwd_bnbwd_apply: coefficient vectors must be 16-byte aligned");
    const BnBwdRecompute bw = {9, bn_mean, bn_invstd, bn_gamma, dgamma, dbeta};
    return conv_fwd_impl(d, x, x2, w_packed, nullptr, g, dz, nullptr, nullptr, ws, ws_bytes, stream, sign_bits, nullptr, nullptr, nullptr, &bw);
}

int mvf_conv2d_nhwc_fwd_resmask(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const float* bias,
                                const void* residual, const unsigned char* res_sign_bits, void* y, void* ws, size_t ws_bytes,
                                void* stream) {
    MVF_REQUIRE(!res_sign_bits || (residual && d && d->in_dil <= 1), MVF_EINVAL, "conv2d_resmask: needs a residual and a stride-1 / non-dilated launch");
    return conv_fwd_impl(d, x, x2, w_packed, bias, residual, y, nullptr, nullptr, ws, ws_bytes, stream, res_sign_bits);
}

int mvf_conv2d_nhwc_fwd_resmask_gate(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const float* bias,
                                     const void* residual, const unsigned char* res_sign_bits, const unsigned char* out_gate_bits, void* y,
                                     void* ws, size_t ws_bytes, void* stream) {
    MVF_REQUIRE(d && residual && out_gate_bits && d->in_dil <= 1 && !d->relu, MVF_EINVAL, "conv2d_resmask_gate: needs a residual, the gate bits and a stride-1 / non-dilated launch without ReLU");
    MVF_REQUIRE(((uintptr_t)out_gate_bits | (uintptr_t)(res_sign_bits ? res_sign_bits : out_gate_bits)) % 16 == 0 || d->cout % 64 != 0, MVF_EINVAL,
                "conv2d_resmask_gate: gate byte rows must be 16-byte aligned");
    return conv_fwd_impl(d, x, x2, w_packed, bias, residual, y, nullptr, nullptr, ws, ws_bytes, stream, res_sign_bits, nullptr, nullptr, nullptr, nullptr, out_gate_bits);
}

int mvf_conv2d_nhwc_fwd_resmask_gate_colsums(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const void* residual,
                                             const unsigned char* res_sign_bits, const unsigned char* out_gate_bits, void* y, float* sums_part, void* ws, size_t ws_bytes,
                                             void* stream) {
    MVF_REQUIRE(d && residual && out_gate_bits && sums_part && d->in_dil <= 1 && !d->relu, MVF_EINVAL,
                "conv2d_resmask_gate_colsums: needs a residual, the gate bits, a partial buffer and a stride-1 launch without ReLU");
    MVF_REQUIRE(((uintptr_t)out_gate_bits | (uintptr_t)(res_sign_bits ? res_sign_bits : out_gate_bits)) % 16 == 0 || d->cout % 64 != 0, MVF_EINVAL,
                "conv2d_resmask_gate_colsums: gate byte rows must be 16-byte aligned");      // (the epilogue stages both byte planes with 16-byte buffer loads, as _gate does)
    // [.][.][0] = sum gm, [.][.][1] = sum gm^2 of the stored values (the dz3-free backward takes dgamma from its weight-gradient GEMM: mvf_bn_bwd_dzfree_sums)
    return conv_fwd_impl(d, x, x2, w_packed, nullptr, residual, y, sums_part, nullptr, ws, ws_bytes, stream, res_sign_bits, nullptr, nullptr, nullptr, nullptr, out_gate_bits);
}

int mvf_conv2d_nhwc_dgrad_bnsums_split(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const float* bias, void* y,
                                       const void* bn_z, const float* bn_mean, const float* bn_invstd, const float* bn_scale, const float* bn_shift,
                                       float* sums_part, void* ws, size_t ws_bytes, void* stream) {
    MVF_REQUIRE(d && bn_z && bn_mean && bn_invstd && bn_scale && bn_shift && sums_part, MVF_EINVAL, "conv2d_dgrad_bnsums_split: NULL argument");
    MVF_REQUIRE(d->in_dil <= 1 && !d->relu && d->kh == 1 && d->kw == 1 && d->stride == 1, MVF_EINVAL, "conv2d_dgrad_bnsums_split: a pointwise stride-1 launch");
    const BnBwdSums b = {bn_z, bn_mean, bn_invstd, bn_scale, bn_shift};
    return conv_fwd_impl(d, x, x2, w_packed, bias, nullptr, y, sums_part, nullptr, ws, ws_bytes, stream, nullptr, &b);
}

int mvf_conv2d_nhwc_fwd_ws(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed,
                           const float* bias, const void* residual, void* y, void* ws, size_t ws_bytes, void* stream) {
    return conv_fwd_impl(d, x, x2, w_packed, bias, residual, y, nullptr, nullptr, ws, ws_bytes, stream);
}

int mvf_conv2d_stats_rows(const mvf_conv_desc_t* d) {
    if (!d || d->n <= 0 || d->ho <= 0 || d->wo <= 0) return 0;
    if (d->in_dil > 1) {                 // strided data gradient: one run of 128-row partials per parity class, back to back
        long rows = 0;
        for (int ph = 0; ph < d->in_dil; ++ph)
            for (int pw = 0; pw < d->in_dil; ++pw) {
                const long ho = (d->ho - ph + d->in_dil - 1) / d->in_dil, wo = (d->wo - pw + d->in_dil - 1) / d->in_dil;
                if (ho > 0 && wo > 0) rows += ((long)d->n * ho * wo + kBM - 1) / kBM;
            }
        return (int)rows;
    }
    return (int)(((long)d->n * d->ho * d->wo + kBM - 1) / kBM);
}

int mvf_conv2d_nhwc_fwd_stats(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, void* y,
                              float* stats_part, const float* stats_shift, void* ws, size_t ws_bytes, void* stream) {
    MVF_REQUIRE(stats_part, MVF_EINVAL, "conv2d_fwd_stats: stats_part is NULL");
    MVF_REQUIRE(!d || d->in_dil <= 1, MVF_EINVAL, "conv2d_fwd_stats: not for data-gradient launches");
    return conv_fwd_impl(d, x, x2, w_packed, nullptr, nullptr, y, stats_part, stats_shift, ws, ws_bytes, stream);
}

static int conv_fwd_impl(const mvf_conv_desc_t* d, const void* x, const void* x2, const void* w_packed, const float* bias,
                         const void* residual, void* y, float* stats_part, const float* stats_shift, void* ws, size_t ws_bytes,
                         void* stream, const unsigned char* res_mask, const BnBwdSums* bnb, const MvfFuse* mf, const BnApply* ap, const BnBwdRecompute* bw,
                         const unsigned char* out_gate) {
    MVF_REQUIRE(d && x && w_packed && (y || (stats_part && !bnb)), MVF_EINVAL, "conv2d: NULL argument");      // (y may be NULL for a statistics-only pass)
    MVF_REQUIRE(d->dtype == MVF_F32 || d->dtype == MVF_BF16, MVF_EINVAL, "conv2d: bad dtype %d", d->dtype);
    MVF_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0 && d->kh > 0 && d->kw > 0 &&
                    d->stride > 0 && d->pad >= 0, MVF_ESHAPE, "conv2d: bad dims");
    const int dil = d->in_dil > 1 ? d->in_dil : 1;
    MVF_REQUIRE(dil == 1 || d->stride == 1, MVF_EINVAL, "conv2d: in_dil > 1 requires stride 1");
    const int ho = ((d->h - 1) * dil + 1 + 2 * d->pad - d->kh) / d->stride + 1 + (dil - 1), wo = ((d->w - 1) * dil + 1 + 2 * d->pad - d->kw) / d->stride + 1 + (dil - 1);
    MVF_REQUIRE(d->ho > 0 && d->wo > 0 && d->ho <= ho && d->wo <= wo, MVF_ESHAPE,
                "conv2d: ho,wo = %d,%d inconsistent with input %dx%d k%dx%d s%d p%d (max %d,%d)", d->ho, d->wo, d->h,
                d->w, d->kh, d->kw, d->stride, d->pad, ho, wo);
    const int ue = d->dtype == MVF_F32 ? 4 : 8, ce = ue * 8, esz = d->dtype == MVF_F32 ? 4 : 2;
    // every row of a K chunk must start on a 16-byte boundary: pixel pitch a multiple of the unit, or (stem view of the
    // padded NHWC4 input) an even pixel pitch walked two pixels at a time over an even-width image
    const bool pitch_ok = d->x_pix_stride > 0 && (d->x_pix_stride % ue == 0 ||
                          ((d->x_pix_stride * 2) % ue == 0 && d->stride % 2 == 0 && d->kw == 1 && d->pad == 0 && d->w % 2 == 0));
    MVF_REQUIRE(d->cin % ue == 0 && pitch_ok, MVF_ESHAPE,
                "conv2d: cin=%d / x_pix_stride=%d do not give 16-byte aligned K chunks (unit = %d elements)", d->cin, d->x_pix_stride, ue);
    MVF_REQUIRE(((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)(x2 ? x2 : x)) % 16 == 0, MVF_EINVAL, "conv2d: pointers must be 16-byte aligned");
    if (d->split_c) {
        MVF_REQUIRE(x2 && d->kh == 1 && d->kw == 1 && d->split_c % ce == 0 && d->split_c <= d->cin && d->x2_pix_stride % ue == 0 && d->x2_pix_stride >= d->split_c,
                    MVF_EINVAL, "conv2d: split_c=%d needs x2, a 1x1 kernel and a multiple of %d channels", d->split_c, ce);
    }
    MVF_REQUIRE(d->cout % 4 == 0 && ((uintptr_t)y | (uintptr_t)(residual ? residual : y) | (uintptr_t)(bias ? (const void*)bias : y)) % 16 == 0, MVF_ESHAPE,
                "conv2d: cout=%d must be a multiple of 4 and y/residual/bias 16-byte aligned (row-wise vector epilogue)", d->cout);
    MVF_REQUIRE((long)d->n * d->ho * d->wo < (1L << 31) && (long)d->n * d->h * d->w < (1L << 31), MVF_ESHAPE, "conv2d: too many pixels");
    ConvArgs a = {};
    a.x = (const char*)x; a.x2 = (const char*)x2; a.w = (const char*)w_packed; a.res = (const char*)residual;
    a.bias = bias; a.y = (char*)y;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.KH = d->kh; a.KW = d->kw;
    a.stride = d->stride; a.pad = d->pad; a.Ho = d->ho; a.Wo = d->wo; a.xps = d->x_pix_stride;
    a.split_c = d->split_c; a.x2ps = d->x2_pix_stride; a.relu = d->relu; a.dil = dil;
    a.res_c0 = d->res_c0 > 0 ? d->res_c0 : 0;
    a.res_mask = res_mask;
    a.out_gate = out_gate;
    MVF_REQUIRE(d->x_c0 == 0 || (d->split_c > 0 && d->x_c0 > 0 && d->x_c0 <= d->split_c && d->x_c0 % ue == 0 && d->x_pix_stride >= d->cin - d->x_c0), MVF_EINVAL,
                "conv2d: x_c0=%d needs a split operand (split_c=%d), 0 < x_c0 <= split_c, a multiple of %d, and x rows of >= cin - x_c0 channels", d->x_c0, d->split_c, ue);
    a.x_c0 = d->x_c0;
    a.mask_lds = 1;
    static const int prio_on = mvf_policy_int("conv_prio", 0);
#ifdef MVF_CONV_ABLATE
    a.prio = prio_on;                 // ablation builds: bit 0 priority, bits 1-5 remove loads / MFMAs / epilogue parts (wrong results)
#else
    a.prio = prio_on & 1;             // product builds: only the wave-priority experiment
#endif
    a.M = d->n * d->ho * d->wo;
    a.cpt = (d->cin + ce - 1) / ce;
    a.nchunks = d->kh * d->kw * a.cpt;
    a.wK = (long)d->kh * d->kw * d->cin;
    (void)esz;
    a.pad_w = d->pad; a.w_kh0 = 0; a.w_kw0 = 0; a.w_ts = 1; a.w_kwfull = d->kw;
    a.stats_part = stats_part; a.stats_shift = stats_shift;
    a.stats_rows = stats_part ? mvf_conv2d_stats_rows(d) : 0;
    if (mf) {
        a.mvf_coef = mf->coef; a.mvf_cs = mf->cs; a.mvf_T = mf->T; a.mvf_act = mf->act;
    }
    if (bw) {
        a.bw_mode = bw->mode; a.bn_mean = bw->mean; a.bn_invstd = bw->invstd; a.bw_gamma = bw->gamma; a.bw_dgamma = bw->dgamma; a.bw_dbeta = bw->dbeta;
    }
    if (ap) {
        a.ap_scale = ap->scale; a.ap_shift = ap->shift; a.ap_rscale = ap->rscale; a.ap_rshift = ap->rshift; a.ap_bits = ap->bits;
    }
    if (bnb) {
        a.bn_z = (const char*)bnb->z; a.bn_mean = bnb->mean; a.bn_invstd = bnb->invstd; a.bn_scale = bnb->scale; a.bn_shift = bnb->shift;
    }
    hipStream_t st = (hipStream_t)stream;
    SkHost skh = {ws, ws_bytes};
    auto launch = [&](const ConvArgs& aa) -> int {
        const bool narrow = d->cout <= 64;
        if (d->dtype == MVF_F32) {
            if (narrow) return launch_conv<float, 4, 1, 1, 2>(aa, skh, st);
            return launch_conv<float, 2, 2, 2, 2>(aa, skh, st);
        }
        if (narrow) return launch_conv<bf16_t, 4, 1, 1, 2>(aa, skh, st);
        return launch_conv<bf16_t, 2, 2, 2, 2>(aa, skh, st);
    };
    // the stem (7 x 1 taps over the 32-"channel" view of the padded NHWC4 operand, stride 2): its own direct kernel (stem_direct.hip)
    if (dil == 1 && d->dtype == MVF_BF16 && d->kh == 7 && d->kw == 1 && d->cin == 32 && d->x_pix_stride == 4 && d->stride == 2 && d->pad == 0 &&
        d->cout == 64 && !d->split_c && !residual && !res_mask && !bnb && !mf && !ap && !bw && y && d->res_c0 <= 0) {
        const int epi = (stats_part && !bias && !d->relu) ? 1 : ((bias && d->relu && !stats_part) ? 4 : 0);
        if (epi) {
            StemDirectArgs s = {x, w_packed, y, bias, stats_part, stats_shift, a.stats_rows, epi, d->n, d->h, d->w, d->ho, d->wo, a.wK};
            const int rc = mvf_internal::stem_direct_launch(s, st);
            if (rc != -1) return rc;
        }
    }
    // the two sum passes of a z3-free bottleneck over a 64-channel input (layer1's conv3: nothing is stored): their own kernel (pw_sums.hip)
    if (dil == 1 && d->dtype == MVF_BF16 && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->cin == 64 && !d->split_c && !y && stats_part &&
        !bias && !d->relu && !bnb && !mf && !ap && d->ho == d->h && d->wo == d->w) {
        const bool sums10 = bw && bw->mode == 10 && residual && res_mask;
        if (sums10 || (!bw && !residual && !res_mask)) {
            PwSumsArgs s = {x, w_packed, sums10 ? residual : nullptr, sums10 ? res_mask : nullptr, sums10 ? bw->mean : stats_shift, sums10 ? bw->invstd : nullptr,
                            stats_part, a.stats_rows, sums10 ? 1 : 0, a.M, d->cout, 64, d->x_pix_stride};
            const int rc = mvf_internal::pw_sums_launch(s, st);
            if (rc != -1) return rc;
        }
    }
    // layer1's 3x3 (64 -> 64 channels, stride 1) and its data gradient: the direct kernel with register-resident weights (conv3x3_c64.hip)
    if (dil == 1 && d->dtype == MVF_BF16 && d->kh == 3 && d->kw == 3 && d->cin == 64 && d->cout == 64 && d->stride == 1 && d->pad == 1 &&
        d->ho == d->h && d->wo == d->w && !d->split_c && !residual && !res_mask && !mf && !ap && !bw && y && d->res_c0 <= 0) {
        int epi = 0;
        if (bnb) epi = (!bias && !d->relu && stats_part) ? 6 : 0;
        else if (stats_part) epi = (!bias && !d->relu) ? 1 : 0;
        else if (bias && d->relu) epi = 4;
        else if (!bias && !d->relu) epi = 2;
        if (epi) {
            Conv3x3C64Args s = {x, w_packed, y, bias, stats_part, stats_shift, bnb ? bnb->z : nullptr, bnb ? bnb->mean : nullptr, bnb ? bnb->invstd : nullptr,
                                bnb ? bnb->scale : nullptr, bnb ? bnb->shift : nullptr, a.stats_rows, epi, d->n, d->h, d->w, d->x_pix_stride, a.wK};
            const int rc = mvf_internal::conv3x3_c64_launch(s, st);
            if (rc != -1) return rc;
        }
    }
    if (dil == 1) return launch(a);
    // Data gradient of a stride-s conv: output pixel (ih, iw) only sees taps with (ih - pad + kh) % s == 0, so the s*s
    // parity classes (ih % s, iw % s) are independent plain stride-1 convs over the gradient map with ceil(k/s)-tap
    // kernels: s*s launches doing k*k/(s*s) of the zero-upsampled work (9 taps instead of 36 for a 3x3 stride-2 conv).
    const int s_ = dil;
    long part_rows = 0;                  // statistics partials of the classes already launched (mvf_conv2d_stats_rows order)
    for (int ph = 0; ph < s_; ++ph)
        for (int pw = 0; pw < s_; ++pw) {
            ConvArgs c = a;
            c.dil = 1;
            const int kh0 = ((d->pad - ph) % s_ + s_) % s_, kw0 = ((d->pad - pw) % s_ + s_) % s_;
            c.KH = kh0 < d->kh ? (d->kh - kh0 + s_ - 1) / s_ : 0;
            c.KW = kw0 < d->kw ? (d->kw - kw0 + s_ - 1) / s_ : 0;
            if (c.KH == 0 || c.KW == 0) c.KH = c.KW = 0;
            c.w_kh0 = kh0; c.w_kw0 = kw0; c.w_ts = s_; c.w_kwfull = d->kw;
            c.pad = -((ph - d->pad + kh0) / s_);          // exact: (ph - pad + kh0) is a multiple of s
            c.pad_w = -((pw - d->pad + kw0) / s_);
            c.stride = 1;
            c.Ho = (d->ho - ph + s_ - 1) / s_;
            c.Wo = (d->wo - pw + s_ - 1) / s_;
            if (c.Ho <= 0 || c.Wo <= 0) continue;
            c.M = d->n * c.Ho * c.Wo;
            c.nchunks = c.KH * c.KW * c.cpt;
            c.o_s = s_; c.o_ph = ph; c.o_pw = pw; c.o_hfull = d->ho; c.o_wfull = d->wo;
            if (a.stats_part) c.stats_part = a.stats_part + part_rows * 2;          // channel-major: this class's run of rows inside every channel
            part_rows += ((long)c.M + kBM - 1) / kBM;
            int rc = launch(c);
            if (rc) return rc;
        }
    return MVF_OK;
}

int mvf_pack_conv_weight(const float* w_oihw, int cout, int cin, int kh, int kw, int kw_pad, int cin_pad,
                         const float* scale, void* w_packed, int dtype, void* stream) {
    MVF_REQUIRE(w_oihw && w_packed && cout > 0 && cin > 0 && kh > 0 && kw > 0 && kw_pad >= kw && cin_pad >= cin,
                MVF_EINVAL, "pack_conv_weight: bad argument");
    const long total = (long)cout * kh * kw_pad * cin_pad;
    const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, kh, kw, kw_pad, cin_pad, scale, (float*)w_packed);
    else if (dtype == MVF_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, kh, kw, kw_pad, cin_pad, scale, (bf16_t*)w_packed);
    else {
        mvf_set_error("pack_conv_weight: bad dtype %d", dtype);
        return MVF_EINVAL;
    }
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

int mvf_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int c,
                float* scale, float* shift, void* stream) {
    MVF_REQUIRE(gamma && beta && mean && var && scale && shift && c > 0, MVF_EINVAL, "bn_fold: bad argument");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps, c, scale, shift);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
}

}  // extern "C"
