// Probe: the 8-phase 256 x 256 LDS-DMA GEMM main loop (cdna_hip_programming.md section 5, "The 256^2 8-phase template") written
// against THIS repo's operand conventions (128-byte K chunks, source-side XOR swizzle, 32x32x16 MFMAs with swapped operands,
// per-row buffer offsets with out-of-range = zero fill) before it goes into conv_tile.  Standalone: C[M][N] = A[M][K] * B[N][K]^T.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm8p_probe gemm8p_probe.hip && ./gemm8p_probe [M N K]
//
// VAR 0: two-barrier loop (chunk k+1 in flight under chunk k, vmcnt(0) + barrier per chunk) = conv_igemm_big2_kernel's structure.
// VAR 1: 8 phases per two chunks: half-tile DMAs 7 phases ahead, counted vmcnt(6) once per chunk, two wave groups one barrier apart.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include <algorithm>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

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
__device__ __forceinline__ void glds16s(const i32x4 rs, const unsigned lds_dst, const unsigned voff, const unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// register barrier: the accumulators are opaque here, so no MFMA on them is scheduled across this point (the MFMA builtin touches
// no memory, so neither s_barrier nor a "memory" clobber orders it; sched_barrier(0) only binds the last scheduler, not the sinking passes)
#define PIN(x, y) asm volatile("" : "+v"(x), "+v"(y))

constexpr int kTileBytes = 256 * 128;        // one operand tile of one chunk
constexpr int kBufBytes = 2 * kTileBytes;    // A + B
constexpr unsigned kOOB = 0x80000000u;

template <int VAR, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, float* __restrict__ C,
                                                   int M, int N, int K, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tn_i = tile % tiles_n, tm_i = tile / tiles_n;
    const int m0 = tm_i * 256, n0 = tn_i * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                 // 2 x 4 waves, 128 rows x 64 columns each
    const int nchunks = K / 64;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const i32x4 rs_a = rsrc_words(A, (unsigned)((long)M * K * 2));
    const i32x4 rs_b = rsrc_words(B, (unsigned)((long)N * K * 2));
    // loader rows: half-tile kinds in staging order {B0, A0, B1, A1}; a kind is 128 rows = 2 wave instructions of 8 rows per wave
    const int lrow8 = lane >> 3, pos = lane & 7;
    unsigned a_off[2][2], b_off[2][2], a_lds[2][2], b_lds[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int idx0 = (wave * 2 + n) * 8;
            const int arow0 = (idx0 >> 6) * 128 + h * 64 + (idx0 & 63), arow = arow0 + lrow8;
            const int brow0 = (idx0 >> 5) * 64 + h * 32 + (idx0 & 31), brow = brow0 + lrow8;
            a_off[h][n] = (m0 + arow) < M ? (unsigned)((long)(m0 + arow) * K * 2 + ((pos ^ ((arow >> 1) & 7)) << 4)) : kOOB;
            b_off[h][n] = (n0 + brow) < N ? (unsigned)((long)(n0 + brow) * K * 2 + ((pos ^ ((brow >> 1) & 7)) << 4)) : kOOB;
            a_lds[h][n] = lds0 + arow0 * 128;
            b_lds[h][n] = lds0 + kTileBytes + brow0 * 128;
        }
    // one half-tile (2 DMA instructions per thread): kind k of chunk c into buffer c & 1; past the last chunk the offsets are pushed out of range
    auto stage = [&](int kind, int c, bool in_loop = false) {
        if ((ABL & 1) && in_loop) return;
        const unsigned buf = (unsigned)(c & 1) * kBufBytes;
        const unsigned koff = c < nchunks ? (unsigned)c * 128u : kOOB;
        if (VAR != 0 && !(ABL & 8)) {
            // the per-lane offset never changes; the chunk is a scalar offset (not range-checked); past the end the last chunk is re-read
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(c < nchunks ? c : nchunks - 1) * 128u;
            if (kind & 1) {
                glds16s(rs_a, a_lds[kind >> 1][0] + buf, a_off[kind >> 1][0], so);
                glds16s(rs_a, a_lds[kind >> 1][1] + buf, a_off[kind >> 1][1], so);
            } else {
                glds16s(rs_b, b_lds[kind >> 1][0] + buf, b_off[kind >> 1][0], so);
                glds16s(rs_b, b_lds[kind >> 1][1] + buf, b_off[kind >> 1][1], so);
            }
            return;
        }
        auto off = [&](unsigned o) { return ((o | koff) & kOOB) ? kOOB : o + koff; };
        if (kind & 1) {
            glds16(rs_a, a_lds[kind >> 1][0] + buf, off(a_off[kind >> 1][0]));
            glds16(rs_a, a_lds[kind >> 1][1] + buf, off(a_off[kind >> 1][1]));
        } else {
            glds16(rs_b, b_lds[kind >> 1][0] + buf, off(b_off[kind >> 1][0]));
            glds16(rs_b, b_lds[kind >> 1][1] + buf, off(b_off[kind >> 1][1]));
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // operand fetch: row = lane & 31 of a 32-row fragment, 16-byte unit (2 ks + lane >> 5) ^ ((row >> 1) & 7)
    const int fsw = (lane >> 1) & 7, hh = lane >> 5;
    const char* a_base = smem + (wm * 128 + (lane & 31)) * 128;
    const char* b_base = smem + kTileBytes + (wn * 64 + (lane & 31)) * 128;
    int ko[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ko[ks] = ((2 * ks + hh) ^ fsw) << 4;
    auto fetch_a = [&](int buf, int i, uint4 (&f)[4]) {
        if (ABL & 2) { for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(f[ks].x)); return; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const uint4*>(a_base + buf * kBufBytes + i * 4096 + ko[ks]);
    };
    auto fetch_b = [&](int buf, int j, uint4 (&f)[4]) {
        if (ABL & 2) { for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(f[ks].x)); return; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const uint4*>(b_base + buf * kBufBytes + j * 4096 + ko[ks]);
    };
    auto mma = [&](f32x16& d, const uint4& fa, const uint4& fb) {
        if (ABL & 4) { asm volatile("" : "+v"(d) : "v"(fa.x), "v"(fb.x)); return; }
        bf16x8 av, bv;
        __builtin_memcpy(&av, &fa, 16);
        __builtin_memcpy(&bv, &fb, 16);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, d, 0, 0, 0);
    };

    if constexpr (VAR == 0) {
        // two barriers per chunk: wait for chunk c, barrier, issue chunk c + 1, multiply chunk c
        for (int k = 0; k < 4; ++k) stage(k, 0);
        for (int c = 0; c < nchunks; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int k = 0; k < 4; ++k) stage(k, c + 1);
            const int buf = c & 1;
            uint4 fa[4][4], fb[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j) fetch_b(buf, j, fb[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i) fetch_a(buf, i, fa[i]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma(acc[i][j], fa[i][ks], fb[j][ks]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        if constexpr (VAR == 3) {
        // ---- 4 phases per two chunks (16 MFMAs each), two wave groups one barrier apart.  Phase X(c): read A{0,1}(c) + B0(c), stage B0(c+1) then
        // A1(c+1); phase Y(c): read A{2,3}(c) + B1(c+1), stage B1(c+2) then A0(c+2).  Every phase waits vmcnt(6) (three half-tiles stay in flight:
        // activations get two phases of lead, weights one) and retires its 8 A reads before its first barrier (lgkmcnt(4)) so that the A slot can be
        // restaged one phase later.
        stage(2, 0); stage(1, 0); stage(0, 0); stage(3, 0); stage(2, 1); stage(1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        uint4 fa[2][4] = {}, fb0[4] = {}, fb1a[4] = {}, fb1b[4] = {};
        unsigned tsv[2][4][4] = {};
        unsigned long tq = 0;
        int tc = 0;
        (void)tq; (void)tc;
        fetch_b(0, 1, fb1a);
        if (wm == 1) __builtin_amdgcn_s_barrier();
#define TS3(p, k) do { if constexpr ((ABL & 32) != 0) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tq) :: "memory"); if (tc == 8) tsv[0][p][k] = (unsigned)tq; if (tc == 9) tsv[1][p][k] = (unsigned)tq; } } while (0)
#define PHASE_X(c_, buf_, cur_)                                                                  \
        tc = (c_); TS3(0, 0);                                                                    \
        fetch_a(buf_, 0, fa[0]); fetch_a(buf_, 1, fa[1]);                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        fetch_b(buf_, 0, fb0);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        stage(0, (c_) + 1, true); stage(3, (c_) + 1, true);                                      \
        TS3(0, 1);                                                                               \
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(4)" ::: "memory");                              \
        __builtin_amdgcn_s_barrier();                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
        TS3(0, 2);                                                                               \
        PIN(acc[0][0], acc[1][0]); PIN(acc[0][1], acc[1][1]);                                    \
        if (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);                                          \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                       \
            mma(acc[0][0], fa[0][ks], fb0[ks]); mma(acc[1][0], fa[1][ks], fb0[ks]);              \
            mma(acc[0][1], fa[0][ks], cur_[ks]); mma(acc[1][1], fa[1][ks], cur_[ks]);            \
        }                                                                                        \
        if (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);                                          \
        PIN(acc[0][0], acc[1][0]); PIN(acc[0][1], acc[1][1]);                                    \
        TS3(0, 3);                                                                               \
        __builtin_amdgcn_s_barrier();
#define PHASE_Y(c_, buf_, cur_, nxt_)                                                            \
        TS3(1, 0);                                                                               \
        fetch_a(buf_, 2, fa[0]); fetch_a(buf_, 3, fa[1]);                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        fetch_b((buf_) ^ 1, 1, nxt_);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        stage(2, (c_) + 2, true); stage(1, (c_) + 2, true);                                      \
        TS3(1, 1);                                                                               \
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(4)" ::: "memory");                              \
        __builtin_amdgcn_s_barrier();                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
        TS3(1, 2);                                                                               \
        PIN(acc[2][0], acc[3][0]); PIN(acc[2][1], acc[3][1]);                                    \
        if (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);                                          \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                       \
            mma(acc[2][0], fa[0][ks], fb0[ks]); mma(acc[3][0], fa[1][ks], fb0[ks]);              \
            mma(acc[2][1], fa[0][ks], cur_[ks]); mma(acc[3][1], fa[1][ks], cur_[ks]);            \
        }                                                                                        \
        if (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);                                          \
        PIN(acc[2][0], acc[3][0]); PIN(acc[2][1], acc[3][1]);                                    \
        TS3(1, 3);                                                                               \
        __builtin_amdgcn_s_barrier();
        unsigned long w0 = 0, w1 = 0, r0 = 0, r1 = 0;
        if constexpr ((ABL & 64) != 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(w0), "=s"(r0) :: "memory");
        for (int c = 0; c < nchunks; c += 2) {
            PHASE_X(c, 0, fb1a)
            PHASE_Y(c, 0, fb1a, fb1b)
            if (c + 1 < nchunks) {
                PHASE_X(c + 1, 1, fb1b)
                PHASE_Y(c + 1, 1, fb1b, fb1a)
            }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
        if constexpr ((ABL & 64) != 0) {
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(w1), "=s"(r1) :: "memory");
            if (lane == 0 && wave == 0) {
                unsigned long* dbg = reinterpret_cast<unsigned long*>(C) + blockIdx.x * 2;
                dbg[0] = w1 - w0;
                dbg[1] = r1 - r0;
            }
            return;
        }
        if constexpr ((ABL & 32) != 0) {
            if (lane == 0 && blockIdx.x == 0 && (wave == 0 || wave == 4)) {
                unsigned* dbg = reinterpret_cast<unsigned*>(C) + (wave >> 2) * 32;
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) dbg[(cc * 4 + pp) * 4 + kk] = tsv[cc][pp][kk];
            }
            return;
        }
        } else
        if constexpr (ABL & 32) {
        unsigned tsv[2][4][4] = {};
        unsigned long tq = 0;
#define TS(p, k) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tq) :: "memory"); if (c == 8) tsv[0][p][k] = (unsigned)tq; if (c == 9) tsv[1][p][k] = (unsigned)tq; } while (0)
        // prologue: chunk 0 whole, chunk 1's first three half-tiles; chunk 0 has landed after the counted wait
        for (int k = 0; k < 4; ++k) stage(k, 0);
        for (int k = 0; k < 3; ++k) stage(k, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();           // the second wave group runs one barrier behind the first
        uint4 fb0[4] = {}, fb1[4] = {}, fa0[2][4] = {}, fa1[2][4] = {};
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            // ---- phase 1: B0 + A0 -> C[0..1][0]; stage A1 of chunk c + 1
            TS(0, 0);
            fetch_b(buf, 0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            fetch_a(buf, 0, fa0[0]);
            fetch_a(buf, 1, fa0[1]);
            __builtin_amdgcn_sched_barrier(0);
            stage(3, c + 1, true);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");     // the B0 reads are back: B0 may be restaged in the next phase
            TS(0, 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TS(0, 2);
            PIN(acc[0][0], acc[1][0]);
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mma(acc[0][0], fa0[0][ks], fb0[ks]);
                mma(acc[1][0], fa0[1][ks], fb0[ks]);
            }
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);
            PIN(acc[0][0], acc[1][0]);
            TS(0, 3);
            __builtin_amdgcn_s_barrier();
            // ---- phase 2: B1 -> C[0..1][1]; stage B0 of chunk c + 2
            TS(1, 0);
            fetch_b(buf, 1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            stage(0, c + 2, true);
            TS(1, 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TS(1, 2);
            PIN(acc[0][1], acc[1][1]);
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mma(acc[0][1], fa0[0][ks], fb1[ks]);
                mma(acc[1][1], fa0[1][ks], fb1[ks]);
            }
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);
            PIN(acc[0][1], acc[1][1]);
            TS(1, 3);
            __builtin_amdgcn_s_barrier();
            // ---- phase 3: A1 -> C[2..3][1]; stage A0 of chunk c + 2
            TS(2, 0);
            fetch_a(buf, 2, fa1[0]);
            fetch_a(buf, 3, fa1[1]);
            __builtin_amdgcn_sched_barrier(0);
            stage(1, c + 2, true);
            TS(2, 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TS(2, 2);
            PIN(acc[2][1], acc[3][1]);
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mma(acc[2][1], fa1[0][ks], fb1[ks]);
                mma(acc[3][1], fa1[1][ks], fb1[ks]);
            }
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);
            PIN(acc[2][1], acc[3][1]);
            TS(2, 3);
            __builtin_amdgcn_s_barrier();
            // ---- phase 4: (B0 kept) -> C[2..3][0]; stage B1 of chunk c + 2; chunk c + 1 has landed after the counted wait
            TS(3, 0);
            stage(2, c + 2, true);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            TS(3, 1);
            __builtin_amdgcn_s_barrier();
            TS(3, 2);
            PIN(acc[2][0], acc[3][0]);
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mma(acc[2][0], fa1[0][ks], fb0[ks]);
                mma(acc[3][0], fa1[1][ks], fb0[ks]);
            }
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);
            PIN(acc[2][0], acc[3][0]);
            TS(3, 3);
            __builtin_amdgcn_s_barrier();
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
        if (lane == 0 && blockIdx.x == 0 && (wave == 0 || wave == 4)) {
            unsigned* dbg = reinterpret_cast<unsigned*>(C) + (wave >> 2) * 32;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) dbg[(cc * 4 + pp) * 4 + kk] = tsv[cc][pp][kk];
        }
        return;
        } else {
        // prologue: chunk 0 whole, chunk 1's first three half-tiles; chunk 0 has landed after the counted wait
        for (int k = 0; k < 4; ++k) stage(k, 0);
        for (int k = 0; k < 3; ++k) stage(k, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();           // the second wave group runs one barrier behind the first
        uint4 fb0[4] = {}, fb1[4] = {}, fa0[2][4] = {}, fa1[2][4] = {};
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            // ---- phase 1: B0 + A0 -> C[0..1][0]; stage A1 of chunk c + 1
            fetch_b(buf, 0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            fetch_a(buf, 0, fa0[0]);
            fetch_a(buf, 1, fa0[1]);
            __builtin_amdgcn_sched_barrier(0);
            stage(3, c + 1, true);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");     // the B0 reads are back: B0 may be restaged in the next phase
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PIN(acc[0][0], acc[1][0]);
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mma(acc[0][0], fa0[0][ks], fb0[ks]);
                mma(acc[1][0], fa0[1][ks], fb0[ks]);
            }
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);
            PIN(acc[0][0], acc[1][0]);
            __builtin_amdgcn_s_barrier();
            // ---- phase 2: B1 -> C[0..1][1]; stage B0 of chunk c + 2
            fetch_b(buf, 1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            stage(0, c + 2, true);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PIN(acc[0][1], acc[1][1]);
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mma(acc[0][1], fa0[0][ks], fb1[ks]);
                mma(acc[1][1], fa0[1][ks], fb1[ks]);
            }
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);
            PIN(acc[0][1], acc[1][1]);
            __builtin_amdgcn_s_barrier();
            // ---- phase 3: A1 -> C[2..3][1]; stage A0 of chunk c + 2
            fetch_a(buf, 2, fa1[0]);
            fetch_a(buf, 3, fa1[1]);
            __builtin_amdgcn_sched_barrier(0);
            stage(1, c + 2, true);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PIN(acc[2][1], acc[3][1]);
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mma(acc[2][1], fa1[0][ks], fb1[ks]);
                mma(acc[3][1], fa1[1][ks], fb1[ks]);
            }
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);
            PIN(acc[2][1], acc[3][1]);
            __builtin_amdgcn_s_barrier();
            // ---- phase 4: (B0 kept) -> C[2..3][0]; stage B1 of chunk c + 2; chunk c + 1 has landed after the counted wait
            stage(2, c + 2, true);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            PIN(acc[2][0], acc[3][0]);
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                mma(acc[2][0], fa1[0][ks], fb0[ks]);
                mma(acc[3][0], fa1[1][ks], fb0[ks]);
            }
            if (!(ABL & 16)) __builtin_amdgcn_s_setprio(0);
            PIN(acc[2][0], acc[3][0]);
            __builtin_amdgcn_s_barrier();
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // plain epilogue (not the subject): pixel = lane & 31, channel = 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of the 32 x 32 fragment
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = m0 + wm * 128 + i * 32 + (lane & 31);
            if (row >= M) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + wn * 64 + j * 32 + 8 * g + 4 * (lane >> 5);
                if (col < N) *reinterpret_cast<float4*>(C + (long)row * N + col) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            }
        }
}

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <int VAR, int ABL = 0>
float run(const uint16_t* dA, const uint16_t* dB, float* dC, int M, int N, int K, int iters) {
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    const int lds = 2 * kBufBytes;
    CK(hipFuncSetAttribute((const void*)gemm_kernel<VAR, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_kernel<VAR, ABL>), dim3(tiles_m * tiles_n), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_kernel<VAR, ABL>), dim3(tiles_m * tiles_n), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    int M = argc > 3 ? atoi(argv[1]) : 4096, N = argc > 3 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    if (K % 64) { printf("K must be a multiple of 64\n"); return 1; }
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hB) v = f2bf(rnd());
    uint16_t *dA, *dB;
    float* dC;
    CK(hipMalloc(&dA, hA.size() * 2));
    CK(hipMalloc(&dB, hB.size() * 2));
    CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hC((size_t)M * N);
    const double flop = 2.0 * M * N * K;
    for (int round = 0; round < 3; ++round) {
        for (int var = 0; var < 3; ++var) {
            CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
            const float ms = var == 0 ? run<0>(dA, dB, dC, M, N, K, 20) : var == 1 ? run<1>(dA, dB, dC, M, N, K, 20) : run<3>(dA, dB, dC, M, N, K, 20);
            CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
            // check 64 scattered rows x all columns against a double reference (asymmetric operands: a transpose cannot pass)
            double maxerr = 0, maxref = 0;
            for (int t = 0; t < 64; ++t) {
                const int r = (int)(((long)t * 7919 + 13) % M);
                for (int c = 0; c < N; c += 37) {
                    double ref = 0;
                    for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)r * K + k]) * bf2f(hB[(size_t)c * K + k]);
                    maxerr = std::max(maxerr, std::abs(ref - hC[(size_t)r * N + c]));
                    maxref = std::max(maxref, std::abs(ref));
                }
            }
            printf("M %d N %d K %d  var %d  %.3f ms  %.1f TF/s  max err %.3g (ref max %.3g) %s\n", M, N, K, var == 2 ? 3 : var, ms, flop / ms * 1e-9, maxerr, maxref,
                   maxerr <= 2e-3 * maxref ? "OK" : "MISMATCH");
        }
    }
    {
        CK(hipMemset(dC, 0, 4096));
        run<1, 32>(dA, dB, dC, M, N, K, 1);
        unsigned h[64];
        CK(hipMemcpy(h, dC, sizeof(h), hipMemcpyDeviceToHost));
        const unsigned t00 = h[0];
        for (int g = 0; g < 2; ++g) {
            printf("trace group %d (cycles since group 0's first stamp; per phase: start, before barrier 1, after barrier 1 + lgkmcnt(0), after MFMA issue):\n", g);
            for (int i = 0; i < 8; ++i) printf("  chunk %d phase %d: %6u %6u %6u %6u\n", 8 + i / 4, i % 4 + 1, h[g * 32 + i * 4] - t00, h[g * 32 + i * 4 + 1] - t00, h[g * 32 + i * 4 + 2] - t00, h[g * 32 + i * 4 + 3] - t00);
        }
    }
    {
        CK(hipMemset(dC, 0, 4096));
        run<3, 32>(dA, dB, dC, M, N, K, 1);
        unsigned h[64];
        CK(hipMemcpy(h, dC, sizeof(h), hipMemcpyDeviceToHost));
        const unsigned t00 = h[0];
        for (int g = 0; g < 2; ++g) {
            printf("VAR 3 trace group %d (per phase X, Y: start, before barrier 1, after barrier 1 + lgkmcnt(0), after MFMA issue):\n", g);
            for (int i = 0; i < 8; ++i) if (i % 4 < 2) printf("  chunk %d phase %c: %6u %6u %6u %6u\n", 8 + i / 4, i % 4 ? 'Y' : 'X', h[g * 32 + i * 4] - t00, h[g * 32 + i * 4 + 1] - t00, h[g * 32 + i * 4 + 2] - t00, h[g * 32 + i * 4 + 3] - t00);
        }
        {
            CK(hipMemset(dC, 0, 1 << 16));
            const float tms = run<3, 64>(dA, dB, dC, M, N, K, 1);
            unsigned long hh[512];
            CK(hipMemcpy(hh, dC, sizeof(hh), hipMemcpyDeviceToHost));
            double cyc = 0, rt = 0; int nb = std::min(256, ((M + 255) / 256) * ((N + 255) / 256));
            for (int b = 0; b < nb; ++b) { cyc += hh[2 * b]; rt += hh[2 * b + 1]; }
            cyc /= nb; rt /= nb;
            printf("VAR 3 main loop: %.0f shader cycles, %.0f realtime ticks (100 MHz -> %.1f us) -> %.3f GHz; %.1f cycles per chunk; kernel %.1f us\n", cyc, rt, rt / 100.0, cyc / (rt * 10.0), cyc / (K / 64), tms * 1e3);
        }
        const float v0 = run<3, 0>(dA, dB, dC, M, N, K, 20), v1 = run<3, 1>(dA, dB, dC, M, N, K, 20), v2 = run<3, 2>(dA, dB, dC, M, N, K, 20), v3 = run<3, 3>(dA, dB, dC, M, N, K, 20),
                    v4 = run<3, 4>(dA, dB, dC, M, N, K, 20), v16 = run<3, 16>(dA, dB, dC, M, N, K, 20);
        printf("VAR 3 ablation ms: full %.3f | noDMA %.3f | noREAD %.3f | mfma+barriers %.3f | noMFMA %.3f | no setprio %.3f\n", v0, v1, v2, v3, v4, v16);
    }
    // timing ablations of the 8-phase loop (wrong results by construction): 1 no DMA in the loop, 2 no operand reads, 4 no MFMAs
    const float t0 = run<1, 0>(dA, dB, dC, M, N, K, 20), t1 = run<1, 1>(dA, dB, dC, M, N, K, 20), t2 = run<1, 2>(dA, dB, dC, M, N, K, 20),
                t3 = run<1, 3>(dA, dB, dC, M, N, K, 20), t4 = run<1, 4>(dA, dB, dC, M, N, K, 20), t5 = run<1, 5>(dA, dB, dC, M, N, K, 20),
                t6 = run<1, 6>(dA, dB, dC, M, N, K, 20), t7 = run<1, 7>(dA, dB, dC, M, N, K, 20);
    const float u0 = run<1, 8>(dA, dB, dC, M, N, K, 20), u1 = run<1, 16>(dA, dB, dC, M, N, K, 20), u2 = run<1, 24>(dA, dB, dC, M, N, K, 20);
    printf("variants ms (TF/s): scalar-offset DMA %.3f (%.0f) | VALU-offset DMA %.3f (%.0f) | no setprio %.3f (%.0f) | VALU-offset + no setprio %.3f (%.0f)\n",
           t0, flop / t0 * 1e-9, u0, flop / u0 * 1e-9, u1, flop / u1 * 1e-9, u2, flop / u2 * 1e-9);
    printf("ablation ms: full %.3f | noDMA %.3f | noREAD %.3f | noDMA+noREAD (mfma+barriers) %.3f | noMFMA %.3f | DMA only... noMFMA+noDMA %.3f | noMFMA+noREAD %.3f | skeleton %.3f\n",
           t0, t1, t2, t3, t4, t5, t6, t7);
    return 0;
}
