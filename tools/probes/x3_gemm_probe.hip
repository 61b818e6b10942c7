// Probe: fp32 GEMM on the bf16 matrix cores (three-term exact bf16 split, six partial products; conv_tile X3 of csrc/conv_nhwc.hip) as a
// standalone C[M][N] = A[M][K] * B[N][K]^T, to find what bounds the loop.  128 x 128 tile, 4 waves, 32-channel (128-byte) K chunks.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o x3_gemm_probe x3_gemm_probe.hip && ./x3_gemm_probe [M N K]
//
// VAR 0: the engine's loop -- one chunk ahead in registers, split when the chunk is WRITTEN to LDS (three 64-byte bf16 planes per row),
//        single LDS buffer, two barriers per chunk.
// VAR 1: fp32 tiles by LDS-DMA into two buffers (chunk k + 1 in flight under chunk k, one barrier per chunk), split when a fragment is
//        FETCHED (two ds_read_b128 = 8 fp32 per lane and k-step -> three bf16x8 in registers): no staging registers, no plane traffic,
//        but every wave splits its own fragments (2 x redundant).
// ABL (timing ablation, wrong results): 1 no global loads, 2 no split arithmetic, 4 no matrix instructions, 8 no LDS fragment reads,
//        16 no LDS writes (VAR 0).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    uint32_t u;
    __builtin_memcpy(&u, &b, 4);
    return u;
}
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
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float sub1(float x, float y) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
template <int ABL>
__device__ __forceinline__ void split4(const uint4& v, uint2 (&pl)[3]) {
    float r[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const unsigned a = pack2(r[0], r[1]), b = pack2(r[2], r[3]);
        pl[p] = make_uint2(a, b);
        if (p < 2 && !(ABL & 2)) {
            // scalar v_sub_f32 through asm: left to itself hipcc SLP-packs these four into v_pk_add_f32, and packed fp32 VALU beside matrix
            // instructions is an anti-lever on this chip (MI355X_MICROARCH.md: +26 cycles per two v_pk_add_f32 in an MFMA gap)
            r[0] = sub1(r[0], __uint_as_float(a << 16)); r[1] = sub1(r[1], __uint_as_float(a & 0xffff0000u));
            r[2] = sub1(r[2], __uint_as_float(b << 16)); r[3] = sub1(r[3], __uint_as_float(b & 0xffff0000u));
        }
    }
}
// 8 fp32 (two 16-byte units) -> hi / mid / lo bf16x8
template <int ABL>
__device__ __forceinline__ void split8(const uint4& v0, const uint4& v1, bf16x8 (&f)[3]) {
    uint2 a[3], b[3];
    split4<ABL>(v0, a);
    split4<ABL>(v1, b);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const uint4 w = make_uint4(a[p].x, a[p].y, b[p].x, b[p].y);
        __builtin_memcpy(&f[p], &w, 16);
    }
}

// VAR 2: the 256 x 256 tile by 8 waves (4 x 2, 64 x 128 outputs each), fp32 tiles by LDS-DMA into two 64 KB buffers, split at fetch: half the
//        L2 -> LDS bytes per flop of the 128 x 128 tile (fp32 operands at 128 x 128 need 12 TB/s of L2 -> CU traffic for the matrix peak / 6).
template <int ABL>
__global__ __launch_bounds__(512) void gemm256_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                 // 4 x 2 waves: rows wm * 64, columns wn * 128
    const int lrow = tid >> 3, q = tid & 7;                  // 64 loader rows per pass, 4 passes per 256-row tile
    const int nchunks = K / 32;
    char* As = smem;                 // [2][256][128]
    char* Bs = smem + 2 * 256 * 128;
    const i32x4 gA = rsrc_words(A + (long)m0 * K, (unsigned)((long)256 * K * 4));
    const i32x4 gB = rsrc_words(B + (long)n0 * K, (unsigned)((long)256 * K * 4));
    const unsigned ldsA = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)As + wave * 8 * 128);
    const unsigned ldsB = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Bs + wave * 8 * 128);
    const int qs = q ^ ((lrow >> 1) & 7);
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto dma = [&](int c, int buf) {
        if (ABL & 1) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned off = (unsigned)(((lrow + 64 * i) * K + c * 32 + qs * 4) * 4);
            glds16(gA, ldsA + (unsigned)((buf * 256 + 64 * i) * 128), off);
            glds16(gB, ldsB + (unsigned)((buf * 256 + 64 * i) * 128), off);
        }
    };
    const int fr = lane & 31, fsw = (fr >> 1) & 7;
    constexpr int PA_[6] = {0, 2, 1, 0, 1, 0}, PB_[6] = {2, 0, 1, 1, 0, 0};
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int u0 = 4 * ks + 2 * (lane >> 5);
            const int k0 = ((u0) ^ fsw) << 4, k1 = ((u0 + 1) ^ fsw) << 4;
            bf16x8 fa[2][3], fb[4][3];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* pa = As + (buf * 256 + wm * 64 + i * 32 + fr) * 128;
                uint4 a0, a1;
                if (ABL & 8) { a0 = make_uint4(0x3f800000u + lane, 0x3f000000u, 0x40000000u, 0x3f800000u + i); a1 = a0; }
                else { a0 = *reinterpret_cast<const uint4*>(pa + k0); a1 = *reinterpret_cast<const uint4*>(pa + k1); }
                split8<ABL>(a0, a1, fa[i]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* pb = Bs + (buf * 256 + wn * 128 + j * 32 + fr) * 128;
                uint4 b0, b1;
                if (ABL & 8) { b0 = make_uint4(0x3f800000u + lane, 0x3f000000u, 0x40000000u, 0x3f800000u + j); b1 = b0; }
                else { b0 = *reinterpret_cast<const uint4*>(pb + k0); b1 = *reinterpret_cast<const uint4*>(pb + k1); }
                split8<ABL>(b0, b1, fb[j]);
            }
            if (!(ABL & 4)) {
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][PB_[t]], fa[i][PA_[t]], acc[i][j], 0, 0, 0);
            }
        }
    };
    dma(0, 0);
    for (int c = 0; c < nchunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 1 < nchunks) dma(c + 1, (c + 1) & 1);
        compute(c & 1);
    }
    const int lc = lane & 31, lr = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + lc, n = n0 + wn * 128 + j * 32 + 8 * (r >> 2) + 4 * lr + (r & 3);
                C[(long)m * N + n] = acc[i][j][r];
            }
}

template <int ABL>
float run256(const float* A, const float* B, float* C, int M, int N, int K, int reps) {
    const int tiles_n = N / 256, tiles = (M / 256) * tiles_n;
    const size_t lds = 2 * 2 * 256 * 128;
    auto k = gemm256_kernel<ABL>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(512), lds, 0, A, B, C, M, N, K, tiles_n);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(512), lds, 0, A, B, C, M, N, K, tiles_n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

// VAR 3: 128 x 128 tile, fp32 tiles by LDS-DMA (two buffers, TWO chunks in flight), split at fetch, SOFTWARE-PIPELINED over k-steps inside every wave:
//        the fragments of step s + 1 are read and split (VALU) between the matrix instructions of step s (sched_group_barrier interleave), so the
//        split hides under the matrix pipe instead of alternating with it.  One barrier per chunk.
template <int ABL, int ILV>
__global__ __launch_bounds__(256, 2) void gemm_sp_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 3, q = tid & 7;
    const int nchunks = K / 32;
    char* As = smem;                 // [2][128][128]
    char* Bs = smem + 2 * 128 * 128;
    const i32x4 gA = rsrc_words(A + (long)m0 * K, (unsigned)((long)128 * K * 4));
    const i32x4 gB = rsrc_words(B + (long)n0 * K, (unsigned)((long)128 * K * 4));
    const unsigned ldsA = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)As + wave * 8 * 128);
    const unsigned ldsB = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Bs + wave * 8 * 128);
    const int qs = q ^ ((lrow >> 1) & 7);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto dma = [&](int c, int buf) {
        if (ABL & 1) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned off = (unsigned)(((lrow + 32 * i) * K + c * 32 + qs * 4) * 4);
            glds16(gA, ldsA + (unsigned)((buf * 128 + 32 * i) * 128), off);
            glds16(gB, ldsB + (unsigned)((buf * 128 + 32 * i) * 128), off);
        }
    };
    const int fr = lane & 31, fsw = (fr >> 1) & 7;
    const char* a_base = As + (wm * 64 + fr) * 128;
    const char* b_base = Bs + (wn * 64 + fr) * 128;
    int ko[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int u0 = 4 * ks + 2 * (lane >> 5);
        ko[ks][0] = (u0 ^ fsw) << 4;
        ko[ks][1] = ((u0 + 1) ^ fsw) << 4;
    }
    struct Raw { uint4 a[2][2], b[2][2]; };
    auto read_raw = [&](int buf, int ks, Raw& r) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            r.a[i][0] = *reinterpret_cast<const uint4*>(a_base + (buf * 128 + i * 32) * 128 + ko[ks][0]);
            r.a[i][1] = *reinterpret_cast<const uint4*>(a_base + (buf * 128 + i * 32) * 128 + ko[ks][1]);
            r.b[i][0] = *reinterpret_cast<const uint4*>(b_base + (buf * 128 + i * 32) * 128 + ko[ks][0]);
            r.b[i][1] = *reinterpret_cast<const uint4*>(b_base + (buf * 128 + i * 32) * 128 + ko[ks][1]);
        }
    };
    struct Frag { bf16x8 a[2][3], b[2][3]; };
    auto split_all = [&](const Raw& r, Frag& f) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            split8<ABL>(r.a[i][0], r.a[i][1], f.a[i]);
            split8<ABL>(r.b[i][0], r.b[i][1], f.b[i]);
        }
    };
    constexpr int PA_[6] = {0, 2, 1, 0, 1, 0}, PB_[6] = {2, 0, 1, 1, 0, 0};
    auto mma_all = [&](const Frag& f) {
        if (ABL & 4) return;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[j][PB_[t]], f.a[i][PA_[t]], acc[i][j], 0, 0, 0);
    };
    // one pipelined step: matrix instructions of `cur`, reads + split of the next step's fragments in their shadow
    auto step = [&](const Frag& cur, Frag& nxt, int nbuf, int nks, bool has_next) {
        Raw r;
        if (has_next) read_raw(nbuf, nks, r);
        mma_all(cur);
        if (has_next) split_all(r, nxt);
        if (ILV) {
            // 8 LDS reads first, then 24 x (1 matrix instruction + ILV VALU): the split's ~180 VALU ride between the 24 matrix instructions
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, ILV, 0);
            }
        }
    };
    dma(0, 0);
    if (nchunks > 1) dma(1, 1);
    if (nchunks > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    Frag f0, f1;
    {
        Raw r;
        read_raw(0, 0, r);
        split_all(r, f0);
    }
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        step(f0, f1, buf, 1, true);                                  // (c, ks0): next = (c, ks1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // chunk c + 1 has landed (issued one chunk ago)
        __syncthreads();                                             // ... for every wave, and every wave is done reading chunk c's buffer
        if (c + 2 < nchunks) dma(c + 2, buf);
        step(f1, f0, buf ^ 1, 0, c + 1 < nchunks);                   // (c, ks1): next = (c + 1, ks0)
    }
    const int lc = lane & 31, lr = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + lc, n = n0 + wn * 64 + j * 32 + 8 * (r >> 2) + 4 * lr + (r & 3);
                C[(long)m * N + n] = acc[i][j][r];
            }
}

template <int ABL, int ILV>
float run_sp(const float* A, const float* B, float* C, int M, int N, int K, int reps) {
    const int tiles_n = N / 128, tiles = (M / 128) * tiles_n;
    const size_t lds = 2 * 2 * 128 * 128;
    auto k = gemm_sp_kernel<ABL, ILV>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), lds, 0, A, B, C, M, N, K, tiles_n);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), lds, 0, A, B, C, M, N, K, tiles_n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int VAR, int ABL, int WPS>
__global__ __launch_bounds__(256, WPS) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 3, q = tid & 7;
    const int nchunks = K / 32;
    constexpr unsigned kOOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (long)m0 * K), 0, (unsigned)((long)128 * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (long)n0 * K), 0, (unsigned)((long)128 * K * 4), 0x00020000);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int PA_[6] = {0, 2, 1, 0, 1, 0}, PB_[6] = {2, 0, 1, 1, 0, 0};
    auto mma6 = [&](const bf16x8 (&fa)[2][3], const bf16x8 (&fb)[2][3]) {
        if (ABL & 4) return;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][PB_[t]], fa[i][PA_[t]], acc[i][j], 0, 0, 0);
    };

    if constexpr (VAR == 0) {
        char* As = smem;                 // [3][128][64]
        char* Bs = smem + 3 * 128 * 64;
        uint4 sa[4], sb[4];
        auto load = [&](int c) {
            if (ABL & 1) {
                if (c == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { sa[i] = make_uint4(0x3f800000u + tid, 0x3f000000u, 0x40000000u, 0x3f800000u + i); sb[i] = sa[i]; }
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned off = (unsigned)(((lrow + 32 * i) * K + c * 32 + q * 4) * 4);
                const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(rsA, c < nchunks ? off : kOOB, 0, 0);
                const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(rsB, c < nchunks ? off : kOOB, 0, 0);
                sa[i] = make_uint4(va.x, va.y, va.z, va.w);
                sb[i] = make_uint4(vb.x, vb.y, vb.z, vb.w);
            }
        };
        auto store = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lrow + 32 * i;
                uint2 pa[3], pb[3];
                split4<ABL>(sa[i], pa);
                split4<ABL>(sb[i], pb);
                if (ABL & 16) continue;
                const int o = row * 64 + ((((q >> 1) ^ ((row >> 2) & 3)) << 4) | ((q & 1) << 3));
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    *reinterpret_cast<uint2*>(As + p * 128 * 64 + o) = pa[p];
                    *reinterpret_cast<uint2*>(Bs + p * 128 * 64 + o) = pb[p];
                }
            }
        };
        const int fr = lane & 31, fx = (fr >> 2) & 3;
        auto compute = [&]() {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ko = ((2 * ks + (lane >> 5)) ^ fx) << 4;
                bf16x8 fa[2][3], fb[2][3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        uint4 va, vb;
                        if (ABL & 8) { va = make_uint4(lane, ks, p, i); vb = va; }
                        else {
                            va = *reinterpret_cast<const uint4*>(As + (p * 128 + wm * 64 + i * 32 + fr) * 64 + ko);
                            vb = *reinterpret_cast<const uint4*>(Bs + (p * 128 + wn * 64 + i * 32 + fr) * 64 + ko);
                        }
                        __builtin_memcpy(&fa[i][p], &va, 16);
                        __builtin_memcpy(&fb[i][p], &vb, 16);
                    }
                mma6(fa, fb);
            }
        };
        load(0);
        store();
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const bool more = c + 1 < nchunks;
            if (more) load(c + 1);
            compute();
            if (more) {
                __syncthreads();
                store();
            }
            __syncthreads();
        }
    } else {
        // VAR 1: fp32 tiles by LDS-DMA, two buffers; position p of row r holds unit p ^ ((r >> 1) & 7) (source-side swizzle)
        char* As = smem;                 // [2][128][128]
        char* Bs = smem + 2 * 128 * 128;
        const i32x4 gA = rsrc_words(A + (long)m0 * K, (unsigned)((long)128 * K * 4));
        const i32x4 gB = rsrc_words(B + (long)n0 * K, (unsigned)((long)128 * K * 4));
        const unsigned ldsA = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)As + wave * 8 * 128);
        const unsigned ldsB = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)Bs + wave * 8 * 128);
        const int qs = q ^ ((lrow >> 1) & 7);
        auto dma = [&](int c, int buf) {
            if (ABL & 1) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned off = (unsigned)(((lrow + 32 * i) * K + c * 32 + qs * 4) * 4);
                glds16(gA, ldsA + (unsigned)((buf * 128 + 32 * i) * 128), off);
                glds16(gB, ldsB + (unsigned)((buf * 128 + 32 * i) * 128), off);
            }
        };
        const int fr = lane & 31, fsw = (fr >> 1) & 7;
        auto compute = [&](int buf) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int u0 = 4 * ks + 2 * (lane >> 5);
                const int k0 = ((u0) ^ fsw) << 4, k1 = ((u0 + 1) ^ fsw) << 4;
                bf16x8 fa[2][3], fb[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    uint4 a0, a1, b0, b1;
                    if (ABL & 8) { a0 = make_uint4(0x3f800000u + lane, 0x3f000000u, 0x40000000u, 0x3f800000u + i); a1 = a0; b0 = a0; b1 = a0; }
                    else {
                        const char* pa = As + (buf * 128 + wm * 64 + i * 32 + fr) * 128;
                        const char* pb = Bs + (buf * 128 + wn * 64 + i * 32 + fr) * 128;
                        a0 = *reinterpret_cast<const uint4*>(pa + k0); a1 = *reinterpret_cast<const uint4*>(pa + k1);
                        b0 = *reinterpret_cast<const uint4*>(pb + k0); b1 = *reinterpret_cast<const uint4*>(pb + k1);
                    }
                    split8<ABL>(a0, a1, fa[i]);
                    split8<ABL>(b0, b1, fb[i]);
                }
                mma6(fa, fb);
            }
        };
        dma(0, 0);
        for (int c = 0; c < nchunks; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (c + 1 < nchunks) dma(c + 1, (c + 1) & 1);
            compute(c & 1);
        }
    }
    // epilogue: D^T (operands swapped): lane & 31 -> m, register -> n
    const int lc = lane & 31, lr = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + lc, n = n0 + wn * 64 + j * 32 + 8 * (r >> 2) + 4 * lr + (r & 3);
                C[(long)m * N + n] = acc[i][j][r];
            }
}

template <int VAR, int ABL, int WPS>
float run(const float* A, const float* B, float* C, int M, int N, int K, int reps) {
    const int tiles_n = N / 128, tiles = (M / 128) * tiles_n;
    const size_t lds = VAR == 0 ? 2 * 3 * 128 * 64 : 2 * 2 * 128 * 128;
    auto k = gemm_kernel<VAR, ABL, WPS>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), lds, 0, A, B, C, M, N, K, tiles_n);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), lds, 0, A, B, C, M, N, K, tiles_n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    int M = argc > 3 ? atoi(argv[1]) : 4096, N = argc > 3 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f + ((s >> 3) & 0xff) * 1e-6f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hC((size_t)M * N);
    const double flop = 2.0 * M * N * K;
    auto check = [&](const char* name, float ms) {
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (int t = 0; t < 32; ++t) {
            const int r = (int)(((long)t * 7919 + 13) % M);
            for (int c = 0; c < N; c += 97) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k];
                maxerr = std::max(maxerr, std::abs(ref - hC[(size_t)r * N + c]));
                maxref = std::max(maxref, std::abs(ref));
            }
        }
        printf("%-44s %.3f ms  %6.1f TF/s (fp32-equivalent)  max err / max %.2e\n", name, ms, flop / ms * 1e-9, maxerr / maxref);
    };
    for (int round = 0; round < 2; ++round) {
        check("VAR 0 split at store, 1 buffer, 2 waves/SIMD", run<0, 0, 2>(dA, dB, dC, M, N, K, 10));
        check("VAR 0 split at store, 1 buffer, 3 waves/SIMD", run<0, 0, 3>(dA, dB, dC, M, N, K, 10));
        check("VAR 1 DMA + split at fetch, 2 buffers", run<1, 0, 2>(dA, dB, dC, M, N, K, 10));
        if (M % 256 == 0 && N % 256 == 0) check("VAR 2 256 x 256 tile, DMA + split at fetch", run256<0>(dA, dB, dC, M, N, K, 10));
        check("VAR 3 pipelined over k-steps, compiler order", run_sp<0, 0>(dA, dB, dC, M, N, K, 10));
        check("VAR 3 pipelined, 1 MFMA : 6 VALU", run_sp<0, 6>(dA, dB, dC, M, N, K, 10));
        check("VAR 3 pipelined, 1 MFMA : 8 VALU", run_sp<0, 8>(dA, dB, dC, M, N, K, 10));
    }
    printf("  VAR 3 (1:8) no split arithmetic %.3f ms, no MFMA %.3f ms, no DMA %.3f ms\n", run_sp<2, 8>(dA, dB, dC, M, N, K, 10), run_sp<4, 8>(dA, dB, dC, M, N, K, 10),
           run_sp<1, 8>(dA, dB, dC, M, N, K, 10));
    if (M % 256 == 0 && N % 256 == 0) {
        printf("  VAR 2 no DMA               %.3f ms\n", run256<1>(dA, dB, dC, M, N, K, 10));
        printf("  VAR 2 no split arithmetic  %.3f ms\n", run256<2>(dA, dB, dC, M, N, K, 10));
        printf("  VAR 2 no MFMA              %.3f ms\n", run256<4>(dA, dB, dC, M, N, K, 10));
        printf("  VAR 2 no fragment reads    %.3f ms\n", run256<8>(dA, dB, dC, M, N, K, 10));
        printf("  VAR 2 MFMA only            %.3f ms\n", run256<11>(dA, dB, dC, M, N, K, 10));
    }
    printf("ablations (wrong results on purpose):\n");
    printf("  VAR 0 no global loads      %.3f ms\n", run<0, 1, 3>(dA, dB, dC, M, N, K, 10));
    printf("  VAR 0 no split arithmetic  %.3f ms\n", run<0, 2, 3>(dA, dB, dC, M, N, K, 10));
    printf("  VAR 0 no MFMA              %.3f ms\n", run<0, 4, 3>(dA, dB, dC, M, N, K, 10));
    printf("  VAR 0 no fragment reads    %.3f ms\n", run<0, 8, 3>(dA, dB, dC, M, N, K, 10));
    printf("  VAR 0 no LDS writes        %.3f ms\n", run<0, 16, 3>(dA, dB, dC, M, N, K, 10));
    printf("  VAR 0 no loads, no split   %.3f ms\n", run<0, 3, 3>(dA, dB, dC, M, N, K, 10));
    printf("  VAR 0 MFMA only (no loads / split / reads / writes) %.3f ms\n", run<0, 27, 3>(dA, dB, dC, M, N, K, 10));
    printf("  VAR 1 no DMA               %.3f ms\n", run<1, 1, 2>(dA, dB, dC, M, N, K, 10));
    printf("  VAR 1 no split arithmetic  %.3f ms\n", run<1, 2, 2>(dA, dB, dC, M, N, K, 10));
    printf("  VAR 1 no MFMA              %.3f ms\n", run<1, 4, 2>(dA, dB, dC, M, N, K, 10));
    printf("  VAR 1 no fragment reads    %.3f ms\n", run<1, 8, 2>(dA, dB, dC, M, N, K, 10));
    printf("  VAR 1 MFMA only            %.3f ms\n", run<1, 11, 2>(dA, dB, dC, M, N, K, 10));
    return 0;
}
