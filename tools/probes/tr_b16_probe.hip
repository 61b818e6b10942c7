// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/tr_b16_probe tools/probes/tr_b16_probe.hip
// Probe of gfx950's ds_read_b64_tr_b16 lane/element mapping (run on the GPU box; prints the table the wgrad kernel relies on).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: lane l points at elements 4l..4l+3 (8 bytes each, consecutive)
    // mode 1: a [16 k][64 n] row-major bf16 matrix (row pitch 128 B); lane l points at row (l&15)/4*... see host printout
    unsigned addr;
    if (mode == 0) addr = l * 8;
    else {
        const int g = l >> 4, i = l & 15;
        const int row = i >> 2, colq = i & 3;                     // 4 rows x (4 col-quads) per 16-lane group
        addr = (unsigned)((row + 4 * (g >> 1)) * 128 + (colq * 4 + 16 * (g & 1)) * 2);
    }
    addr += (unsigned)(uintptr_t)lds;                             // LDS base of this array (static shared: usually 0)
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = (uint16_t)(v.x & 0xffff);
    out[l * 4 + 1] = (uint16_t)(v.x >> 16);
    out[l * 4 + 2] = (uint16_t)(v.y & 0xffff);
    out[l * 4 + 3] = (uint16_t)(v.y >> 16);
}

int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
