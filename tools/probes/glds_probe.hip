// Probe: LDS-DMA (buffer_load_dwordx4 ... lds) semantics on gfx950 -- lane-linear destination (M0 base + lane*16), and what
// an out-of-range buffer offset writes (the conv loader relies on ZERO fill for padding taps).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/glds_probe.hip -o tools/probes/glds_probe && tools/probes/glds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void probe(const unsigned* src, unsigned src_bytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * 64 * 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2 * 64 * 4; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
    // lane l reads unit (l ^ 5); lanes 8..15 read out of range
    unsigned voff = (unsigned)((lane ^ 5) * 16);
    if (lane >= 8 && lane < 16) voff = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 64 * 4), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 2 * 64 * 4; i += 64) out[i] = lds[i];
}

int main() {
    unsigned h[64 * 4], *d, *o, r[2 * 64 * 4];
    for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    hipMalloc(&d, sizeof h);
    hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, (unsigned)sizeof h, o);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[i] != 0xdeadbeefu;
    printf("first slab untouched: %s\n", bad ? "NO" : "yes");
    bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
            unsigned want = (l >= 8 && l < 16) ? 0u : 1000 + (l ^ 5) * 4 + k;
            if (r[256 + l * 4 + k] != want) {
                if (bad < 8) printf("lane %d dword %d: got %u (0x%x) want %u\n", l, k, r[256 + l * 4 + k], r[256 + l * 4 + k], want);
                ++bad;
            }
        }
    printf("lane-linear + zero-fill: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return bad != 0;
}
