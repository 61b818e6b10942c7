"""The stem conv alone at the C3 shape (256 frames, 224 x 224): times mvf_conv2d_nhwc_fwd_stats / _fwd_ws (bias + ReLU) with HIP events.
usage: python tools/stem_bench.py [iters]   (MVF_POLICY=stem_direct=0 -> the implicit-GEMM kernel)"""
import ctypes as C
import sys

import torch

from mvfnet_amd import _lib

lib, check = _lib.lib, _lib.check
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
n, h, w = 256, 224, 224
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hp, wp, ho, wo = h + 6, (w + 9) // 2 * 2, 112, 112
x = torch.randn(n, 3, h, w, device="cuda")
wt = torch.randn(64, 3, 7, 7, device="cuda") * 0.1
xp = torch.empty(n, hp, wp, 4, device="cuda", dtype=torch.bfloat16)
check(lib.mvf_stem_prep(P(x), n, 3, h, w, 3, wp, P(xp), 1, None))
wpk = torch.empty(64, 7, 8, 4, device="cuda", dtype=torch.bfloat16)
check(lib.mvf_pack_conv_weight(P(wt), 64, 3, 7, 7, 8, 4, None, P(wpk), 1, None))
d = _lib.ConvDesc(n, hp, wp, 32, 64, 7, 1, 2, 0, ho, wo, 4, 1, 0, 0, 0, 0, 0)
rows = lib.mvf_conv2d_stats_rows(C.byref(d))
z = torch.empty(n * ho * wo, 64, device="cuda", dtype=torch.bfloat16)
part = torch.empty(64, rows, 2, device="cuda")
shift = torch.zeros(64, device="cuda")
bias = torch.zeros(64, device="cuda")
for name in ("train", "infer"):
    d.relu = 1 if name == "infer" else 0
    def go():
        if name == "train":
            check(lib.mvf_conv2d_nhwc_fwd_stats(C.byref(d), P(xp), None, P(wpk), P(z), P(part), P(shift), None, 0, None))
        else:
            check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), P(xp), None, P(wpk), P(bias), None, P(z), None, 0, None))
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("stem %s: %.1f us  (%.2f TB/s of output bytes)" % (name, us, n * ho * wo * 128 / us / 1e6))
# [r4] the stem's weight gradient (kernel + slab reduce): MVF_POLICY=wgrad_stem_direct=0 -> the implicit GEMM
dz = torch.randn(n * ho * wo, 64, device="cuda").bfloat16()
wsz = lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d))
wws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
dw = torch.empty(64, 3, 7, 7, device="cuda")
d.relu = 0
def gow():
    check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), P(dz), P(xp), None, 7, 3, 8, 4, P(dw), P(wws), wsz, None))
for _ in range(3):
    gow()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    gow()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
print("stem wgrad: %.1f us  (%.2f TB/s of dz + input bytes, %.0f TF/s of the padded K = 224 product)" % (us, (n * ho * wo * 128 + xp.numel() * 2) / us / 1e6, 2.0 * n * ho * wo * 64 * 224 / us / 1e6))
