"""layer1's 3x3 conv alone at the C3 shape (256 frames, 56 x 56 x 64): forward + statistics, data gradient + BatchNorm sums.
usage: python tools/c3_bench.py [iters]   (MVF_POLICY=conv3x3_direct=0 -> the implicit-GEMM kernel)"""
import ctypes as C
import sys

import torch

from mvfnet_amd import _lib

lib, check = _lib.lib, _lib.check
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
n, h, w = 256, 56, 56
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = n * h * w
x = torch.randn(m, 64, device="cuda").bfloat16()
wt = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
wpk = torch.empty(64, 3, 3, 64, device="cuda", dtype=torch.bfloat16)
check(lib.mvf_pack_conv_weight(P(wt), 64, 64, 3, 3, 3, 64, None, P(wpk), 1, None))
d = _lib.ConvDesc(n, h, w, 64, 64, 3, 3, 1, 1, h, w, 64, 1, 0, 0, 0, 0, 0)
rows = lib.mvf_conv2d_stats_rows(C.byref(d))
z = torch.empty(m, 64, device="cuda", dtype=torch.bfloat16)
zb = torch.randn(m, 64, device="cuda").bfloat16()
part = torch.empty(64, rows, 2, device="cuda")
v = [torch.zeros(64, device="cuda") for _ in range(4)]
ws = torch.empty(lib.mvf_conv2d_workspace_bytes(C.byref(d)), dtype=torch.uint8, device="cuda")
for name in ("fwd+stats", "plain", "dgrad+bnsums"):
    def go():
        if name == "fwd+stats":
            check(lib.mvf_conv2d_nhwc_fwd_stats(C.byref(d), P(x), None, P(wpk), P(z), P(part), P(v[0]), P(ws), ws.numel(), None))
        elif name == "plain":
            check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), P(x), None, P(wpk), None, None, P(z), P(ws), ws.numel(), None))
        else:
            check(lib.mvf_conv2d_nhwc_dgrad_bnsums(C.byref(d), P(x), P(wpk), P(z), P(zb), P(v[0]), P(v[1]), P(v[2]), P(v[3]), P(part), P(ws), ws.numel(), None))
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
    print("c3 %s: %.1f us  (%.0f TF/s)" % (name, us, 2.0 * m * 64 * 576 / us / 1e6))
