#!/usr/bin/env python3
"""Single weight-gradient launches at the C3 shapes (HIP-event timing of kernel + reduce; run under rocprofv3 --kernel-trace --stats to
split the two): python tools/wgbench.py [substring]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvfnet_amd import _lib  # noqa: E402

lib, check = _lib.lib, _lib.check
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
SHAPES = [("l2.c1", 28, 512, 128, 1), ("l2.ds", 28, 256, 512, 1), ("l3.c2", 14, 256, 256, 3), ("l3.c1", 14, 1024, 256, 1), ("l3.c3", 14, 256, 1024, 1), ("l4.c2", 7, 512, 512, 3), ("l4.c3", 7, 512, 2048, 1),
          ("l2.c2", 28, 128, 128, 3), ("l2.c3", 28, 128, 512, 1), ("l1.c2", 56, 64, 64, 3), ("l1.c3", 56, 64, 256, 1)]


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    n, dt = 256, 1
    for name, hw, cin, cout, k in SHAPES:
        if only and only not in name:
            continue
        m = n * hw * hw
        d = _lib.ConvDesc(n, hw, hw, cin, cout, k, k, 1, k // 2, hw, hw, cin, dt, 0, 0, 0, 0, 0)
        gen = torch.Generator(device="cuda").manual_seed(m + cin)
        x = torch.randn(m, cin, device="cuda", generator=gen).bfloat16()
        dz = torch.randn(m, cout, device="cuda", generator=gen).bfloat16()
        ws = torch.empty(lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device="cuda")
        dw = torch.empty(cout, cin, k, k, device="cuda")
        fn = lambda: check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), P(dz), P(x), None, k, cin, k, cin, P(dw), P(ws), ws.numel(), None))  # noqa: E731
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2.0 * m * cout * k * k * cin
        import hashlib
        dig = hashlib.sha256(dw.cpu().numpy().tobytes()).hexdigest()[:12]
        print("%-6s wgrad M%-7d N%-5d K%-5d %8.1f us  %6.1f TF/s   slab %.1f MB  dw %s" % (name, m, cout, k * k * cin, t, fl / t / 1e6, ws.numel() / 1e6, dig))


if __name__ == "__main__":
    main()
