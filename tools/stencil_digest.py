#!/usr/bin/env python3
"""Digest of the bf16 NHWC stencil launches over a set of shapes and variants (plain, + statistics, transposed + gated addend + output gate [+ sums]): printed as
JSON.  tests/test_mvf_gpu.py runs it with MVF_POLICY=stencil_lds=0 / 1 (the switch is read once per process) and compares: the LDS-tiled kernel must reproduce the chunked
kernel bit for bit (outputs) and to fp32 summation order (statistics)."""
import ctypes as C
import hashlib
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from mvfnet_amd import _lib as L  # noqa: E402

lib, check = L.lib, L.check
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
BF = torch.bfloat16
CASES = [(2, 8, 14, 14, 1024, 128), (1, 16, 14, 14, 1024, 128), (2, 8, 28, 28, 512, 64), (3, 8, 7, 7, 2048, 256), (2, 4, 14, 14, 256, 32), (1, 8, 9, 11, 128, 16),
         (1, 16, 28, 28, 512, 64)]


def main():
    res = {}
    for case in CASES:
        nc, t, h, w, c, cs = case
        nt, m = nc * t, nc * t * h * w
        gen = torch.Generator().manual_seed(m + cs)
        x = torch.randn(m, c, generator=gen).cuda().to(BF)
        dy = torch.randn(m, cs, generator=gen).cuda().to(BF)
        add = torch.randn(m, c, generator=gen).cuda().to(BF)
        abits = torch.randint(0, 16, (m, c // 4), generator=gen, dtype=torch.uint8).cuda()
        gate = torch.randint(0, 16, (m, c // 4), generator=gen, dtype=torch.uint8).cuda()
        z = (torch.randn(m, c, generator=gen) * 1.2 + 0.3).cuda().to(BF)
        mean, invstd, shift = (torch.randn(c, generator=gen) * 0.3).cuda(), (torch.rand(c, generator=gen) + 0.4).cuda(), (torch.randn(cs, generator=gen) * 0.1).cuda()
        wt, wh, ww = (torch.randn(cs, 3, generator=gen).cuda() for _ in range(3))
        sc, sh = (torch.rand(cs, generator=gen) + 0.5).cuda(), (torch.randn(cs, generator=gen) * 0.2).cuda()
        d = L.MvfDesc(nt, c, h, w, t, cs, L.MODE_BITS["THW"], L.MVF_NHWC, L.MVF_BF16)
        outs = {}
        y = torch.zeros(m, cs, device="cuda", dtype=BF)
        check(lib.mvf_nhwc_stencil(C.byref(d), P(x), c, P(y), cs, P(wt), P(wh), P(ww), None, None, 0, None, 0, None, None))
        outs["plain"] = y.clone()
        check(lib.mvf_nhwc_stencil(C.byref(d), P(x), c, P(y), cs, P(wt), P(wh), P(ww), P(sc), P(sh), 0, None, 0, None, None))
        outs["hardswish"] = y.clone()
        rows = lib.mvf_nhwc_stencil_stats_rows(C.byref(d), c, cs)
        part = torch.zeros(cs, rows, 2, device="cuda")
        check(lib.mvf_nhwc_stencil_stats(C.byref(d), P(x), c, P(y), cs, P(wt), P(wh), P(ww), P(part), P(shift), None))
        outs["stats_y"] = y.clone()
        stats = part.double().sum(1)
        o = torch.zeros(m, c, device="cuda", dtype=BF)
        check(lib.mvf_nhwc_stencil(C.byref(d), P(dy), cs, P(o), c, P(wt), P(wh), P(ww), None, None, 1, P(add), c, P(abits), None))
        outs["transposed_addend"] = o[:, :cs].clone()
        check(lib.mvf_nhwc_stencil_gate(C.byref(d), P(dy), cs, P(o), c, P(wt), P(wh), P(ww), None, None, 1, P(add), c, P(abits), P(gate), None))
        outs["transposed_gated"] = o[:, :cs].clone()
        rows2 = lib.mvf_nhwc_stencil_stats_rows(C.byref(d), cs, c)
        part3 = torch.zeros(cs, rows2, 2, device="cuda")
        check(lib.mvf_nhwc_stencil_gate_colsums(C.byref(d), P(dy), cs, P(o), c, P(wt), P(wh), P(ww), 1, P(add), c, P(abits), P(gate), P(part3), None))
        outs["transposed_gated_colsums"] = o[:, :cs].clone()
        colsums = part3.double().sum(1)
        torch.cuda.synchronize()
        res[str(case)] = dict(digest={k: hashlib.sha256(v.contiguous().view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16] for k, v in outs.items()},
                              stats=stats.cpu().flatten().tolist(), colsums=colsums.cpu().flatten().tolist(), rows=[rows, rows2])
    print(json.dumps(res))


if __name__ == "__main__":
    main()
