#!/usr/bin/env python3
"""Micro-benchmarks of single C-ABI kernels at the R50 8x8 x 32-clip (C3) tensor sizes: HIP-event timing, algorithmic GB/s.

    python tools/kbench.py bn        # BatchNorm streaming kernels (apply / backward reduce / backward apply) per stage
    python tools/kbench.py conv [substring]     # single implicit-GEMM launches (forward + statistics, data gradients) per stage
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvfnet_amd import _lib  # noqa: E402

lib, check = _lib.lib, _lib.check
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def bench_conv(dt=1, only=None):
    """Single conv launches at the C3 shapes: (name, pixels M as n x h x w, cin, cout, k, kind); kind: 'stats' = forward + BN statistics
    (EPI 1), 'plain' = plain data gradient (EPI 2), 'res' = data gradient + gated residual (EPI 3), 'bnb' = data gradient + BN sums (EPI 6)."""
    from mvfnet_amd._lib import ConvDesc
    tdt = torch.bfloat16 if dt else torch.float32
    esz = 2 if dt else 4
    dev = "cuda"
    shapes = [("l1.c3 fwd", 56, 64, 256, 1, "stats"), ("l1.c1 fwd", 56, 256, 64, 1, "stats"), ("l1.c1 dgrad", 56, 64, 256, 1, "res"),
              ("l1.c3 dgrad", 56, 256, 64, 1, "bnb"), ("l2.c3 fwd", 28, 128, 512, 1, "stats"), ("l2.c1 dgrad", 28, 128, 512, 1, "res"),
              ("l3.c3 fwd", 14, 256, 1024, 1, "stats"), ("l3.c1 fwd", 14, 1024, 256, 1, "stats"), ("l3.c1 dgrad", 14, 256, 1024, 1, "res"),
              ("l3.c3 dgrad", 14, 1024, 256, 1, "bnb"), ("l3.c1 dgrad+sums (proxy)", 14, 256, 1024, 1, "bnb"), ("l3.c2 fwd", 14, 256, 256, 3, "stats"), ("l3.c2 dgrad", 14, 256, 256, 3, "bnb"),
              ("l4.c3 fwd", 7, 512, 2048, 1, "stats"), ("l2.c2 fwd", 28, 128, 128, 3, "stats"), ("l1.c2 fwd", 56, 64, 64, 3, "stats"),
              ("l1.c3 apply", 56, 64, 256, 1, "infer_res"), ("l2.c3 apply", 28, 128, 512, 1, "infer_res"), ("l3.c3 apply", 14, 256, 1024, 1, "infer_res"),
              ("l4.c3 apply", 7, 512, 2048, 1, "infer_res"), ("l3.c3 plain", 14, 256, 1024, 1, "plain"), ("l2.c3 plain", 28, 128, 512, 1, "plain"),
              ("l1.c3 plain", 56, 64, 256, 1, "plain")]
    n = int(os.environ.get("KBENCH_FRAMES", "256"))          # 256 = the C3 / C4 step; other values probe the tile-count quantisation
    for name, hw, cin, cout, k, kind in shapes:
        if only and only not in name:
            continue
        m = n * hw * hw
        d = ConvDesc(n, hw, hw, cin, cout, k, k, 1, k // 2, hw, hw, cin, dt, 0, 0, 0, 0, 0)
        x = torch.randn(m, cin, device=dev).to(tdt)
        wp = (torch.randn(cout, k * k * cin, device=dev) * 0.05).to(tdt)
        y = torch.empty(m, cout, device=dev, dtype=tdt)
        ws = torch.empty(lib.mvf_conv2d_workspace_bytes(None), dtype=torch.uint8, device=dev)
        rows = lib.mvf_conv2d_stats_rows(C.byref(d))
        part = torch.empty(cout, rows, 2, device=dev)
        shift = torch.zeros(cout, device=dev)
        nbytes = esz * (m * cin + m * cout + cout * k * k * cin)
        if kind == "stats":
            fn = lambda: check(lib.mvf_conv2d_nhwc_fwd_stats(C.byref(d), P(x), None, P(wp), P(y), P(part), P(shift), P(ws), ws.numel(), None))  # noqa: E731
        elif kind == "infer_res":      # bias + residual + ReLU (the inference epilogue): what a fused bn3-apply pass of conv3 would cost, minus the sign bits
            res = torch.randn(m, cout, device=dev).to(tdt)
            bias = torch.randn(cout, device=dev)
            nbytes += esz * m * cout
            d = ConvDesc(n, hw, hw, cin, cout, k, k, 1, k // 2, hw, hw, cin, dt, 1, 0, 0, 0, 0)
            fn = lambda: check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), P(x), None, P(wp), P(bias), P(res), P(y), P(ws), ws.numel(), None))  # noqa: E731
        elif kind == "plain":
            fn = lambda: check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), P(x), None, P(wp), None, None, P(y), P(ws), ws.numel(), None))  # noqa: E731
        elif kind == "res":
            res = torch.randn(m, cout, device=dev).to(tdt)
            bits = torch.randint(0, 16, (m, cout // 4), device=dev, dtype=torch.uint8)
            nbytes += esz * m * cout + m * cout // 4
            fn = lambda: check(lib.mvf_conv2d_nhwc_fwd_resmask(C.byref(d), P(x), None, P(wp), None, P(res), P(bits), P(y), P(ws), ws.numel(), None))  # noqa: E731
        else:
            z = torch.randn(m, cout, device=dev).to(tdt)
            v = [torch.rand(cout, device=dev) + 0.5 for _ in range(4)]
            nbytes += esz * m * cout
            fn = lambda: check(lib.mvf_conv2d_nhwc_dgrad_bnsums(C.byref(d), P(x), P(wp), P(y), P(z), P(v[0]), P(v[1]), P(v[2]), P(v[3]), P(part), P(ws), ws.numel(), None))  # noqa: E731
        t = timeit(fn, reps=30, warm=5)
        fl = 2.0 * m * cout * k * k * cin
        print("%-12s %-5s M%-7d K%-5d N%-5d %8.1f us  %6.1f TF/s  %5.2f TB/s" % (name, kind, m, k * k * cin, cout, t, fl / t / 1e6, nbytes / t / 1e6))


def bench_bn(dt=1):
    tdt = torch.bfloat16 if dt else torch.float32
    esz = 2 if dt else 4
    dev = "cuda"
    # (rows, channels) of the block-final tensors (bn3 / residual) and the mid-block ones (bn1 / bn2) per stage, NT = 256
    stages = [("layer1", 256 * 56 * 56, 256, 64), ("layer2", 256 * 28 * 28, 512, 128), ("layer3", 256 * 14 * 14, 1024, 256), ("layer4", 256 * 7 * 7, 2048, 512)]
    for name, m, c, cm in stages:
        for cc, tag in ((c, "final"), (cm, "mid")):
            z = torch.randn(m, cc, device=dev).to(tdt)
            r = torch.randn(m, cc, device=dev).to(tdt)
            g = torch.randn(m, cc, device=dev).to(tdt)
            out, dz = torch.empty_like(z), torch.empty_like(z)
            bits = torch.zeros(m, cc // 4, dtype=torch.uint8, device=dev)
            f = lambda: torch.rand(cc, device=dev) + 0.5  # noqa: E731
            scale, shift, mean, invstd, gamma = f(), f() - 1, f() - 1, f(), f()
            dg, db = torch.empty(cc, device=dev), torch.empty(cc, device=dev)
            ws = torch.empty(lib.mvf_bn_workspace_bytes(m, cc), dtype=torch.uint8, device=dev)
            nb = m * cc * esz
            if tag == "final":
                t = timeit(lambda: check(lib.mvf_bn_apply_bits(P(z), m, cc, P(scale), P(shift), P(r), None, None, 1, P(out), P(bits), dt, None)))
                print("%-7s %-5s apply+res+bits  %8.1f us  %6.2f TB/s" % (name, tag, t, (3 * nb + m * cc // 4) / t / 1e6))
                t = timeit(lambda: check(lib.mvf_bn_apply_bits(P(z), m, cc, P(scale), P(shift), P(r), None, None, 1, P(out), None, dt, None)))
                print("%-7s %-5s apply+res        %8.1f us  %6.2f TB/s   (no sign bits)" % (name, tag, t, 3 * nb / t / 1e6))
                t = timeit(lambda: check(lib.mvf_bn_bwd_reduce(P(g), cc, P(z), P(bits), m, cc, P(mean), P(invstd), P(scale), P(shift), 4, None, P(dg), P(db), P(ws), ws.numel(), dt, None)))
                print("%-7s %-5s bwd_reduce<4>   %8.1f us  %6.2f TB/s" % (name, tag, t, (2 * nb + m * cc // 4) / t / 1e6))
                t = timeit(lambda: check(lib.mvf_bn_bwd_apply_masked(P(g), cc, P(z), P(bits), m, cc, P(gamma), P(mean), P(invstd), P(scale), P(shift), P(dg), P(db), 4, P(dz), dt, None)))
                print("%-7s %-5s bwd_apply<4>    %8.1f us  %6.2f TB/s" % (name, tag, t, (3 * nb + m * cc // 4) / t / 1e6))
            else:
                t = timeit(lambda: check(lib.mvf_bn_apply_bits(P(z), m, cc, P(scale), P(shift), None, None, None, 1, P(out), None, dt, None)))
                print("%-7s %-5s apply           %8.1f us  %6.2f TB/s" % (name, tag, t, 2 * nb / t / 1e6))
                t = timeit(lambda: check(lib.mvf_bn_bwd_apply_masked(P(g), cc, P(z), None, m, cc, P(gamma), P(mean), P(invstd), P(scale), P(shift), P(dg), P(db), 2, P(dz), dt, None)))
                print("%-7s %-5s bwd_apply<2>    %8.1f us  %6.2f TB/s" % (name, tag, t, 3 * nb / t / 1e6))
            del z, r, g, out, dz, bits


def bench_bnwg():
    """[r4] BatchNorm backward apply + weight gradient in one pass (mvf_bn_bwd_apply_wgrad / mvf_bn_bwd_pair_wgrad) against the two kernels it
    replaces (mvf_bn_bwd_apply_masked / mvf_bn_bwd_pair + mvf_conv2d_nhwc_wgrad), each alone on the stream, at the C3 shapes of layer1 / layer2."""
    from mvfnet_amd._lib import ConvDesc
    dev, bf = "cuda", torch.bfloat16
    # (name, pixels, c = channels of dz, k = conv input channels, mask mode, BatchNorms, both convs fused)
    shapes = [("l1.c3 (plain)", 256 * 56 * 56, 256, 64, 4, 1, True), ("l1.0 c3 + downsample (pair)", 256 * 56 * 56, 256, 64, 4, 2, True),
              ("l2.c3 (plain)", 256 * 28 * 28, 512, 128, 4, 1, True), ("l2.0 c3 (pair, downsample apart)", 256 * 28 * 28, 512, 128, 4, 2, False),
              ("l1.c1 (x 256 wide)", 256 * 56 * 56, 64, 256, 2, 1, True), ("l2.c1 (x 512 wide)", 256 * 28 * 28, 128, 512, 2, 1, True),
              ("l2.0 c1 (x 256 wide)", 256 * 56 * 56, 128, 256, 2, 1, True)]
    for name, m, c, k, mode, nbn, both in shapes:
        g = torch.randn(m, c, device=dev).to(bf)
        z = [(torch.randn(m, c, device=dev) * 1.3).to(bf) for _ in range(nbn)]
        x = [torch.randn(m, k, device=dev).to(bf) for _ in range(nbn)]
        bits = torch.randint(0, 16, (m, c // 4), device=dev, dtype=torch.uint8)
        par = [[(torch.rand(c, device=dev) + 0.5) for _ in range(7)] for _ in range(nbn)]      # gamma mean invstd scale shift dgamma dbeta
        dz = [torch.empty(m, c, device=dev, dtype=bf) for _ in range(nbn)]
        ws = torch.empty(2 * lib.mvf_bn_workspace_bytes(m, c), dtype=torch.uint8, device=dev)
        d = ConvDesc(1, m, 1, k, c, 1, 1, 1, 0, m, 1, k, 1, 0, 0, 0, 0, 0)
        wsz = lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d))
        wws = torch.empty(wsz, dtype=torch.uint8, device=dev)
        dw = [torch.empty(c, k, device=dev) for _ in range(nbn)]
        ns = lib.mvf_bn_bwd_wgrad_splits(m, c, k, nbn, mode)
        nb = lib.mvf_bn_bwd_wgrad_slab_bytes(m, c, k, nbn, mode)
        slabs = [torch.empty(nb // 4, device=dev) for _ in range(nbn)]
        ym = bits if mode == 4 else None

        def apply_only():
            if nbn == 1:
                p = par[0]
                check(lib.mvf_bn_bwd_apply_masked(P(g), c, P(z[0]), P(ym), m, c, P(p[0]), P(p[1]), P(p[2]), P(p[3]), P(p[4]), P(p[5]), P(p[6]), mode, P(dz[0]), 1, None))
            else:
                a, b = par
                check(lib.mvf_bn_bwd_pair(P(g), c, P(z[0]), P(z[1]), P(bits), m, c, P(a[0]), P(a[1]), P(a[2]), P(a[5]), P(a[6]), P(b[0]), P(b[1]), P(b[2]), P(b[5]), P(b[6]),
                                          P(dz[0]), P(dz[1]), P(ws), ws.numel(), 1, None))

        def wgrads():
            for i in range(nbn if both else 1):
                check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), P(dz[i]), P(x[i]), None, 1, k, 1, k, P(dw[i]), P(wws), wsz, None))

        def fused():
            if nbn == 1:
                p = par[0]
                check(lib.mvf_bn_bwd_apply_wgrad(P(g), c, P(z[0]), P(ym), m, c, P(p[0]), P(p[1]), P(p[2]), P(p[3]), P(p[4]), P(p[5]), P(p[6]), mode, P(dz[0]),
                                                 P(x[0]), k, k, P(slabs[0]), nb, 1, None))
            else:
                a, b = par
                check(lib.mvf_bn_bwd_pair_wgrad(P(g), c, P(z[0]), P(z[1]), P(bits), m, c, P(a[0]), P(a[1]), P(a[2]), P(a[5]), P(a[6]), P(b[0]), P(b[1]), P(b[2]), P(b[5]), P(b[6]),
                                                P(dz[0]), P(dz[1]), P(x[0]), k, P(x[1]) if both else None, k, k, P(slabs[0]), P(slabs[1]) if both else None, nb,
                                                P(ws), ws.numel(), 1, None))

        def reduces():
            for i in range(nbn if both else 1):
                check(lib.mvf_wgrad_slab_reduce(P(slabs[i]), ns, c, k, P(dw[i]), None))

        ta, tw, tf, tr = timeit(apply_only), timeit(wgrads), timeit(fused), timeit(reduces)
        nx = nbn if both else 1
        by_f = 2 * m * c * (1 + 2 * nbn) + (m * c // 4 if mode == 4 else 0) + 2 * m * k * nx        # g + z + dz (+ bits) + x
        if nbn == 2:
            by_f += 2 * m * c * 3 + m * c // 4                                                     # the pair's reduce pass: g, z_a, z_b, bits
        print("%-36s M %7d c %4d k %4d  apply %6.1f + wgrad %6.1f = %6.1f us | fused %6.1f + slab reduce %5.1f us (%d splits)  %5.2f TB/s  saves %6.1f us" % (
            name, m, c, k, ta, tw, ta + tw, tf, tr, ns, by_f / tf / 1e6, ta + tw - tf - tr))


def bench_c3bwd():
    """[r4] The one-pass backward of a z3-free block's last conv (mvf_conv1x1_bwd_fused, csrc/pw_bwd_fused.hip) against the three launches it replaces
    (conv + BatchNorm-backward apply, data gradient + BatchNorm sums, weight gradient), each alone on the stream, at layer1's C3 shape."""
    from mvfnet_amd._lib import ConvDesc
    dev, bf = "cuda", torch.bfloat16
    cin, cout = 64, 256
    for frames in (int(os.environ.get("KBENCH_FRAMES", "256")),):
        m = frames * 56 * 56
        a2 = torch.relu(torch.randn(m, cin, device=dev)).to(bf)
        g = torch.randn(m, cout, device=dev).to(bf)
        z2 = torch.randn(m, cin, device=dev).to(bf)
        bits = torch.randint(0, 16, (m, cout // 4), device=dev, dtype=torch.uint8)
        w = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
        wp = torch.empty(cout, 1, 1, cin, dtype=bf, device=dev)
        wd = torch.empty(cin, 1, 1, cout, dtype=bf, device=dev)
        check(lib.mvf_pack_conv_weight(P(w), cout, cin, 1, 1, 1, cin, None, P(wp), 1, None))
        check(lib.mvf_pack_conv_weight_dgrad(P(w), cout, cin, 1, 1, P(wd), 1, None))
        p3 = [(torch.rand(cout, device=dev) + 0.5) for _ in range(5)]          # gamma mean invstd dgamma dbeta
        p2 = [(torch.rand(cin, device=dev) + 0.5) for _ in range(4)]           # mean invstd scale shift
        d = ConvDesc(frames, 56, 56, cin, cout, 1, 1, 1, 0, 56, 56, cin, 1, 0, 0, 0, 0, 0)
        dd = ConvDesc(frames, 56, 56, cout, cin, 1, 1, 1, 0, 56, 56, cout, 1, 0, 0, 0, 0, 0)
        ws = torch.empty(lib.mvf_conv2d_workspace_bytes(None), dtype=torch.uint8, device=dev)
        dz3 = torch.empty(m, cout, dtype=bf, device=dev)
        dx = torch.empty(m, cin, dtype=bf, device=dev)
        rows2 = lib.mvf_conv2d_stats_rows(C.byref(dd))
        part = torch.empty(rows2, cin, 2, device=dev)
        wsz = lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d))
        wws = torch.empty(wsz, dtype=torch.uint8, device=dev)
        dw = torch.empty(cout, cin, device=dev)
        ns = lib.mvf_conv1x1_bwd_fused_splits(m, cout, cin)
        spart = torch.empty(cin, 2 * ns, 2, device=dev)
        slabs = torch.empty(ns * cout * cin, device=dev)

        def apply_():
            check(lib.mvf_conv2d_nhwc_fwd_bnbwd_apply(C.byref(d), P(a2), None, P(wp), P(g), P(bits), P(p3[0]), P(p3[1]), P(p3[2]), P(p3[3]), P(p3[4]), P(dz3),
                                                      P(ws), ws.numel(), None))

        def dgrad():
            check(lib.mvf_conv2d_nhwc_dgrad_bnsums(C.byref(dd), P(dz3), P(wd), P(dx), P(z2), P(p2[0]), P(p2[1]), P(p2[2]), P(p2[3]), P(part), P(ws), ws.numel(), None))

        def wgrad():
            check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), P(dz3), P(a2), None, 1, cin, 1, cin, P(dw), P(wws), wsz, None))

        def fused():
            check(lib.mvf_conv1x1_bwd_fused(P(a2), cin, P(wp), P(g), cout, P(bits), m, cout, cin, P(p3[0]), P(p3[1]), P(p3[2]), P(p3[3]), P(p3[4]), P(z2),
                                            P(p2[0]), P(p2[1]), P(p2[2]), P(p2[3]), P(dx), P(spart), 2 * ns, P(slabs), slabs.numel() * 4, 1, None))

        def reduce_():
            check(lib.mvf_wgrad_slab_reduce(P(slabs), ns, cout, cin, P(dw), None))

        ta, td, tw, tf, tr = timeit(apply_), timeit(dgrad), timeit(wgrad), timeit(fused), timeit(reduce_)
        by_f = 2 * m * (cout + 3 * cin) + m * cout // 4
        print("l1.c3 backward  M %7d  apply %6.1f + dgrad+bn %6.1f + wgrad %6.1f = %6.1f us | fused %6.1f + slab reduce %5.1f us (%d splits)  %5.2f TB/s %6.1f TF/s  saves %6.1f us"
              % (m, ta, td, tw, ta + td + tw, tf, tr, ns, by_f / tf / 1e6, 6.0 * m * cout * cin / tf / 1e6, ta + td + tw - tf - tr))


def bench_wgrad(dt=0, only=None):
    """Single weight-gradient launches (kernel + slab reduce) at the C3 shapes, fp32 by default (`wgrad` / `wgrad16`); A/B the fp32 kernels
    with MVF_POLICY=wgrad_x3=0 (the exact-fp32 MFMA kernel) against the default (three-term bf16 splits on the bf16 matrix cores)."""
    from mvfnet_amd._lib import ConvDesc
    tdt = torch.bfloat16 if dt else torch.float32
    dev = "cuda"
    shapes = [("l1.c1", 56, 256, 64, 1), ("l1.c2", 56, 64, 64, 3), ("l1.c3", 56, 64, 256, 1), ("l2.c1", 28, 512, 128, 1), ("l2.c2", 28, 128, 128, 3),
              ("l2.c3", 28, 128, 512, 1), ("l3.c1", 14, 1024, 256, 1), ("l3.c2", 14, 256, 256, 3), ("l3.c3", 14, 256, 1024, 1),
              ("l4.c1", 7, 2048, 512, 1), ("l4.c2", 7, 512, 512, 3), ("l4.c3", 7, 512, 2048, 1)]
    n = int(os.environ.get("KBENCH_FRAMES", "256"))
    tot = 0.0
    for name, hw, cin, cout, k in shapes:
        if only and only not in name:
            continue
        m = n * hw * hw
        d = ConvDesc(n, hw, hw, cin, cout, k, k, 1, k // 2, hw, hw, cin, dt, 0, 0, 0, 0, 0)
        x = torch.randn(m, cin, device=dev).to(tdt)
        dz = torch.randn(m, cout, device=dev).to(tdt)
        ws = torch.empty(lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
        dw = torch.empty(cout, cin, k, k, device=dev)
        fn = lambda: check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), P(dz), P(x), None, k, cin, k, cin, P(dw), P(ws), ws.numel(), None))  # noqa: E731
        t = timeit(fn, reps=20, warm=3)
        fl = 2.0 * m * cout * k * k * cin
        tot += t
        print("%-6s M%-7d K%-5d N%-5d %8.1f us  %6.1f TF/s" % (name, m, k * k * cin, cout, t, fl / t / 1e6))
    print("sum %.1f us" % tot)


def bench_mvf():
    """[r4] The engine's MVF stencil (mvf_nhwc_stencil: plain, and transposed + gated addend as the backward runs it) on the layer3 / layer4 shapes,
    reading its slice (a) where it lies today -- the first cs channels of every c-channel pixel row (256 B of every 2 KB) -- and (b) from a COMPACT
    [m][cs] copy: what a producer-side slice copy could buy."""
    from mvfnet_amd._lib import MvfDesc
    dev, bf = "cuda", torch.bfloat16
    for name, hw, c in (("layer3.1+", 14, 1024), ("layer4.1+", 7, 2048), ("layer3.0", 28, 512)):
        nt, T, cs = 256, 8, c // 8
        m = nt * hw * hw
        d = MvfDesc(nt, c, hw, hw, T, cs, 7, _lib.MVF_NHWC, 1)
        x = torch.randn(m, c, device=dev).to(bf)
        xs = x[:, :cs].contiguous()
        y = torch.empty(m, cs, device=dev, dtype=bf)
        dxp = torch.randn(m, c, device=dev).to(bf)
        g = torch.randn(m, c, device=dev).to(bf)
        gs = g[:, :cs].contiguous()
        bits = torch.randint(0, 16, (m, c // 4), device=dev, dtype=torch.uint8)
        w = [torch.randn(cs, 3, device=dev) for _ in range(3)]
        by = 2 * m * cs * 2
        t0 = timeit(lambda: check(lib.mvf_nhwc_stencil(C.byref(d), P(x), c, P(y), cs, P(w[0]), P(w[1]), P(w[2]), None, None, 0, None, 0, None, None)))
        t1 = timeit(lambda: check(lib.mvf_nhwc_stencil(C.byref(d), P(xs), cs, P(y), cs, P(w[0]), P(w[1]), P(w[2]), None, None, 0, None, 0, None, None)))
        print("%-10s M%-7d cs %-4d stencil   in place %6.1f us (%4.2f TB/s)   compact input %6.1f us (%4.2f TB/s)" % (name, m, cs, t0, by / t0 / 1e6, t1, by / t1 / 1e6))
        by = 3 * m * cs * 2
        t0 = timeit(lambda: check(lib.mvf_nhwc_stencil(C.byref(d), P(y), cs, P(dxp), c, P(w[0]), P(w[1]), P(w[2]), None, None, 1, P(g), c, P(bits), None)))
        t1 = timeit(lambda: check(lib.mvf_nhwc_stencil(C.byref(d), P(y), cs, P(xs), cs, P(w[0]), P(w[1]), P(w[2]), None, None, 1, P(gs), cs, None, None)))
        print("%-10s M%-7d cs %-4d stencil^T in place %6.1f us (%4.2f TB/s)   compact in / out / addend (no gate) %6.1f us (%4.2f TB/s)" % (name, m, cs, t0, by / t0 / 1e6, t1, by / t1 / 1e6))
        # the floor of ANY kernel of this size: a plain copy of the compact slice
        t2 = timeit(lambda: y.copy_(xs))
        print("%-10s            copy of the compact slice (torch) %6.1f us (%4.2f TB/s)" % (name, t2, 2 * m * cs * 2 / t2 / 1e6))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "bn"
    if what == "mvf":
        bench_mvf()
        sys.exit(0)
    if what in ("wgrad", "wgrad16"):
        bench_wgrad(1 if what == "wgrad16" else 0, sys.argv[2] if len(sys.argv) > 2 else None)
        sys.exit(0)
    if what == "bnwg":
        bench_bnwg()
        sys.exit(0)
    if what == "c3bwd":
        bench_c3bwd()
        sys.exit(0)
    if what == "bn":
        bench_bn(1)
    elif what == "conv":
        bench_conv(1, sys.argv[2] if len(sys.argv) > 2 else None)
