"""GPU parity of the HIP training path: primitives vs the CPU oracle (torch autograd on CPU = the reference's own
arithmetic), a bottleneck block vs golden vectors, and the whole train step vs golden vectors."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from cases import BLOCK_CASES
from helpers import bf16_storage_oracle, golden, policy_env, rel_err, rel_l2
from mvfnet_amd import synth

pytestmark = pytest.mark.gpu


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


# ------------------------------------------------------------------------------------------------ primitives
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 14, 10, 64), (2, 9, 12, 8), (5, 21, 17, 32)], ids=str)
def test_maxpool_bwd_with_bn_sums_equals_scatter_then_reduce(shape, dtype):
    """Stem backward, pool(relu(bn(z))) (reference resnet.py:461-466): the scatter that also accumulates the BatchNorm's backward
    sums writes the same ga bit for bit, and dgamma / dbeta equal the separate mvf_bn_bwd_reduce pass over (ga, z)."""
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    n, h, w, c = shape
    dt = 0 if dtype == torch.float32 else 1
    gen = torch.Generator().manual_seed(h * w + c)
    dev = "cuda"
    z = (torch.randn(n, h, w, c, generator=gen) * 1.3 + 0.2).to(dev, dtype)
    m = n * h * w
    gamma, beta = (torch.rand(c, generator=gen) + 0.5).to(dev), (torch.randn(c, generator=gen) * 0.3).to(dev)
    mean, invstd, scale, shift = (torch.empty(c, device=dev) for _ in range(4))
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    ws = torch.empty(lib.mvf_bn_workspace_bytes(m, c), dtype=torch.uint8, device=dev)
    check(lib.mvf_bn_train_stats(P(z), m, c, P(gamma), P(beta), C.c_float(1e-5), C.c_float(0.1), P(rm), P(rv), P(mean), P(invstd), P(scale), P(shift),
                                 P(ws), ws.numel(), dt, None))
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty(n, ho, wo, c, device=dev, dtype=dtype)
    am = torch.empty(n, ho, wo, c, device=dev, dtype=torch.uint8)
    check(lib.mvf_maxpool_bn_relu_fwd(P(z), n, h, w, c, P(scale), P(shift), P(out), P(am), dt, None))
    g = torch.randn(n, ho, wo, c, generator=gen).to(dev, dtype)
    ga_ref = torch.empty_like(z)
    check(lib.mvf_maxpool_bn_relu_bwd(P(am), P(g), n, h, w, c, P(ga_ref), dt, None))
    dg_ref, db_ref = torch.empty(c, device=dev), torch.empty(c, device=dev)
    check(lib.mvf_bn_bwd_reduce(P(ga_ref), c, P(z), None, m, c, P(mean), P(invstd), P(scale), P(shift), 2, None, P(dg_ref), P(db_ref), P(ws), ws.numel(), dt, None))
    rows = lib.mvf_maxpool_bwd_sums_rows(n, h)
    assert rows == (n * h + 7) // 8
    part = torch.full((c, rows, 2), float("nan"), device=dev)
    ga = torch.empty_like(z)
    check(lib.mvf_maxpool_bn_relu_bwd_sums(P(am), P(g), n, h, w, c, P(ga), P(z), P(mean), P(invstd), P(scale), P(shift), P(part), dt, None))
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    check(lib.mvf_bn_bwd_finalize(P(part), rows, c, P(dg), P(db), None))
    torch.cuda.synchronize()
    assert torch.equal(ga, ga_ref)
    assert torch.isfinite(part).all()
    assert rel_err(dg.cpu().numpy(), dg_ref.cpu().numpy()) < 2e-5 and rel_err(db.cpu().numpy(), db_ref.cpu().numpy()) < 2e-5
    bad = lib.mvf_maxpool_bn_relu_bwd_sums(P(am), P(g), n, h, w, 12, P(ga), P(z), P(mean), P(invstd), P(scale), P(shift), P(part), dt, None)
    assert bad == -5          # MVF_EUNSUPPORTED: 256 % (c/4) != 0
    # without a materialised ga: same sums, and the apply pass that gathers again equals the masked apply over the stored ga
    part2 = torch.full((c, rows, 2), float("nan"), device=dev)
    check(lib.mvf_maxpool_bn_relu_bwd_sums(P(am), P(g), n, h, w, c, None, P(z), P(mean), P(invstd), P(scale), P(shift), P(part2), dt, None))
    assert torch.equal(part2, part)
    dz_ref, dz = torch.empty_like(z), torch.empty_like(z)
    check(lib.mvf_bn_bwd_apply_masked(P(ga_ref), c, P(z), None, m, c, P(gamma), P(mean), P(invstd), P(scale), P(shift), P(dg), P(db), 2, P(dz_ref), dt, None))
    check(lib.mvf_maxpool_bn_relu_bwd_apply(P(am), P(g), n, h, w, c, P(z), P(gamma), P(mean), P(invstd), P(scale), P(shift), P(dg), P(db), P(dz), dt, None))
    torch.cuda.synchronize()
    assert torch.equal(dz, dz_ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("mc", [(8 * 28 * 28, 512), (1000, 256), (37, 12), (16 * 56 * 56, 256), (301, 2048)], ids=str)
def test_bn_bwd_pair_equals_two_separate_backwards(mc, dtype):
    """Downsample block, out = relu(bn3(z3) + bnd(zd)) (reference resnet.py:227-233): the paired backward reads g and the sign
    bits once for both BatchNorms and must reproduce the two separate passes bit for bit (and, in fp32, torch autograd)."""
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    m, c = mc
    dt = 0 if dtype == torch.float32 else 1
    gen = torch.Generator().manual_seed(m + c)
    za, zb = torch.randn(m, c, generator=gen) * 1.5 + 0.3, torch.randn(m, c, generator=gen) * 0.7 - 0.2
    ga, ba, gb, bb = (torch.rand(c, generator=gen) + 0.5 for _ in range(4))
    g = torch.randn(m, c, generator=gen)
    dev = "cuda"
    zag, zbg, gg = za.to(dev, dtype), zb.to(dev, dtype), g.to(dev, dtype)
    ws = torch.empty(2 * lib.mvf_bn_workspace_bytes(m, c), dtype=torch.uint8, device=dev)
    st = {}
    for k, z, gam, bet in (("a", zag, ga, ba), ("b", zbg, gb, bb)):
        d = dict(gamma=gam.to(dev), beta=bet.to(dev), rm=torch.zeros(c, device=dev), rv=torch.ones(c, device=dev))
        for n in ("mean", "invstd", "scale", "shift"):
            d[n] = torch.empty(c, device=dev)
        check(lib.mvf_bn_train_stats(P(z), m, c, P(d["gamma"]), P(d["beta"]), C.c_float(1e-5), C.c_float(0.1), P(d["rm"]), P(d["rv"]), P(d["mean"]), P(d["invstd"]),
                                     P(d["scale"]), P(d["shift"]), P(ws), ws.numel(), dt, None))
        st[k] = d
    out, bits = torch.empty_like(zag), torch.empty(m, c // 4, dtype=torch.uint8, device=dev)
    check(lib.mvf_bn_apply_bits(P(zag), m, c, P(st["a"]["scale"]), P(st["a"]["shift"]), P(zbg), P(st["b"]["scale"]), P(st["b"]["shift"]), 1, P(out), P(bits), dt, None))
    sep = {}
    for k, z in (("a", zag), ("b", zbg)):
        d = st[k]
        dg, db, dz = torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty_like(z)
        check(lib.mvf_bn_bwd_reduce(P(gg), c, P(z), P(bits), m, c, P(d["mean"]), P(d["invstd"]), P(d["scale"]), P(d["shift"]), 4, None, P(dg), P(db), P(ws), ws.numel(), dt, None))
        check(lib.mvf_bn_bwd_apply_masked(P(gg), c, P(z), P(bits), m, c, P(d["gamma"]), P(d["mean"]), P(d["invstd"]), P(d["scale"]), P(d["shift"]), P(dg), P(db), 4, P(dz), dt, None))
        sep[k] = (dg, db, dz)
    pa = [torch.empty(c, device=dev) for _ in range(4)]
    dza, dzb = torch.full_like(zag, 7.0), torch.full_like(zbg, 7.0)
    a, b = st["a"], st["b"]
    check(lib.mvf_bn_bwd_pair(P(gg), c, P(zag), P(zbg), P(bits), m, c, P(a["gamma"]), P(a["mean"]), P(a["invstd"]), P(pa[0]), P(pa[1]),
                              P(b["gamma"]), P(b["mean"]), P(b["invstd"]), P(pa[2]), P(pa[3]), P(dza), P(dzb), P(ws), ws.numel(), dt, None))
    torch.cuda.synchronize()
    assert torch.equal(pa[0], sep["a"][0]) and torch.equal(pa[1], sep["a"][1]) and torch.equal(pa[2], sep["b"][0]) and torch.equal(pa[3], sep["b"][1])
    assert torch.equal(dza, sep["a"][2]) and torch.equal(dzb, sep["b"][2])
    small = lib.mvf_bn_bwd_pair(P(gg), c, P(zag), P(zbg), P(bits), m, c, P(a["gamma"]), P(a["mean"]), P(a["invstd"]), P(pa[0]), P(pa[1]),
                                P(b["gamma"]), P(b["mean"]), P(b["invstd"]), P(pa[2]), P(pa[3]), P(dza), P(dzb), P(ws), ws.numel() // 2 - 256, dt, None)
    assert small == -3          # MVF_EWS
    if dtype == torch.float32 and m <= 1000:
        zat, zbt = za.clone().requires_grad_(True), zb.clone().requires_grad_(True)
        gat, gbt = ga.clone().requires_grad_(True), gb.clone().requires_grad_(True)
        y = F.relu(F.batch_norm(zat, None, None, gat, ba, True, 0.1, 1e-5) + F.batch_norm(zbt, None, None, gbt, bb, True, 0.1, 1e-5))
        y.backward(g)
        assert rel_err(dza.cpu().numpy(), zat.grad.numpy()) < 5e-5 and rel_err(dzb.cpu().numpy(), zbt.grad.numpy()) < 5e-5
        assert rel_err(pa[0].cpu().numpy(), gat.grad.numpy()) < 5e-5 and rel_err(pa[2].cpu().numpy(), gbt.grad.numpy()) < 5e-5


@pytest.mark.parametrize("shape", [(4, 9, 7, 16), (8, 14, 14, 64), (2, 5, 5, 260)], ids=str)
def test_bn_train_forward_backward_vs_oracle(shape):
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    n, h, w, c = shape
    g = torch.Generator().manual_seed(c)
    z = torch.randn(n, c, h, w, generator=g) * 2 + torch.randn(1, c, 1, 1, generator=g) * 3     # large means: stresses the variance
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    rm, rv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    zt = z.clone().requires_grad_(True)
    gt, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.relu(F.batch_norm(zt, rm_ref, rv_ref, gt, bt, True, 0.1, 1e-5))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    m = n * h * w
    zg = nhwc(z).cuda().view(m, c)
    dev = "cuda"
    rmg, rvg, gg, bg = rm.cuda(), rv.cuda(), gamma.cuda(), beta.cuda()
    mean, invstd, scale, shift = (torch.empty(c, device=dev) for _ in range(4))
    ws = torch.empty(lib.mvf_bn_workspace_bytes(m, c), dtype=torch.uint8, device=dev)
    check(lib.mvf_bn_train_stats(P(zg), m, c, P(gg), P(bg), C.c_float(1e-5), C.c_float(0.1), P(rmg), P(rvg), P(mean), P(invstd), P(scale), P(shift),
                                 P(ws), ws.numel(), 0, None))
    out = torch.empty_like(zg)
    check(lib.mvf_bn_apply(P(zg), m, c, P(scale), P(shift), None, None, None, 1, P(out), 0, None))
    assert rel_err(out.view(n, h, w, c).permute(0, 3, 1, 2).cpu().numpy(), y.detach().numpy()) < 1e-5
    assert rel_err(rmg.cpu().numpy(), rm_ref.numpy()) < 1e-5 and rel_err(rvg.cpu().numpy(), rv_ref.numpy()) < 1e-5
    dyg = nhwc(dy).cuda().view(m, c)
    dgam, dbet = torch.empty(c, device=dev), torch.empty(c, device=dev)
    check(lib.mvf_bn_bwd_reduce(P(dyg), c, P(zg), None, m, c, P(mean), P(invstd), P(scale), P(shift), 2, None, P(dgam), P(dbet), P(ws), ws.numel(), 0, None))
    dz = torch.empty_like(zg)
    check(lib.mvf_bn_bwd_apply(P(dyg), c, P(zg), m, c, P(gg), P(mean), P(invstd), P(scale), P(shift), P(dgam), P(dbet), 2, P(dz), 0, None))
    assert rel_err(dgam.cpu().numpy(), gt.grad.numpy()) < 5e-5
    assert rel_err(dbet.cpu().numpy(), bt.grad.numpy()) < 5e-5
    assert rel_err(dz.view(n, h, w, c).permute(0, 3, 1, 2).cpu().numpy(), zt.grad.numpy()) < 5e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("mc", [(300, 64), (77, 260), (1000, 256)], ids=str)
def test_bn_sign_bits_equal_the_tensor_mask(mc, dtype):
    """mvf_bn_apply_bits writes bit j of byte [row][k] <=> out[row][4k+j] > 0, and mask_mode 4 (bits) of the backward
    kernels, the conv residual gate and the stencil addend gate reproduce mask_mode 1 (the tensor itself) exactly."""
    from mvfnet_amd import _lib
    from mvfnet_amd._lib import ConvDesc, MvfDesc
    lib, check = _lib.lib, _lib.check
    m, c = mc
    if dtype == torch.bfloat16 and c % 8:
        pytest.skip("bf16 engine tensors have c % 8 == 0")
    dt = 0 if dtype == torch.float32 else 1
    g = torch.Generator().manual_seed(m + c)
    dev = "cuda"
    z = torch.randn(m, c, generator=g).to(dtype).cuda()
    res = torch.randn(m, c, generator=g).to(dtype).cuda()
    scale, shift = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    out, bits = torch.empty_like(z), torch.zeros(m, c // 4, dtype=torch.uint8, device=dev)
    check(lib.mvf_bn_apply_bits(P(z), m, c, P(scale), P(shift), P(res), None, None, 1, P(out), P(bits), dt, None))
    want = (out.float() > 0).view(m, c // 4, 4).to(torch.uint8)
    packed = want[..., 0] | (want[..., 1] << 1) | (want[..., 2] << 2) | (want[..., 3] << 3)
    assert torch.equal(bits, packed)
    # backward reductions / apply: bits == tensor mask
    gy = torch.randn(m, c, generator=g).to(dtype).cuda()
    mean, invstd, gamma = torch.randn(c, generator=g).cuda(), (torch.rand(c, generator=g) + 0.5).cuda(), (torch.rand(c, generator=g) + 0.5).cuda()
    ws = torch.empty(lib.mvf_bn_workspace_bytes(m, c), dtype=torch.uint8, device=dev)
    sums = []
    for mode, ym in ((1, out), (4, bits)):
        dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
        check(lib.mvf_bn_bwd_reduce(P(gy), c, P(z), P(ym), m, c, P(mean), P(invstd), P(scale), P(shift), mode, None, P(dg), P(db), P(ws), ws.numel(), dt, None))
        dz = torch.empty_like(z)
        check(lib.mvf_bn_bwd_apply_masked(P(gy), c, P(z), P(ym), m, c, P(gamma), P(mean), P(invstd), P(scale), P(shift), P(dg), P(db), mode, P(dz), dt, None))
        sums.append((dg.clone(), db.clone(), dz.clone()))
    torch.cuda.synchronize()
    for a, b in zip(sums[0], sums[1]):
        assert torch.equal(a, b)
    # conv epilogue: residual gated by the bits (1x1 conv, identity-free check against an explicit masked residual), skip columns < res_c0
    n_, h_, w_ = 1, 1, m
    cin = 64
    xw = (torch.randn(m, cin, generator=g) * 0.1).to(dtype).cuda()
    wgt = (torch.randn(c, cin, generator=g) * 0.1).cuda()
    wp = torch.empty(c, 1, 1, cin, dtype=dtype, device=dev)
    check(lib.mvf_pack_conv_weight(P(wgt), c, cin, 1, 1, 1, cin, None, P(wp), dt, None))
    ws2 = torch.empty(max(lib.mvf_conv2d_workspace_bytes(C.byref(ConvDesc(n_, h_, w_, cin, c, 1, 1, 1, 0, h_, w_, cin, dt, 0, 0, 0, 0))), 1), dtype=torch.uint8, device=dev)
    ws2.zero_()
    for res_c0 in (0, 8):
        d = ConvDesc(n_, h_, w_, cin, c, 1, 1, 1, 0, h_, w_, cin, dt, 0, 0, 0, 0, res_c0)
        y1, y2 = torch.empty(m, c, dtype=dtype, device=dev), torch.empty(m, c, dtype=dtype, device=dev)
        check(lib.mvf_conv2d_nhwc_fwd_resmask(C.byref(d), P(xw), None, P(wp), None, P(gy), P(bits), P(y1), P(ws2), ws2.numel(), None))
        gm = (gy.float() * (out.float() > 0)).to(dtype)
        gm[:, :res_c0] = 0
        d0 = ConvDesc(n_, h_, w_, cin, c, 1, 1, 1, 0, h_, w_, cin, dt, 0, 0, 0, 0, 0)
        check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d0), P(xw), None, P(wp), None, P(gm), P(y2), P(ws2), ws2.numel(), None))
        torch.cuda.synchronize()
        assert torch.equal(y1, y2), res_c0
    # stencil addend gate: out = stencil(x) + addend * mask on the first cs channels
    T, hh, ww, cs = 2, 5, m // 10 if m % 10 == 0 else 1, 8
    nt = T                       # one clip of two frames, 5 x (m/10) pixels each
    ran_stencil = hh * ww * nt == m
    assert ran_stencil or m % 10, "stencil sub-check must run for the m % 10 == 0 cases"
    if ran_stencil:
        md = MvfDesc(nt, c, hh, ww, T, cs, 7, _lib.MVF_NHWC, dt)
        xs = torch.randn(m, cs, generator=g).to(dtype).cuda()
        wt3 = torch.randn(cs, 3, generator=g).cuda()
        o1, o2 = torch.zeros(m, c, dtype=dtype, device=dev), torch.zeros(m, c, dtype=dtype, device=dev)
        check(lib.mvf_nhwc_stencil(C.byref(md), P(xs), cs, P(o1), c, P(wt3), P(wt3), P(wt3), None, None, 1, P(gy), c, P(bits), None))
        gmf = (gy.float() * (out.float() > 0)).to(dtype)
        check(lib.mvf_nhwc_stencil(C.byref(md), P(xs), cs, P(o2), c, P(wt3), P(wt3), P(wt3), None, None, 1, P(gmf), c, None, None))
        torch.cuda.synchronize()
        assert torch.equal(o1[:, :cs], o2[:, :cs])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 9, 7, 256, 64, 1), (3, 10, 10, 64, 64, 3), (1, 20, 20, 128, 128, 3)], ids=str)
def test_dgrad_bnsums_equals_separate_reduce(shape, dtype):
    """mvf_conv2d_nhwc_dgrad_bnsums: same data gradient bit for bit, and the epilogue's BatchNorm-backward sums equal
    mvf_bn_bwd_reduce (mask mode 2) on that output."""
    from mvfnet_amd import _lib
    from mvfnet_amd._lib import ConvDesc
    lib, check = _lib.lib, _lib.check
    n, h, w, cout, cin, k = shape
    dt = 0 if dtype == torch.float32 else 1
    g = torch.Generator().manual_seed(1)
    pad, m = k // 2, n * h * w
    dz = (torch.randn(m, cout, generator=g) * 0.5).to(dtype).cuda()
    wgt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).cuda()
    wd = torch.empty(cin, k, k, cout, dtype=dtype, device="cuda")
    check(lib.mvf_pack_conv_weight_dgrad(P(wgt), cout, cin, k, k, P(wd), dt, None))
    z = torch.randn(m, cin, generator=g).to(dtype).cuda()
    mean, invstd = torch.randn(cin, generator=g).cuda() * 0.1, (torch.rand(cin, generator=g) + 0.5).cuda()
    scale, shift = (torch.rand(cin, generator=g) + 0.5).cuda(), torch.randn(cin, generator=g).cuda() * 0.3
    d = ConvDesc(n, h, w, cout, cin, k, k, 1, k - 1 - pad, h, w, cout, dt, 0, 0, 0, 0, 0)
    ws = torch.zeros(max(lib.mvf_conv2d_workspace_bytes(C.byref(d)), 1), dtype=torch.uint8, device="cuda")
    y1, y2 = torch.empty(m, cin, dtype=dtype, device="cuda"), torch.empty(m, cin, dtype=dtype, device="cuda")
    rows = lib.mvf_conv2d_stats_rows(C.byref(d))
    part = torch.zeros(rows, cin, 2, device="cuda")
    check(lib.mvf_conv2d_nhwc_dgrad_bnsums(C.byref(d), P(dz), P(wd), P(y1), P(z), P(mean), P(invstd), P(scale), P(shift), P(part), P(ws), ws.numel(), None))
    dg, db = torch.empty(cin, device="cuda"), torch.empty(cin, device="cuda")
    check(lib.mvf_bn_bwd_finalize(P(part), rows, cin, P(dg), P(db), None))
    check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), P(dz), None, P(wd), None, None, P(y2), P(ws), ws.numel(), None))
    ws2 = torch.empty(lib.mvf_bn_workspace_bytes(m, cin), dtype=torch.uint8, device="cuda")
    dg2, db2 = torch.empty(cin, device="cuda"), torch.empty(cin, device="cuda")
    check(lib.mvf_bn_bwd_reduce(P(y2), cin, P(z), None, m, cin, P(mean), P(invstd), P(scale), P(shift), 2, None, P(dg2), P(db2), P(ws2), ws2.numel(), dt, None))
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    assert rel_err(dg.cpu().numpy(), dg2.cpu().numpy()) < 2e-6 and rel_err(db.cpu().numpy(), db2.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 13, 11, 128, 64, 3, 2), (3, 8, 8, 64, 128, 1, 2), (1, 28, 28, 256, 256, 3, 2), (2, 7, 9, 64, 64, 3, 2)], ids=str)
def test_strided_dgrad_bnsums_equals_separate_reduce(shape, dtype):
    """The same for a stride-2 conv's data gradient, which runs as four parity classes scattering into the full-resolution map:
    identical gradient, and the four runs of partial rows finalise to mvf_bn_bwd_reduce's sums."""
    from mvfnet_amd import _lib
    from mvfnet_amd._lib import ConvDesc
    lib, check = _lib.lib, _lib.check
    n, h, w, cout, cin, k, st = shape                      # h, w = the conv's INPUT map (= the data gradient's output map)
    dt = 0 if dtype == torch.float32 else 1
    g = torch.Generator().manual_seed(3)
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // st + 1, (w + 2 * pad - k) // st + 1
    m_out, m_in = n * ho * wo, n * h * w
    dz = (torch.randn(m_out, cout, generator=g) * 0.5).to(dtype).cuda()
    wgt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).cuda()
    wd = torch.empty(cin, k, k, cout, dtype=dtype, device="cuda")
    check(lib.mvf_pack_conv_weight_dgrad(P(wgt), cout, cin, k, k, P(wd), dt, None))
    z = torch.randn(m_in, cin, generator=g).to(dtype).cuda()
    mean, invstd = torch.randn(cin, generator=g).cuda() * 0.1, (torch.rand(cin, generator=g) + 0.5).cuda()
    scale, shift = (torch.rand(cin, generator=g) + 0.5).cuda(), torch.randn(cin, generator=g).cuda() * 0.3
    d = ConvDesc(n, ho, wo, cout, cin, k, k, 1, k - 1 - pad, h, w, cout, dt, 0, 0, 0, st, 0)
    ws = torch.zeros(max(lib.mvf_conv2d_workspace_bytes(C.byref(d)), 1), dtype=torch.uint8, device="cuda")
    y1 = torch.full((m_in, cin), 7.0, dtype=dtype, device="cuda")
    y2 = torch.full((m_in, cin), 7.0, dtype=dtype, device="cuda")
    rows = lib.mvf_conv2d_stats_rows(C.byref(d))
    assert rows >= (m_in + 127) // 128
    part = torch.full((rows, cin, 2), float("nan"), device="cuda")          # every partial row must be written
    check(lib.mvf_conv2d_nhwc_dgrad_bnsums(C.byref(d), P(dz), P(wd), P(y1), P(z), P(mean), P(invstd), P(scale), P(shift), P(part), P(ws), ws.numel(), None))
    dg, db = torch.empty(cin, device="cuda"), torch.empty(cin, device="cuda")
    check(lib.mvf_bn_bwd_finalize(P(part), rows, cin, P(dg), P(db), None))
    check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), P(dz), None, P(wd), None, None, P(y2), P(ws), ws.numel(), None))
    ws2 = torch.empty(lib.mvf_bn_workspace_bytes(m_in, cin), dtype=torch.uint8, device="cuda")
    dg2, db2 = torch.empty(cin, device="cuda"), torch.empty(cin, device="cuda")
    check(lib.mvf_bn_bwd_reduce(P(y2), cin, P(z), None, m_in, cin, P(mean), P(invstd), P(scale), P(shift), 2, None, P(dg2), P(db2), P(ws2), ws2.numel(), dt, None))
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    assert torch.isfinite(part).all()
    assert rel_err(dg.cpu().numpy(), dg2.cpu().numpy()) < 2e-6 and rel_err(db.cpu().numpy(), db2.cpu().numpy()) < 2e-6


# (n, h, w, cin, cout, k, stride, pad)
GRAD_CASES = [(2, 8, 8, 64, 32, 1, 1, 0), (2, 9, 9, 32, 64, 3, 1, 1), (2, 10, 10, 32, 32, 3, 2, 1), (3, 8, 8, 64, 128, 1, 2, 0),
              (1, 14, 14, 256, 256, 3, 1, 1), (2, 7, 7, 132, 36, 3, 1, 1),
              (150, 1, 1, 64, 64, 1, 1, 0), (3, 2, 33, 64, 96, 3, 1, 1), (5, 13, 11, 128, 64, 3, 2, 1), (2, 20, 20, 64, 256, 1, 1, 0),
              # [r3] shapes of the 256 x 256 weight-gradient tile (cout % 256 == 0, K % 256 == 0, cin % 64 == 0): 6 / 7 / 11 pixel chunks incl.
              # ragged last ones, stride 2, more than one split (M = 1452 -> 256-row splits), 512-wide output
              (4, 9, 9, 256, 512, 1, 1, 0), (7, 8, 8, 512, 256, 1, 1, 0), (2, 13, 11, 256, 256, 3, 2, 1), (12, 11, 11, 256, 256, 1, 1, 0),
              # [r4] the direct 64 -> 64 channel 3x3 weight-gradient kernel (bf16; fp32 takes the implicit GEMM): the C3 frame size, a ragged last band
              # (H % 4 != 0), narrow / tiny frames, more bands than workgroups (n = 40 frames of 56 rows = 560 bands on 512 workgroups)
              (2, 56, 56, 64, 64, 3, 1, 1), (3, 7, 12, 64, 64, 3, 1, 1), (1, 5, 4, 64, 64, 3, 1, 1), (40, 56, 24, 64, 64, 3, 1, 1)]


@pytest.mark.parametrize("case", GRAD_CASES, ids=lambda c: "n%d_%dx%d_c%d_o%d_k%d_s%d" % c[:7])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_conv_dgrad_wgrad_vs_oracle(case, dtype):
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    n, h, w, cin, cout, k, stride, pad = case
    if dtype == torch.bfloat16 and (cin % 8 or cout % 8):
        pytest.skip("bf16 needs channel counts that are multiples of 8")
    dt = 0 if dtype == torch.float32 else 1
    tol = 5e-5 if dtype == torch.float32 else 1e-2
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, wt, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ho, wo = y.shape[2:]
    xg, dyg, wg = nhwc(x.detach()).cuda().to(dtype), nhwc(dy).cuda().to(dtype), wt.detach().cuda()
    # wgrad
    d = _lib.ConvDesc(n, h, w, cin, cout, k, k, stride, pad, ho, wo, cin, dt, 0, 0, 0, 0)
    ws = torch.empty(lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device="cuda")
    dw = torch.empty(cout, cin, k, k, device="cuda")
    check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), P(dyg), P(xg), None, k, cin, k, cin, P(dw), P(ws), ws.numel(), None))
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < tol
    # dgrad = forward kernel on dz with flipped/transposed weights (parity-class decomposition for stride 2)
    wd = torch.empty(cin, k, k, cout, device="cuda", dtype=dtype)
    check(lib.mvf_pack_conv_weight_dgrad(P(wg), cout, cin, k, k, P(wd), dt, None))
    dd = _lib.ConvDesc(n, ho, wo, cout, cin, k, k, 1, k - 1 - pad, h, w, cout, dt, 0, 0, 0, stride if stride > 1 else 0)
    dx = torch.full((n, h, w, cin), float("nan"), device="cuda", dtype=dtype)
    check(lib.mvf_conv2d_nhwc_fwd(C.byref(dd), P(dyg), None, P(wd), None, None, P(dx), None))
    assert rel_err(dx.float().cpu().permute(0, 3, 1, 2).numpy(), x.grad.numpy()) < tol


@pytest.mark.parametrize("shape", [(3, 64, 64), (2, 96, 96), (2, 224, 224), (5, 32, 64)], ids=str)
def test_stem_wgrad_direct_bf16_vs_fp64_conv_weight_gradient(shape):
    """[r4] The stem's weight gradient on the direct kernel (csrc/wgrad_stem.hip; reference resnet.py:448-452, autograd of Conv2d(3, 64, 7, 2, 3)) in bf16
    storage: against torch's conv weight gradient in fp64 on the same bf16-rounded operands (the kernel multiplies exactly those and sums in fp32)."""
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    n, h, w = shape
    g = torch.Generator().manual_seed(h * 7 + w)
    bf = torch.bfloat16
    x = torch.randn(n, 3, h, w, generator=g).to(bf)
    ho, wo = h // 2, w // 2
    dy = torch.randn(n, 64, ho, wo, generator=g).to(bf)
    wt = torch.zeros(64, 3, 7, 7, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, stride=2, padding=3).backward(dy.double())
    hp, wp = h + 6, w + 8
    xp = torch.empty(n, hp, wp, 4, device="cuda", dtype=bf)
    xg = x.float().cuda()
    check(lib.mvf_stem_prep(P(xg), n, 3, h, w, 3, wp, P(xp), 1, None))
    d = _lib.ConvDesc(n, hp, wp, 32, 64, 7, 1, 2, 0, ho, wo, 4, 1, 0, 0, 0, 0, 0)
    ws = torch.full((lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d)) // 4,), float("nan"), device="cuda")
    dw = torch.full((64, 3, 7, 7), float("nan"), device="cuda")
    dyg = nhwc(dy).cuda()
    check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), P(dyg), P(xp), None, 7, 3, 8, 4, P(dw), P(ws), ws.numel() * 4, None))
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all()
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < 2e-5


def test_stem_wgrad_maxpool_head_sgd_vs_oracle():
    from mvfnet_amd import _lib
    from oracle import net_torch
    lib, check = _lib.lib, _lib.check
    g = torch.Generator().manual_seed(5)
    # stem weight gradient through the padded NHWC4 view
    n, h, w = 2, 32, 32
    x = torch.randn(n, 3, h, w, generator=g)
    wt = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).requires_grad_(True)
    y = F.conv2d(x, wt, stride=2, padding=3)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    hp, wp = h + 6, h + 8
    xp = torch.empty(n, hp, wp, 4, device="cuda")
    xg = x.cuda()
    check(lib.mvf_stem_prep(P(xg), n, 3, h, w, 3, wp, P(xp), 0, None))
    d = _lib.ConvDesc(n, hp, wp, 32, 64, 7, 1, 2, 0, 16, 16, 4, 0, 0, 0, 0, 0)
    ws = torch.empty(lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device="cuda")
    dw = torch.empty(64, 3, 7, 7, device="cuda")
    dyg = nhwc(dy).cuda()
    check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), P(dyg), P(xp), None, 7, 3, 8, 4, P(dw), P(ws), ws.numel(), None))
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < 5e-5
    # maxpool(relu(bn(z))) forward + backward
    n, c, h, w = 2, 8, 9, 12
    z = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    sc, sh = torch.randn(c, generator=g), torch.randn(c, generator=g) * 0.3
    a = F.relu(z * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    a.retain_grad()
    pl = F.max_pool2d(a, 3, 2, 1)
    gp = torch.randn(pl.shape, generator=g)
    pl.backward(gp)
    zg, scg, shg = nhwc(z.detach()).cuda(), sc.cuda(), sh.cuda()
    out = torch.empty(n, pl.shape[2], pl.shape[3], c, device="cuda")
    am = torch.empty(n, pl.shape[2], pl.shape[3], c, device="cuda", dtype=torch.uint8)
    check(lib.mvf_maxpool_bn_relu_fwd(P(zg), n, h, w, c, P(scg), P(shg), P(out), P(am), 0, None))
    assert rel_err(out.cpu().permute(0, 3, 1, 2).numpy(), pl.detach().numpy()) < 1e-6
    ga = torch.empty(n, h, w, c, device="cuda")
    gpg = nhwc(gp).cuda()
    check(lib.mvf_maxpool_bn_relu_bwd(P(am), P(gpg), n, h, w, c, P(ga), 0, None))
    # compare where the ReLU is active (ties among zeros route gradient to dead positions; see train_ops.hip)
    ref = a.grad * (a.detach() > 0)
    got = ga.cpu().permute(0, 3, 1, 2) * (a.detach() > 0)
    assert rel_err(got.numpy(), ref.numpy()) < 1e-6
    # head: pool -> fc -> consensus -> CE, forward + backward
    clips, T, hw, c, classes = 3, 4, 9, 32, 10
    feat = torch.randn(clips * T, c, 3, 3, generator=g, requires_grad=True)
    fw = (torch.randn(classes, c, generator=g) * 0.2).requires_grad_(True)
    fb = torch.randn(classes, generator=g).requires_grad_(True)
    labels = torch.tensor([1, 7, 3])
    sd = {"cls_head.new_fc.weight": fw, "cls_head.new_fc.bias": fb}
    loss = F.cross_entropy(net_torch.head(feat, sd, T), labels)
    loss.backward()
    fg, fwg, fbg, lg = nhwc(feat.detach()).cuda(), fw.detach().cuda(), fb.detach().cuda(), labels.cuda()
    dev = "cuda"
    pooled, scores, dsc = torch.empty(clips * T, c, device=dev), torch.empty(clips, classes, device=dev), torch.empty(clips, classes, device=dev)
    lp, lo = torch.empty(clips, device=dev), torch.empty(1, device=dev)
    check(lib.mvf_head_train_fwd(P(fg), clips, T, hw, c, P(fwg), P(fbg), classes, P(lg), None, P(pooled), P(scores), P(dsc), P(lp), P(lo), 0, None))
    assert abs(float(lo) - float(loss.detach())) < 1e-5 * abs(float(loss.detach()))
    dfw, dfb, dpool, dfeat = torch.empty(classes, c, device=dev), torch.empty(classes, device=dev), torch.empty(clips, c, device=dev), torch.empty_like(fg)
    check(lib.mvf_head_train_bwd(P(dsc), P(pooled), P(fwg), None, clips, T, hw, c, classes, P(dfw), P(dfb), P(dpool), P(dfeat), 0, None))
    assert rel_err(dfw.cpu().numpy(), fw.grad.numpy()) < 5e-5
    assert rel_err(dfb.cpu().numpy(), fb.grad.numpy()) < 5e-5
    assert rel_err(dfeat.cpu().permute(0, 3, 1, 2).numpy(), feat.grad.numpy()) < 5e-5
    # clip + SGD nesterov, two steps, vs the oracle's restatement of DistOptimizerHook + torch.optim.SGD
    nparam = 5000
    p0 = torch.randn(nparam, generator=g)
    params, mom = {"p": p0.clone()}, {}
    pg, bufg = p0.clone().cuda(), torch.zeros(nparam, device=dev)
    norm = torch.zeros(2, device=dev)
    ws = torch.empty(lib.mvf_sgd_workspace_bytes(nparam), dtype=torch.uint8, device=dev)
    for it in range(2):
        gr = torch.randn(nparam, generator=g) * (3.0 if it == 0 else 0.1)          # step 0 clips, step 1 does not
        tot = net_torch.sgd_nesterov_step(params, {"p": gr * 2}, mom, world_size=2)
        grg = (gr * 2).cuda()
        check(lib.mvf_sgd_nesterov_step(P(pg), P(grg), P(bufg), nparam, C.c_float(0.5), C.c_float(40.0), C.c_float(0.015), C.c_float(0.9),
                                        C.c_float(1e-4), int(it == 0), P(norm), P(ws), ws.numel(), None))
        assert abs(float(norm[0]) - float(tot)) < 1e-5 * float(tot)
        assert rel_err(pg.cpu().numpy(), params["p"].numpy()) < 1e-6


# ------------------------------------------------------------------------------------------------ bottleneck block
def _block(name):
    from mvfnet_amd.backbones.resnet import Bottleneck
    from mvfnet_amd.modules import MVF
    N, T, Cin, planes, H, W, stride = BLOCK_CASES[name]
    down = None
    if stride != 1 or Cin != planes * 4:
        down = nn.Sequential(nn.Conv2d(Cin, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
    blk = Bottleneck(Cin, planes, stride, 1, down)
    blk.conv1 = MVF(blk.conv1, T, Cin, 0.125, True, False, "THW")
    sd = blk.state_dict()
    pre = "block/%s/" % name
    vals = synth.synth_state_dict({pre + k: tuple(v.shape) for k, v in sd.items()})
    blk.load_state_dict({k: torch.from_numpy(vals[pre + k]) for k in sd})
    return blk.cuda().train()


@pytest.mark.parametrize("name", sorted(BLOCK_CASES))
def test_bottleneck_train_block_matches_reference_golden(name):
    from mvfnet_amd.train_engine import BlockTrainer
    N, T, Cin, planes, H, W, stride = BLOCK_CASES[name]
    g = golden("block_cases.npz")
    blk = _block(name)
    tr = BlockTrainer(blk)
    x = torch.from_numpy(synth.synth_tensor("block_x/" + name, (N * T, Cin, H, W))).cuda()
    y = tr.forward(x)
    assert rel_err(y.cpu().numpy(), g[name + "/train/y"]) < 1e-5
    dy = torch.from_numpy(synth.synth_tensor("block_dy/" + name, tuple(y.shape))).cuda()
    dx = tr.backward(dy)
    assert rel_err(dx.cpu().numpy(), g[name + "/train/dx"]) < 1e-4
    for pn, p in blk.named_parameters():
        assert rel_err(tr.grad_of(p).cpu().numpy(), g[name + "/train/grad/" + pn]) < 2e-4, pn
    for bn_, b in blk.named_buffers():
        ref = g[name + "/train/buf/" + bn_]
        if ref.dtype.kind == "i":
            assert int(b) == int(ref), bn_
        else:
            assert rel_err(b.cpu().numpy(), ref) < 1e-5, bn_


# ------------------------------------------------------------------------------------------------ whole network
def _model(depth, T, dropout=0.0):
    import mvfnet_amd
    m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(depth, T, dropout_ratio=dropout), None, dict(average_clips=None))
    sd = m.state_dict()
    pre = "r%d/" % depth
    vals = synth.synth_state_dict({pre + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals[pre + k]) for k in sd}, strict=True)
    return m.cuda().train()


def test_c1_train_two_steps_vs_reference_golden():
    """R50 4x16, 2 clips 224^2: loss, per-stage activations (batch-stat BN), every parameter's gradient norm, then the
    clip + SGD-nesterov update and a second step -- against the reference's own run (tests/golden/net_cases.npz)."""
    g = golden("net_cases.npz")
    m = _model(50, 4)
    eng = m.train_engine()
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 224, 224)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2)).cuda()
    stages = {}
    loss = eng.forward(imgs, labels, stages=stages)
    assert abs(float(loss) - float(g["c1/train/loss/0"])) < 2e-5 * float(g["c1/train/loss/0"])
    for k, v in stages.items():
        a = v.float().cpu().permute(0, 3, 1, 2).contiguous().numpy().astype(np.float64).ravel()
        ref = g["c1/train/stage/" + k]
        assert abs(a.mean() - ref[0]) < 1e-4 * ref[2] and abs(np.sqrt((a * a).mean()) - ref[1]) < 1e-4 * ref[2], k
        idx = np.linspace(0, a.size - 1, 16).astype(np.int64)
        assert np.abs(a[idx] - ref[3:]).max() < 2e-4 * ref[2], k
    eng.backward()
    names, ref_norms = list(g["c1/train/grad_names"]), g["c1/train/grad_norms"]
    params = dict(m.named_parameters())
    worst = 0.0
    for nme, r in zip(names, ref_norms):
        got = float(eng.grad_of(params[nme]).double().norm())
        err = abs(got - r) / max(r, 1e-6)
        worst = max(worst, err)
        # early layers are ill-conditioned (2 clips, 53 batch-stat BNs): the reference itself moves by 4e-3 under op
        # reordering and 2e-2 between fp32 and fp64; late layers must be tight
        tol = 3e-3 if (nme.startswith("cls_head") or nme.startswith("backbone.layer4.2")) else 3e-2
        assert err < tol, (nme, got, r)
    for k in g.files:
        if k.startswith("c1/train/grad/"):
            nme = k[len("c1/train/grad/"):]
            # element-wise: tight only for the head; MVF tap / BN gradients inside layer3/4 are sums of O(1e5) signed terms
            # that cancel to ~1e-3 (measured: reference-vs-oracle op-order noise alone is ~1e-2 there)
            tol = 2e-3 if nme.startswith("cls_head") else 5e-2
            assert rel_err(eng.grad_of(params[nme]).cpu().numpy(), g[k]) < tol, nme
    norm = eng.step()
    assert abs(float(norm[0]) - float(g["c1/train/total_norm/0"])) < 2e-3 * float(g["c1/train/total_norm/0"])
    loss1 = eng.forward(imgs, labels)
    assert abs(float(loss1) - float(g["c1/train/loss/1"])) < 5e-3 * float(g["c1/train/loss/1"])     # chaotic regime, see oracle test
    eng.backward()
    norm = eng.step()
    assert abs(float(norm[0]) - float(g["c1/train/total_norm/1"])) < 2e-2 * float(g["c1/train/total_norm/1"])
    sd = m.state_dict()
    for k in g.files:
        if k.startswith("c1/train/after2/"):
            nme = k[len("c1/train/after2/"):]
            a = sd[nme].detach().float().cpu().numpy().ravel()
            if g[k].dtype.kind == "i":
                assert int(a[0]) == int(g[k][0]), nme
            else:
                assert rel_err(a[: g[k].size], g[k]) < 0.1, nme


def test_norm_eval_training_vs_reference_golden():
    """backbone norm_eval=True (reference resnet.py:496-505): every BatchNorm -- the 53 of the ResNet and the 9 inside the MVF
    modules -- normalises with its running statistics and leaves them alone, gamma / beta still train.  Two clip + SGD-nesterov
    steps against the reference's own run (tests/golden/normeval_cases.npz).  Without batch statistics the network is
    well-conditioned, so the tolerances are tight everywhere."""
    import mvfnet_amd
    g = golden("normeval_cases.npz")
    cfg = mvfnet_amd.mvfnet_config(50, 4, dropout_ratio=0.0)
    cfg["backbone"]["norm_eval"] = True
    m = mvfnet_amd.build_recognizer(cfg, None, dict(average_clips=None))
    sd = m.state_dict()
    vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True)
    m = m.cuda().train()
    bns = [x for x in m.backbone.modules() if isinstance(x, torch.nn.modules.batchnorm._BatchNorm)]
    assert len(bns) == 62 and not any(x.training for x in bns)
    before = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}
    eng = m.train_engine()
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 96, 96, seed=77)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2)).cuda()
    loss = eng.forward(imgs, labels)
    assert abs(float(loss) - float(g["loss/0"])) < 2e-5 * float(g["loss/0"])
    eng.backward()
    params = dict(m.named_parameters())
    # [r4] Gradient tolerances.  The reference's own fp32 run of this step equals its fp64 run to 1e-6 (tests/golden/make_normeval_fp64_golden.py),
    # and with the exact-fp32 MFMA convs (MVF_POLICY=f32_x3=0) the engine is within 5e-4 of both: 2e-3 asserted (the tight leg below runs that path in a
    # child process).  The default fp32 convs (three-term bf16 split on the bf16 matrix cores) differ from the MFMA ones by ~1e-6 per conv output --
    # both equally close to an fp64 convolution (test_fp32_conv_on_the_bf16_matrix_cores_is_as_accurate_as_the_fp32_mfma) -- which on THIS
    # input flips ONE ReLU decision of layer4.0's bn2 (a 72 x 512 tensor at 96^2 input: one element is 1.2e-2 of its gradient's norm;
    # tools/probes/normeval_blocks.py): every gradient upstream of it then moves by ~5e-3.  That is fp32 sign luck at a kink, not accuracy:
    # the loose bound is asserted here, the conv- and block-level comparisons stay at 1e-6 / 2e-4.
    import os
    tight = os.environ.get("MVF_F32_X3", "1") == "0"
    tol_n, tol_g = (2e-3, 2e-3) if tight else (1.5e-2, 2e-2)
    for nme, r in zip(list(g["grad_names"]), g["grad_norms"]):
        got = float(eng.grad_of(params[nme]).double().norm())
        assert abs(got - r) < tol_n * max(r, 1e-6), (nme, got, r)
    for k in g.files:
        if k.startswith("grad/"):
            assert rel_err(eng.grad_of(params[k[5:]]).cpu().numpy(), g[k]) < tol_g, k
    g64 = golden("normeval_fp64.npz")                      # the same step by the reference in DOUBLE precision
    for k in g64.files:
        if k.startswith("grad/"):
            assert rel_err(eng.grad_of(params[k[5:]]).cpu().numpy().astype(np.float64), g64[k]) < tol_g, k
    norm = eng.step()
    assert abs(float(norm[0]) - float(g["total_norm/0"])) < (1e-3 if tight else 5e-3) * float(g["total_norm/0"])
    loss1 = eng.forward(imgs, labels)
    assert abs(float(loss1) - float(g["loss/1"])) < 2e-3 * float(g["loss/1"])
    eng.backward()
    norm = eng.step()
    assert abs(float(norm[0]) - float(g["total_norm/1"])) < (5e-3 if tight else 1e-2) * float(g["total_norm/1"])
    sd = m.state_dict()
    for k, v in before.items():
        assert torch.equal(sd[k], v), k                       # frozen statistics (and step counters) did not move
    for k in g.files:
        if k.startswith("after2/"):
            a = sd[k[7:]].detach().float().cpu().numpy().ravel()
            assert rel_err(a[: g[k].size], g[k]) < (2e-3 if tight else 5e-3), k


def test_norm_eval_training_vs_reference_golden_tight_on_the_exact_fp32_mfma():
    """The tight leg of the test above: the same step in a child process with MVF_POLICY=f32_x3=0 (the switch is read once per process)."""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "test_norm_eval_training_vs_reference_golden and not tight",
                        "-p", "no:cacheprovider"], env=policy_env(f32_x3=0), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_norm_eval_default_fp32_path_tight_over_seeds():
    """[r5] The TIGHT end-to-end gradient gate of the shipped fp32 path (convs and weight gradients as exact three-term bf16 splits on the bf16
    matrix cores, the default).  The frozen-statistics step on six inputs, each against the REFERENCE's own double-precision run
    (tests/golden/normeval_seeds_fp64.npz, make_normeval_seeds_golden.py).  Every input has ReLU pre-activations within 1e-7 ... 1e-8 of zero in
    every stage (recorded in the golden as margin/*), so on any ONE input an fp32 path may take one kink decision the other way -- in layer4's
    72 x 512 tensors that is ~1e-2 of a gradient's norm (seed 77 is the documented case: test_norm_eval_training_vs_reference_golden).  An accuracy
    regression moves every input; sign luck moves one.  Asserted: all 206 gradient norms and ten named gradients within 2e-3 on at least five
    of the six inputs (and the median input's worst error), the loose 2e-2 on all six."""
    import mvfnet_amd
    g = golden("normeval_seeds_fp64.npz")
    cfg = mvfnet_amd.mvfnet_config(50, 4, dropout_ratio=0.0)
    cfg["backbone"]["norm_eval"] = True
    m = mvfnet_amd.build_recognizer(cfg, None, dict(average_clips=None))
    sd = m.state_dict()
    vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True)
    m = m.cuda().train()
    eng = m.train_engine()
    params = dict(m.named_parameters())
    labels = torch.from_numpy(synth.synth_labels(2)).cuda()
    names = list(g["grad_names"])

    def sample(a, size):
        a = a.ravel()
        if a.size > size:        # the generator's fixed random sample of a large tensor's elements
            a = a[np.sort(np.random.RandomState(a.size % 65521).choice(a.size, size, replace=False))]
        return a

    worst = {}
    for seed in [int(s_) for s_ in g["seeds"]]:
        tag = "s%d/" % seed
        imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 96, 96, seed=seed)).cuda()
        loss = eng.forward(imgs, labels)
        assert abs(float(loss) - float(g[tag + "loss"])) < 2e-5 * float(g[tag + "loss"]), seed
        eng.backward()
        torch.cuda.synchronize()
        en = max(abs(float(eng.grad_of(params[n]).double().norm()) - r) / max(r, 1e-6) for n, r in zip(names, g[tag + "grad_norms"]))
        eg = max(rel_err(sample(eng.grad_of(params[k[len(tag) + 5:]]).cpu().numpy().astype(np.float64), 16384), g[k])
                 for k in g.files if k.startswith(tag + "grad/"))
        worst[seed] = (en, eg)
        print("seed %d: worst gradient-norm error %.2e, worst gradient error %.2e (vs the reference in fp64)" % (seed, en, eg))
    errs = sorted(max(v) for v in worst.values())
    assert errs[-1] < 2e-2, worst
    assert errs[-2] < 2e-3, worst                       # at most one input may sit on a kink
    assert errs[len(errs) // 2] < 1.5e-3, worst          # (measured on the round's kernels: 3e-4 ... 1.1e-3 on five inputs, seed 77 at 1.3e-2)


def test_norm_eval_training_bf16_close_to_fp32():
    import mvfnet_amd
    g = golden("normeval_cases.npz")
    cfg = mvfnet_amd.mvfnet_config(50, 4, dropout_ratio=0.0)
    cfg["backbone"]["norm_eval"] = True
    m = mvfnet_amd.build_recognizer(cfg, None, dict(average_clips=None))
    sd = m.state_dict()
    vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True)
    m = m.cuda().train()
    eng = m.train_engine(dtype=torch.bfloat16)
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 96, 96, seed=77)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2)).cuda()
    loss = eng.forward(imgs, labels)
    assert abs(float(loss) - float(g["loss/0"])) < 2e-2 * float(g["loss/0"])
    eng.backward()
    norm = eng.step()
    assert abs(float(norm[0]) - float(g["total_norm/0"])) < 5e-2 * float(g["total_norm/0"])


def test_frozen_stages_training_vs_reference_golden():
    """frozen_stages=1 (reference resnet.py:515-527): stem + layer1 in eval mode and excluded from training -- a prefix of the flat
    parameter buffer, so norm / clip / weight decay / momentum / update run on the rest only.  Two steps against the reference's own
    run (clip active: total norm 131 > 40); layers 2-4 keep batch-statistics BN with 2 clips, hence the c1-style tolerances."""
    import mvfnet_amd
    g = golden("frozen_cases.npz")
    cfg = mvfnet_amd.mvfnet_config(50, 4, dropout_ratio=0.0)
    cfg["backbone"]["frozen_stages"] = 1
    m = mvfnet_amd.build_recognizer(cfg, None, dict(average_clips=None))
    sd = m.state_dict()
    vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True)
    m = m.cuda().train()
    frozen = sorted(n for n, p in m.named_parameters() if not p.requires_grad)
    assert frozen == sorted(g["frozen_names"])
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    eng = m.train_engine()
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 96, 96, seed=78)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2)).cuda()
    loss = eng.forward(imgs, labels)
    assert abs(float(loss) - float(g["loss/0"])) < 1e-4 * float(g["loss/0"])
    eng.backward()
    params = dict(m.named_parameters())
    for nme, r in zip(list(g["grad_names"]), g["grad_norms"]):
        got = float(eng.grad_of(params[nme]).double().norm())
        tol = 3e-3 if (nme.startswith("cls_head") or nme.startswith("backbone.layer4.2")) else 3e-2
        assert abs(got - r) < tol * max(r, 1e-6), (nme, got, r)
    norm = eng.step()
    assert abs(float(norm[0]) - float(g["total_norm/0"])) < 5e-3 * float(g["total_norm/0"])
    loss1 = eng.forward(imgs, labels)
    assert abs(float(loss1) - float(g["loss/1"])) < 2e-2 * float(g["loss/1"])
    eng.backward()
    eng.step()
    sd = m.state_dict()
    for k in frozen + ["backbone.bn1.running_mean", "backbone.layer1.2.bn3.running_var", "backbone.layer1.0.bn1.num_batches_tracked"]:
        assert torch.equal(sd[k], sd0[k]), k                      # excluded parameters and frozen statistics did not move
    assert not torch.equal(sd["backbone.layer2.0.bn1.running_mean"], sd0["backbone.layer2.0.bn1.running_mean"])
    for k in g.files:
        if k.startswith("after2/"):
            a = sd[k[7:]].detach().float().cpu().numpy().ravel()
            assert rel_err(a[: g[k].size], g[k]) < 0.1, k


def test_norm_frozen_scattered_exclusions_vs_reference_golden():
    """norm_eval=True + norm_frozen=True (reference resnet.py:496-505): every BatchNorm in eval mode and its weight / bias excluded
    from training -- 124 parameters in SCATTERED places of model.parameters().  The optimizer's segment kernel leaves them (and their
    momentum) untouched and keeps their gradients out of the clip norm (max_norm 5 so the clip is active).  Two steps against the
    reference's own run."""
    import mvfnet_amd
    g = golden("normfrozen_cases.npz")
    cfg = mvfnet_amd.mvfnet_config(50, 4, dropout_ratio=0.0)
    cfg["backbone"]["norm_eval"] = True
    cfg["backbone"]["norm_frozen"] = True
    m = mvfnet_amd.build_recognizer(cfg, None, dict(average_clips=None))
    sd = m.state_dict()
    vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True)
    m = m.cuda().train()
    frozen = sorted(n for n, p in m.named_parameters() if not p.requires_grad)
    assert frozen == sorted(g["frozen_names"]) and len(frozen) == 124
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    eng = m.train_engine(max_norm=5.0)
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 96, 96, seed=79)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2)).cuda()
    loss = eng.forward(imgs, labels)
    assert abs(float(loss) - float(g["loss/0"])) < 1e-4 * float(g["loss/0"])
    eng.backward()
    params = dict(m.named_parameters())
    for nme, r in zip(list(g["grad_names"]), g["grad_norms"]):
        got = float(eng.grad_of(params[nme]).double().norm())
        assert abs(got - r) < 2e-3 * max(r, 1e-6), (nme, got, r)        # frozen statistics: no batch-statistics chaos
    norm = eng.step()
    assert abs(float(norm[0]) - float(g["total_norm/0"])) < 1e-3 * float(g["total_norm/0"])
    loss1 = eng.forward(imgs, labels)
    assert abs(float(loss1) - float(g["loss/1"])) < 2e-3 * float(g["loss/1"])
    eng.backward()
    norm = eng.step()
    assert abs(float(norm[0]) - float(g["total_norm/1"])) < 5e-3 * float(g["total_norm/1"])
    sd = m.state_dict()
    for k in frozen:
        assert torch.equal(sd[k], sd0[k]), k                      # excluded parameters did not move
    for k in g.files:
        if k.startswith("after2/"):
            a = sd[k[7:]].detach().float().cpu().numpy().ravel()
            assert rel_err(a[: g[k].size], g[k]) < 2e-3, k
    opt = eng.optimizer_state_dict()                              # torch creates no state for parameters without a gradient
    idx = {n: i for i, (n, _) in enumerate(m.named_parameters())}
    assert all(idx[n] not in opt["state"] for n in frozen) and len(opt["state"]) == len(idx) - len(frozen)


def test_forward_train_autograd_api_and_external_optimizer():
    """The reference's flow: losses = model(img_group, label); loss.backward(); clip; optimizer.step() with torch SGD."""
    m = _model(50, 4)
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 64, 64, seed=2)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2, seed=2)).cuda()
    opt = torch.optim.SGD(m.parameters(), lr=0.015, momentum=0.9, weight_decay=1e-4, nesterov=True)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out = m(imgs, labels, return_loss=True)
        out["loss_cls"].backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 40.0)
        opt.step()
        losses.append(float(out["loss_cls"]))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    # after training, eval-mode inference runs through the inference engine with the updated weights
    m.eval()
    s = m(imgs, None, return_loss=False)
    assert s.shape == (2, 400) and np.isfinite(s).all()


def test_r101_16x4_train_loss_vs_oracle_and_runner_resume(tmp_path):
    """BASELINE config 4 family (R101, T=16; reduced to 2 clips at 64^2): train-mode loss vs the CPU oracle, then the
    runner shell: 2 iterations, checkpoint, resume."""
    from mvfnet_amd.runner import Runner
    from oracle import net_torch
    m = _model(101, 16)
    cpu_sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 16, 64, 64, seed=4))
    labels = torch.from_numpy(synth.synth_labels(2, seed=4))
    with torch.no_grad():
        ref = float(net_torch.forward_train(imgs, labels, cpu_sd, 101))
    run = Runner(m, work_dir=str(tmp_path), ckpt_interval=1, log_interval=0, warmup_iters=10)
    loss = run.engine.forward(imgs.cuda(), labels.cuda())
    assert abs(float(loss) - ref) < 1e-4 * abs(ref)
    run.engine.backward()
    loader = [dict(img_group=imgs.cuda(), label=labels.cuda())] * 2
    run.run(loader, max_epochs=1)
    assert run.iter == 2 and (tmp_path / "epoch_1.pth").exists() and (tmp_path / "latest.pth").exists()
    w_after = m.backbone.conv1.weight.detach().clone()
    m2 = _model(101, 16)
    run2 = Runner(m2, work_dir=str(tmp_path), log_interval=0)
    run2.resume(str(tmp_path / "latest.pth"))
    assert run2.epoch == 1 and run2.iter == 2 and torch.equal(m2.backbone.conv1.weight, w_after)
    assert torch.equal(run2.engine.flat_mom, run.engine.flat_mom)


# ------------------------------------------------------------------------------------------------ bf16 storage
# (the bf16 bottleneck-block comparison lives in tests/test_bf16_parity_gpu.py: against an oracle that rounds where the engine rounds)


def _ragged_setup(shape, dtype):
    from oracle import net_torch
    b, t, h, w = shape
    m = _model(50, t)
    eng = m.train_engine(dtype=dtype)
    imgs_np, labels_np = synth.synth_clip_batch(b, t, h, w), synth.synth_labels(b)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    leaves = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            leaves[k] = v.requires_grad_(True)
    ref_stages = {}
    ref_loss = net_torch.forward_train(torch.from_numpy(imgs_np), torch.from_numpy(labels_np), sd, depth=50, T=t, new_buffers={}, stages=ref_stages)
    ref_loss.backward()
    stages = {}
    loss = eng.forward(torch.from_numpy(imgs_np).cuda(), torch.from_numpy(labels_np).cuda(), stages=stages)
    got = {k: v.float().cpu().permute(0, 3, 1, 2).contiguous().numpy() for k, v in stages.items() if k in ref_stages}
    eng.backward()
    params = dict(m.named_parameters())
    errs = {}
    for k, leaf in leaves.items():
        if k in params and leaf.grad is not None:
            r = float(leaf.grad.double().norm())
            errs[k] = abs(float(eng.grad_of(params[k]).double().norm()) - r) / max(r, 1e-6)
    assert len(errs) > 150
    return float(loss), float(ref_loss.detach()), got, {k: v.detach().numpy() for k, v in ref_stages.items()}, errs, (sd, imgs_np, labels_np, t)


RAGGED = [(3, 3, 80, 112), (5, 2, 64, 96)]


@pytest.mark.parametrize("shape", RAGGED, ids=lambda s: "b%d_t%d_%dx%d" % s)
def test_train_step_ragged_shapes_vs_oracle(shape):
    """Sizes that divide nothing: 9-10 frames of non-square inputs -> 108-pixel layer4 maps (less than one 128-row tile),
    ragged K / M tails in every conv, weight-gradient chunk and BatchNorm column plan.  Loss, stage outputs and all
    gradient norms against the CPU oracle (oracle/net_torch.py, fp32 torch autograd) computed here."""
    loss, ref_loss, got, ref, errs, _ = _ragged_setup(shape, torch.float32)
    assert abs(loss - ref_loss) < 5e-5 * abs(ref_loss)
    for k in got:
        assert rel_err(got[k], ref[k]) < 2e-4, k
    head = [v for k, v in errs.items() if k.startswith("cls_head") or k.startswith("backbone.layer4.2")]
    med, worst = np.median(list(errs.values())), max(errs.values())
    assert max(head) < 3e-3, max(head)
    assert med < 3e-3 and worst < 5e-2, (med, worst)


@pytest.mark.parametrize("shape", RAGGED, ids=lambda s: "b%d_t%d_%dx%d" % s)
def test_train_step_ragged_shapes_bf16_deviates_like_bf16_storage(shape):
    """bf16 engine on the same ragged shapes.  On the synthetic-weight network bf16 STORAGE alone moves the stage outputs by
    0.3 % (maxpool) ... 2 % (layer1) ... 44 % (layer4) in relative L2 -- measured with the CPU oracle whose conv inputs /
    weights / outputs and ReLU outputs are rounded to bf16 (helpers.bf16_storage_oracle).  The engine must show THAT deviation
    profile (within 15 % per stage), be closer to the emulation than to fp32, and keep the loss within 5 %."""
    from oracle import net_torch
    loss, ref_loss, got, ref, errs, (sd, imgs_np, labels_np, t) = _ragged_setup(shape, torch.bfloat16)
    assert abs(loss - ref_loss) < 5e-2 * abs(ref_loss)      # 10-frame batches, 2x3 layer4 maps: the loss itself moves by 1-3 %
    emu = {}
    with torch.no_grad(), bf16_storage_oracle():
        net_torch.forward_train(torch.from_numpy(imgs_np), torch.from_numpy(labels_np), {k: v.detach() for k, v in sd.items()}, depth=50, T=t,
                                new_buffers={}, stages=emu)
    for k in got:
        d_eng, d_emu = rel_l2(got[k], ref[k]), rel_l2(emu[k].numpy(), ref[k])
        assert abs(d_eng - d_emu) < 0.15 * d_emu + 1e-4, (k, d_eng, d_emu)
        if k != "maxpool":
            assert rel_l2(got[k], emu[k].numpy()) < 0.7 * d_eng, (k, rel_l2(got[k], emu[k].numpy()), d_eng)
    vals = list(errs.values())
    assert all(np.isfinite(vals)) and np.median(vals) < 0.1, np.median(vals)


def test_c1_train_bf16_loss_and_gradients_track_reference():
    g = golden("net_cases.npz")
    m = _model(50, 4)
    eng = m.train_engine(dtype=torch.bfloat16)
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 224, 224)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2)).cuda()
    loss = eng.forward(imgs, labels)
    assert abs(float(loss) - float(g["c1/train/loss/0"])) < 1e-2 * float(g["c1/train/loss/0"])
    eng.backward()
    params = dict(m.named_parameters())
    names, ref = list(g["c1/train/grad_names"]), g["c1/train/grad_norms"]
    for nme, r in zip(names, ref):
        got = float(eng.grad_of(params[nme]).double().norm())
        assert np.isfinite(got), nme
        if nme.startswith("cls_head"):
            assert abs(got - r) / max(r, 1e-6) < 2e-2, (nme, got, r)
    # the backbone gradients of a bf16 step are NOT compared with the fp32 reference here: on this synthetic network bf16 storage
    # alone moves them by O(1) (chaotic 2-clip batch statistics); tests/test_bf16_parity_gpu.py compares every block's gradients
    # with an oracle that rounds where the engine rounds, at a few %
    norm = eng.step()
    assert abs(float(norm[0]) - float(g["c1/train/total_norm/0"])) < 5e-2 * float(g["c1/train/total_norm/0"])
    assert float(eng.forward(imgs, labels)) < float(loss)              # the step reduces the loss


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_side_stream_overlap_is_bit_identical_to_single_stream(dtype):
    """Race detector.  Nothing in the step uses atomics, so training with the weight gradients / MVF tap gradients / weight
    packs on the side stream must reproduce the single-stream run BIT for bit (losses, parameters, BatchNorm buffers) over
    several optimizer steps -- and the loss on the fixed batch must go down."""
    imgs = torch.from_numpy(synth.synth_clip_batch(4, 4, 128, 128)).cuda()
    labels = torch.from_numpy(synth.synth_labels(4)).cuda()
    res = []
    for overlap in (True, False):
        torch.manual_seed(0)                                     # same dropout masks
        m = _model(50, 4, dropout=0.5)
        eng = m.train_engine(dtype=dtype)
        eng.overlap_wgrad = overlap
        losses = [float(eng.train_step(imgs, labels)) for _ in range(6)]
        torch.cuda.synchronize()
        res.append((losses, eng.flat_params.clone(), torch.cat([b.flatten().float() for b in m.buffers()])))
    assert all(np.isfinite(res[0][0])) and res[0][0][-1] < res[0][0][0]
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_engine_switch_variants_reproduce_the_default_step(dtype):
    """The backward variants behind the engine's A/B switches compute the same step: the unpaired bn3 / downsample-BN backward is BIT-identical to the default
    over three optimizer steps; the others agree to summation order or bf16 noise as stated at each."""
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 96, 96)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2)).cuda()

    def run(steps=3, **attrs):
        torch.manual_seed(0)
        m = _model(50, 4, dropout=0.5)
        eng = m.train_engine(dtype=dtype)
        for k, v in attrs.items():
            assert hasattr(eng, k)
            setattr(eng, k, v)
        losses = [float(eng.train_step(imgs, labels)) for _ in range(steps)]
        torch.cuda.synchronize()
        return losses, eng.flat_params.clone()

    base = run()
    # [r3] fuse_bn3_apply: bn3's apply + residual + ReLU as the epilogue of a second conv3 pass (0 = the pass over z3, 2 = every block incl. layer4)
    for attrs in (dict(pair_bn_bwd=False), dict(use_plan=False)):
        got = run(**attrs)
        assert got[0] == base[0], attrs
        assert torch.equal(got[1], base[1]), attrs
    base0 = run(z3_free=0)           # (with z3 stored everywhere: the apply fusion alone changes no bit)
    for attrs in (dict(fuse_bn3_apply=0), dict(fuse_bn3_apply=2)):
        got = run(z3_free=0, **attrs)
        assert got[0] == base0[0], attrs
        assert torch.equal(got[1], base0[1]), attrs
    # [r3] z3_free: the plain blocks' bn3 backward on the recomputed conv3 sums dgamma / dbeta per 128-row tile instead of per row band: another
    # summation order (one step; the forward -- statistics-only pass + fused apply -- is bit-identical, so the loss is)
    # [r3] in bf16 the layer1 statistics-only pass runs on csrc/pw_sums.hip (channel = lane sums): batch statistics to summation order, which the
    # 2-clip bf16 network amplifies (rounding flips) -- the loss then agrees to 1e-2 instead of bit for bit
    ref = run(steps=1)
    for z3f in (0, 2):
        got = run(steps=1, z3_free=z3f)
        if dtype == torch.float32:
            assert got[0] == ref[0], z3f
        else:
            assert abs(got[0][0] - ref[0][0]) < 1e-2 * abs(ref[0][0]), (z3f, got[0], ref[0])
        assert rel_l2(got[1].cpu().numpy(), ref[1].cpu().numpy()) < (1e-5 if dtype == torch.float32 else 1e-2), z3f       # (measured 3.6e-3)
    # [r4] fuse_bnwg: layer1 / layer2's pointwise weight gradients inside the BatchNorm-backward apply pass (bf16 only; a no-op in fp32): every dz
    # is bit-identical, the weight gradients are summed per persistent workgroup instead of per GEMM split -- the first step's loss is the same
    # bit for bit, the updated parameters to fp32 summation order
    got, ref = run(steps=1, fuse_bnwg=0), run(steps=1)
    assert got[0] == ref[0]
    if dtype == torch.float32:
        assert torch.equal(got[1], ref[1])
    else:
        assert rel_l2(got[1].cpu().numpy(), ref[1].cpu().numpy()) < 1e-6
        # ... also with layer1's plain blocks on stored z3 + the fused pass instead of the z3-free path ([r5] both sides with bn3's statistics from a conv3 pass:
        # the Gram form of the default, checked below, exists for z3-free blocks only)
        got8, ref8 = run(steps=1, fuse_bnwg=15, gram_stats=False), run(steps=1, gram_stats=False)
        assert abs(got8[0][0] - ref8[0][0]) < 1e-2 * abs(ref8[0][0]) and rel_l2(got8[1].cpu().numpy(), ref8[1].cpu().numpy()) < 1e-2
        # [r5] gram_stats: bn3's batch statistics of the z3-free blocks from the Gram matrix of a2 instead of a statistics pass of conv3 -- the statistics of the
        # UNROUNDED z3, 1e-7 of fp64 where the pass over the bf16 tensor is at 4e-5 (tests/test_dzfree_gpu.py): a 4e-5 change of two BatchNorms' statistics, which
        # this 2-clip batch-statistics network amplifies like any other (measured: loss 1.3e-2 against the pass, parameters 6e-3)
        assert abs(ref8[0][0] - ref[0][0]) < 3e-2 * abs(ref[0][0]) and rel_l2(ref8[1].cpu().numpy(), ref[1].cpu().numpy()) < 2e-2
    # [r4] fuse_c3_bwd: the z3-free blocks' conv3 backward in ONE pass (csrc/pw_bwd_fused.hip; bf16 only, a no-op in fp32): the same dz3 bit for bit,
    # kept on chip; its data gradient bit for bit, bn2's sums and the weight gradient summed per persistent workgroup (fp32 summation order).  (With the
    # downsample block's z3-free mode off on both sides: it depends on this switch and changes the FORWARD's statistics pass.)
    got, ref = run(steps=1, fuse_c3_bwd=0, z3_free_ds=0), run(steps=1, z3_free_ds=0)
    assert got[0] == ref[0]
    if dtype == torch.float32:
        assert torch.equal(got[1], ref[1])
    else:
        assert rel_l2(got[1].cpu().numpy(), ref[1].cpu().numpy()) < 1e-2
    # [r4] z3_free_ds: layer1.0 (the downsample block whose two convs are 64 -> 256 pointwise) without a stored z3 and without the paired BatchNorm
    # backward: per branch the sums pass on the recomputed conv + the one-pass backward (bf16 only; a no-op in fp32)
    got, ref = run(steps=1, z3_free_ds=0), run(steps=1)
    if dtype == torch.float32:
        assert got[0] == ref[0] and torch.equal(got[1], ref[1])
    else:
        assert abs(got[0][0] - ref[0][0]) < 1e-2 * abs(ref[0][0]) and rel_l2(got[1].cpu().numpy(), ref[1].cpu().numpy()) < 1e-2
    # [r5] dzfree: bn3's backward of the plain blocks of layer2 ... layer4 without the dz3 tensor (csrc/bn_dzfree.hip; bf16 only, a no-op in fp32): the same
    # forward (loss bit for bit); dz3 is no longer rounded to bf16 on its way into the two GEMMs, a.W and G are instead -- parameters to bf16 noise.
    # gate_producer=False: the dz3-free blocks' own sums pass writes gm instead of the block above gating what it hands down -- the same gm bit for bit.
    got, ref = run(steps=1, dzfree=0), run(steps=1)
    assert got[0] == ref[0]
    if dtype == torch.float32:
        assert torch.equal(got[1], ref[1])
    else:
        assert rel_l2(got[1].cpu().numpy(), ref[1].cpu().numpy()) < 1e-2
    got = run(steps=1, gate_producer=False)
    assert got[0] == ref[0] and torch.equal(got[1], ref[1])
    # [r5] bn3's backward sums from the producers' column sums + the weight-gradient GEMM in EVERY dz3-free block (default: large ones) instead of a pass over
    # (gm, z3): the same forward with the Gram statistics off on both sides (with them on, such a block of layer2 also drops its first conv3 pass: next line)
    got, refq = run(steps=1, dzfree_q=2, gram_stats=False), run(steps=1, gram_stats=False)
    assert got[0] == refq[0] and rel_l2(got[1].cpu().numpy(), refq[1].cpu().numpy()) < (1e-6 if dtype == torch.float32 else 1e-2)
    got = run(steps=1, dzfree_q=2)
    assert abs(got[0][0] - ref[0][0]) < (1e-6 if dtype == torch.float32 else 3e-2) * abs(ref[0][0])
    assert rel_l2(got[1].cpu().numpy(), ref[1].cpu().numpy()) < (1e-6 if dtype == torch.float32 else 2e-2)
    got = run(steps=1, fuse_mvf_stats=False)            # [r5] MVF's BatchNorm statistics from a pass over y instead of the stencil launch: fp32 summation order
    assert abs(got[0][0] - ref[0][0]) < (1e-6 if dtype == torch.float32 else 1e-2) * abs(ref[0][0])
    assert rel_l2(got[1].cpu().numpy(), ref[1].cpu().numpy()) < (1e-3 if dtype == torch.float32 else 1e-2)      # (batch statistics in another summation order, amplified by the 2-clip network)


def test_two_bucket_gradient_exchange_matches_flat_allreduce_single_rank():
    """One rank, RCCL process group, exchange forced: the engine launches the tail bucket (layer3 + layer4 + head) from the
    side stream while backward is still running and the head bucket after it -- parameters after two steps must be bit-identical
    to the un-overlapped single flat all-reduce (world size 1: the collective is the identity, the test is the stream choreography)."""
    import os
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 96, 96)).cuda()
        labels = torch.from_numpy(synth.synth_labels(2)).cuda()
        outs = []
        for overlap in (False, True):
            m = _model(50, 4)
            eng = m.train_engine(dtype=torch.bfloat16)
            eng.force_allreduce = True
            eng.overlap_allreduce = overlap
            losses = [float(eng.train_step(imgs, labels)) for _ in range(2)]
            assert eng._tail_off and 0 < eng._tail_off < eng.flat_grads.numel() // 4      # the tail bucket is most of the buffer
            torch.cuda.synchronize()
            outs.append((losses, eng.flat_params.clone(), eng.flat_grads.clone()))
        assert outs[0][0] == outs[1][0]
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_batched_weight_pack_equals_the_per_conv_packs(dtype):
    """mvf_pack_conv_weights_batched (one launch for every forward / data-gradient pack of the step) writes exactly what the
    per-conv pack entry points write."""
    import mvfnet_amd
    m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, 4), None, dict(average_clips=None)).cuda().train()
    eng = m.train_engine(dtype=dtype) if getattr(m, "_train_engine", None) is None else m.train_engine()
    convs = [eng.stem] + [cv for blk in eng.blocks for cv in blk.convs()]
    for cv in convs:
        cv.pack(need_dgrad=True)
    torch.cuda.synchronize()
    want = [(cv.wp.clone(), None if cv.wd is None else cv.wd.clone()) for cv in convs]
    for cv in convs:
        if cv.wp is not cv.w:
            cv.wp.fill_(7.0)
        if cv.wd is not None:
            cv.wd.fill_(7.0)
    eng._pack_all(0)
    eng._pack_all(1)
    torch.cuda.synchronize()
    for cv, (wp, wd) in zip(convs, want):
        assert torch.equal(cv.wp, wp)
        if wd is not None:
            assert torch.equal(cv.wd, wd)


@pytest.mark.gpu
def test_side_stream_really_runs_beside_the_launch_stream():
    """streams.concurrent_stream: the stream it returns overtakes a spin kernel on the launch stream (a stream that shares the
    launch stream's hardware queue cannot), also after many other streams were created (what torch.distributed does)."""
    from mvfnet_amd.streams import concurrent_stream, runs_beside
    main = torch.cuda.current_stream()
    junk = [torch.cuda.Stream() for _ in range(7)]          # shift the round-robin stream -> queue assignment
    s = concurrent_stream(main)
    assert runs_beside(main, s)
    s2 = concurrent_stream(main, avoid=[s])
    assert runs_beside(main, s2) and runs_beside(s, s2)
    del junk


@pytest.mark.gpu
def test_process_group_does_not_cost_the_stream_overlap():
    """With a torch.distributed process group up (RCCL, one rank) the train step must run at the single-process speed: the
    gradient exchange of one rank is free, so any gap means the side stream lost its hardware queue (mvfnet_amd/streams.py) or the
    queue configuration went wrong (GPU_MAX_HW_QUEUES=8 cost 46 %).  bench.py in two child processes, 15 % tolerance."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(extra_env, launcher):
        env = dict(os.environ, **extra_env)
        cmd = launcher + [os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--clips", "16"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=repo)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]

    plain = run({}, [sys.executable])
    port = str(29600 + os.getpid() % 300)
    dist_ms = run({"BENCH_FORCE_DIST": "1"}, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                                              "--master-addr", "127.0.0.1", "--master-port", port])
    if dist_ms >= 1.15 * plain:                    # one retry of both legs: a noisy neighbour must not fail the suite
        plain = min(plain, run({}, [sys.executable]))
        dist_ms = min(dist_ms, run({"BENCH_FORCE_DIST": "1"}, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                                                                "--master-addr", "127.0.0.1", "--master-port", str(int(port) + 1)]))
    assert dist_ms < 1.25 * plain, (plain, dist_ms)       # (what this guards against costs 46 %)


@pytest.mark.gpu
@pytest.mark.parametrize("env", ["wgrad_dma=0", "wgrad_dma=2", "wgrad_stages=2", "wgrad_stages=4", "wgrad_x3=0,wgrad_dma_f32=0", "wgrad_p4=0", "wgrad_big=0", "wgrad_x3=0",
                                 "wgrad3x3_direct=0", "wgrad3x3_r=4,wgrad3x3_wgs=96"],
                         ids=["register_staged_wgrad", "lds_dma_wgrad_everywhere", "two_buffer_lds_dma_wgrad", "four_stage_lds_dma_ring", "register_staged_wgrad_f32",
                              "big_tile_two_barrier_loop", "no_big_tile", "fp32_mfma_wgrad_lds_dma", "layer1_3x3_wgrad_on_the_implicit_gemm",
                              "direct_3x3_wgrad_four_row_bands_96_workgroups"])
def test_wgrad_loader_variants_forced_by_env(env):
    """The weight-gradient loader choice is a per-process policy; both forced settings re-run this file's gradient comparisons
    (conv weight gradients vs the oracle, whole-network goldens) in a child process."""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "conv_dgrad_wgrad or c1_train or norm_eval_training or bottleneck_train or stem_wgrad", "-p", "no:cacheprovider"],
                       env=policy_env(**dict(kv.split("=") for kv in env.split(","))), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


_WX3_CHILD = r"""
import ctypes as C, sys, numpy as np, torch
from mvfnet_amd import _lib
lib, check = _lib.lib, _lib.check
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
out = {}
for (n, h, cin, cout, k, stride) in SHAPES:
    gen = torch.Generator().manual_seed(cin + cout + k)
    ho = (h + 2 * (k // 2) - k) // stride + 1
    x = torch.randn(n, h, h, cin, generator=gen).cuda()
    dz = torch.randn(n, ho, ho, cout, generator=gen).cuda()
    d = _lib.ConvDesc(n, h, h, cin, cout, k, k, stride, k // 2, ho, ho, cin, 0, 0, 0, 0, 0)
    ws = torch.empty(lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device="cuda")
    dw = torch.empty(cout, cin, k, k, device="cuda")
    check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), p(dz), p(x), None, k, cin, k, cin, p(dw), p(ws), ws.numel(), None))
    torch.cuda.synchronize()
    out["%d_%d_%d_%d" % (cin, cout, k, stride)] = dw.cpu().numpy()
np.savez(sys.argv[1], **out)
"""
# (n, h, cin, cout, k, stride): pointwise long / short contraction, 64-wide tiles on either side, 3x3 with padding taps, stride 2, ragged channel tails
_WX3_SHAPES = [(16, 14, 256, 1024, 1, 1), (4, 28, 64, 256, 1, 1), (4, 28, 256, 64, 1, 1), (4, 14, 128, 128, 3, 1), (3, 13, 64, 96, 3, 2), (2, 9, 132, 36, 3, 1)]


@pytest.mark.gpu
def test_fp32_weight_gradient_on_the_bf16_matrix_cores_is_as_accurate_as_the_fp32_mfma(tmp_path):
    """[r4] wgrad_x3_kernel (the default fp32 weight gradient): dz and x are split exactly into three bf16 terms each on their way into LDS and the
    contraction over pixels is six partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Against an fp64 weight gradient of the
    same operands the error must be the fp32 ACCUMULATION error -- no larger than what the exact-fp32 MFMA kernel (MVF_POLICY=wgrad_x3=0, run in a
    child process: the switch is read once per process) leaves -- over contractions of 196 ... 3136 pixels per split."""
    import os
    import subprocess
    import sys
    src = "SHAPES = %r\n" % (_WX3_SHAPES,) + _WX3_CHILD
    res = {}
    for tag, val in (("x3", "1"), ("mfma", "0")):
        f = str(tmp_path / (tag + ".npz"))
        env = dict(os.environ, MVF_POLICY="wgrad_x3=%s" % val)
        env.update(PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        r = subprocess.run([sys.executable, "-c", src, f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(f)
    for (n, h, cin, cout, k, stride) in _WX3_SHAPES:
        gen = torch.Generator().manual_seed(cin + cout + k)
        ho = (h + 2 * (k // 2) - k) // stride + 1
        x = torch.randn(n, h, h, cin, generator=gen).double().permute(0, 3, 1, 2)
        dz = torch.randn(n, ho, ho, cout, generator=gen).double().permute(0, 3, 1, 2)
        ref = torch.nn.grad.conv2d_weight(x, (cout, cin, k, k), dz, stride=stride, padding=k // 2).numpy()
        key = "%d_%d_%d_%d" % (cin, cout, k, stride)
        e3 = np.linalg.norm(res["x3"][key] - ref) / np.linalg.norm(ref)
        e1 = np.linalg.norm(res["mfma"][key] - ref) / np.linalg.norm(ref)
        assert not np.array_equal(res["x3"][key], res["mfma"][key])           # (two different kernels did run)
        assert e3 < 2e-6 and e3 < 1.5 * e1 + 2e-8, (key, e3, e1)
        assert np.abs(res["x3"][key] - ref).max() / np.abs(ref).max() < 2e-6


# ------------------------------------------------------------------------------------------------ round 2: optimizer wire format, runner shell
@pytest.mark.parametrize("tag", ["plain", "paramwise"])
def test_replaying_the_references_optimizer_steps_reproduces_its_checkpoint(tag):
    """tests/golden/ref_ckpt_block_*.pth: the reference's own Bottleneck+MVF trained for two steps by the torch SGD its
    build_optimizer makes (plain, and with paramwise_options = one param group per parameter), saved by its save_checkpoint.
    The same two steps through BlockTrainer + the fused clip/SGD kernel (segment form for the param-wise multipliers) must land on
    the same parameters, BatchNorm buffers and momentum buffers; the optimizer state written back has torch's layout."""
    import os
    from helpers import GOLDEN
    from mvfnet_amd.runner import paramwise_multipliers
    from mvfnet_amd.train_engine import BlockTrainer
    name = "l3_like"
    N, T, Cin, planes, H, W, stride = BLOCK_CASES[name]
    g = np.load(os.path.join(GOLDEN, "ref_ckpt_block.npz"))
    ck = torch.load(os.path.join(GOLDEN, "ref_ckpt_block_%s.pth" % tag), weights_only=False)
    blk = _block(name)
    tr = BlockTrainer(blk)
    tr.lr, tr.momentum, tr.weight_decay, tr.max_norm, tr.nesterov = 0.015, 0.9, 1e-4, 40.0, True
    if tag == "paramwise":
        tr.set_param_options(paramwise_multipliers(blk, dict(bias_lr_mult=2.0, bias_decay_mult=0.0, norm_decay_mult=0.0)))
    x = torch.from_numpy(synth.synth_tensor("block_x/" + name, (N * T, Cin, H, W))).cuda()
    for step in range(2):
        y = tr.forward(x)
        tr.backward(torch.from_numpy(synth.synth_tensor("block_dy/" + name, tuple(y.shape), seed=step)).cuda())
        tr.apply_sgd()
        for k, v in blk.state_dict().items():
            if k.endswith("num_batches_tracked"):
                continue
            ref = g["%s/step1/%s" % (tag, k)] if step == 0 else g["%s/%s" % (tag, k)]
            assert rel_err(v.cpu().numpy(), ref) < (2e-5 if step == 0 else 2e-4), (step, k)
    opt = tr.optimizer_state_dict()
    assert len(opt["param_groups"]) == len(ck["optimizer"]["param_groups"]) and len(opt["state"]) == len(ck["optimizer"]["state"])
    for i, st in ck["optimizer"]["state"].items():
        assert rel_err(opt["state"][i]["momentum_buffer"].numpy(), st["momentum_buffer"].numpy()) < 2e-4, i
    # and the other direction: the reference's optimizer entry loads into a fresh engine
    tr2 = BlockTrainer(_block(name))
    tr2.load_optimizer_state_dict(ck["optimizer"])
    assert tr2.steps > 0
    for p, p2 in zip(blk.parameters(), tr2.model.parameters()):
        i0 = tr2.grad_of(p2).storage_offset()
        want = ck["optimizer"]["state"][[id(q) for q in tr2.model.parameters()].index(id(p2))]["momentum_buffer"]
        assert torch.equal(tr2.flat_mom[i0:i0 + p2.numel()].view(p2.shape).cpu(), want)


def test_inference_then_training_then_inference_uses_the_updated_head_and_backbone():
    """an inference before the train engine exists must not leave a stale packed copy of ANY weight behind -- after an
    optimizer step the eval path has to score with the updated backbone AND the updated FC (checked against the CPU oracle run on
    the model's own state_dict)."""
    from oracle import net_torch
    m = _model(50, 4)
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 96, 96, seed=3))
    labels = torch.from_numpy(synth.synth_labels(2, seed=3))
    m.eval()
    s0 = m(imgs.cuda(), None, return_loss=False)
    m.train()
    eng = m.train_engine(lr=0.5)                     # a big step, so stale weights are unmistakable
    eng.train_step(imgs.cuda(), labels.cuda())
    m.eval()
    s1 = m(imgs.cuda(), None, return_loss=False)
    with torch.no_grad():
        ref = net_torch.forward_test(imgs, {k: v.detach().cpu() for k, v in m.state_dict().items()}, 50, 4).numpy()
    assert rel_err(s1, ref) < 2e-4
    assert rel_err(s1, s0) > 1e-2                     # the step really moved the scores


def test_autograd_api_gradients_are_not_doubled_and_stale_backward_is_refused():
    """loss.backward() through Recognizer2D.forward_train: .grad = one copy of the engine's gradient (also after attach_grads(),
    where .grad aliases the flat buffer); a second backward / a backward after the next forward raises."""
    m = _model(50, 4)
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 64, 64)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2)).cuda()
    out = m(imgs, labels, return_loss=True)
    out["loss_cls"].backward()
    eng = m.train_engine()
    w = m.backbone.layer4[2].conv3.weight
    assert torch.equal(w.grad, eng.grad_of(w)) and w.grad.data_ptr() != eng.grad_of(w).data_ptr()
    want = eng.grad_of(w).clone()
    eng.attach_grads()
    out = m(imgs, labels, return_loss=True)
    out["loss_cls"].backward()
    # (the second pass is not bit-identical to the first: the BatchNorm running means the conv epilogues shift their sums by have
    # moved, and this 2-clip 64^2 network is chaotic -- doubling would read 1.0 here)
    assert rel_l2(w.grad.cpu().numpy(), want.cpu().numpy()) < 0.3 and w.grad.data_ptr() == eng.grad_of(w).data_ptr()
    with pytest.raises(RuntimeError):
        out["loss_cls"].backward()
    stale = m(imgs, labels, return_loss=True)["loss_cls"]
    m(imgs, labels, return_loss=True)
    with pytest.raises(RuntimeError, match="activations are gone"):
        stale.backward()
    # options passed to an existing engine are applied, not dropped
    assert m.train_engine(lr=0.123, max_norm=7.0) is eng and eng.lr == 0.123 and eng.max_norm == 7.0
    with pytest.raises(RuntimeError):
        m.train_engine(dtype=torch.bfloat16)


def test_head_loss_matches_cross_entropy():
    m = _model(50, 4)
    g = torch.Generator().manual_seed(5)
    scores = torch.randn(6, 400, generator=g) * 3
    labels = torch.randint(0, 400, (6, 1), generator=g)
    out = m.cls_head.loss(scores.cuda(), labels.cuda())
    assert abs(float(out["loss_cls"]) - float(F.cross_entropy(scores, labels.squeeze(1)))) < 1e-5


def test_train_network_shim_runs_the_config_end_to_end(tmp_path):
    """train_network(model, dataset, cfg) with the shipped config's sections (optimizer + paramwise_options, optimizer_config,
    lr_config, checkpoint_config, log_config, fp16 -> bf16 engine, total_epochs), host batches uploaded by the prefetcher;
    then resume_from picks the checkpoint up (torch-SGD optimizer entry) and continues."""
    from mvfnet_amd.runner import Config, train_network
    torch.manual_seed(0)
    batches = [dict(img_group=torch.from_numpy(synth.synth_clip_batch(2, 4, 64, 64, seed=i)), label=torch.from_numpy(synth.synth_labels(2, seed=i)))
               for i in range(3)]
    cfg = Config(optimizer=dict(type="SGD", lr=0.015, momentum=0.9, weight_decay=1e-4, nesterov=True,
                                paramwise_options=dict(bias_lr_mult=2.0, bias_decay_mult=0.0, norm_decay_mult=0.0)),
                 optimizer_config=dict(grad_clip=dict(max_norm=40, norm_type=2)),
                 lr_config=dict(policy="step", step=[90, 130], warmup="linear", warmup_iters=10, warmup_ratio=0.01),
                 checkpoint_config=dict(interval=1), log_config=dict(interval=1), total_epochs=2, work_dir=str(tmp_path),
                 fp16=dict(loss_scale=512.0), data=dict(videos_per_gpu=2, workers_per_gpu=0), resume_from=None, load_from=None)
    logs = []
    m = _model(50, 4)
    run = train_network(m, batches, cfg, distributed=False, validate=False, logger=logs.append)
    assert run.epoch == 2 and run.iter == 6 and run.engine.tdtype == torch.bfloat16 and run.engine.param_options
    assert len(logs) == 6 and all("loss_cls" in l for l in logs)
    ck = torch.load(str(tmp_path / "latest.pth"), weights_only=False)
    assert set(ck) == {"meta", "state_dict", "optimizer"} and len(ck["optimizer"]["param_groups"]) == len(list(m.parameters()))
    assert ck["optimizer"]["param_groups"][1]["weight_decay"] == 0.0           # backbone.bn1.weight: norm_decay_mult = 0
    m2 = _model(50, 4)
    cfg2 = Config(dict(cfg, resume_from=str(tmp_path / "latest.pth"), total_epochs=3))
    run2 = train_network(m2, batches, cfg2, logger=logs.append)
    assert run2.epoch == 3 and run2.iter == 9
    assert torch.isfinite(run2.engine.flat_params).all()


def test_fp16_optimizer_hook_maps_to_the_bf16_engine():
    """codes/core/fp16/hooks.py:12-136 -> Fp16OptimizerHook(before_run, after_train_iter) with a torch optimizer: the model trains
    through the bf16-storage engine, master weights and gradients stay fp32, the loss goes down."""
    from mvfnet_amd.dist import Fp16OptimizerHook
    m = _model(50, 4)
    hook = Fp16OptimizerHook(grad_clip=dict(max_norm=40, norm_type=2), loss_scale=512.0, distributed=False)
    hook.before_run(m)
    assert m.fp16_enabled and m.train_engine().tdtype == torch.bfloat16
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 64, 64, seed=5)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2, seed=5)).cuda()
    opt = torch.optim.SGD(m.parameters(), lr=0.015, momentum=0.9, weight_decay=1e-4, nesterov=True)
    losses = []
    for _ in range(3):
        out = m(imgs, labels, return_loss=True)
        total = hook.after_train_iter(m, opt, out["loss_cls"])
        losses.append(float(out["loss_cls"]))
        assert float(total) > 0 and all(p.grad.dtype == torch.float32 for p in m.parameters())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert all(p.dtype == torch.float32 for p in m.parameters())


def test_eight_step_training_trajectory_vs_oracle_norm_eval():
    """Eight clip + SGD-nesterov steps (norm_eval: frozen statistics keep the synthetic network well-conditioned, so a trajectory is
    comparable at all) through the fused engine vs the CPU oracle stepping the same state_dict: every loss of the trajectory and the
    parameters after it -- no drift of forward, backward, clip or momentum over steps; the bf16 engine follows within its noise."""
    import mvfnet_amd
    from oracle import net_torch
    cfg = mvfnet_amd.mvfnet_config(50, 4, dropout_ratio=0.0)
    cfg["backbone"]["norm_eval"] = True
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 64, 64, seed=31))
    labels = torch.from_numpy(synth.synth_labels(2, seed=31))

    def fresh():
        m = mvfnet_amd.build_recognizer(cfg, None, dict(average_clips=None))
        sd = m.state_dict()
        vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
        m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True)
        return m.cuda().train()

    LR = 0.003          # (the config's 0.015 makes this 2-clip problem overshoot: 5.8, 1.8, 0.7, 1.2, 8.9, ... -- fp32 still agrees to 2e-3, bf16 noise does not)
    m = fresh()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    mom, ref_losses = {}, []
    torch.set_num_threads(min(16, torch.get_num_threads()))
    for _ in range(8):
        for p in params.values():
            p.grad = None
        f = net_torch.backbone(imgs.reshape(-1, 3, 64, 64), sd, 50, 4, training=False)
        loss = F.cross_entropy(net_torch.head(f, sd, 4), labels.squeeze(1))
        loss.backward()
        with torch.no_grad():
            net_torch.sgd_nesterov_step(params, {k: v.grad for k, v in params.items()}, mom, lr=LR)
        ref_losses.append(float(loss.detach()))
    eng = m.train_engine(lr=LR)
    losses = [float(eng.train_step(imgs.cuda(), labels.cuda())) for _ in range(8)]
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 2e-3 * abs(b), (losses, ref_losses)
    assert ref_losses[-1] < 0.7 * ref_losses[0]                       # the trajectory really trains
    got = dict(m.named_parameters())
    for k in ("backbone.conv1.weight", "backbone.layer2.1.conv2.weight", "backbone.layer3.2.conv1.shift_conv.weight", "backbone.layer4.2.bn3.weight",
              "cls_head.new_fc.weight"):
        assert rel_err(got[k].detach().cpu().numpy(), params[k].detach().numpy()) < 5e-3, k
    m16 = fresh()
    e16 = m16.train_engine(dtype=torch.bfloat16, lr=LR)
    l16 = [float(e16.train_step(imgs.cuda(), labels.cuda())) for _ in range(8)]
    for a, b in zip(l16, ref_losses):
        assert abs(a - b) < 5e-2 * abs(b) + 2e-2, (l16, ref_losses)


# ------------------------------------------------------------------------------------------------ round 3
def test_optimizer_hooks_with_the_engine_backed_optimizer_keep_the_gradient_clip():
    """[r3, advisor] The reference's _dist_train pairing -- build_optimizer(model, cfg.optimizer) + DistOptimizerHook(grad_clip) (train.py:
    159-196, dist_utils.py:52-67) -- with build_optimizer's EngineSGD: the clip must reach the update (it acted on autograd's copies of the
    gradients and was silently dropped).  With a max_norm far below the gradient norm, the hook path must equal the fused
    train_step(max_norm) bit for bit, differ from an un-clipped step, and return the pre-clip norm like clip_grad_norm_."""
    from mvfnet_amd.dist import DistOptimizerHook, Fp16OptimizerHook
    from mvfnet_amd.runner import build_optimizer
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 64, 64, seed=9)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2, seed=9)).cuda()
    ocfg = dict(type="SGD", lr=0.015, momentum=0.9, weight_decay=1e-4, nesterov=True)

    def run(kind, max_norm):
        m = _model(50, 4)
        m.train()
        if kind == "fused":
            eng = m.train_engine(lr=0.015, momentum=0.9, weight_decay=1e-4, max_norm=max_norm)
            eng.dropout = 0.0
            for _ in range(2):
                eng.train_step(imgs, labels)
            return eng.flat_params.clone(), float(eng.norm_out[0])
        m.cls_head.dropout = None
        hook = (DistOptimizerHook if kind == "hook" else Fp16OptimizerHook)(grad_clip=None if max_norm is None else dict(max_norm=max_norm, norm_type=2))
        opt = build_optimizer(m, ocfg)
        total = None
        for _ in range(2):
            out = m(imgs, labels, return_loss=True)
            total = hook.after_train_iter(m, opt, out["loss_cls"])
        assert opt.engine.max_norm is None                    # the hook leaves the optimizer as it found it
        return opt.engine.flat_params.clone(), (None if total is None else float(total))

    p_fused, n_fused = run("fused", 0.05)
    p_hook, n_hook = run("hook", 0.05)
    p_free, n_free = run("hook", None)
    assert n_fused > 1.0                                      # the clip is active: the norm is far above max_norm
    assert torch.equal(p_hook, p_fused) and n_hook == n_fused
    assert n_free is None and not torch.equal(p_free, p_fused)
    with pytest.raises(NotImplementedError):
        m = _model(50, 4)
        out = m(imgs, labels, return_loss=True)
        DistOptimizerHook(grad_clip=dict(max_norm=1.0, norm_type=1)).after_train_iter(m, build_optimizer(m, ocfg), out["loss_cls"])


@pytest.mark.parametrize("name", sorted(BLOCK_CASES))
def test_standalone_bottleneck_forward_is_an_autograd_node_over_the_hip_block(name):
    """[r3] reference resnet.py:208-244: `block(x)` on its own.  Output, input gradient, every parameter gradient (through autograd:
    y.backward(dy)) and the running statistics against the reference's own run of the block (tests/golden/block_cases.npz)."""
    N, T, Cin, planes, H, W, stride = BLOCK_CASES[name]
    g = golden("block_cases.npz")
    blk = _block(name)
    x = torch.from_numpy(synth.synth_tensor("block_x/" + name, (N * T, Cin, H, W))).cuda().requires_grad_(True)
    y = blk(x)
    assert y.requires_grad and rel_err(y.detach().cpu().numpy(), g[name + "/train/y"]) < 1e-5
    y.backward(torch.from_numpy(synth.synth_tensor("block_dy/" + name, tuple(y.shape))).cuda())
    assert rel_err(x.grad.cpu().numpy(), g[name + "/train/dx"]) < 1e-4
    for pn, p in blk.named_parameters():
        assert p.grad is not None and rel_err(p.grad.cpu().numpy(), g[name + "/train/grad/" + pn]) < 2e-4, pn
    for bn_, b in blk.named_buffers():
        ref = g[name + "/train/buf/" + bn_]
        assert (int(b) == int(ref)) if ref.dtype.kind == "i" else (rel_err(b.cpu().numpy(), ref) < 1e-5), bn_
    # eval mode (folded running statistics) on a FRESH block against the reference's eval run (the train forward above has updated the
    # running statistics of `blk`, as it must); a second backward of the first output is refused
    with torch.no_grad():
        ye = _block(name).eval()(x.detach())
    assert rel_err(ye.cpu().numpy(), g[name + "/eval/y"]) < 1e-4
    with pytest.raises(RuntimeError):
        y.backward(torch.ones_like(y))
    with pytest.raises(RuntimeError):
        blk(x.detach().cpu())


def test_dist_eval_hook_with_the_references_signature_and_validate_registration(tmp_path):
    """[r3] eval_hooks.py:86-104 / train.py:192-196: DistEvalTopKAccuracyHook(dataset, interval, k, dist) scores dataset[i] one by one in eval
    mode and reads the labels from dataset.video_infos; train_network(validate=True) registers it for a Dataset object under cfg.data.val."""
    from mvfnet_amd.evaluation import DistEvalTopKAccuracyHook, top_k_accuracy
    from mvfnet_amd.runner import Config, train_network

    class Val(object):
        def __init__(self, n):
            self.video_infos = [dict(label=int(l)) for l in synth.synth_labels(n, seed=3).ravel()]

        def __len__(self):
            return len(self.video_infos)

        def __getitem__(self, i):
            return dict(img_group=torch.from_numpy(synth.synth_clip_batch(1, 4, 64, 64, seed=40 + i)[0]), label=torch.tensor([self.video_infos[i]["label"]]))

    val = Val(5)
    m = _model(50, 4)
    with pytest.raises(TypeError):
        DistEvalTopKAccuracyHook(dict(type="RawFramesDataset"))
    hook = DistEvalTopKAccuracyHook(val, interval=1, k=(1, 5), dist=False)

    class R(object):
        epoch, model = 1, m
    out = hook.after_train_epoch(R())
    m.eval()
    rows = [m(val[i]["img_group"].unsqueeze(0).cuda(), None, return_loss=False).squeeze() for i in range(5)]
    want = top_k_accuracy(rows, [v["label"] for v in val.video_infos], k=(1, 5))
    assert out["top1 acc"] == float(want[0]) and out["top5 acc"] == float(want[1]) and m.training is False
    batches = [dict(img_group=torch.from_numpy(synth.synth_clip_batch(2, 4, 64, 64, seed=i)), label=torch.from_numpy(synth.synth_labels(2, seed=i))) for i in range(2)]
    cfg = Config(optimizer=dict(type="SGD", lr=0.001, momentum=0.9, weight_decay=1e-4, nesterov=True), optimizer_config=dict(grad_clip=dict(max_norm=40, norm_type=2)),
                 lr_config=dict(policy="step", step=[90]), checkpoint_config=dict(interval=0), log_config=dict(interval=0), total_epochs=2, eval_interval=2,
                 work_dir=str(tmp_path), data=dict(videos_per_gpu=2, workers_per_gpu=0, val=val))
    logs = []
    run = train_network(_model(50, 4), batches, cfg, distributed=False, validate=True, logger=logs.append)
    assert len(run.hooks) == 1 and isinstance(run.hooks[0], DistEvalTopKAccuracyHook)
    assert len(run.hooks[0].history) == 1 and run.hooks[0].history[0]["epoch"] == 2                 # eval_interval = 2: after the second epoch only
    assert any("Epoch(val) [2]" in l and "top1 acc" in l for l in logs)
    with pytest.raises(NotImplementedError):
        train_network(_model(50, 4), batches, Config(dict(cfg, data=dict(videos_per_gpu=2, val=dict(type="RawFramesDataset")))), validate=True)


def test_device_prefetcher_slot_reuse_waits_for_the_consumers_kernels():
    """[r3] DevicePrefetcher keeps two persistent device buffers per key: the upload of batch k + 2 into the slot batch k used must wait (on the
    copy stream) for the consumer's ASYNCHRONOUS work on batch k.  The consumer parks a ~20 ms spin kernel in front of its read of every batch,
    so an upload that does not wait would overwrite the data first; pinned and pageable host batches alternate."""
    from mvfnet_amd.runner import DevicePrefetcher
    host = []
    for i in range(7):
        t = torch.full((1 << 20,), float(i + 1))
        host.append(dict(x=t.pin_memory() if i % 2 else t, tag=i))
    sums, ptrs = [], set()
    for b in DevicePrefetcher(host):
        assert b["x"].is_cuda and b["tag"] == len(sums)
        ptrs.add(b["x"].data_ptr())
        torch.cuda._sleep(40_000_000)                       # the consumer's kernels are far behind the host
        sums.append(b["x"].double().sum())
    torch.cuda.synchronize()
    assert [float(s) for s in sums] == [float((i + 1) * (1 << 20)) for i in range(7)]
    assert len(ptrs) == 2                                   # two persistent landing buffers, no per-batch allocation


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 9, 7, 64, 256, False), (3, 6, 6, 128, 96, True), (1, 14, 14, 256, 1024, True), (2, 7, 7, 512, 2048, False),
                                   (5, 5, 5, 32, 64, False)], ids=str)
def test_conv_bnapply_pass_equals_bn_apply_on_the_stored_conv_output(shape, dtype):
    """[r3] mvf_conv2d_nhwc_fwd_bnapply (reference resnet.py:229-244: conv3 -> bn3 -> += identity -> relu, as a second pass of the conv): `out` and the
    sign bits equal mvf_bn_apply_bits on the z3 the first pass (mvf_conv2d_nhwc_fwd_stats) stored, BIT FOR BIT -- with a plain identity and with a
    downsample branch whose BatchNorm is applied to the residual operand."""
    from mvfnet_amd import _lib
    from mvfnet_amd._lib import ConvDesc
    lib, check = _lib.lib, _lib.check
    n, h, w, cin, cout, with_rbn = shape
    dt = 0 if dtype == torch.float32 else 1
    g = torch.Generator().manual_seed(cin + cout)
    m = n * h * w
    x = torch.randn(m, cin, generator=g).to(dtype).cuda()
    wgt = (torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).cuda()
    wp = torch.empty(cout, 1, 1, cin, dtype=dtype, device="cuda")
    check(lib.mvf_pack_conv_weight(P(wgt), cout, cin, 1, 1, 1, cin, None, P(wp), dt, None))
    res = torch.randn(m, cout, generator=g).to(dtype).cuda()
    scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.3).cuda()
    rscale = (torch.rand(cout, generator=g) + 0.5).cuda() if with_rbn else None
    rshift = (torch.randn(cout, generator=g) * 0.3).cuda() if with_rbn else None
    d = ConvDesc(n, h, w, cin, cout, 1, 1, 1, 0, h, w, cin, dt, 0, 0, 0, 0, 0)
    ws = torch.zeros(max(lib.mvf_conv2d_workspace_bytes(C.byref(d)), 1), dtype=torch.uint8, device="cuda")
    rows = lib.mvf_conv2d_stats_rows(C.byref(d))
    part = torch.empty(cout, rows, 2, device="cuda")
    z3 = torch.empty(m, cout, dtype=dtype, device="cuda")
    check(lib.mvf_conv2d_nhwc_fwd_stats(C.byref(d), P(x), None, P(wp), P(z3), P(part), None, P(ws), ws.numel(), None))
    want, wbits = torch.empty_like(z3), torch.empty(m, cout // 4, dtype=torch.uint8, device="cuda")
    check(lib.mvf_bn_apply_bits(P(z3), m, cout, P(scale), P(shift), P(res), P(rscale), P(rshift), 1, P(want), P(wbits), dt, None))
    got, gbits = torch.full_like(z3, 7.0), torch.full((m, cout // 4), 255, dtype=torch.uint8, device="cuda")
    check(lib.mvf_conv2d_nhwc_fwd_bnapply(C.byref(d), P(x), None, P(wp), P(scale), P(shift), P(res), P(rscale), P(rshift), P(got), P(gbits), P(ws), ws.numel(), None))
    torch.cuda.synchronize()
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    assert torch.equal(gbits, wbits)
    assert 0.2 < float((want > 0).float().mean()) < 0.8          # the ReLU and the bits are exercised on both sides


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 9, 7, 64, 256), (3, 56, 56, 64, 256), (2, 9, 7, 64, 128), (3, 6, 6, 128, 96), (1, 14, 14, 256, 1024), (5, 5, 5, 32, 64),
                                   (4, 16, 16, 128, 512)], ids=str)
def test_bn_backward_on_the_recomputed_conv_equals_the_backward_on_the_stored_output(shape, dtype):
    """[r3] mvf_conv2d_nhwc_fwd_bnbwd_sums / _apply (BatchNorm backward of bn3 without a stored z3) against mvf_bn_bwd_reduce / mvf_bn_bwd_apply_masked
    (mask mode 4) on the z3 the forward stored: dz3 BIT FOR BIT given the same dgamma / dbeta, the sums to fp32 summation order; and the
    statistics-only first pass (y = NULL) leaves the partial sums unchanged."""
    from mvfnet_amd import _lib
    from mvfnet_amd._lib import ConvDesc
    lib, check = _lib.lib, _lib.check
    n, h, w, cin, cout = shape
    dt = 0 if dtype == torch.float32 else 1
    g_ = torch.Generator().manual_seed(cin * 3 + cout)
    m = n * h * w
    x = torch.randn(m, cin, generator=g_).to(dtype).cuda()
    wgt = (torch.randn(cout, cin, 1, 1, generator=g_) * (2.0 / cin) ** 0.5).cuda()
    wp = torch.empty(cout, 1, 1, cin, dtype=dtype, device="cuda")
    check(lib.mvf_pack_conv_weight(P(wgt), cout, cin, 1, 1, 1, cin, None, P(wp), dt, None))
    gout = torch.randn(m, cout, generator=g_).to(dtype).cuda()
    bits = torch.randint(0, 16, (m, cout // 4), generator=g_, dtype=torch.uint8).cuda()
    gamma = (torch.rand(cout, generator=g_) + 0.5).cuda()
    mean, invstd = (torch.randn(cout, generator=g_) * 0.1).cuda(), (torch.rand(cout, generator=g_) + 0.5).cuda()
    zero = torch.zeros(cout, device="cuda")
    d = ConvDesc(n, h, w, cin, cout, 1, 1, 1, 0, h, w, cin, dt, 0, 0, 0, 0, 0)
    ws = torch.zeros(max(lib.mvf_conv2d_workspace_bytes(C.byref(d)), 1), dtype=torch.uint8, device="cuda")
    rows = lib.mvf_conv2d_stats_rows(C.byref(d))
    part, part0 = torch.empty(cout, rows, 2, device="cuda"), torch.full((cout, rows, 2), float("nan"), device="cuda")
    z3 = torch.empty(m, cout, dtype=dtype, device="cuda")
    check(lib.mvf_conv2d_nhwc_fwd_stats(C.byref(d), P(x), None, P(wp), P(z3), P(part), None, P(ws), ws.numel(), None))
    check(lib.mvf_conv2d_nhwc_fwd_stats(C.byref(d), P(x), None, P(wp), None, P(part0), None, P(ws), ws.numel(), None))      # statistics only
    torch.cuda.synchronize()
    if dtype == torch.bfloat16 and cin == 64 and cout in (128, 256):
        # [r3] these two sum passes run on their own kernel (csrc/pw_sums.hip: channel = lane, one partial row per workgroup): same column sums,
        # another summation order and partial layout
        assert torch.isfinite(part0).all()
        assert rel_err(part0.double().sum(1).cpu().numpy(), part.double().sum(1).cpu().numpy()) < 2e-6
    else:
        assert torch.equal(part, part0)
    # reference: reduce + apply on the stored z3
    bws = torch.empty(lib.mvf_bn_workspace_bytes(m, cout), dtype=torch.uint8, device="cuda")
    dg, db = torch.empty(cout, device="cuda"), torch.empty(cout, device="cuda")
    check(lib.mvf_bn_bwd_reduce(P(gout), cout, P(z3), P(bits), m, cout, P(mean), P(invstd), P(zero), P(zero), 4, None, P(dg), P(db), P(bws), bws.numel(), dt, None))
    want = torch.empty_like(z3)
    check(lib.mvf_bn_bwd_apply_masked(P(gout), cout, P(z3), P(bits), m, cout, P(gamma), P(mean), P(invstd), P(zero), P(zero), P(dg), P(db), 4, P(want), dt, None))
    # recompute path
    sp = torch.full((cout, rows, 2), float("nan"), device="cuda")
    check(lib.mvf_conv2d_nhwc_fwd_bnbwd_sums(C.byref(d), P(x), None, P(wp), P(gout), P(bits), P(mean), P(invstd), P(sp), P(ws), ws.numel(), None))
    dg2, db2 = torch.empty(cout, device="cuda"), torch.empty(cout, device="cuda")
    check(lib.mvf_bn_bwd_finalize(P(sp), rows, cout, P(dg2), P(db2), None))
    got = torch.full_like(z3, 7.0)
    check(lib.mvf_conv2d_nhwc_fwd_bnbwd_apply(C.byref(d), P(x), None, P(wp), P(gout), P(bits), P(gamma), P(mean), P(invstd), P(dg), P(db), P(got), P(ws), ws.numel(), None))
    torch.cuda.synchronize()
    assert torch.isfinite(sp).all()
    assert rel_err(dg2.cpu().numpy(), dg.cpu().numpy()) < 3e-6 and rel_err(db2.cpu().numpy(), db.cpu().numpy()) < 3e-6
    if dtype == torch.bfloat16:
        assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    else:           # fp32: the two kernels contract a * (g - d0 - (z - mu) * kx) into fused multiply-adds differently (last-bit differences)
        assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < 1e-6


# ------------------------------------------------------------------------------------------------ [r4] advisor findings of round 3
def test_standalone_block_call_inside_a_model_leaves_the_models_engine_in_charge():
    """A block of a model that already has a train engine, called on its own (feature hooks, debugging): the block's trainer must not
    re-home the parameters -- they stay views of the MODEL engine's flat buffer, the next optimizer step still moves what the block
    reads, and the block's gradients are copies (the model engine's flat gradient is untouched by the stand-alone call)."""
    m = _model(50, 4, dropout=0.0)
    eng = m.train_engine()
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 64, 64)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2)).cuda()
    eng.train_step(imgs, labels)
    blk = m.backbone.layer1[1]
    lo, hi = eng.flat_params.data_ptr(), eng.flat_params.data_ptr() + eng.flat_params.numel() * 4
    before = [p.data_ptr() for p in blk.parameters()]
    grads_before = eng.flat_grads.clone()
    x = torch.randn(8, 256, 16, 16, device="cuda", requires_grad=True)
    y = blk(x)
    y.sum().backward()
    torch.cuda.synchronize()
    assert [p.data_ptr() for p in blk.parameters()] == before and all(lo <= a < hi for a in before)
    assert torch.equal(eng.flat_grads, grads_before)                      # the stand-alone backward wrote its own buffer
    assert all(p.grad is not None and p.grad.data_ptr() != eng.grad_of(p).data_ptr() for p in blk.parameters())
    w0 = blk.conv2.weight.detach().clone()
    y0 = blk(x.detach()).detach().clone()
    eng.train_step(imgs, labels)                                          # the model's optimizer still owns the block's weights
    torch.cuda.synchronize()
    assert not torch.equal(blk.conv2.weight.detach(), w0)
    assert not torch.equal(blk(x.detach()).detach(), y0)                  # and the stand-alone forward reads the UPDATED weights
    with pytest.raises(RuntimeError):
        blk._trainer.apply_sgd()                                          # a store that does not own its parameters has no optimizer


def test_engine_backed_optimizer_zero_grad_drops_the_autograd_copies():
    from mvfnet_amd.runner import EngineSGD
    m = _model(50, 4, dropout=0.0)
    opt = EngineSGD(m, lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 64, 64)).cuda()
    labels = torch.from_numpy(synth.synth_labels(2)).cuda()
    norms = []
    for _ in range(2):
        opt.zero_grad()
        assert all(p.grad is None for p in m.parameters())
        loss = m(imgs, labels, return_loss=True)["loss_cls"]
        loss.backward()
        norms.append(float(m.backbone.conv1.weight.grad.norm()))
        opt.step()
    assert norms[1] < 1.9 * norms[0]              # .grad is this step's gradient, not the running sum of all steps


def test_trailing_partial_batch_reuses_the_engines_buffers():
    """drop_last=False (the reference's loader): the epoch's last, smaller batch must not allocate a second resident set of activation
    buffers, and -- with the running statistics put back to what a fresh model holds, so that the statistics sums are shifted by the same
    constants -- its loss and gradients equal a fresh engine's on the same clips BIT for bit (the small-M launches take the direct-kernel
    fallbacks; the views into the larger allocations change nothing)."""
    for dtype in (torch.float32, torch.bfloat16):
        m = _model(50, 4, dropout=0.0)
        buffers0 = {k: v.clone() for k, v in m.named_buffers()}
        eng = m.train_engine(dtype=dtype)
        imgs = torch.from_numpy(synth.synth_clip_batch(3, 4, 96, 96)).cuda()
        labels = torch.from_numpy(synth.synth_labels(3)).cuda()
        eng.forward(imgs, labels)
        eng.backward()
        torch.cuda.synchronize()
        n_alloc = len(eng._caps)
        mem = torch.cuda.memory_allocated()
        with torch.no_grad():
            for k, v in m.named_buffers():
                v.copy_(buffers0[k])
        l2 = eng.forward(imgs[:1], labels[:1])
        eng.backward()
        torch.cuda.synchronize()
        assert len(eng._caps) == n_alloc                                   # every call site re-used its allocation
        assert torch.cuda.memory_allocated() - mem < 64 << 20
        g2 = eng.flat_grads.clone()
        l3 = eng.forward(imgs, labels)                                     # and the full batch again, in the same storage
        eng.backward()
        torch.cuda.synchronize()
        assert len(eng._caps) == n_alloc and torch.isfinite(l3).all()
        m1 = _model(50, 4, dropout=0.0)
        e1 = m1.train_engine(dtype=dtype)
        l1 = e1.forward(imgs[:1], labels[:1])
        e1.backward()
        torch.cuda.synchronize()
        assert float(l1) == float(l2)
        assert torch.equal(g2, e1.flat_grads)
        # [r5] every shape asked for under one key is a view of one allocation: a second shape INSIDE one step would alias the first
        e1.buf("alias_probe", (4, 8))
        with pytest.raises(RuntimeError, match="inside one step"):
            e1.buf("alias_probe", (2, 8))
        e1.forward_count += 1                    # ... the next step may (the partial batch above did)
        e1.buf("alias_probe", (2, 8))


def test_z3_free_block_gradients_match_stored_z3_block():
    """[r3 advisor] The z3-free path takes bn3's statistics from one kernel's rounded z3 (pw_sums.hip) and apply / backward from the
    recomputed conv's: individual z3 elements may differ by one bf16 ulp between the passes.  End to end on layer1's bf16 block shape
    (256 -> 64 -> 256, 56 x 56) the gradients of z3_free = 1 and = 0 agree to the bf16 noise floor."""
    from mvfnet_amd.backbones.resnet import Bottleneck
    from mvfnet_amd.train_engine import BlockTrainer
    res = {}
    for z3f in (0, 1):
        torch.manual_seed(3)
        blk = Bottleneck(256, 64).cuda().train()
        with torch.no_grad():
            for bn in (blk.bn1, blk.bn2, blk.bn3):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.normal_(0, 0.2)
        tr = BlockTrainer(blk, dtype=torch.bfloat16)
        tr.z3_free = z3f
        assert tr.blk.z3_free(tr) == bool(z3f)
        x = torch.randn(8, 256, 56, 56, device="cuda")
        dy = torch.randn(8, 256, 56, 56, device="cuda")
        y = tr.forward(x).float().clone()
        dx = tr.backward(dy).float().clone()
        torch.cuda.synchronize()
        res[z3f] = (y, dx, tr.flat_grads.clone())
    assert rel_l2(res[1][0].cpu().numpy(), res[0][0].cpu().numpy()) < 4e-3          # out: bf16 ulp flips of single elements
    assert rel_l2(res[1][1].cpu().numpy(), res[0][1].cpu().numpy()) < 1e-2
    assert rel_l2(res[1][2].cpu().numpy(), res[0][2].cpu().numpy()) < 1e-2


# ------------------------------------------------------------------------------------------------ [r4] BatchNorm backward apply + weight gradient in one pass
def _bnwg_inputs(m, c, k, seed, nbn=1):
    gen = torch.Generator().manual_seed(seed)
    dev = "cuda"
    bf = torch.bfloat16
    t = dict(g=torch.randn(m, c, generator=gen).to(dev, bf), bits=torch.randint(0, 16, (m, c // 4), generator=gen, dtype=torch.uint8).to(dev))
    for b in range(nbn):
        t["z%d" % b] = (torch.randn(m, c, generator=gen) * 1.4 + 0.3).to(dev, bf)
        t["x%d" % b] = torch.randn(m, k, generator=gen).to(dev, bf)
        t["gamma%d" % b] = (torch.rand(c, generator=gen) + 0.5).to(dev)
        t["mean%d" % b] = (torch.randn(c, generator=gen) * 0.3).to(dev)
        t["invstd%d" % b] = (torch.rand(c, generator=gen) + 0.4).to(dev)
        t["scale%d" % b] = t["gamma%d" % b] * t["invstd%d" % b]
        t["shift%d" % b] = (torch.randn(c, generator=gen) * 0.2).to(dev)
    return t


@pytest.mark.parametrize("case", [(3000, 256, 64, 4), (64 * 57 + 17, 512, 128, 4), (5000, 64, 256, 2), (4099, 128, 512, 2), (777, 256, 256, 2), (256, 256, 64, 4)], ids=str)
def test_bn_bwd_apply_wgrad_equals_separate_apply_and_wgrad(case):
    """mvf_bn_bwd_apply_wgrad (csrc/bnbwd_wgrad.hip; autograd of resnet.py:213-244): dz BIT-identical to mvf_bn_bwd_apply_masked, the weight
    gradient (slabs summed by mvf_wgrad_slab_reduce) against dz^T x in fp32 and against mvf_conv2d_nhwc_wgrad on the same dz; ragged row
    counts, two column tiles (c = 512), two k tiles (k = 512), both gate modes."""
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    m, c, k, mode = case
    t = _bnwg_inputs(m, c, k, seed=m + c + k)
    dev = "cuda"
    ws = torch.empty(lib.mvf_bn_workspace_bytes(m, c), dtype=torch.uint8, device=dev)
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    ymask = t["bits"] if mode == 4 else None
    check(lib.mvf_bn_bwd_reduce(P(t["g"]), c, P(t["z0"]), P(ymask), m, c, P(t["mean0"]), P(t["invstd0"]), P(t["scale0"]), P(t["shift0"]), mode, None,
                                P(dg), P(db), P(ws), ws.numel(), 1, None))
    dz_ref = torch.empty(m, c, device=dev, dtype=torch.bfloat16)
    check(lib.mvf_bn_bwd_apply_masked(P(t["g"]), c, P(t["z0"]), P(ymask), m, c, P(t["gamma0"]), P(t["mean0"]), P(t["invstd0"]), P(t["scale0"]), P(t["shift0"]),
                                      P(dg), P(db), mode, P(dz_ref), 1, None))
    ns = lib.mvf_bn_bwd_wgrad_splits(m, c, k, 1, mode)
    assert ns > 0
    nb = lib.mvf_bn_bwd_wgrad_slab_bytes(m, c, k, 1, mode)
    assert nb >= ns * c * k * 4
    slabs = torch.full((nb // 4,), float("nan"), device=dev)
    dz = torch.full((m, c), 7.0, device=dev, dtype=torch.bfloat16)
    check(lib.mvf_bn_bwd_apply_wgrad(P(t["g"]), c, P(t["z0"]), P(ymask), m, c, P(t["gamma0"]), P(t["mean0"]), P(t["invstd0"]), P(t["scale0"]), P(t["shift0"]),
                                     P(dg), P(db), mode, P(dz), P(t["x0"]), k, k, P(slabs), nb, 1, None), "bn_bwd_apply_wgrad")
    dw = torch.full((c, k, 1, 1), float("nan"), device=dev)
    check(lib.mvf_wgrad_slab_reduce(P(slabs), ns, c, k, P(dw), None))
    torch.cuda.synchronize()
    assert torch.equal(dz.view(torch.int16), dz_ref.view(torch.int16))
    want = dz_ref.float().t() @ t["x0"].float()
    assert torch.isfinite(dw).all()
    assert rel_err(dw.view(c, k).cpu().numpy(), want.cpu().numpy()) < 2e-5
    # and the GEMM it replaces, on the same dz
    d = _lib.ConvDesc(1, m, 1, k, c, 1, 1, 1, 0, m, 1, k, 1, 0, 0, 0, 0, 0)
    wsz = lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d))
    wws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    dw2 = torch.empty(c, k, 1, 1, device=dev)
    check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), P(dz_ref), P(t["x0"]), None, 1, k, 1, k, P(dw2), P(wws), wsz, None))
    torch.cuda.synchronize()
    assert rel_err(dw.cpu().numpy(), dw2.cpu().numpy()) < 2e-5
    # shapes that are not built say so (the engine then keeps the separate kernels); fp32 storage is refused
    assert lib.mvf_bn_bwd_wgrad_splits(m, 96, 64, 1, 4) == 0 and lib.mvf_bn_bwd_wgrad_splits(m, 1024, 256, 1, 4) == 0
    assert lib.mvf_bn_bwd_apply_wgrad(P(t["g"]), c, P(t["z0"]), P(ymask), m, c, P(t["gamma0"]), P(t["mean0"]), P(t["invstd0"]), P(t["scale0"]), P(t["shift0"]),
                                      P(dg), P(db), mode, P(dz), P(t["x0"]), k, k, P(slabs), nb, 0, None) == -5


@pytest.mark.parametrize("case", [(3000, 256, 64, True), (64 * 40 + 5, 512, 128, False), (50176, 256, 64, True)], ids=str)
def test_bn_bwd_pair_wgrad_equals_pair_and_two_wgrads(case):
    """mvf_bn_bwd_pair_wgrad: dgamma / dbeta / dz of both BatchNorms BIT-identical to mvf_bn_bwd_pair, both weight gradients (or, k = 128,
    conv a's only) against dz^T x in fp32."""
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    m, c, k, both = case
    t = _bnwg_inputs(m, c, k, seed=m + c, nbn=2)
    dev = "cuda"
    nbw = lib.mvf_bn_workspace_bytes(m, c)
    ws = torch.empty(2 * nbw, dtype=torch.uint8, device=dev)
    ref = dict(dga=torch.empty(c, device=dev), dba=torch.empty(c, device=dev), dgb=torch.empty(c, device=dev), dbb=torch.empty(c, device=dev),
               dza=torch.empty(m, c, device=dev, dtype=torch.bfloat16), dzb=torch.empty(m, c, device=dev, dtype=torch.bfloat16))
    got = {k_: torch.full_like(v, 3.0) for k_, v in ref.items()}
    check(lib.mvf_bn_bwd_pair(P(t["g"]), c, P(t["z0"]), P(t["z1"]), P(t["bits"]), m, c, P(t["gamma0"]), P(t["mean0"]), P(t["invstd0"]), P(ref["dga"]), P(ref["dba"]),
                              P(t["gamma1"]), P(t["mean1"]), P(t["invstd1"]), P(ref["dgb"]), P(ref["dbb"]), P(ref["dza"]), P(ref["dzb"]), P(ws), ws.numel(), 1, None))
    ns = lib.mvf_bn_bwd_wgrad_splits(m, c, k, 2, 4)
    nb = lib.mvf_bn_bwd_wgrad_slab_bytes(m, c, k, 2, 4)
    assert ns > 0
    sa, sb = torch.full((nb // 4,), float("nan"), device=dev), torch.full((nb // 4,), float("nan"), device=dev)
    check(lib.mvf_bn_bwd_pair_wgrad(P(t["g"]), c, P(t["z0"]), P(t["z1"]), P(t["bits"]), m, c, P(t["gamma0"]), P(t["mean0"]), P(t["invstd0"]), P(got["dga"]), P(got["dba"]),
                                    P(t["gamma1"]), P(t["mean1"]), P(t["invstd1"]), P(got["dgb"]), P(got["dbb"]), P(got["dza"]), P(got["dzb"]),
                                    P(t["x0"]), k, P(t["x1"]) if both else None, k, k, P(sa), P(sb) if both else None, nb, P(ws), ws.numel(), 1, None), "pair_wgrad")
    dwa, dwb = torch.empty(c, k, 1, 1, device=dev), torch.empty(c, k, 1, 1, device=dev)
    check(lib.mvf_wgrad_slab_reduce(P(sa), ns, c, k, P(dwa), None))
    if both:
        check(lib.mvf_wgrad_slab_reduce(P(sb), ns, c, k, P(dwb), None))
    torch.cuda.synchronize()
    for k_ in ref:
        assert torch.equal(got[k_].view(torch.int16 if got[k_].dtype == torch.bfloat16 else torch.int32),
                           ref[k_].view(torch.int16 if ref[k_].dtype == torch.bfloat16 else torch.int32)), k_
    assert rel_err(dwa.view(c, k).cpu().numpy(), (ref["dza"].float().t() @ t["x0"].float()).cpu().numpy()) < 2e-5
    if both:
        assert rel_err(dwb.view(c, k).cpu().numpy(), (ref["dzb"].float().t() @ t["x1"].float()).cpu().numpy()) < 2e-5
