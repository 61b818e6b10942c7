"""GPU parity of the one-pass backward of a z3-free bottleneck's last conv (csrc/pw_bwd_fused.hip, mvf_conv1x1_bwd_fused): autograd of
Bottleneck.forward (reference codes/models/backbones/resnet.py:229-244) for out = relu(bn3(conv3(a2)) + identity), a2 = relu(bn2(z2)) --
against the three launches it replaces (mvf_conv2d_nhwc_fwd_bnbwd_apply -> mvf_conv2d_nhwc_dgrad_bnsums -> weight gradient), which are themselves
pinned to the reference by tests/test_train_gpu.py, and against an fp32 restatement with the same rounding points."""
import ctypes as C

import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _inputs(m, seed):
    gen = torch.Generator().manual_seed(seed)
    dev, bf = "cuda", torch.bfloat16
    cin, cout = 64, 256
    t = dict(a2=torch.relu(torch.randn(m, cin, generator=gen)).to(dev, bf), g=torch.randn(m, cout, generator=gen).to(dev, bf),
             bits=torch.randint(0, 16, (m, cout // 4), generator=gen, dtype=torch.uint8).to(dev),
             z2=(torch.randn(m, cin, generator=gen) * 1.2 + 0.1).to(dev, bf),
             w=(torch.randn(cout, cin, 1, 1, generator=gen) * (2.0 / cin) ** 0.5).to(dev),
             gamma=(torch.rand(cout, generator=gen) + 0.5).to(dev), mean=(torch.randn(cout, generator=gen) * 0.1).to(dev),
             invstd=(torch.rand(cout, generator=gen) + 0.5).to(dev),
             mean2=(torch.randn(cin, generator=gen) * 0.1).to(dev), invstd2=(torch.rand(cin, generator=gen) + 0.5).to(dev),
             scale2=(torch.rand(cin, generator=gen) + 0.5).to(dev), shift2=(torch.randn(cin, generator=gen) * 0.3).to(dev))
    return t


def _three_launches(lib, check, ConvDesc, t, m):
    """bn3's sums + the un-fused apply / data gradient + bn2 sums / weight gradient through the C ABI."""
    cin, cout = 64, 256
    d = ConvDesc(1, m, 1, cin, cout, 1, 1, 1, 0, m, 1, cin, 1, 0, 0, 0, 0, 0)
    wp = torch.empty(cout, 1, 1, cin, dtype=torch.bfloat16, device="cuda")
    check(lib.mvf_pack_conv_weight(P(t["w"]), cout, cin, 1, 1, 1, cin, None, P(wp), 1, None))
    wd = torch.empty(cin, 1, 1, cout, dtype=torch.bfloat16, device="cuda")
    check(lib.mvf_pack_conv_weight_dgrad(P(t["w"]), cout, cin, 1, 1, P(wd), 1, None))
    ws = torch.zeros(max(lib.mvf_conv2d_workspace_bytes(C.byref(d)), 1), dtype=torch.uint8, device="cuda")
    rows = lib.mvf_conv2d_stats_rows(C.byref(d))
    sp = torch.empty(cout, rows, 2, device="cuda")
    check(lib.mvf_conv2d_nhwc_fwd_bnbwd_sums(C.byref(d), P(t["a2"]), None, P(wp), P(t["g"]), P(t["bits"]), P(t["mean"]), P(t["invstd"]), P(sp), P(ws), ws.numel(), None))
    dg, db = torch.empty(cout, device="cuda"), torch.empty(cout, device="cuda")
    check(lib.mvf_bn_bwd_finalize(P(sp), rows, cout, P(dg), P(db), None))
    dz3 = torch.empty(m, cout, dtype=torch.bfloat16, device="cuda")
    check(lib.mvf_conv2d_nhwc_fwd_bnbwd_apply(C.byref(d), P(t["a2"]), None, P(wp), P(t["g"]), P(t["bits"]), P(t["gamma"]), P(t["mean"]), P(t["invstd"]), P(dg), P(db),
                                              P(dz3), P(ws), ws.numel(), None))
    dd = ConvDesc(1, m, 1, cout, cin, 1, 1, 1, 0, m, 1, cout, 1, 0, 0, 0, 0, 0)
    ws2 = torch.zeros(max(lib.mvf_conv2d_workspace_bytes(C.byref(dd)), 1), dtype=torch.uint8, device="cuda")
    rows2 = lib.mvf_conv2d_stats_rows(C.byref(dd))
    part = torch.zeros(rows2, cin, 2, device="cuda")
    dx = torch.empty(m, cin, dtype=torch.bfloat16, device="cuda")
    check(lib.mvf_conv2d_nhwc_dgrad_bnsums(C.byref(dd), P(dz3), P(wd), P(dx), P(t["z2"]), P(t["mean2"]), P(t["invstd2"]), P(t["scale2"]), P(t["shift2"]), P(part),
                                           P(ws2), ws2.numel(), None))
    dg2, db2 = torch.empty(cin, device="cuda"), torch.empty(cin, device="cuda")
    check(lib.mvf_bn_bwd_finalize(P(part), rows2, cin, P(dg2), P(db2), None))
    torch.cuda.synchronize()
    return dict(wp=wp, dg=dg, db=db, dz3=dz3, dx=dx, dg2=dg2, db2=db2)


@pytest.mark.parametrize("m", [64 * 40, 300, 64 * 57 + 17, 3 * 56 * 56, 130, 256 * 56 * 56], ids=str)          # (the last: layer1 of the C3 step at full size)
def test_conv1x1_bwd_fused_equals_apply_dgrad_wgrad(m):
    from mvfnet_amd import _lib
    lib, check, ConvDesc = _lib.lib, _lib.check, _lib.ConvDesc
    cin, cout = 64, 256
    t = _inputs(m, seed=m)
    ref = _three_launches(lib, check, ConvDesc, t, m)
    ns = lib.mvf_conv1x1_bwd_fused_splits(m, cout, cin)
    assert ns > 0
    dx = torch.full((m, cin), 7.0, dtype=torch.bfloat16, device="cuda")
    spart = torch.full((cin, 2 * ns, 2), float("nan"), device="cuda")
    slabs = torch.full((ns * cout * cin,), float("nan"), device="cuda")
    check(lib.mvf_conv1x1_bwd_fused(P(t["a2"]), cin, P(ref["wp"]), P(t["g"]), cout, P(t["bits"]), m, cout, cin, P(t["gamma"]), P(t["mean"]), P(t["invstd"]),
                                    P(ref["dg"]), P(ref["db"]), P(t["z2"]), P(t["mean2"]), P(t["invstd2"]), P(t["scale2"]), P(t["shift2"]), P(dx), P(spart),
                                    2 * ns, P(slabs), slabs.numel() * 4, 1, None), "conv1x1_bwd_fused")
    dg2, db2 = torch.empty(cin, device="cuda"), torch.empty(cin, device="cuda")
    check(lib.mvf_bn_bwd_finalize(P(spart), 2 * ns, cin, P(dg2), P(db2), None))
    dw = torch.full((cout, cin, 1, 1), float("nan"), device="cuda")
    check(lib.mvf_wgrad_slab_reduce(P(slabs), ns, cout, cin, P(dw), None))
    torch.cuda.synchronize()
    assert torch.isfinite(dx.float()).all() and torch.isfinite(spart).all() and torch.isfinite(dw).all()
    # the data gradient: the same dz3 (bit-identical arithmetic), contracted in ONE chain over the 256 channels in ascending order like the un-fused
    # kernel's K loop -> bit for bit
    a, b = dx.float(), ref["dx"].float()
    flips = int((dx.view(torch.int16) != ref["dx"].view(torch.int16)).sum())
    assert flips == 0, (flips, rel_err(a.cpu().numpy(), b.cpu().numpy()))
    # bn2's sums over what was stored
    assert rel_err(dg2.cpu().numpy(), ref["dg2"].cpu().numpy()) < 2e-5 and rel_err(db2.cpu().numpy(), ref["db2"].cpu().numpy()) < 2e-5
    gq = torch.where(t["z2"].float() * t["scale2"] + t["shift2"] > 0, a, torch.zeros_like(a)).double()
    want_db = gq.sum(0)
    want_dg = (gq * ((t["z2"].double() - t["mean2"].double()) * t["invstd2"].double())).sum(0)
    assert rel_err(db2.cpu().numpy(), want_db.float().cpu().numpy()) < 2e-5 and rel_err(dg2.cpu().numpy(), want_dg.float().cpu().numpy()) < 2e-5
    # the weight gradient of the un-fused dz3
    want_dw = ref["dz3"].float().t() @ t["a2"].float()
    assert rel_err(dw.view(cout, cin).cpu().numpy(), want_dw.cpu().numpy()) < 2e-5


def test_conv1x1_bwd_fused_against_fp32_restatement():
    """The same pass against plain torch fp32 with the rounding points of the path (z3, dz3 and da2 rounded to bf16)."""
    from mvfnet_amd import _lib
    lib, check, ConvDesc = _lib.lib, _lib.check, _lib.ConvDesc
    m, cin, cout = 2 * 28 * 28, 64, 256
    t = _inputs(m, seed=5)
    ref = _three_launches(lib, check, ConvDesc, t, m)
    bf = torch.bfloat16
    wq = t["w"].view(cout, cin).to(bf).float()
    z3 = (t["a2"].float() @ wq.t()).to(bf).float()
    mask = torch.stack([(t["bits"] >> j) & 1 for j in range(4)], dim=-1).reshape(m, cout).float()
    gm = t["g"].float() * mask
    ca, cd, ck = t["gamma"] * t["invstd"], ref["db"] / m, t["invstd"] * ref["dg"] / m
    dz3 = (ca * (gm - cd - (z3 - t["mean"]) * ck)).to(bf).float()
    want_dx = (dz3 @ wq).to(bf).float()
    ns = lib.mvf_conv1x1_bwd_fused_splits(m, cout, cin)
    dx = torch.empty(m, cin, dtype=bf, device="cuda")
    spart = torch.empty(cin, 2 * ns, 2, device="cuda")
    slabs = torch.empty(ns * cout * cin, device="cuda")
    check(lib.mvf_conv1x1_bwd_fused(P(t["a2"]), cin, P(ref["wp"]), P(t["g"]), cout, P(t["bits"]), m, cout, cin, P(t["gamma"]), P(t["mean"]), P(t["invstd"]),
                                    P(ref["dg"]), P(ref["db"]), P(t["z2"]), P(t["mean2"]), P(t["invstd2"]), P(t["scale2"]), P(t["shift2"]), P(dx), P(spart),
                                    2 * ns, P(slabs), slabs.numel() * 4, 1, None), "conv1x1_bwd_fused")
    dw = torch.empty(cout, cin, 1, 1, device="cuda")
    check(lib.mvf_wgrad_slab_reduce(P(slabs), ns, cout, cin, P(dw), None))
    torch.cuda.synchronize()
    assert rel_err(dx.float().cpu().numpy(), want_dx.cpu().numpy()) < 4e-3          # one bf16 ulp where a rounding of z3 / dz3 fell the other way
    assert rel_err(dw.view(cout, cin).cpu().numpy(), (dz3.t() @ t["a2"].float()).cpu().numpy()) < 2e-3
    # shapes that are not built say so, and fp32 storage is refused
    assert lib.mvf_conv1x1_bwd_fused_splits(m, 512, 128) == 0 and lib.mvf_conv1x1_bwd_fused_splits(m, 256, 128) == 0
    assert lib.mvf_conv1x1_bwd_fused(P(t["a2"]), cin, P(ref["wp"]), P(t["g"]), cout, P(t["bits"]), m, cout, cin, P(t["gamma"]), P(t["mean"]), P(t["invstd"]),
                                     P(ref["dg"]), P(ref["db"]), P(t["z2"]), P(t["mean2"]), P(t["invstd2"]), P(t["scale2"]), P(t["shift2"]), P(dx), P(spart),
                                     2 * ns, P(slabs), slabs.numel() * 4, 0, None) == -5


@pytest.mark.parametrize("m", [64 * 33 + 5, 2 * 56 * 56], ids=str)
def test_conv1x1_bwd_fused_without_input_batchnorm(m):
    """z_in = NULL (a downsample branch reading the block input, reference resnet.py:227-228): the same data gradient and weight gradient bit for bit, no
    sums taken (the partial-row buffer is not touched)."""
    from mvfnet_amd import _lib
    lib, check, ConvDesc = _lib.lib, _lib.check, _lib.ConvDesc
    cin, cout = 64, 256
    t = _inputs(m, seed=m + 1)
    ref = _three_launches(lib, check, ConvDesc, t, m)
    ns = lib.mvf_conv1x1_bwd_fused_splits(m, cout, cin)
    out = {}
    for with_bn in (True, False):
        dx = torch.full((m, cin), 7.0, dtype=torch.bfloat16, device="cuda")
        spart = torch.full((cin, 2 * ns, 2), 3.0, device="cuda")
        slabs = torch.full((ns * cout * cin,), float("nan"), device="cuda")
        check(lib.mvf_conv1x1_bwd_fused(P(t["a2"]), cin, P(ref["wp"]), P(t["g"]), cout, P(t["bits"]), m, cout, cin, P(t["gamma"]), P(t["mean"]), P(t["invstd"]),
                                        P(ref["dg"]), P(ref["db"]), P(t["z2"]) if with_bn else None, P(t["mean2"]) if with_bn else None,
                                        P(t["invstd2"]) if with_bn else None, P(t["scale2"]) if with_bn else None, P(t["shift2"]) if with_bn else None,
                                        P(dx), P(spart) if with_bn else None, 2 * ns if with_bn else 0, P(slabs), slabs.numel() * 4, 1, None), "conv1x1_bwd_fused")
        dw = torch.empty(cout, cin, 1, 1, device="cuda")
        check(lib.mvf_wgrad_slab_reduce(P(slabs), ns, cout, cin, P(dw), None))
        torch.cuda.synchronize()
        out[with_bn] = (dx, dw, spart)
    assert torch.equal(out[True][0].view(torch.int16), out[False][0].view(torch.int16)) and torch.equal(out[True][0].view(torch.int16), ref["dx"].view(torch.int16))
    assert torch.equal(out[True][1], out[False][1])
    assert bool((out[False][2] == 3.0).all())


@pytest.mark.parametrize("m", [64 * 33 + 5, 300, 256 * 56 * 56], ids=str)
def test_bnbwd_sums_pair_equals_two_sums_passes(m):
    """mvf_conv1x1_bnbwd_sums_pair (csrc/pw_sums_pair.hip): dgamma / dbeta of bn3 and bn_d against two mvf_conv2d_nhwc_fwd_bnbwd_sums passes and against fp64 sums
    over the recomputed, bf16-rounded conv outputs."""
    from mvfnet_amd import _lib
    lib, check, ConvDesc = _lib.lib, _lib.check, _lib.ConvDesc
    cin, cout = 64, 256
    t = _inputs(m, seed=m + 2)
    gen = torch.Generator().manual_seed(m)
    xb = torch.randn(m, cin, generator=gen).to("cuda", torch.bfloat16)
    wb = (torch.randn(cout, cin, 1, 1, generator=gen) * (2.0 / cin) ** 0.5).cuda()
    mean_b, invstd_b = (torch.randn(cout, generator=gen) * 0.1).cuda(), (torch.rand(cout, generator=gen) + 0.5).cuda()
    wpa, wpb = torch.empty(cout, 1, 1, cin, dtype=torch.bfloat16, device="cuda"), torch.empty(cout, 1, 1, cin, dtype=torch.bfloat16, device="cuda")
    check(lib.mvf_pack_conv_weight(P(t["w"]), cout, cin, 1, 1, 1, cin, None, P(wpa), 1, None))
    check(lib.mvf_pack_conv_weight(P(wb), cout, cin, 1, 1, 1, cin, None, P(wpb), 1, None))
    d = ConvDesc(1, m, 1, cin, cout, 1, 1, 1, 0, m, 1, cin, 1, 0, 0, 0, 0, 0)
    ws = torch.zeros(max(lib.mvf_conv2d_workspace_bytes(C.byref(d)), 1), dtype=torch.uint8, device="cuda")
    rows = lib.mvf_conv2d_stats_rows(C.byref(d))
    want = []
    for inp, wp, mu, rs in ((t["a2"], wpa, t["mean"], t["invstd"]), (xb, wpb, mean_b, invstd_b)):
        sp = torch.empty(cout, rows, 2, device="cuda")
        check(lib.mvf_conv2d_nhwc_fwd_bnbwd_sums(C.byref(d), P(inp), None, P(wp), P(t["g"]), P(t["bits"]), P(mu), P(rs), P(sp), P(ws), ws.numel(), None))
        dg, db = torch.empty(cout, device="cuda"), torch.empty(cout, device="cuda")
        check(lib.mvf_bn_bwd_finalize(P(sp), rows, cout, P(dg), P(db), None))
        want.append((dg, db))
    ns = lib.mvf_conv1x1_bwd_fused_splits(m, cout, cin)
    pa, pb = torch.full((cout, 2 * ns, 2), float("nan"), device="cuda"), torch.full((cout, 2 * ns, 2), float("nan"), device="cuda")
    check(lib.mvf_conv1x1_bnbwd_sums_pair(P(t["a2"]), cin, P(wpa), P(xb), cin, P(wpb), P(t["g"]), cout, P(t["bits"]), m, cout, cin, P(t["mean"]), P(t["invstd"]),
                                          P(mean_b), P(invstd_b), P(pa), P(pb), 2 * ns, 1, None), "sums pair")
    p1 = torch.full((cout, 2 * ns, 2), float("nan"), device="cuda")          # the one-branch form (x_in = NULL): bn3's sums alone, bit for bit the pair's
    check(lib.mvf_conv1x1_bnbwd_sums_pair(P(t["a2"]), cin, P(wpa), None, 0, None, P(t["g"]), cout, P(t["bits"]), m, cout, cin, P(t["mean"]), P(t["invstd"]),
                                          None, None, P(p1), None, 2 * ns, 1, None), "sums, one branch")
    got = []
    for part in (pa, pb):
        dg, db = torch.empty(cout, device="cuda"), torch.empty(cout, device="cuda")
        check(lib.mvf_bn_bwd_finalize(P(part), 2 * ns, cout, P(dg), P(db), None))
        got.append((dg, db))
    torch.cuda.synchronize()
    assert torch.equal(p1, pa)
    mask = torch.stack([(t["bits"] >> j) & 1 for j in range(4)], dim=-1).reshape(m, cout).double()
    gm = t["g"].double() * mask
    for (dg, db), (wdg, wdb), (inp, w, mu, rs) in zip(got, want, ((t["a2"], t["w"], t["mean"], t["invstd"]), (xb, wb, mean_b, invstd_b))):
        assert torch.isfinite(dg).all() and torch.isfinite(db).all()
        assert rel_err(dg.cpu().numpy(), wdg.cpu().numpy()) < 3e-6 and rel_err(db.cpu().numpy(), wdb.cpu().numpy()) < 3e-6
        z = (inp.float() @ w.view(cout, cin).to(torch.bfloat16).float().t()).to(torch.bfloat16).double()
        ref_db = gm.sum(0)
        ref_dg = (gm * ((z - mu.double()) * rs.double())).sum(0)
        assert rel_err(db.cpu().numpy(), ref_db.float().cpu().numpy()) < 3e-6 and rel_err(dg.cpu().numpy(), ref_dg.float().cpu().numpy()) < 2e-5
