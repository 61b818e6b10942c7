"""GPU parity of the implicit-GEMM conv + fused epilogue and the small network ops against the CPU oracle
(torch.nn.functional on CPU -- the same ATen arithmetic the reference calls)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import policy_env, policy_set, policy_str, rel_err

pytestmark = pytest.mark.gpu

# (n, h, w, cin, cout, k, stride, pad)
CONV_CASES = [
    (2, 14, 14, 64, 64, 1, 1, 0),       # narrow tile (Cout 64)
    (3, 9, 7, 128, 256, 1, 1, 0),       # M tail (189 rows)
    (2, 14, 14, 256, 512, 1, 2, 0),     # downsample 1x1 stride 2
    (2, 12, 12, 64, 64, 3, 1, 1),       # 3x3
    (2, 13, 11, 128, 128, 3, 2, 1),     # 3x3 stride 2, odd sizes
    (1, 7, 7, 512, 2048, 1, 1, 0),      # many n tiles
    (2, 8, 8, 48, 96, 3, 1, 1),         # cin not a multiple of the chunk, cout not a multiple of the tile
    (4, 7, 7, 2048, 512, 1, 1, 0),      # long K
    (150, 1, 1, 64, 64, 1, 1, 0),       # 1x1 maps: a 128-row tile spans 128 images (row -> image division by 1)
    (3, 2, 33, 32, 96, 3, 1, 1),        # two-row maps, Wo not a power of two, every tap masked somewhere
    (2, 5, 5, 64, 128, 5, 1, 2),        # 5x5 kernel (25 taps in the kh / kw bit masks)
    (1, 40, 3, 64, 64, 3, 2, 1),        # tall thin map, stride 2
    # [r3] shapes the 256 x 256 tiles take when forced (Cout % 256 == 0, whole 64-channel K chunks): 3x3 with 18 / 9 (odd) / 27 chunks,
    # ragged M, stride 2, and a single-chunk K (the four-phase loop's prologue alone stages more than that)
    (2, 14, 14, 128, 256, 3, 1, 1),
    (3, 9, 11, 64, 256, 3, 2, 1),
    (1, 14, 14, 192, 256, 3, 1, 1),
    (2, 7, 7, 64, 512, 3, 1, 1),
    (5, 6, 6, 64, 256, 1, 1, 0),
    (2, 16, 16, 128, 256, 1, 1, 0),
]


def _run_conv(x_nhwc, w_oihw, scale, bias, residual, relu, stride, pad, dtype, x2=None, split_c=0):
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    dt = _lib.MVF_F32 if dtype == torch.float32 else _lib.MVF_BF16
    n, h, w, cin = x_nhwc.shape
    cout, _, kh, kw = w_oihw.shape
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    wp = torch.empty(cout, kh, kw, cin, dtype=dtype, device="cuda")
    check(lib.mvf_pack_conv_weight(p(w_oihw), cout, cin, kh, kw, kw, cin, p(scale), p(wp), dt, None))
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    y = torch.empty(n, ho, wo, cout, dtype=dtype, device="cuda")
    d = _lib.ConvDesc(n, h, w, cin, cout, kh, kw, stride, pad, ho, wo, cin, dt, int(relu), split_c, split_c)
    check(lib.mvf_conv2d_nhwc_fwd(C.byref(d), p(x_nhwc), p(x2), p(wp), p(bias), p(residual), p(y), None))
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "n%d_%dx%d_c%d_o%d_k%d_s%d" % c[:7])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("epi", ["plain", "bias_relu", "bias_res_relu"])
def test_conv_igemm_vs_oracle(case, dtype, epi):
    n, h, w, cin, cout, k, stride, pad = case
    if dtype == torch.bfloat16 and cin % 8:
        pytest.skip("bf16 needs cin % 8 == 0")
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(n, cin, h, w, generator=g)
    wgt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g) * 0.2 if epi != "plain" else None
    ref = F.conv2d(x, wgt * scale.view(-1, 1, 1, 1), bias, stride=stride, padding=pad)
    res = torch.randn(ref.shape, generator=g) if epi == "bias_res_relu" else None
    if res is not None:
        ref = ref + res
    if epi != "plain":
        ref = F.relu(ref)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda().to(dtype)
    rg = res.permute(0, 2, 3, 1).contiguous().cuda().to(dtype) if res is not None else None
    y = _run_conv(xg, wgt.cuda(), scale.cuda(), bias.cuda() if bias is not None else None, rg, epi != "plain", stride, pad, dtype)
    got = y.float().cpu().permute(0, 3, 1, 2).numpy()
    tol = 2e-5 if dtype == torch.float32 else 1e-2          # north_star: 1e-3 fp32 / 1e-2 bf16
    assert rel_err(got, ref.numpy()) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 7, 7, 256, 128, 64), (3, 9, 7, 512, 256, 128), (2, 6, 6, 256, 256, 64)], ids=str)
def test_conv_split_a_operand(dtype, shape):
    """1x1 conv reading channels [0,Cs) from the compact MVF slice buffer and the rest from x ([r3] also at Cout = 256: the 256 x 256 tiles)."""
    g = torch.Generator().manual_seed(7)
    n, h, w, cin, cout, cs = shape
    x = torch.randn(n, h, w, cin, generator=g)
    sl = torch.randn(n, h, w, cs, generator=g)
    wgt = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    xin = x.clone()
    xin[..., :cs] = sl
    ref = F.conv2d(xin.permute(0, 3, 1, 2), wgt)
    y = _run_conv(x.cuda().to(dtype), wgt.cuda(), None, None, None, False, 1, 0, dtype, x2=sl.cuda().to(dtype), split_c=cs)
    assert rel_err(y.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < (2e-5 if dtype == torch.float32 else 1e-2)


def test_conv_full_size_linearity_and_oracle_rows():
    """BASELINE C2 size (layer3 MVF.net: 256 images x 14x14, 1024 -> 256): full-tensor properties + oracle on a slice."""
    torch.manual_seed(3)
    n, h, w, cin, cout = 256, 14, 14, 1024, 256
    x1 = torch.randn(n, h, w, cin, device="cuda")
    x2 = torch.randn(n, h, w, cin, device="cuda")
    wgt = torch.randn(cout, cin, 1, 1, device="cuda") * (2.0 / cin) ** 0.5
    y1 = _run_conv(x1, wgt, None, None, None, False, 1, 0, torch.float32)
    y2 = _run_conv(x2, wgt, None, None, None, False, 1, 0, torch.float32)
    y12 = _run_conv(x1 + 2 * x2, wgt, None, None, None, False, 1, 0, torch.float32)
    lin = (y1 + 2 * y2 - y12).abs().max() / y12.abs().max()
    assert float(lin) < 1e-5                                      # linearity in the input
    ref = F.conv2d(x1[:2].cpu().permute(0, 3, 1, 2), wgt.cpu())    # oracle on 2 images
    assert rel_err(y1[:2].cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < 2e-5
    ref = F.conv2d(x1[-1:].cpu().permute(0, 3, 1, 2), wgt.cpu())   # ... and on the last image (tail tiles)
    assert rel_err(y1[-1:].cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < 2e-5


def test_stem_maxpool_head_ops_vs_oracle():
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    g = torch.Generator().manual_seed(11)
    # max-pool (odd and even sizes)
    for (n, h, w, c) in [(2, 9, 11, 8), (1, 112, 112, 64)]:
        x = torch.randn(n, c, h, w, generator=g)
        ref = F.max_pool2d(x, 3, 2, 1)
        y = torch.empty(n, ref.shape[2], ref.shape[3], c, device="cuda")
        xg = x.permute(0, 2, 3, 1).contiguous().cuda()           # keep alive: p() only takes the address
        check(lib.mvf_maxpool3x3s2_nhwc(p(xg), n, h, w, c, p(y), 0, None))
        assert torch.equal(y.cpu().permute(0, 3, 1, 2), ref)
    # head: avgpool + fc + segment mean, then clip averaging
    clips, T, hw, c, classes = 6, 4, 9, 64, 10
    feat = torch.randn(clips * T, c, 3, 3, generator=g)
    fw, fb = torch.randn(classes, c, generator=g) * 0.1, torch.randn(classes, generator=g)
    ref = F.linear(F.adaptive_avg_pool2d(feat, 1).flatten(1), fw, fb).reshape(clips, T, classes).mean(1)
    pooled = torch.empty(clips, c, device="cuda")
    out = torch.empty(clips, classes, device="cuda")
    fg, fwg, fbg = feat.permute(0, 2, 3, 1).contiguous().cuda(), fw.cuda(), fb.cuda()
    check(lib.mvf_head_pool_fc(p(fg), clips, T, hw, c, p(fwg), p(fbg), classes, p(pooled), p(out), 0, None))
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-5
    for kind, fn in ((1, lambda s: s.mean(0, keepdim=True)), (2, lambda s: F.softmax(s, 1).mean(0, keepdim=True))):
        avg = torch.empty(1, classes, device="cuda")
        check(lib.mvf_average_clip(p(out), clips, classes, kind, p(avg), None))
        assert rel_err(avg.cpu().numpy(), fn(ref).numpy()) < 1e-5


@pytest.mark.parametrize("shape", [(3, 56, 56, 64), (5, 56, 56, 128), (2, 64, 64, 64), (3, 16, 16, 64), (4, 8, 8, 64)], ids=lambda s: "n%d_%dx%d_pitch%d" % s)
def test_conv3x3_c64_direct_vs_oracle_and_the_implicit_gemm(shape):
    """[r3] layer1's 3x3 (resnet.py:213-224 conv2 at planes = 64) and its data gradient run on a direct kernel (csrc/conv3x3_c64.hip: padded
    window staged once per row band, the wave's weights in registers); MVF_POLICY=conv3x3_direct=0 sends the same calls to the implicit-GEMM kernel.
    Every epilogue (statistics, plain, bias + ReLU, data gradient + BatchNorm-backward sums) against torch on the bf16-rounded operands and
    against the other kernel; a pixel pitch wider than the 64 channels read (the input as a slice of a wider tensor)."""
    import os
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    n, h, w, pitch = shape
    g = torch.Generator().manual_seed(h + n)
    xw = torch.randn(n, h, w, pitch, generator=g).bfloat16().cuda()
    x = xw[..., :64].float().cpu().permute(0, 3, 1, 2)
    wt = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    bias = (torch.randn(64, generator=g) * 0.2).cuda()
    shift = (torch.randn(64, generator=g) * 0.1).cuda()
    wpk = torch.empty(64, 3, 3, 64, device="cuda", dtype=torch.bfloat16)
    wg = wt.cuda()
    check(lib.mvf_pack_conv_weight(P(wg), 64, 64, 3, 3, 3, 64, None, P(wpk), 1, None))
    wdg = torch.empty(64, 3, 3, 64, device="cuda", dtype=torch.bfloat16)
    check(lib.mvf_pack_conv_weight_dgrad(P(wg), 64, 64, 3, 3, P(wdg), 1, None))
    wb = wt.bfloat16().float()
    ref = F.conv2d(x, wb, padding=1)
    ref_d = F.conv_transpose2d(x, wb, padding=1)                       # the data gradient of conv(., wt) applied to `x` as the incoming gradient
    zb = torch.randn(n * h * w, 64, generator=g).bfloat16().cuda()     # the pre-activation whose ReLU gate / xhat the sums use
    mu, rs = (torch.randn(64, generator=g) * 0.1).cuda(), (torch.rand(64, generator=g) + 0.5).cuda()
    sc, sh = (torch.randn(64, generator=g)).cuda(), (torch.randn(64, generator=g) * 0.3).cuda()
    m = n * h * w

    def run(direct, **more):
        with policy_set(conv3x3_direct=1 if direct else 0, **more):
            d = _lib.ConvDesc(n, h, w, 64, 64, 3, 3, 1, 1, h, w, pitch, 1, 0, 0, 0, 0, 0)
            rows = lib.mvf_conv2d_stats_rows(C.byref(d))
            new = lambda: torch.full((m, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
            z, y2, y4, dx = new(), new(), new(), new()
            part = torch.full((64, rows, 2), float("nan"), device="cuda")
            sums = torch.full((64, rows, 2), float("nan"), device="cuda")
            check(lib.mvf_conv2d_nhwc_fwd_stats(C.byref(d), P(xw), None, P(wpk), P(z), P(part), P(shift), None, 0, None))
            check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), P(xw), None, P(wpk), None, None, P(y2), None, 0, None))
            check(lib.mvf_conv2d_nhwc_dgrad_bnsums(C.byref(d), P(xw), P(wdg), P(dx), P(zb), P(mu), P(rs), P(sc), P(sh), P(sums), None, 0, None))
            d.relu = 1
            check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), P(xw), None, P(wpk), P(bias), None, P(y4), None, 0, None))
            torch.cuda.synchronize()
            return z, part.double().sum(1), y2, y4, dx, sums.double().sum(1)

    r1, r0 = run(True), run(False)
    # workgroups whose band ranges straddle frames (5 bands each; a frame has h / 4 or h / 8): same outputs bit for bit, sums to summation order
    r5 = run(True, conv3x3_bpw=5)
    for a_, b_ in zip((r5[0], r5[2], r5[3], r5[4]), (r1[0], r1[2], r1[3], r1[4])):
        assert torch.equal(a_, b_)
    assert rel_err(r5[1].cpu().numpy(), r1[1].cpu().numpy()) < 1e-5 and rel_err(r5[5].cpu().numpy(), r1[5].cpu().numpy()) < 1e-5
    nchw = lambda t: t.float().cpu().reshape(n, h, w, 64).permute(0, 3, 1, 2).numpy()
    for z, st, y2, y4, dx, bs in (r1, r0):
        assert rel_err(nchw(z), ref.numpy()) < 6e-3 and rel_err(nchw(y2), ref.numpy()) < 6e-3
        assert rel_err(nchw(y4), F.relu(ref + bias.cpu().view(1, -1, 1, 1)).numpy()) < 6e-3
        assert rel_err(nchw(dx), ref_d.numpy()) < 6e-3
        dz = z.double() - shift.double()                              # statistics of the STORED values, minus the shift
        assert rel_err(st[:, 0].cpu().numpy(), dz.sum(0).cpu().numpy()) < 1e-4
        assert rel_err(st[:, 1].cpu().numpy(), (dz * dz).sum(0).cpu().numpy()) < 1e-5
        gm = dx.double() * ((zb.float() * sc + sh) > 0)               # the gate in fp32, as the kernels (and bn_apply) evaluate it
        xhat = ((zb.float() - mu) * rs).double()
        assert rel_err(bs[:, 0].cpu().numpy(), gm.sum(0).cpu().numpy()) < 1e-4
        assert rel_err(bs[:, 1].cpu().numpy(), (gm * xhat).sum(0).cpu().numpy()) < 1e-4
    for a_, b_ in zip((r1[0], r1[2], r1[3], r1[4]), (r0[0], r0[2], r0[3], r0[4])):      # same products, same k order
        assert (a_ != b_).float().mean().item() < 1e-3 and rel_err(a_.float().cpu().numpy(), b_.float().cpu().numpy()) < 1e-3


def test_conv3x3_c64_direct_full_c3_size_oracle_rows():
    """[r4] The direct 3x3 kernel at the FULL C3 launch (256 frames x 56 x 56 x 64: `conv3x3_c64_kernel<.., 56, 4>` with every workgroup walking its
    14 bands of many frames) against the oracle on whole frames -- the first two, one in the middle (a workgroup's frame boundary) and the last (tail bands) --
    for the forward + statistics epilogue and for the data gradient + BatchNorm-backward sums; statistics against the stored output."""
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    n, h, w = 256, 56, 56
    g = torch.Generator(device="cuda").manual_seed(5)
    xw = torch.randn(n, h, w, 64, device="cuda", generator=g).bfloat16()
    wt = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    shift = torch.randn(64, device="cuda", generator=g) * 0.1
    wpk = torch.empty(64, 3, 3, 64, device="cuda", dtype=torch.bfloat16)
    wdg = torch.empty(64, 3, 3, 64, device="cuda", dtype=torch.bfloat16)
    check(lib.mvf_pack_conv_weight(P(wt), 64, 64, 3, 3, 3, 64, None, P(wpk), 1, None))
    check(lib.mvf_pack_conv_weight_dgrad(P(wt), 64, 64, 3, 3, P(wdg), 1, None))
    m = n * h * w
    zb = torch.randn(m, 64, device="cuda", generator=g).bfloat16()
    mu, rs = torch.randn(64, device="cuda", generator=g) * 0.1, torch.rand(64, device="cuda", generator=g) + 0.5
    sc, sh = torch.randn(64, device="cuda", generator=g), torch.randn(64, device="cuda", generator=g) * 0.3
    d = _lib.ConvDesc(n, h, w, 64, 64, 3, 3, 1, 1, h, w, 64, 1, 0, 0, 0, 0, 0)
    rows = lib.mvf_conv2d_stats_rows(C.byref(d))
    z = torch.full((m, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
    dx = torch.full((m, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
    part = torch.full((64, rows, 2), float("nan"), device="cuda")
    sums = torch.full((64, rows, 2), float("nan"), device="cuda")
    check(lib.mvf_conv2d_nhwc_fwd_stats(C.byref(d), P(xw), None, P(wpk), P(z), P(part), P(shift), None, 0, None))
    check(lib.mvf_conv2d_nhwc_dgrad_bnsums(C.byref(d), P(xw), P(wdg), P(dx), P(zb), P(mu), P(rs), P(sc), P(sh), P(sums), None, 0, None))
    torch.cuda.synchronize()
    assert torch.isfinite(z.float()).all() and torch.isfinite(dx.float()).all() and torch.isfinite(part).all() and torch.isfinite(sums).all()
    wb = wt.bfloat16().float().cpu()
    for f in (0, 1, 127, 128, 255):
        xf = xw[f].float().cpu().permute(2, 0, 1)[None]
        ref = F.conv2d(xf, wb, padding=1)[0].permute(1, 2, 0).numpy()
        ref_d = F.conv_transpose2d(xf, wb, padding=1)[0].permute(1, 2, 0).numpy()
        assert rel_err(z.view(n, h, w, 64)[f].float().cpu().numpy(), ref) < 6e-3, f
        assert rel_err(dx.view(n, h, w, 64)[f].float().cpu().numpy(), ref_d) < 6e-3, f
    dz = z.double() - shift.double()
    st = part.double().sum(1)
    assert rel_err(st[:, 0].cpu().numpy(), dz.sum(0).cpu().numpy()) < 1e-4 and rel_err(st[:, 1].cpu().numpy(), (dz * dz).sum(0).cpu().numpy()) < 1e-5
    gm = dx.double() * ((zb.float() * sc + sh) > 0)
    bs = sums.double().sum(1)
    assert rel_err(bs[:, 0].cpu().numpy(), gm.sum(0).cpu().numpy()) < 1e-4
    assert rel_err(bs[:, 1].cpu().numpy(), (gm * ((zb.float() - mu) * rs).double()).sum(0).cpu().numpy()) < 1e-4


@pytest.mark.parametrize("shape", [(3, 64, 64), (2, 32, 32), (1, 48, 80), (2, 224, 224)], ids=lambda s: "n%d_%dx%d" % s)
def test_stem_direct_conv_vs_oracle_and_the_implicit_gemm(shape):
    """[r3] The bf16 stem (resnet.py:420-431 conv1) runs on its own direct kernel (csrc/stem_direct.hip: input patch staged once, weights in
    registers); MVF_POLICY=stem_direct=0 sends the same call to the implicit-GEMM kernel.  Both against F.conv2d on the bf16-rounded operands; the
    two kernels against each other (same products, same k order: outputs within one bf16 rounding of the fp32 sum, statistics of the stored
    values to summation order).  (1, 48, 80) has no whole statistic rows per row band: the training epilogue falls back, the inference one does not."""
    import os
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    n, h, w = shape
    g = torch.Generator().manual_seed(h + w)
    x = torch.randn(n, 3, h, w, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    bias = (torch.randn(64, generator=g) * 0.2).cuda()
    shift = (torch.randn(64, generator=g) * 0.1).cuda()
    hp, wp = h + 6, (w + 6 + 2 + 1) // 2 * 2
    ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    xg, wg = x.cuda(), wt.cuda()
    xp = torch.empty(n, hp, wp, 4, device="cuda", dtype=torch.bfloat16)
    check(lib.mvf_stem_prep(P(xg), n, 3, h, w, 3, wp, P(xp), 1, None))
    wpk = torch.empty(64, 7, 8, 4, device="cuda", dtype=torch.bfloat16)
    check(lib.mvf_pack_conv_weight(P(wg), 64, 3, 7, 7, 8, 4, None, P(wpk), 1, None))
    ref = F.conv2d(x.bfloat16().float(), wt.bfloat16().float(), stride=2, padding=3)          # fp32 sums of the bf16 products
    m = n * ho * wo

    def run(direct, **more):
        with policy_set(stem_direct=1 if direct else 0, **more):
            d = _lib.ConvDesc(n, hp, wp, 32, 64, 7, 1, 2, 0, ho, wo, 4, 1, 0, 0, 0, 0, 0)
            rows = lib.mvf_conv2d_stats_rows(C.byref(d))
            z = torch.full((m, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
            part = torch.full((64, rows, 2), float("nan"), device="cuda")
            check(lib.mvf_conv2d_nhwc_fwd_stats(C.byref(d), P(xp), None, P(wpk), P(z), P(part), P(shift), None, 0, None))
            d.relu = 1
            y = torch.full((m, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
            check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), P(xp), None, P(wpk), P(bias), None, P(y), None, 0, None))
            torch.cuda.synchronize()
            return z, part.double().sum(1), y

    z1, st1, y1 = run(True)
    z0, st0, y0 = run(False)
    z3, st3, y3 = run(True, stem_tpw=3)          # workgroups that walk 3 tiles (ranges straddle frames): same outputs, sums to summation order
    assert torch.equal(z3, z1) and torch.equal(y3, y1) and rel_err(st3.cpu().numpy(), st1.cpu().numpy()) < 1e-5
    nchw = lambda t: t.float().cpu().reshape(n, ho, wo, 64).permute(0, 3, 1, 2).numpy()
    for z, st, y in ((z1, st1, y1), (z0, st0, y0)):
        assert rel_err(nchw(z), ref.numpy()) < 6e-3
        assert rel_err(nchw(y), F.relu(ref + bias.cpu().view(1, -1, 1, 1)).numpy()) < 6e-3
        dz = z.double() - shift.double()                              # the statistics are those of the STORED values, minus the shift
        assert rel_err(st[:, 0].cpu().numpy(), dz.sum(0).cpu().numpy()) < 1e-4
        assert rel_err(st[:, 1].cpu().numpy(), (dz * dz).sum(0).cpu().numpy()) < 1e-5
    # the two kernels: same k order -> identical except where the fp32 sums differ in their last bits right at a rounding boundary
    assert (z1 != z0).float().mean().item() < 1e-3 and rel_err(z1.float().cpu().numpy(), z0.float().cpu().numpy()) < 1e-3
    assert (y1 != y0).float().mean().item() < 1e-3


@pytest.mark.parametrize("case", [
    (256, 14, 14, 256, 256, 3, 1, 1),     # 784 tiles on 512 slots: 1 full wave + stream-K tail of 272 tiles
    (256, 7, 7, 512, 512, 3, 1, 1),       # 392 tiles: everything is tail
    (256, 14, 14, 1024, 256, 1, 1, 0),    # K = 32 chunks
    (200, 7, 7, 2048, 512, 1, 1, 0),      # ragged M (9800 rows, 77 tiles x 4), K = 64 chunks
    (256, 14, 14, 2048, 64, 1, 1, 0),     # narrow tile kernel, 392 tiles
], ids=lambda c: "n%d_%dx%d_c%d_o%d_k%d" % c[:6])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_conv_streamk_tail_matches_plain_launch_and_oracle(case, dtype):
    """The stream-K decomposition of the last tile wave (with workspace) must equal the one-tile-per-workgroup launch
    (without) up to fp32 summation order, for the fused bias+residual+ReLU epilogue, and match the oracle on a slice."""
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    n, h, w, cin, cout, k, stride, pad = case
    dt = _lib.MVF_F32 if dtype == torch.float32 else _lib.MVF_BF16
    gen = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(n, h, w, cin, device="cuda", generator=gen).to(dtype)
    wgt = torch.randn(cout, cin, k, k, device="cuda", generator=gen) * (2.0 / (cin * k * k)) ** 0.5
    bias = torch.randn(cout, device="cuda", generator=gen) * 0.1
    res = torch.randn(n, h, w, cout, device="cuda", generator=gen).to(dtype)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    wp = torch.empty(cout, k, k, cin, dtype=dtype, device="cuda")
    check(lib.mvf_pack_conv_weight(p(wgt), cout, cin, k, k, k, cin, None, p(wp), dt, None))
    d = _lib.ConvDesc(n, h, w, cin, cout, k, k, stride, pad, h, w, cin, dt, 1, 0, 0, 0)
    y0 = torch.empty(n, h, w, cout, dtype=dtype, device="cuda")
    check(lib.mvf_conv2d_nhwc_fwd(C.byref(d), p(x), None, p(wp), p(bias), p(res), p(y0), None))
    ws = torch.zeros(lib.mvf_conv2d_workspace_bytes(C.byref(d)), dtype=torch.uint8, device="cuda")   # zeroed: the error word is only written by a stream-K launch
    for rep in range(3):                                   # repeated launches reuse (and must re-zero) the flags
        y1 = torch.full_like(y0, float("nan"))
        check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), p(x), None, p(wp), p(bias), p(res), p(y1), p(ws), ws.numel(), None))
        torch.cuda.synchronize()
        err_flag = ws.view(torch.int32)[(ws.numel() - 4096) // 4 + 512].item()
        assert err_flag == 0, "stream-K spin gave up"
        assert torch.isfinite(y1.float()).all()
        assert rel_err(y1.float().cpu().numpy(), y0.float().cpu().numpy()) < (1e-5 if dtype == torch.float32 else 1e-2)   # fp32: summation order only
    sl = slice(n - 2, n)
    ref = F.relu(F.conv2d(x[sl].float().cpu().permute(0, 3, 1, 2), (wp.float().cpu().permute(0, 3, 1, 2)), bias.cpu(), stride=stride, padding=pad)
                 + res[sl].float().cpu().permute(0, 3, 1, 2))
    assert rel_err(y1[sl].float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) < (2e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("env", ["conv_glds=0", "conv_glds=1,conv_glds_nb=1", "conv_glds=1,conv_glds_nb=2", "conv_big=1", "conv_big2=1,conv_big2_force=1",
                                 "conv_big2=1,conv_big2_force=1,conv_p4=0", "stem_direct=0,conv3x3_direct=0"],
                         ids=["register_staged_only", "lds_dma_1buf_everywhere", "lds_dma_2buf_everywhere", "tile_256x128_everywhere",
                              "tile_256x256_four_phase_everywhere", "tile_256x256_two_barrier_everywhere", "no_direct_stem_or_layer1_3x3"])
def test_conv_kernel_variants_forced_by_env(env):
    """The loader variant is a per-process policy (environment, read once), so each forced policy re-runs this file's oracle
    comparisons in a child process: register staging only, the LDS-DMA loop with one and two buffers for EVERY launch (the
    default policy uses it from 32 K chunks on), the 8-wave 256 x 128 LDS-DMA tile for every Cout >= 128 launch and the 8-wave
    256 x 256 tile (default policy: bf16, >= 16 chunks, tile counts that fill the 256 single-workgroup slots) for every bf16
    Cout % 256 == 0 launch whatever its size."""
    import os
    import subprocess
    import sys
    child_env = dict(os.environ, MVF_POLICY=policy_str(**dict(kv.split("=") for kv in env.split(","))))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "not forced_by_env",
                        "-p", "no:cacheprovider"], env=child_env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------ MVF fused into the conv's loader
# (clips, T, h, w, C, cout, cs): ragged M (rows past the last tile), clips shorter than a tile, both tile shapes, several K chunks of MVF
MVF_FUSED_CASES = [(2, 4, 7, 7, 128, 64, 64), (3, 3, 5, 9, 256, 128, 128), (1, 8, 14, 14, 512, 256, 64), (40, 2, 2, 2, 128, 128, 64), (2, 1, 6, 6, 256, 64, 64)]


@pytest.mark.parametrize("case", MVF_FUSED_CASES, ids=lambda c: "n%d_t%d_%dx%d_c%d_o%d_cs%d" % c)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("mode,use_hs", [("THW", True), ("TH", True), ("T", False)])
def test_mvf_fused_into_conv_loader_equals_stencil_then_conv(case, dtype, mode, use_hs):
    """mvf_conv2d_nhwc_fwd_mvf (MVF-proper computed inside the 1x1 conv's A-operand loader, reference MVF.py:104-138 + resnet.py:213-215)
    against the two-launch path it replaces (mvf_fwd_infer_slice into a slice buffer, then the split-A conv) and against the numpy
    oracle of MVF-proper followed by a torch conv."""
    from mvfnet_amd import _lib
    from oracle import mvf_numpy
    lib, check = _lib.lib, _lib.check
    clips, T, h, w, Cc, cout, cs = case
    if dtype == torch.float32:
        cs = cs // 2 if cs > 64 else 32                        # fp32 chunks are 32 channels
    nt = clips * T
    dt = _lib.MVF_F32 if dtype == torch.float32 else _lib.MVF_BF16
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    g = torch.Generator().manual_seed(nt * 7 + Cc + cs)
    x = torch.randn(nt, h, w, Cc, generator=g).to(dtype).cuda()
    wt, wh, ww = (torch.randn(cs, 3, generator=g).cuda() * 0.5 for _ in range(3))
    bits = {"T": 1, "TH": 3, "THW": 7}[mode]
    wh_, ww_ = (wh if bits & 2 else None), (ww if bits & 4 else None)
    scale, shift = ((torch.rand(cs, generator=g) + 0.5).cuda(), (torch.randn(cs, generator=g) * 0.3).cuda()) if use_hs else (None, None)
    wgt = (torch.randn(cout, Cc, 1, 1, generator=g) * (2.0 / Cc) ** 0.5).cuda()
    bias = (torch.randn(cout, generator=g) * 0.2).cuda()
    wp = torch.empty(cout, 1, 1, Cc, dtype=dtype, device="cuda")
    check(lib.mvf_pack_conv_weight(p(wgt), cout, Cc, 1, 1, 1, Cc, None, p(wp), dt, None))
    # two launches: stencil -> slice buffer -> split-A conv
    sl = torch.empty(nt, h, w, cs, dtype=dtype, device="cuda")
    md = _lib.MvfDesc(nt, Cc, h, w, T, cs, bits, _lib.MVF_NHWC, dt)
    check(lib.mvf_fwd_infer_slice(C.byref(md), p(x), p(sl), p(wt), p(wh_), p(ww_), p(scale), p(shift), None))
    y2 = torch.empty(nt, h, w, cout, dtype=dtype, device="cuda")
    d2 = _lib.ConvDesc(nt, h, w, Cc, cout, 1, 1, 1, 0, h, w, Cc, dt, 1, cs, cs)
    check(lib.mvf_conv2d_nhwc_fwd(C.byref(d2), p(x), p(sl), p(wp), p(bias), None, p(y2), None))
    # one launch
    z3 = torch.zeros(cs, 3, device="cuda")
    coef = torch.cat([wt, wh_ if wh_ is not None else z3, ww_ if ww_ is not None else z3,
                      scale.view(cs, 1) if use_hs else torch.ones(cs, 1, device="cuda"),
                      shift.view(cs, 1) if use_hs else torch.zeros(cs, 1, device="cuda"), torch.zeros(cs, 1, device="cuda")], 1).contiguous()
    y1 = torch.empty(nt, h, w, cout, dtype=dtype, device="cuda")
    d1 = _lib.ConvDesc(nt, h, w, Cc, cout, 1, 1, 1, 0, h, w, Cc, dt, 1, 0, 0)
    ws = torch.zeros(max(lib.mvf_conv2d_workspace_bytes(C.byref(d1)), 16), dtype=torch.uint8, device="cuda")
    check(lib.mvf_conv2d_nhwc_fwd_mvf(C.byref(d1), p(x), p(wp), p(bias), p(coef), cs, T, int(use_hs), p(y1), p(ws), ws.numel(), None))
    torch.cuda.synchronize()
    # (fp32: the fused-loader kernel multiplies on the fp32 MFMA, the stencil + conv path on the bf16 matrix cores with 3-term splits, [r4] X3:
    # two fp32-accurate paths with different summation orders)
    assert rel_err(y1.float().cpu().numpy(), y2.float().cpu().numpy()) < (3e-6 if dtype == torch.float32 else 8e-3)
    # oracle: numpy MVF-proper (eval BN folded into scale / shift) on the stored input, then the conv in fp32
    xn = x.float().cpu().permute(0, 3, 1, 2).contiguous().numpy()
    s = xn[:, :cs].reshape(clips, T, cs, h, w).astype(np.float64)
    y = np.zeros_like(s)
    for dim, wv, on in ((1, wt, True), (3, wh, bits & 2), (4, ww, bits & 4)):
        if not on:
            continue
        wv = wv.double().cpu().numpy()
        for j in range(3):
            sh_ = np.zeros_like(s)
            dd = j - 1
            src = [slice(None)] * 5
            dst = [slice(None)] * 5
            n_ = s.shape[dim]
            if abs(dd) < n_:
                src[dim] = slice(max(dd, 0), n_ + min(dd, 0))
                dst[dim] = slice(max(-dd, 0), n_ + min(-dd, 0))
                sh_[tuple(dst)] = s[tuple(src)]
            y += wv[:, j].reshape(1, 1, cs, 1, 1) * sh_
    if use_hs:
        u = y * scale.double().cpu().numpy().reshape(1, 1, cs, 1, 1) + shift.double().cpu().numpy().reshape(1, 1, cs, 1, 1)
        y = u * np.clip(u + 3.0, 0.0, 6.0) / 6.0
    xin = xn.copy()
    o = torch.from_numpy(y.reshape(nt, cs, h, w)).to(dtype).float().numpy()          # the engine stores / stages o in `dtype`
    xin[:, :cs] = o
    wq = wgt.to(dtype).float().cpu()
    ref = F.relu(F.conv2d(torch.from_numpy(xin), wq, bias.cpu())).permute(0, 2, 3, 1).numpy()
    assert rel_err(y1.float().cpu().numpy(), ref) < (2e-5 if dtype == torch.float32 else 1e-2)


# ------------------------------------------------------------------------------------------------ [r3] race screen of the four-phase loops
def _race_child():
    """Runs in a child process with the 256 x 256 tiles forced: the SAME conv / weight gradient 40 times at the BASELINE layer3 size while a
    copy kernel on another stream perturbs the timing; every output must equal the first bit for bit (an LDS hazard in the ping-pong loops --
    a read ahead of its DMA, a slot restaged under a late reader -- shows up as a rare different tile, never as a deterministic error)."""
    from mvfnet_amd import _lib
    lib, check = _lib.lib, _lib.check
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    torch.manual_seed(5)
    n, hw, cin, cout, k = 256, 14, 256, 256, 3
    m = n * hw * hw
    x = torch.randn(n, hw, hw, cin, device="cuda").bfloat16()
    wp = (torch.randn(cout, k, k, cin, device="cuda") * 0.03).bfloat16()
    dz = torch.randn(m, cout, device="cuda").bfloat16()
    d = _lib.ConvDesc(n, hw, hw, cin, cout, k, k, 1, 1, hw, hw, cin, 1, 0, 0, 0, 0, 0)
    ws = torch.empty(lib.mvf_conv2d_workspace_bytes(C.byref(d)), dtype=torch.uint8, device="cuda")
    rows = lib.mvf_conv2d_stats_rows(C.byref(d))
    shift = torch.zeros(cout, device="cuda")
    wws = torch.empty(lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device="cuda")
    junk_a, junk_b = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"), torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    first = None
    for it in range(40):
        if it % 3:
            with torch.cuda.stream(side):
                junk_b.copy_(junk_a)                       # ~0.1 ms of HBM streaming beside the GEMMs, on and off
        y = torch.empty(m, cout, device="cuda", dtype=torch.bfloat16)
        part = torch.empty(cout, rows, 2, device="cuda")
        check(lib.mvf_conv2d_nhwc_fwd_stats(C.byref(d), p(x), None, p(wp), p(y), p(part), p(shift), p(ws), ws.numel(), None))
        dw = torch.empty(cout, cin, k, k, device="cuda")
        check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), p(dz), p(x), None, k, cin, k, cin, p(dw), p(wws), wws.numel(), None))
        torch.cuda.synchronize()
        if first is None:
            first = (y.clone(), part.clone(), dw.clone())
            assert torch.isfinite(y.float()).all() and torch.isfinite(dw).all() and float(dw.abs().max()) > 0
        else:
            assert torch.equal(y, first[0]) and torch.equal(part, first[1]), "conv tile differs in run %d" % it
            assert torch.equal(dw, first[2]), "weight gradient differs in run %d" % it
    print("race screen ok")


def test_four_phase_loops_are_run_to_run_bit_identical_under_a_perturbing_stream():
    import os
    import subprocess
    import sys
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_conv_gpu as t; t._race_child()" % (
        os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=policy_env(conv_big2=1, conv_big2_force=1), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "race screen ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------ [r4] fp32 on the bf16 matrix cores (conv_tile X3)
_X3_CHILD = r"""
import ctypes as C, sys, numpy as np, torch
from mvfnet_amd import _lib
lib, check = _lib.lib, _lib.check
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
out = {}
for (n, h, cin, cout, k) in SHAPES:
    gen = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(n, h, h, cin, generator=gen).cuda()
    w = (torch.randn(cout, cin, k, k, generator=gen) / (cin * k * k) ** 0.5).cuda()
    wp = torch.empty(cout, k, k, cin, device="cuda")
    check(lib.mvf_pack_conv_weight(p(w), cout, cin, k, k, k, cin, None, p(wp), 0, None))
    d = _lib.ConvDesc(n, h, h, cin, cout, k, k, 1, k // 2, h, h, cin, 0, 0, 0, 0, 0)
    y = torch.empty(n, h, h, cout, device="cuda")
    check(lib.mvf_conv2d_nhwc_fwd(C.byref(d), p(x), None, p(wp), None, None, p(y), None))
    torch.cuda.synchronize()
    out["%d_%d_%d" % (cin, cout, k)] = y.cpu().numpy()
np.savez(sys.argv[1], **out)
"""
_X3_SHAPES = [(4, 14, 64, 256, 1), (2, 14, 1024, 256, 1), (2, 14, 256, 256, 3), (2, 7, 512, 512, 3), (3, 9, 96, 40, 3)]


@pytest.mark.gpu
def test_fp32_conv_on_the_bf16_matrix_cores_is_as_accurate_as_the_fp32_mfma(tmp_path):
    """conv_tile X3 (the default fp32 path): every fp32 operand is split exactly into three bf16 terms and a product is the six partial
    products of order <= 2^-16 on v_mfma_f32_32x32x16_bf16.  Against an fp64 convolution of the same operands the error must be the fp32
    ACCUMULATION error, i.e. no larger than what the exact-fp32 MFMA path (MVF_POLICY=f32_x3=0, run in a child process: the switch is read once
    per process) leaves -- K = 64 ... 4608, a ragged channel tail included -- and both far inside the 1e-5 every fp32 test allows."""
    import os
    import subprocess
    import sys
    src = "SHAPES = %r\n" % (_X3_SHAPES,) + _X3_CHILD
    res = {}
    for tag, val in (("x3", "1"), ("mfma", "0")):
        f = str(tmp_path / (tag + ".npz"))
        env = dict(os.environ, MVF_POLICY="f32_x3=%s" % val)       # (ONLY this switch: a forced tile / loader policy of a parent test run would send both legs to one kernel)
        env.update(PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        r = subprocess.run([sys.executable, "-c", src, f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(f)
    for (n, h, cin, cout, k) in _X3_SHAPES:
        gen = torch.Generator().manual_seed(cin + cout + k)
        x = torch.randn(n, h, h, cin, generator=gen)
        w = torch.randn(cout, cin, k, k, generator=gen) / (cin * k * k) ** 0.5
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=k // 2).permute(0, 2, 3, 1).numpy()
        key = "%d_%d_%d" % (cin, cout, k)
        e3 = np.linalg.norm(res["x3"][key] - ref) / np.linalg.norm(ref)
        e1 = np.linalg.norm(res["mfma"][key] - ref) / np.linalg.norm(ref)
        assert not np.array_equal(res["x3"][key], res["mfma"][key])           # (two different kernels did run)
        assert e3 < 2e-6 and e3 < 1.5 * e1 + 2e-8, (key, e3, e1)        # (measured: K = 1024 4.9e-7 against the fp32 MFMA's 5.7e-7)
        assert np.abs(res["x3"][key] - ref).max() / np.abs(ref).max() < 2e-6
