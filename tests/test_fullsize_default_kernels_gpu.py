"""[r6] Full-size, DIRECT oracle rows for the kernels that carry the bf16 train step by default since round 5 -- at the sizes where the
product's own rules select them (nothing is forced through an environment switch):

  * mvf_nhwc_apply_lds (csrc/mvf_nhwc.hip: every MVF stencil launch of the C3 / C4 step) against oracle/mvf_numpy.py on channel subsets
    of the full BASELINE tensors: plain, BatchNorm + hard-swish, + batch statistics, the transposed stencil with the gated skip-connection
    addend, + the output gate, + the column sums (MVF.py:104-137 and its autograd transpose);
  * the register-chunked fallback for a clip that does not fit a 32-bit buffer descriptor, and the tile one step under that limit;
  * two-block chains of layer3- and layer2-shaped bottlenecks at the full C3 pixel counts (M = 50 176 / 200 704) through BlockTrainer --
    dz3-free backward, sums from the weight-gradient GEMM, bn3's statistics from the Gram matrix, gated hand-down -- every block against the
    CPU restatement that rounds where the engine rounds (oracle/net_torch.py under helpers.bf16_storage_oracle), teacher-forced from the
    engine's own boundary tensors, plus fp64 checks of the batch statistics (resnet.py:208-244 under autograd)."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import bf16_storage_oracle, rel_err, rel_l2

pytestmark = pytest.mark.gpu

P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
BF = torch.bfloat16
# one bf16 rounding of the stored value (8 significand bits: at most 2^-8 of the element, so of the tensor's scale) + fp32 arithmetic; north_star's budget is 1e-2
TOL_ONE_ROUNDING = 4.5e-3
FULL_SHAPES = [(32, 8, 512, 28, 28), (32, 8, 1024, 14, 14), (32, 8, 2048, 7, 7), (16, 16, 1024, 14, 14)]


def _L():
    from mvfnet_amd import _lib as L
    return L


def _bits(t):
    """(m, c / 4) gate bytes -> (m, c) bool: bit i of byte q = channel 4 q + i."""
    return ((t.unsqueeze(-1) >> torch.arange(4, device=t.device, dtype=torch.uint8)) & 1).reshape(t.shape[0], -1).bool()


def _sub(mat, nt, h, w, chans):
    """channels `chans` of a channels-last matrix (m, C) as the oracle's (nt, len(chans), h, w) float64 array."""
    return mat.view(nt, h, w, -1)[..., chans].permute(0, 3, 1, 2).double().cpu().numpy()


def _uses_tile(L, d, x_c, out_c):
    rb, cw = C.c_int(0), C.c_int(0)
    return bool(L.lib.mvf_nhwc_stencil_tile_plan(C.byref(d), x_c, out_c, C.byref(rb), C.byref(cw))), rb.value, cw.value


# the other MVF shapes the bench's configurations launch: C4's layer2 / layer4 (R101 16x4, 16 clips per GPU) and the reference's own 12 clips per GPU (layer3)
OTHER_SHAPES = [(16, 16, 512, 28, 28), (16, 16, 2048, 7, 7), (12, 8, 1024, 14, 14)]


@pytest.mark.parametrize("shape", FULL_SHAPES + OTHER_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_lds_tiled_stencil_full_size_vs_oracle(shape):
    from oracle import mvf_numpy
    L = _L()
    lib, check = L.lib, L.check
    N, T, c, h, w = shape
    cs, nt = c // 8, N * T
    m = nt * h * w
    gen = torch.Generator(device="cuda").manual_seed(m + cs)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=gen)      # noqa: E731
    x = (rnd(m, c) * 1.3 + 0.2).to(BF)
    dy = rnd(m, cs).to(BF)
    add = rnd(m, c).to(BF)
    abits = torch.randint(0, 16, (m, c // 4), device="cuda", generator=gen, dtype=torch.uint8)
    gate = torch.randint(0, 16, (m, c // 4), device="cuda", generator=gen, dtype=torch.uint8)
    wt, wh, ww = (rnd(cs, 3) for _ in range(3))
    sc, sh = torch.rand(cs, device="cuda", generator=gen) + 0.5, rnd(cs) * 0.2
    gamma, beta = torch.rand(cs, device="cuda", generator=gen) + 0.5, rnd(cs) * 0.2
    d = L.MvfDesc(nt, c, h, w, T, cs, L.MODE_BITS["THW"], L.MVF_NHWC, L.MVF_BF16)
    # the product's own rule puts both directions of this shape on the LDS tile
    fwd_tile, bwd_tile = _uses_tile(L, d, c, cs), _uses_tile(L, d, cs, c)
    assert (fwd_tile[0] and bwd_tile[0]) or shape in OTHER_SHAPES, (fwd_tile, bwd_tile)      # (OTHER_SHAPES: whichever kernel the rule picks meets the oracle)
    chans = np.array(sorted({0, 1, cs // 2 - 1, cs // 2, cs - 2, cs - 1}))
    tsel = lambda t_: t_[chans].double().cpu().numpy()      # noqa: E731
    taps = dict(wt=tsel(wt), wh=tsel(wh), ww=tsel(ww))
    xs = _sub(x, nt, h, w, chans)
    k = len(chans)
    # ---- (a) plain: y = the three views' tap sum
    y = torch.zeros(m, cs, device="cuda", dtype=BF)
    check(lib.mvf_nhwc_stencil(C.byref(d), P(x), c, P(y), cs, P(wt), P(wh), P(ww), None, None, 0, None, 0, None, None))
    ref_y, cache, _ = mvf_numpy.mvf_forward(xs, T, k, use_hs=False, **taps)
    e_plain = rel_err(_sub(y, nt, h, w, chans), ref_y)
    # ---- (b) + BatchNorm (folded scale / shift) + hard-swish
    yh = torch.zeros(m, cs, device="cuda", dtype=BF)
    check(lib.mvf_nhwc_stencil(C.byref(d), P(x), c, P(yh), cs, P(wt), P(wh), P(ww), P(sc), P(sh), 0, None, 0, None, None))
    ref_h, _, _ = mvf_numpy.mvf_forward(xs, T, k, use_hs=True, gamma=tsel(sc), beta=tsel(sh), running_mean=np.zeros(k), running_var=np.full(k, 1.0 - mvf_numpy.EPS),
                                        training=False, **taps)
    e_hs = rel_err(_sub(yh, nt, h, w, chans), ref_h)
    # ---- (c) + the batch statistics of MVF's BatchNorm3d over what is stored (training mode: MVF.py:131-134)
    rows = lib.mvf_nhwc_stencil_stats_rows(C.byref(d), c, cs)
    part = torch.full((cs, rows, 2), float("nan"), device="cuda")
    rm, rv = rnd(cs) * 0.1, torch.rand(cs, device="cuda", generator=gen) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    ys = torch.zeros(m, cs, device="cuda", dtype=BF)
    st = [torch.empty(cs, device="cuda") for _ in range(4)]
    check(lib.mvf_nhwc_stencil_stats(C.byref(d), P(x), c, P(ys), cs, P(wt), P(wh), P(ww), P(part), P(rm), None))
    check(lib.mvf_bn_train_finalize(P(part), rows, m, cs, P(gamma), P(beta), C.c_float(1e-5), C.c_float(0.1), P(rm), P(rv), P(st[0]), P(st[1]), P(st[2]), P(st[3]), None))
    torch.cuda.synchronize()
    assert torch.equal(ys.view(torch.int16), y.view(torch.int16))
    _, cache_t, (new_rm, new_rv) = mvf_numpy.mvf_forward(xs, T, k, use_hs=True, gamma=tsel(gamma), beta=tsel(beta), running_mean=tsel(rm0), running_var=tsel(rv0),
                                                       training=True, **taps)
    mean64, std64 = ref_y.mean(axis=(0, 2, 3)), ref_y.std(axis=(0, 2, 3))
    e_mean = float(np.abs(tsel(st[0]) - mean64).max() / std64.max())
    e_inv = rel_err(tsel(st[1]), cache_t["invstd"])
    e_run = max(rel_err(tsel(rm), new_rm), rel_err(tsel(rv), new_rv))
    # ---- (d) the transposed stencil of the backward + the gated skip-connection addend (MVF.py:118-129 under autograd; resnet.py:241 out += identity)
    o = torch.zeros(m, c, device="cuda", dtype=BF)
    check(lib.mvf_nhwc_stencil(C.byref(d), P(dy), cs, P(o), c, P(wt), P(wh), P(ww), None, None, 1, P(add), c, P(abits), None))
    ds = mvf_numpy.mvf_backward(_sub(dy, nt, h, w, chans), cache)["dx"]
    keep_add = _bits(abits).view(nt, h, w, c)[..., chans].permute(0, 3, 1, 2).cpu().numpy()
    ref_t = ds + _sub(add, nt, h, w, chans) * keep_add
    e_tr = rel_err(_sub(o, nt, h, w, chans), ref_t)
    o_plain = o[:, :cs].clone()
    # ---- (e) ... with the output gate, and with the column sums of what is stored (the dz3-free block below reads both)
    check(lib.mvf_nhwc_stencil_gate(C.byref(d), P(dy), cs, P(o), c, P(wt), P(wh), P(ww), None, None, 1, P(add), c, P(abits), P(gate), None))
    keep_gate = _bits(gate).view(nt, h, w, c)[..., chans].permute(0, 3, 1, 2).cpu().numpy()
    e_gate = rel_err(_sub(o, nt, h, w, chans), ref_t * keep_gate)
    og = o[:, :cs].clone()
    assert torch.equal(torch.where(_bits(gate)[:, :cs], o_plain, torch.zeros_like(o_plain)).view(torch.int16), og.view(torch.int16))
    rows2 = lib.mvf_nhwc_stencil_stats_rows(C.byref(d), cs, c)
    part2 = torch.full((cs, rows2, 2), float("nan"), device="cuda")
    o.zero_()
    check(lib.mvf_nhwc_stencil_gate_colsums(C.byref(d), P(dy), cs, P(o), c, P(wt), P(wh), P(ww), 1, P(add), c, P(abits), P(gate), P(part2), None))
    torch.cuda.synchronize()
    assert torch.equal(o[:, :cs].contiguous().view(torch.int16), og.view(torch.int16))
    assert torch.count_nonzero(o[:, cs:]) == 0                    # channels >= cs are conv1's data gradient's, not this kernel's
    s1, s2 = part2[:, :, 0].double().sum(1), part2[:, :, 1].double().sum(1)
    e_s1 = rel_l2(s1.cpu().numpy(), og.double().sum(0).cpu().numpy())
    e_s2 = rel_l2(s2.cpu().numpy(), (og.double() ** 2).sum(0).cpu().numpy())
    print("LDS stencil %s tile fwd %s bwd %s: plain %.2e hardswish %.2e | mean %.1e invstd %.1e running %.1e | transposed %.2e gated %.2e | column sums %.1e / %.1e" %
          (shape, fwd_tile[1:], bwd_tile[1:], e_plain, e_hs, e_mean, e_inv, e_run, e_tr, e_gate, e_s1, e_s2))
    assert e_plain < TOL_ONE_ROUNDING and e_hs < TOL_ONE_ROUNDING and e_tr < TOL_ONE_ROUNDING and e_gate < TOL_ONE_ROUNDING
    assert e_mean < 1e-4 and e_inv < 1e-4 and e_run < 1e-4         # statistics of the ROUNDED tensor: the rounding is unbiased, m >= 12 544 samples per channel
    assert e_s1 < 1e-5 and e_s2 < 1e-5


@pytest.mark.parametrize("over", [True, False], ids=["clip_over_2GiB_chunked", "clip_just_under_2GiB_tiled"])
def test_stencil_clip_at_the_32_bit_descriptor_limit(over):
    """One clip of the source tensor one step over / under what a 32-bit buffer descriptor addresses (T * H * W * C * 2 bytes vs 0x7ffffff0): over it the
    launch must take the register-chunked kernel (64-bit addressing), under it the tile with clip-relative offsets up to 2^31 -- both against the oracle
    on channels at both ends of the slice (where the byte offsets are smallest and largest)."""
    from oracle import mvf_numpy
    L = _L()
    lib, check = L.lib, L.check
    T, h, w = 8, 14, 14
    c = 688128 if over else 684672
    cs = c // 8
    assert (T * h * w * c * 2 >= 0x7ffffff0) == over and cs % 16 == 0
    nt = T
    m = nt * h * w
    x = torch.empty(m, c, device="cuda", dtype=BF).normal_(generator=torch.Generator(device="cuda").manual_seed(3))
    wt, wh, ww = (torch.randn(cs, 3, device="cuda") for _ in range(3))
    d = L.MvfDesc(nt, c, h, w, T, cs, L.MODE_BITS["THW"], L.MVF_NHWC, L.MVF_BF16)
    tile = _uses_tile(L, d, c, cs)
    assert tile[0] == (not over), tile
    y = torch.zeros(m, cs, device="cuda", dtype=BF)
    check(lib.mvf_nhwc_stencil(C.byref(d), P(x), c, P(y), cs, P(wt), P(wh), P(ww), None, None, 0, None, 0, None, None))
    torch.cuda.synchronize()
    chans = np.array([0, 1, 2, 3, cs // 2, cs - 4, cs - 3, cs - 2, cs - 1])
    tsel = lambda t_: t_[chans].double().cpu().numpy()      # noqa: E731
    ref, _, _ = mvf_numpy.mvf_forward(_sub(x, nt, h, w, chans), T, len(chans), use_hs=False, wt=tsel(wt), wh=tsel(wh), ww=tsel(ww))
    e = rel_err(_sub(y, nt, h, w, chans), ref)
    print("stencil with a %.3f GiB clip (%s kernel): %.2e" % (T * h * w * c * 2 / 2.0 ** 30, "tiled" if tile[0] else "chunked", e))
    assert e < TOL_ONE_ROUNDING


def _chain(cin, planes, T, with_mvf):
    from mvfnet_amd.backbones.resnet import Bottleneck
    from mvfnet_amd.modules.MVF import MVF
    blks = []
    for _ in range(2):
        blk = Bottleneck(cin, planes)
        if with_mvf:
            blk.conv1 = MVF(blk.conv1, T, cin, 0.125)
        with torch.no_grad():
            for bn in (blk.bn1, blk.bn2, blk.bn3):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.normal_(0, 0.2)
        blks.append(blk)
    return torch.nn.Sequential(*blks).cuda().train()


@pytest.mark.parametrize("stage", ["layer3", "layer2"])
def test_full_c3_size_block_chain_vs_storage_emulation(stage):
    """Two plain bottlenecks in a row at the C3 step's full pixel count (256 frames), bf16 storage, the engine's DEFAULT switches.  The lower block takes its
    gradient gated from the upper one, its bn3 backward sums from the weight-gradient GEMM (mvf_bn_bwd_dzfree_sums) and never forms dz3; in layer2 it also takes
    bn3's batch statistics from the Gram matrix of a2 and never stores z3.  Each block is then re-run on the CPU emulation from the engine's own stored input
    and incoming gradient and compared: output, outgoing gradient, every parameter gradient."""
    from mvfnet_amd.train_engine import BlockTrainer
    from oracle import net_torch
    cin, planes, hw, T, with_mvf = dict(layer3=(1024, 256, 14, 8, True), layer2=(512, 128, 28, 8, False))[stage]
    nt = 256
    m = nt * hw * hw
    torch.manual_seed(17)
    seq = _chain(cin, planes, T, with_mvf)
    sds = [{k: v.detach().cpu().clone() for k, v in blk.state_dict().items()} for blk in seq]
    tr = BlockTrainer(seq, dtype=torch.bfloat16)
    tr.keep_io = True
    x = torch.relu(torch.randn(nt, cin, hw, hw, device="cuda"))
    dy = torch.randn(nt, cin, hw, hw, device="cuda")
    tr.forward(x)
    lo, hi = tr.blks
    # what the product's rules chose at this size
    assert lo.dzfree(tr, m) and hi.dzfree(tr, m) and lo.q_policy(tr, m)
    assert lo.gram_fwd(tr, m) == (stage == "layer2") and (lo.saved["z3"] is None) == (stage == "layer2") and hi.saved["z3"] is not None
    # bn3's batch statistics of the lower block against fp64 on the unrounded z3 = a2 W^T (Gram form in layer2, the conv epilogue's sums in layer3)
    a2 = lo.saved["a2"].double()
    w3 = seq[0].conv3.weight.detach().view(cin, planes).to(BF).double()
    zx = a2 @ w3.t()
    mean64, inv64 = zx.mean(0), 1.0 / torch.sqrt(zx.var(0, unbiased=False) + 1e-5)
    e_stat = (rel_l2(lo.b3.mean.cpu().numpy(), mean64.cpu().numpy()), rel_l2(lo.b3.invstd.cpu().numpy(), inv64.cpu().numpy()))
    del a2, zx
    assert max(e_stat) < (1e-6 if stage == "layer2" else 1e-4), e_stat
    tr.backward(dy)
    torch.cuda.synchronize()
    assert hi.gated_out and hi.sums_out == "s1"
    torch.set_num_threads(min(32, torch.get_num_threads()))
    rep = []
    for i, (blk, tb) in enumerate(zip(seq, tr.blks)):
        io = tb.io
        sd, leaves = {}, {}
        for k_, v in sds[i].items():
            v = v.clone()
            if v.dtype.is_floating_point and "running" not in k_:
                v.requires_grad_(True)
                leaves[k_] = v
            sd[k_] = v
        nchw = lambda t_, c_: t_.view(nt, hw, hw, c_).float().cpu().permute(0, 3, 1, 2).contiguous()      # noqa: E731
        xr = nchw(io["x"], cin).requires_grad_(True)
        with bf16_storage_oracle(backward=True):
            y = net_torch.bottleneck(xr, sd, "", 1, T, dict(mode="THW", share=False, use_hs=True) if with_mvf else None, True, {})
            y.backward(nchw(io["g"], cin))
        e_out = rel_l2(nchw(io["out"], cin).numpy(), y.detach().numpy())
        ref_dx = xr.grad
        if io.get("dx_gated"):
            ref_dx = ref_dx * (xr.detach() > 0).to(ref_dx.dtype)
        e_dx = rel_l2(nchw(io["dx"], cin).numpy(), ref_dx.bfloat16().float().numpy())
        params = dict(blk.named_parameters())
        errs = {k_: rel_l2(tr.grad_of(params[k_]).cpu().numpy(), v.grad.numpy()) for k_, v in leaves.items()}
        worst = max(errs, key=errs.get)
        rep.append((e_out, e_dx, errs[worst], worst))
        del y, xr, sd, leaves
    print("full-size %s chain (M = %d), bn3 statistics vs fp64 %.1e / %.1e; per block (lower, upper): out %.2e %.2e | dx %.2e %.2e | worst parameter gradient %.2e (%s) %.2e (%s)" %
          (stage, m, e_stat[0], e_stat[1], rep[0][0], rep[1][0], rep[0][1], rep[1][1], rep[0][2], rep[0][3], rep[1][2], rep[1][3]))
    for e_out, e_dx, e_w, name in rep:
        assert e_out < 5e-3 and e_dx < 4e-2 and e_w < 5e-2, (stage, e_out, e_dx, e_w, name)
