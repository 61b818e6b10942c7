"""CPU: pin the oracle (oracle/) against the golden vectors captured from the imported reference."""
import numpy as np
import pytest
import torch

from cases import BLOCK_CASES, MVF_CASES
from helpers import golden, mvf_case_params, rel_err
from mvfnet_amd import synth
from oracle import mvf_numpy, net_torch

DEGENERATE_BWD = {"thw_h1w1"}
TOL = 2e-5   # fp32 reference vs fp64 oracle arithmetic, relative to tensor scale


@pytest.mark.parametrize("case", MVF_CASES, ids=[c[0] for c in MVF_CASES])
@pytest.mark.parametrize("train", [False, True], ids=["eval", "train"])
def test_mvf_numpy_oracle_matches_reference(case, train):
    name, N, T, C, H, W, alpha, mode, share, use_hs, planes = case
    g = golden("mvf_cases.npz")
    p = mvf_case_params(name, C, alpha, mode, share, use_hs, planes, "id")
    cs = int(C * alpha)
    x = synth.synth_tensor("mvf_x/" + name, (N * T, C, H, W))
    tag = "%s/id/%s" % (name, "train" if train else "eval")
    kw = {}
    if cs:
        kw = dict(wt=p["shift_conv.weight"].reshape(cs, 3), wh=p.get("h_conv.weight"), ww=p.get("w_conv.weight"),
                  gamma=p["bn.weight"], beta=p["bn.bias"], running_mean=p["bn.running_mean"],
                  running_var=p["bn.running_var"])
    else:
        kw = dict(wt=None)
    out, cache, (rm, rv) = mvf_numpy.mvf_forward(x, T, cs, mode=mode, share=share, use_hs=use_hs, training=train, **kw)
    assert rel_err(out, g[tag + "/y"]) < TOL
    if not cs:
        return
    dy = synth.synth_tensor("mvf_dy/%s/id" % name, out.shape)
    r = mvf_numpy.mvf_backward(dy, cache)
    if name in DEGENERATE_BWD:
        # H=W=1: the reference's ATen Conv3d backward on its strided (transposed+split) view returns a dx
        # that disagrees with finite differences (size-1 dims make the stride check ambiguous). The forward
        # still matches; the backward is pinned against autograd of the closed form (fd-verified) instead.
        sd = {k: torch.from_numpy(v).double() if v.dtype == np.float32 else torch.from_numpy(np.asarray(v))
              for k, v in p.items()}
        for k in sd:
            if sd[k].dtype == torch.float64 and "running" not in k:
                sd[k].requires_grad_(True)
        xt = torch.from_numpy(x).double().requires_grad_(True)
        yt = net_torch.mvf_proper(xt, sd, "", T, mode, share, use_hs, train)
        yt.backward(torch.from_numpy(dy).double())
        assert rel_err(r["dx"], xt.grad.numpy()) < 1e-10
        assert rel_err(r["dwt"].reshape(-1), sd["shift_conv.weight"].grad.numpy().reshape(-1)) < 1e-10
        assert rel_err(r["dgamma"], sd["bn.weight"].grad.numpy()) < 1e-10
        return
    assert rel_err(r["dx"], g[tag + "/dx"]) < TOL
    assert rel_err(r["dwt"].reshape(-1), g[tag + "/grad/shift_conv.weight"].reshape(-1)) < TOL
    if not share and mode in ("TH", "THW"):
        assert rel_err(r["dwh"].reshape(-1), g[tag + "/grad/h_conv.weight"].reshape(-1)) < TOL
    if not share and mode == "THW":
        assert rel_err(r["dww"].reshape(-1), g[tag + "/grad/w_conv.weight"].reshape(-1)) < TOL
    if use_hs:
        assert rel_err(r["dgamma"], g[tag + "/grad/bn.weight"]) < TOL
        assert rel_err(r["dbeta"], g[tag + "/grad/bn.bias"]) < TOL
        if train:
            assert rel_err(rm, g[tag + "/buf/bn.running_mean"]) < TOL
            assert rel_err(rv, g[tag + "/buf/bn.running_var"]) < TOL


def _block_sd(name, Cin, planes, stride, requires_grad=True):
    cs = int(Cin * 0.125)
    shapes = {"conv1.net.weight": (planes, Cin, 1, 1), "conv1.shift_conv.weight": (cs, 1, 3, 1, 1),
              "conv1.h_conv.weight": (cs, 1, 1, 3, 1), "conv1.w_conv.weight": (cs, 1, 1, 1, 3),
              "conv2.weight": (planes, planes, 3, 3), "conv3.weight": (planes * 4, planes, 1, 1)}
    bns = {"conv1.bn.": cs, "bn1.": planes, "bn2.": planes, "bn3.": planes * 4}
    if stride != 1 or Cin != planes * 4:
        shapes["downsample.0.weight"] = (planes * 4, Cin, 1, 1)
        bns["downsample.1."] = planes * 4
    for pre, n in bns.items():
        for k in ("weight", "bias", "running_mean", "running_var"):
            shapes[pre + k] = (n,)
        shapes[pre + "num_batches_tracked"] = ()
    pre = "block/%s/" % name
    vals = synth.synth_state_dict({pre + k: v for k, v in shapes.items()})
    sd = {}
    for k in shapes:
        t = torch.from_numpy(vals[pre + k])
        if requires_grad and t.dtype == torch.float32 and "running" not in k:
            t.requires_grad_(True)
        sd[k] = t
    return sd


@pytest.mark.parametrize("name", sorted(BLOCK_CASES))
@pytest.mark.parametrize("train", [False, True], ids=["eval", "train"])
def test_bottleneck_oracle_matches_reference(name, train):
    N, T, Cin, planes, H, W, stride = BLOCK_CASES[name]
    g = golden("block_cases.npz")
    sd = _block_sd(name, Cin, planes, stride)
    x = torch.from_numpy(synth.synth_tensor("block_x/" + name, (N * T, Cin, H, W))).requires_grad_(True)
    nb = {}
    y = net_torch.bottleneck(x, sd, "", stride, T, dict(mode="THW", share=False, use_hs=True), train, nb)
    tag = "%s/%s" % (name, "train" if train else "eval")
    assert rel_err(y.detach().numpy(), g[tag + "/y"]) < TOL
    y.backward(torch.from_numpy(synth.synth_tensor("block_dy/" + name, tuple(y.shape))))
    assert rel_err(x.grad.numpy(), g[tag + "/dx"]) < 1e-4
    for k, t in sd.items():
        if t.requires_grad:
            assert rel_err(t.grad.numpy(), g[tag + "/grad/" + k]) < 1e-4, k
    if train:
        for k, t in nb.items():
            assert rel_err(t.numpy(), g[tag + "/buf/" + k]) < TOL, k


def _net_sd(depth, requires_grad=False):
    g = golden("net_cases.npz")
    if depth == 50:
        keys, shapes = list(g["struct/r50/keys"]), [eval(s) for s in g["struct/r50/shapes"]]
        shp = dict(zip(keys, shapes))
    else:
        from mvfnet_amd.arch import state_dict_shapes
        shp = state_dict_shapes(depth)
    pre = "r%d/" % depth
    vals = synth.synth_state_dict({pre + k: v for k, v in shp.items()})
    sd = {}
    for k in shp:
        t = torch.from_numpy(vals[pre + k])
        if requires_grad and t.dtype == torch.float32 and "running" not in k:
            t.requires_grad_(True)
        sd[k] = t
    return sd


def test_arch_table_matches_reference_state_dict():
    """mvfnet_amd.arch.state_dict_shapes reproduces the reference's key names and shapes (R50), param and
    MVF counts (R50/R101) -- the reference's own known answers: 24.34 M / 43.36 M params (cfg :1-5)."""
    from mvfnet_amd.arch import state_dict_shapes
    g = golden("net_cases.npz")
    shp = state_dict_shapes(50)
    assert sorted(shp) == list(g["struct/r50/keys"])
    assert [str(tuple(shp[k])) for k in sorted(shp)] == list(g["struct/r50/shapes"])
    for depth, tag in ((50, "r50"), (101, "r101")):
        shp = state_dict_shapes(depth)
        npar = sum(int(np.prod(v)) for k, v in shp.items() if "running" not in k and "num_batches" not in k)
        assert npar == int(g["struct/%s/n_params" % tag])
        assert sum(1 for k in shp if k.endswith("shift_conv.weight")) == int(g["struct/%s/n_mvf" % tag])
    assert round(int(g["struct/r50/n_params"]) / 1e6, 2) == 24.34
    assert round(int(g["struct/r101/n_params"]) / 1e6, 2) == 43.36


def test_net_oracle_eval_c1():
    """BASELINE config 1: R50 4x16, 2 clips 224^2, CPU forward."""
    g = golden("net_cases.npz")
    sd = _net_sd(50)
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 224, 224))
    stages = {}
    with torch.no_grad():
        logits = net_torch.forward_test(imgs, sd, 50, 4, None, stages=stages)
    assert rel_err(logits.numpy(), g["c1/eval/logits"]) < TOL
    assert (logits.numpy().argmax(1) == g["c1/eval/logits"].argmax(1)).all()
    for k, v in stages.items():
        a = v.numpy().astype(np.float64).ravel()
        ref = g["c1/eval/stage/" + k]
        assert abs(a.mean() - ref[0]) < 1e-5 * max(1, abs(ref[2]))
        assert abs(np.sqrt((a * a).mean()) - ref[1]) < 1e-5 * ref[2]
        idx = np.linspace(0, a.size - 1, 16).astype(np.int64)
        assert np.abs(a[idx] - ref[3:]).max() < 1e-4 * ref[2]
    with torch.no_grad():
        assert rel_err(net_torch.average_clip(logits, "prob").numpy(), g["c1/eval/prob"]) < TOL
        assert rel_err(net_torch.average_clip(logits, "score").numpy(), g["c1/eval/score"]) < TOL


def test_net_oracle_train_two_steps_c1():
    g = golden("net_cases.npz")
    sd = _net_sd(50, requires_grad=True)
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 224, 224))
    labels = torch.from_numpy(synth.synth_labels(2))
    params = {k: v for k, v in sd.items() if v.requires_grad}
    mom = {}
    for it in range(2):
        for p in params.values():
            p.grad = None
        nb = {}
        loss = net_torch.forward_train(imgs, labels, sd, 50, new_buffers=nb)
        loss.backward()
        # step 0 is a pure forward (tight); step 1 sees parameters updated with the ill-conditioned grads above
        tol = 2e-5 if it == 0 else 2e-3
        assert abs(float(loss.detach()) - float(g["c1/train/loss/%d" % it])) < tol * abs(float(g["c1/train/loss/%d" % it]))
        if it == 0:
            names = list(g["c1/train/grad_names"])
            ref = g["c1/train/grad_norms"]
            for n, r in zip(names, ref):
                got = float(params[n].grad.double().norm())
                assert abs(got - r) <= 2e-3 * max(r, 1e-6), n
            for k in g.files:
                if k.startswith("c1/train/grad/"):
                    # 53 train-mode BN layers over only 8 images make early-layer grads ill-conditioned:
                    # fp32 op-order differences alone move backbone.conv1.weight.grad by ~4e-3 (measured)
                    assert rel_err(params[k[len("c1/train/grad/"):]].grad.numpy(), g[k]) < 2e-2, k
        with torch.no_grad():
            total = net_torch.sgd_nesterov_step({k: v for k, v in params.items()},
                                                {k: v.grad for k, v in params.items()}, mom)
            for k, v in nb.items():
                sd[k] = v
        assert abs(float(total) - float(g["c1/train/total_norm/%d" % it])) < (1e-3 if it == 0 else 1e-2) * float(g["c1/train/total_norm/%d" % it])
    for k in g.files:
        if k.startswith("c1/train/after2/"):
            n = k[len("c1/train/after2/"):]
            a = sd[n].detach().numpy().ravel()
            # chaotic regime: this oracle in fp32 vs fp64 differs by 5e-2 on backbone.conv1.weight after 2 steps
            # (the step-1 grads differ by O(1)); the check pins the update RULE (lr/momentum/wd/clip order), which
            # would be off by far more if wrong, not the low-order bits.
            assert rel_err(a[: g[k].size], g[k]) < 0.1, n


@pytest.mark.parametrize("depth,T,tag", [(50, 8, "r50_t8"), (101, 16, "r101_t16")])
def test_net_oracle_eval_t8_t16(depth, T, tag):
    g = golden("net_cases.npz")
    sd = _net_sd(depth)
    imgs = torch.from_numpy(synth.synth_clip_batch(1, T, 224, 224, seed=depth))
    with torch.no_grad():
        logits = net_torch.forward_test(imgs, sd, depth, T, None)
    assert rel_err(logits.numpy(), g[tag + "/eval/logits"]) < TOL
    assert (logits.numpy().argmax(1) == g[tag + "/eval/logits"].argmax(1)).all()


def test_net_oracle_fcn_testing():
    g = golden("net_cases.npz")
    sd = _net_sd(50)
    vid = torch.from_numpy(synth.synth_tensor("fcn_video", (1, 3 * 2 * 4, 3, 128, 128)))
    with torch.no_grad():
        prob = net_torch.forward_test(vid, sd, 50, 4, "prob", fcn_testing=True)
        scores = net_torch.forward_test(vid, sd, 50, 4, None, fcn_testing=True)
    assert rel_err(prob.numpy(), g["fcn/prob"]) < TOL
    assert rel_err(scores.numpy(), g["fcn/scores"]) < TOL
    assert prob.numpy().argmax() == g["fcn/prob"].argmax()


@pytest.mark.parametrize("case", [c for c in MVF_CASES if int(c[3] * c[6]) and not c[8]], ids=lambda c: c[0])
@pytest.mark.parametrize("train", [False, True], ids=["eval", "train"])
def test_mvf_plain_c_oracle_matches_reference(case, train):
    """oracle/mvf_ref.c (scalar C, double accumulation) against the reference's golden outputs and gradients."""
    from oracle import mvf_c
    name, N, T, C, H, W, alpha, mode, share, use_hs, planes = case
    g = golden("mvf_cases.npz")
    p = mvf_case_params(name, C, alpha, mode, share, use_hs, planes, "id")
    cs = int(C * alpha)
    x = synth.synth_tensor("mvf_x/" + name, (N * T, C, H, W))
    tag = "%s/id/%s" % (name, "train" if train else "eval")
    dy = synth.synth_tensor("mvf_dy/%s/id" % name, x.shape)
    r = mvf_c.forward_backward(x, T, cs, {"T": 1, "TH": 3, "THW": 7}[mode], p["shift_conv.weight"].reshape(cs, 3),
                               p["h_conv.weight"].reshape(cs, 3) if "h_conv.weight" in p else None,
                               p["w_conv.weight"].reshape(cs, 3) if "w_conv.weight" in p else None, use_hs, train,
                               p["bn.weight"], p["bn.bias"], p["bn.running_mean"], p["bn.running_var"], g=dy)
    assert rel_err(r["out"], g[tag + "/y"]) < TOL
    if name in DEGENERATE_BWD:
        return
    assert rel_err(r["dx"], g[tag + "/dx"]) < TOL
    assert rel_err(r["dwt"].reshape(-1), g[tag + "/grad/shift_conv.weight"].reshape(-1)) < TOL
    if mode in ("TH", "THW"):
        assert rel_err(r["dwh"].reshape(-1), g[tag + "/grad/h_conv.weight"].reshape(-1)) < TOL
    if mode == "THW":
        assert rel_err(r["dww"].reshape(-1), g[tag + "/grad/w_conv.weight"].reshape(-1)) < TOL
    if use_hs:
        assert rel_err(r["dgamma"], g[tag + "/grad/bn.weight"]) < TOL and rel_err(r["dbeta"], g[tag + "/grad/bn.bias"]) < TOL
        if train:
            assert rel_err(r["running_mean"], g[tag + "/buf/bn.running_mean"]) < TOL
            assert rel_err(r["running_var"], g[tag + "/buf/bn.running_var"]) < TOL
