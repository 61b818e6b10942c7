#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself on CPU.

Run ONCE in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

Nothing of the reference (source, bytecode) is written anywhere: only inputs-by-formula
(mvfnet_amd.synth) go in and small output arrays come out as .npz fixtures.  The GPU box
never sees /root/reference; tests there read the committed .npz files only.

The reference imports mmcv / torchvision / cv2 at module level;
none is installed, so inert placeholders are put in sys.modules first.  Only
kaiming_init / constant_init / is_str need real behaviour, and they are irrelevant to the
vectors because every parameter is overwritten by synth_state_dict afterwards.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
from mvfnet_amd import synth  # noqa: E402

REF = "/root/reference"


# --------------------------------------------------------------------------- stubs
def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Inert(object):
        def __init__(self, *a, **k):
            pass

    def kaiming_init(module, mode="fan_out", nonlinearity="relu", bias=0, distribution="normal"):
        nn.init.kaiming_normal_(module.weight, mode=mode, nonlinearity=nonlinearity)
        if getattr(module, "bias", None) is not None:
            nn.init.constant_(module.bias, bias)

    def constant_init(module, val, bias=0):
        nn.init.constant_(module.weight, val)
        if getattr(module, "bias", None) is not None:
            nn.init.constant_(module.bias, bias)

    mmcv = mod("mmcv", is_str=lambda x: isinstance(x, str), ProgressBar=_Inert, dump=None, load=None,
               mkdir_or_exist=None, Config=_Inert, __version__="0.4.3", __path__=[])
    mmcv.cnn = mod("mmcv.cnn", kaiming_init=kaiming_init, constant_init=constant_init,
                   normal_init=None, xavier_init=None, __path__=[])
    mmcv.runner = mod("mmcv.runner", OptimizerHook=_Inert, Hook=_Inert, DistSamplerSeedHook=_Inert,
                      Runner=_Inert, obj_from_dict=None, get_dist_info=lambda: (0, 1),
                      load_state_dict=None, load_checkpoint=None, __path__=[])
    mmcv.parallel = mod("mmcv.parallel", DataContainer=_Inert, collate=None, __path__=[])
    mmcv.fileio = mod("mmcv.fileio", __path__=[])
    tv = mod("torchvision", __path__=[])
    tv.models = mod("torchvision.models", __path__=[])
    mod("cv2")


def _import_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        from codes.models import build_recognizer  # noqa
        from codes.models.modules.MVF import MVF  # noqa
        from codes.models.backbones.resnet import Bottleneck  # noqa
    return build_recognizer, MVF, Bottleneck


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def load_synth(module, prefix, seed=0, fc_std=0.05):
    sd = module.state_dict()
    shapes = {prefix + k: tuple(v.shape) for k, v in sd.items()}
    vals = synth.synth_state_dict(shapes, seed=seed, fc_std=fc_std)
    module.load_state_dict({k: torch.from_numpy(vals[prefix + k]) for k in sd}, strict=True)


def t2n(t):
    return t.detach().cpu().numpy().copy()


# --------------------------------------------------------------------------- (i) MVF module cases
from cases import MVF_CASES, BLOCK_CASES  # noqa: E402


def gen_mvf_cases(MVF):
    out = {}
    for (name, N, T, C, H, W, alpha, mode, share, use_hs, planes) in MVF_CASES:
        for net_kind in ("id", "conv"):
            for train in (False, True):
                tag = "%s/%s/%s" % (name, net_kind, "train" if train else "eval")
                net = nn.Identity() if net_kind == "id" else nn.Conv2d(C, planes, 1, bias=False)
                m = quiet(MVF, net, T, C, alpha, use_hs, share, mode)
                load_synth(m, "mvf/%s/" % name)
                m.train(train)
                x = torch.from_numpy(synth.synth_tensor("mvf_x/" + name, (N * T, C, H, W))).requires_grad_(True)
                y = m(x)
                dy = torch.from_numpy(synth.synth_tensor("mvf_dy/%s/%s" % (name, net_kind), tuple(y.shape)))
                y.backward(dy)
                out[tag + "/y"] = t2n(y)
                out[tag + "/dx"] = t2n(x.grad)
                for pn, p in m.named_parameters():
                    if p.grad is not None:      # use_hs=False leaves bn.* unused (MVF.py:131-134)
                        out[tag + "/grad/" + pn] = t2n(p.grad)
                if train:
                    for bn_, b in m.named_buffers():
                        out[tag + "/buf/" + bn_] = t2n(b)
    np.savez_compressed(os.path.join(HERE, "mvf_cases.npz"), **out)
    print("mvf_cases.npz: %d arrays" % len(out))


# --------------------------------------------------------------------------- (ii) bottleneck block
def gen_block(MVF, Bottleneck):
    out = {}
    for name, (N, T, Cin, planes, H, W, stride) in BLOCK_CASES.items():
        for train in (False, True):
            down = None
            if stride != 1 or Cin != planes * 4:
                down = nn.Sequential(nn.Conv2d(Cin, planes * 4, 1, stride=stride, bias=False),
                                     nn.BatchNorm2d(planes * 4))
            blk = Bottleneck(Cin, planes, stride, 1, down, style="pytorch", norm_cfg=dict(type="BN"))
            blk.conv1 = quiet(MVF, blk.conv1, T, Cin, 0.125, True, False, "THW")
            load_synth(blk, "block/%s/" % name)
            blk.train(train)
            x = torch.from_numpy(synth.synth_tensor("block_x/" + name, (N * T, Cin, H, W))).requires_grad_(True)
            y = blk(x)
            dy = torch.from_numpy(synth.synth_tensor("block_dy/" + name, tuple(y.shape)))
            y.backward(dy)
            tag = "%s/%s" % (name, "train" if train else "eval")
            out[tag + "/y"] = t2n(y)
            out[tag + "/dx"] = t2n(x.grad)
            for pn, p in blk.named_parameters():
                out[tag + "/grad/" + pn] = t2n(p.grad)
            if train:
                for bn_, b in blk.named_buffers():
                    out[tag + "/buf/" + bn_] = t2n(b)
    np.savez_compressed(os.path.join(HERE, "block_cases.npz"), **out)
    print("block_cases.npz: %d arrays" % len(out))


# --------------------------------------------------------------------------- (iii)-(vi) full networks
def model_cfg(depth, T, dropout=0.5, fcn=False):
    return dict(
        type="Recognizer2D",
        backbone=dict(type="ResNet", pretrained=None, depth=depth, out_indices=(3,), norm_eval=False,
                      partial_norm=False, norm_cfg=dict(type="BN", requires_grad=True)),
        cls_head=dict(type="TSNClsHead", spatial_size=-1, spatial_type="avg", with_avg_pool=False,
                      temporal_feature_size=1, spatial_feature_size=1, dropout_ratio=dropout, in_channels=2048,
                      init_std=0.01, num_classes=400, fcn_testing=fcn),
        fcn_testing=fcn,
        module_cfg=dict(type="MVF", n_segment=T, alpha=0.125, mvf_freq=(0, 0, 1, 1), mode="THW"))


def stage_stats(a):
    a = a.astype(np.float64).ravel()
    idx = np.linspace(0, a.size - 1, 16).astype(np.int64)
    return np.concatenate([[a.mean(), np.sqrt((a * a).mean()), np.abs(a).max()], a[idx]]).astype(np.float64)


def run_stages(model, imgs):
    """Stage outputs of the reference backbone via forward hooks (maxpool, layer1..4)."""
    feats = {}
    hooks = []
    bb = model.backbone
    for nm in ("maxpool", "layer1", "layer2", "layer3", "layer4"):
        hooks.append(getattr(bb, nm).register_forward_hook(
            lambda mod, inp, outp, nm=nm: feats.__setitem__(nm, t2n(outp))))
    return feats, hooks


def gen_nets(build_recognizer):
    out = {}

    # ---- (iii) config C1: R50 4x16, N=2, 224^2: eval logits, train loss + grads + one SGD step
    T, N, S = 4, 2, 224
    model = quiet(build_recognizer, model_cfg(50, T, dropout=0.0), None, dict(average_clips=None))
    load_synth(model, "r50/")
    sd = model.state_dict()
    out["struct/r50/n_params"] = np.array(sum(p.numel() for p in model.parameters()))
    out["struct/r50/n_mvf"] = np.array(sum(1 for m in model.modules() if type(m).__name__ == "MVF"))
    out["struct/r50/keys"] = np.array(sorted(sd.keys()))
    out["struct/r50/shapes"] = np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())])
    imgs = torch.from_numpy(synth.synth_clip_batch(N, T, S, S))
    labels = torch.from_numpy(synth.synth_labels(N))
    model.eval()
    feats, hooks = run_stages(model, imgs)
    with torch.no_grad():
        logits = model(imgs, None, return_loss=False, return_numpy=False)
    for h in hooks:
        h.remove()
    out["c1/eval/logits"] = t2n(logits)
    for k, v in feats.items():
        out["c1/eval/stage/" + k] = stage_stats(v)
    model.test_cfg = dict(average_clips="prob")
    with torch.no_grad():
        out["c1/eval/prob"] = t2n(model(imgs, None, return_loss=False, return_numpy=False))
    model.test_cfg = dict(average_clips="score")
    with torch.no_grad():
        out["c1/eval/score"] = t2n(model(imgs, None, return_loss=False, return_numpy=False))
    model.test_cfg = dict(average_clips=None)

    # train step (a8 + a13): loss, grads, clip, SGD-nesterov
    model.train()
    feats, hooks = run_stages(model, imgs)
    opt = torch.optim.SGD(model.parameters(), lr=0.015, momentum=0.9, weight_decay=1e-4, nesterov=True)
    for it in range(2):
        opt.zero_grad()
        losses = model(imgs, labels, return_loss=True)
        loss = losses["loss_cls"]
        loss.backward()
        if it == 0:
            for h in hooks:
                h.remove()
            out["c1/train/loss"] = t2n(loss)
            for k, v in feats.items():
                out["c1/train/stage/" + k] = stage_stats(v)
            gn = {}
            for pn, p in model.named_parameters():
                gn[pn] = float(p.grad.double().norm())
            names = sorted(gn)
            out["c1/train/grad_names"] = np.array(names)
            out["c1/train/grad_norms"] = np.array([gn[n] for n in names])
            for pn in ("backbone.conv1.weight", "backbone.layer3.0.conv1.shift_conv.weight",
                       "backbone.layer3.0.conv1.h_conv.weight", "backbone.layer3.0.conv1.w_conv.weight",
                       "backbone.layer3.0.conv1.bn.weight", "backbone.layer3.0.conv1.bn.bias",
                       "backbone.layer4.2.conv1.shift_conv.weight", "backbone.layer4.2.bn3.weight",
                       "cls_head.new_fc.bias"):
                out["c1/train/grad/" + pn] = t2n(dict(model.named_parameters())[pn].grad)
        total = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=40, norm_type=2)
        out["c1/train/total_norm/%d" % it] = t2n(total)
        opt.step()
        out["c1/train/loss/%d" % it] = t2n(loss)
    # post-2-step parameter/buffer samples
    sd = model.state_dict()
    for k in ("backbone.conv1.weight", "backbone.layer3.0.conv1.shift_conv.weight", "backbone.layer3.0.conv1.bn.running_mean",
              "backbone.layer3.0.conv1.bn.running_var", "backbone.layer3.0.conv1.bn.num_batches_tracked",
              "backbone.layer4.2.bn3.running_var", "backbone.bn1.running_mean", "cls_head.new_fc.bias",
              "backbone.layer4.2.conv1.net.weight"):
        a = t2n(sd[k]).ravel()
        out["c1/train/after2/" + k] = a[: min(a.size, 512)]

    # ---- (iv) R50 8x8 N=1 and R101 16x4 N=1 eval logits (224^2)
    for depth, T2, tag in ((50, 8, "r50_t8"), (101, 16, "r101_t16")):
        m2 = quiet(build_recognizer, model_cfg(depth, T2), None, dict(average_clips=None))
        load_synth(m2, "r%d/" % depth)
        m2.eval()
        im2 = torch.from_numpy(synth.synth_clip_batch(1, T2, 224, 224, seed=depth))
        with torch.no_grad():
            out[tag + "/eval/logits"] = t2n(m2(im2, None, return_loss=False, return_numpy=False))
        if depth == 101:
            out["struct/r101/n_params"] = np.array(sum(p.numel() for p in m2.parameters()))
            out["struct/r101/n_mvf"] = np.array(sum(1 for m in m2.modules() if type(m).__name__ == "MVF"))

    # ---- (v) fcn_testing path: 1 video = 3 crops x 2 clips x T frames at 128^2, average_clips='prob'
    T3 = 4
    m3 = quiet(build_recognizer, model_cfg(50, T3, fcn=True), None, dict(average_clips="prob"))
    load_synth(m3, "r50/")
    m3.eval()
    # neutralise the .cuda() in the lazily-built Conv3d (tsn_clshead.py:101-110)
    cls = nn.Conv3d(2048, 400, 1, 1, 0)
    cls.load_state_dict({"weight": m3.cls_head.new_fc.weight.detach()[:, :, None, None, None],
                         "bias": m3.cls_head.new_fc.bias.detach()})
    m3.cls_head.new_cls = cls
    im3 = torch.from_numpy(synth.synth_tensor("fcn_video", (1, 3 * 2 * T3, 3, 128, 128)))
    with torch.no_grad():
        out["fcn/prob"] = t2n(m3(im3, None, return_loss=False, return_numpy=False))
        m3.test_cfg = dict(average_clips=None)
        out["fcn/scores"] = t2n(m3(im3, None, return_loss=False, return_numpy=False))

    np.savez_compressed(os.path.join(HERE, "net_cases.npz"), **out)
    print("net_cases.npz: %d arrays" % len(out))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    build_recognizer, MVF, Bottleneck = _import_reference()
    which = sys.argv[1:] or ["mvf", "block", "nets"]
    if "mvf" in which:
        gen_mvf_cases(MVF)
    if "block" in which:
        gen_block(MVF, Bottleneck)
    if "nets" in which:
        gen_nets(build_recognizer)


if __name__ == "__main__":
    main()
