#!/usr/bin/env python3
"""The frozen-statistics (norm_eval=True, reference resnet.py:496-505) training step of make_normeval_fp64_golden.py on SEVERAL inputs, each run by
the REFERENCE itself in double precision -- the gradient every fp32 path approximates.  One input cannot carry a tight end-to-end bound for an
fp32 path: a 2-clip 96^2 batch puts ~1e7 values through ReLU / hard-swish kinks, the smallest |pre-activation| is ~1e-7 of the tensor's scale on
EVERY input (recorded below as `margin/...`), and in layer4's 72 x 512 tensors one flipped decision is ~1e-2 of that gradient's norm.  Several
inputs separate the two things: an accuracy regression moves the gradients on every input, sign luck at a kink moves them on one.
tests/test_train_gpu.py::test_norm_eval_default_fp32_path_tight_over_seeds asserts the tight bound on all but (at most) one of them.
    python tests/golden/make_normeval_seeds_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from mvfnet_amd import synth  # noqa: E402

SEEDS = (77, 101, 202, 303, 404, 505)
NAMED = ("backbone.layer3.0.conv1.bn.weight", "backbone.layer3.0.conv1.shift_conv.weight", "backbone.layer1.0.bn3.bias", "backbone.bn1.weight",
         "backbone.layer4.2.bn2.weight", "backbone.layer3.4.conv1.bn.weight", "backbone.layer1.0.conv1.weight", "backbone.layer3.2.conv2.weight",
         "backbone.layer4.0.conv2.weight", "backbone.layer2.1.conv3.weight")

SAMPLE = 16384


def sample_indices(size):
    return np.sort(np.random.RandomState(size % 65521).choice(size, SAMPLE, replace=False))


torch.manual_seed(0)
torch.set_num_threads(os.cpu_count())
build_recognizer, MVF, Bottleneck = mg._import_reference()
T, N, S = 4, 2, 96
cfg = mg.model_cfg(50, T, dropout=0.0)
cfg["backbone"]["norm_eval"] = True
model = mg.quiet(build_recognizer, cfg, None, dict(average_clips=None))
mg.load_synth(model, "r50/")
model = model.double()
model.train()

# smallest |ReLU input| relative to that tensor's largest value, per residual stage (forward pre-hooks on the reference's own ReLU modules)
margins = {}


def _hook(stage):
    def pre(mod, args):
        x = args[0].detach()
        margins[stage] = min(margins.get(stage, 1.0), float(x.abs().min() / x.abs().max()))
    return pre


for name in ("layer1", "layer2", "layer3", "layer4"):
    for blk in getattr(model.backbone, name):
        blk.relu.register_forward_pre_hook(_hook(name))

out = {"seeds": np.array(SEEDS)}
labels = torch.from_numpy(synth.synth_labels(N))
for seed in SEEDS:
    margins.clear()
    model.zero_grad()
    imgs = torch.from_numpy(synth.synth_clip_batch(N, T, S, S, seed=seed)).double()
    loss = model(imgs, labels, return_loss=True)["loss_cls"]
    loss.backward()
    params = dict(model.named_parameters())
    names = sorted(params)
    tag = "s%d/" % seed
    out[tag + "loss"] = np.array(float(loss))
    if "grad_names" not in out:
        out["grad_names"] = np.array(names)
    out[tag + "grad_norms"] = np.array([float(params[n].grad.norm()) for n in names])
    for pn in NAMED:
        gr = params[pn].grad.numpy().ravel()
        if gr.size > SAMPLE:          # large conv weights: a fixed random sample of their elements (the test draws the same indices)
            gr = gr[sample_indices(gr.size)]
        out[tag + "grad/" + pn] = gr.copy()
    for k, v in margins.items():
        out[tag + "margin/" + k] = np.array(v)
    print("seed %d: loss %.12f  smallest |relu input| / max per stage: %s" % (seed, float(loss), {k: "%.1e" % v for k, v in sorted(margins.items())}))
np.savez_compressed(os.path.join(HERE, "normeval_seeds_fp64.npz"), **out)
print("normeval_seeds_fp64.npz written (%d seeds)" % len(SEEDS))
