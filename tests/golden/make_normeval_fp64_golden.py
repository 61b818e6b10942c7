#!/usr/bin/env python3
"""The norm_eval training step of make_normeval_golden.py once more, by the REFERENCE itself in DOUBLE precision (model.double(), fp64 inputs):
the gradient every fp32 implementation -- the reference's own fp32 run included -- approximates.  Used to tell an accuracy difference between
two fp32 conv paths (the fp32 MFMA, the 3-term bf16 split on the bf16 matrix cores) from their different luck at the ReLU / hard-swish kinks.
    python tests/golden/make_normeval_fp64_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from mvfnet_amd import synth  # noqa: E402

torch.manual_seed(0)
build_recognizer, MVF, Bottleneck = mg._import_reference()
T, N, S = 4, 2, 96
cfg = mg.model_cfg(50, T, dropout=0.0)
cfg["backbone"]["norm_eval"] = True
model = mg.quiet(build_recognizer, cfg, None, dict(average_clips=None))
mg.load_synth(model, "r50/")
model = model.double()
imgs = torch.from_numpy(synth.synth_clip_batch(N, T, S, S, seed=77)).double()
labels = torch.from_numpy(synth.synth_labels(N))
model.train()
loss = model(imgs, labels, return_loss=True)["loss_cls"]
loss.backward()
out = {"loss": np.array(float(loss))}
names = sorted(pn for pn, _ in model.named_parameters())
params = dict(model.named_parameters())
out["grad_names"] = np.array(names)
out["grad_norms"] = np.array([float(params[n].grad.norm()) for n in names])
for pn in ("backbone.layer3.0.conv1.bn.weight", "backbone.layer3.0.conv1.shift_conv.weight", "backbone.layer1.0.bn3.bias", "backbone.bn1.weight",
           "backbone.layer4.2.bn2.weight", "backbone.layer3.4.conv1.bn.weight", "backbone.layer1.0.conv1.weight", "backbone.layer3.2.conv2.weight"):
    out["grad/" + pn] = params[pn].grad.numpy().copy()
np.savez_compressed(os.path.join(HERE, "normeval_fp64.npz"), **out)
print("normeval_fp64.npz: loss %.12f" % float(loss))
