#!/usr/bin/env python3
"""Golden vectors for training with norm_eval=True + norm_frozen=True (every BatchNorm in eval mode AND its weight / bias excluded
from training: requires_grad False in SCATTERED places of model.parameters(), reference resnet.py:496-505), from the REFERENCE
itself on CPU.  Run in the build container: python tests/golden/make_normfrozen_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from mvfnet_amd import synth  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(os.cpu_count())
build_recognizer, MVF, Bottleneck = mg._import_reference()
T, N, S = 4, 2, 96
cfg = mg.model_cfg(50, T, dropout=0.0)
cfg["backbone"]["norm_eval"] = True
cfg["backbone"]["norm_frozen"] = True
model = mg.quiet(build_recognizer, cfg, None, dict(average_clips=None))
mg.load_synth(model, "r50/")
imgs = torch.from_numpy(synth.synth_clip_batch(N, T, S, S, seed=79))
labels = torch.from_numpy(synth.synth_labels(N))
model.train()
frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
assert frozen and all((".bn" in n or "downsample.1" in n) for n in frozen), frozen[:5]
out = {"frozen_names": np.array(frozen)}
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=0.015, momentum=0.9, weight_decay=1e-4, nesterov=True)
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
for it in range(2):
    opt.zero_grad()
    loss = model(imgs, labels, return_loss=True)["loss_cls"]
    loss.backward()
    if it == 0:
        gn = {pn: float(p.grad.double().norm()) for pn, p in model.named_parameters() if p.grad is not None}
        names = sorted(gn)
        out["grad_names"] = np.array(names)
        out["grad_norms"] = np.array([gn[n] for n in names])
    out["total_norm/%d" % it] = mg.t2n(torch.nn.utils.clip_grad_norm_(params, max_norm=5.0, norm_type=2))      # 5 instead of the config's 40: the clip must be ACTIVE for the excluded gradients to matter
    opt.step()
    out["loss/%d" % it] = mg.t2n(loss)
sd = model.state_dict()
for k in frozen:
    assert torch.equal(sd[k], sd0[k]), k
for k in ("backbone.conv1.weight", "backbone.layer2.0.conv1.weight", "backbone.layer3.0.conv1.net.weight", "backbone.layer3.0.conv1.shift_conv.weight",
          "backbone.layer4.2.conv3.weight", "cls_head.new_fc.bias"):
    a = mg.t2n(sd[k]).ravel()
    out["after2/" + k] = a[: min(a.size, 512)]
np.savez_compressed(os.path.join(HERE, "normfrozen_cases.npz"), **out)
print("normfrozen_cases.npz: %d frozen params, loss" % len(frozen), out["loss/0"], out["loss/1"], "total_norm", out["total_norm/0"], out["total_norm/1"])
