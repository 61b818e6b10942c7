#!/usr/bin/env python3
"""Golden vectors for training with frozen BatchNorm statistics (backbone norm_eval=True, reference resnet.py:496-505), from the
REFERENCE itself on CPU (same stubs and synthetic-weight recipe as make_golden.py).  Run in the build container:
    python tests/golden/make_normeval_golden.py"""
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
model = mg.quiet(build_recognizer, cfg, None, dict(average_clips=None))
mg.load_synth(model, "r50/")
imgs = torch.from_numpy(synth.synth_clip_batch(N, T, S, S, seed=77))
labels = torch.from_numpy(synth.synth_labels(N))
model.train()
assert not any(m.training for m in model.backbone.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm))
out = {}
opt = torch.optim.SGD(model.parameters(), lr=0.015, momentum=0.9, weight_decay=1e-4, nesterov=True)
before = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
for it in range(2):
    opt.zero_grad()
    loss = model(imgs, labels, return_loss=True)["loss_cls"]
    loss.backward()
    if it == 0:
        gn = {pn: float(p.grad.double().norm()) for pn, p in model.named_parameters()}
        names = sorted(gn)
        out["grad_names"] = np.array(names)
        out["grad_norms"] = np.array([gn[n] for n in names])
        for pn in ("backbone.layer3.0.conv1.bn.weight", "backbone.layer3.0.conv1.shift_conv.weight", "backbone.layer1.0.bn3.bias",
                   "backbone.bn1.weight", "backbone.layer4.2.bn2.weight"):
            out["grad/" + pn] = mg.t2n(dict(model.named_parameters())[pn].grad)
    out["total_norm/%d" % it] = mg.t2n(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=40, norm_type=2))
    opt.step()
    out["loss/%d" % it] = mg.t2n(loss)
sd = model.state_dict()
assert all(torch.equal(sd[k], v) for k, v in before.items()), "frozen statistics moved"
for k in ("backbone.conv1.weight", "backbone.layer3.0.conv1.bn.weight", "backbone.layer4.2.bn3.bias", "cls_head.new_fc.bias"):
    a = mg.t2n(sd[k]).ravel()
    out["after2/" + k] = a[: min(a.size, 512)]
np.savez_compressed(os.path.join(HERE, "normeval_cases.npz"), **out)
print("normeval_cases.npz: loss", out["loss/0"], out["loss/1"], "total_norm", out["total_norm/0"])
