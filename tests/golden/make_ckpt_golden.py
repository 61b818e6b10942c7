#!/usr/bin/env python3
"""Generate tests/golden/ref_ckpt_block.pth: a checkpoint WRITTEN BY THE REFERENCE's own save_checkpoint
(codes/utils/checkpoint.py:235-265) for a small model, after two optimizer steps of the optimizer its own build_optimizer
(codes/core/train.py:79-156) makes from the shipped config's optimizer dict -- once plain, once with paramwise_options.

Run ONCE in the build container (where /root/reference exists):   python tests/golden/make_ckpt_golden.py

The model is one reference Bottleneck with an MVF on conv1 (the BLOCK_CASES "l3_like" shape, synth weights): a few thousand
parameters, so the fixture stays small.  What is stored is DATA: the checkpoint file the reference wrote ({'meta', 'state_dict',
'optimizer'} with torch.optim.SGD's state) plus, in a side .npz, the inputs' seeds and the parameter values after the two steps.
tests/test_runner_cpu.py / tests/test_train_gpu.py check that this repo reads it and writes the same layout back."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (stubs + reference import helpers)
from cases import BLOCK_CASES  # noqa: E402
from mvfnet_amd import synth  # noqa: E402


def main():
    build_recognizer, MVF, Bottleneck = MG._import_reference()
    import mmcv
    mmcv.mkdir_or_exist = lambda d: os.makedirs(d, exist_ok=True) if d else None
    import mmcv.runner
    mmcv.runner.obj_from_dict = lambda info, parent, default_args: getattr(parent, dict(info).pop("type"))(
        **{**{k: v for k, v in info.items() if k != "type"}, **default_args})
    from codes.core import train as ref_train
    ref_train.obj_from_dict = mmcv.runner.obj_from_dict
    from codes.utils.checkpoint import save_checkpoint
    import torch.nn as nn
    name = "l3_like"
    N, T, Cin, planes, H, W, stride = BLOCK_CASES[name]
    out = {}
    for tag, pw in (("plain", None), ("paramwise", dict(bias_lr_mult=2.0, bias_decay_mult=0.0, norm_decay_mult=0.0))):
        down = None
        if stride != 1 or Cin != planes * 4:
            down = nn.Sequential(nn.Conv2d(Cin, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        blk = MG.quiet(Bottleneck, Cin, planes, stride, 1, down)
        blk.conv1 = MG.quiet(MVF, blk.conv1, T, Cin, 0.125, True, False, "THW")
        MG.load_synth(blk, "block/%s/" % name)
        blk.train()
        ocfg = dict(type="SGD", lr=0.015, momentum=0.9, weight_decay=0.0001, nesterov=True)        # the shipped config's optimizer (cfg:152-153)
        if pw:
            ocfg["paramwise_options"] = pw
        opt = ref_train.build_optimizer(blk, ocfg)
        x = torch.from_numpy(synth.synth_tensor("block_x/" + name, (N * T, Cin, H, W)))
        for step in range(2):
            opt.zero_grad()
            y = blk(x)
            dy = torch.from_numpy(synth.synth_tensor("block_dy/" + name, tuple(y.shape), seed=step))
            (y * dy).sum().backward()
            torch.nn.utils.clip_grad_norm_(blk.parameters(), max_norm=40, norm_type=2)
            opt.step()
            if step == 0:
                for k, v in blk.state_dict().items():
                    out["%s/step1/%s" % (tag, k)] = MG.t2n(v)
        path = os.path.join(HERE, "ref_ckpt_block_%s.pth" % tag)
        save_checkpoint(blk, path, optimizer=opt, meta=dict(epoch=3, iter=14))
        ck = torch.load(path, weights_only=False)
        ck["meta"].pop("time", None)
        torch.save(ck, path)                 # same content minus the wall-clock stamp (reproducible bytes)
        for k, v in blk.state_dict().items():
            out["%s/%s" % (tag, k)] = MG.t2n(v)
        print(tag, "groups", len(ck["optimizer"]["param_groups"]), "state entries", len(ck["optimizer"]["state"]), os.path.getsize(path), "bytes")
    # torch optimizer state is keyed by POSITION in model.parameters(): pin the reference's parameter order for the full models
    cfg = __import__("mvfnet_amd").mvfnet_config
    for depth, t in ((50, 8), (101, 16)):
        ref = MG.quiet(build_recognizer, cfg(depth, t), None, dict(average_clips=None))
        out["r%d/param_order" % depth] = np.array([k for k, _ in ref.named_parameters()])
    np.savez_compressed(os.path.join(HERE, "ref_ckpt_block.npz"), **out)


if __name__ == "__main__":
    main()
