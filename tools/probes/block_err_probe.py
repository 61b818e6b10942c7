"""Errors of one train-mode bottleneck (BlockTrainer, fp32) against the reference's golden block run, per tensor -- to compare the fp32-MFMA
convs (MVF_F32_X3=0) with the bf16x3 ones (default)."""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "tests", "golden")); sys.path.insert(0, root)
import numpy as np, torch, torch.nn as nn
from cases import BLOCK_CASES
from helpers import golden, rel_err
from mvfnet_amd import synth
from mvfnet_amd.backbones.resnet import Bottleneck
from mvfnet_amd.modules import MVF
from mvfnet_amd.train_engine import BlockTrainer
g = golden("block_cases.npz")
for name in sorted(BLOCK_CASES):
    N, T, Cin, planes, H, W, stride = BLOCK_CASES[name]
    down = None
    if stride != 1 or Cin != planes * 4:
        down = nn.Sequential(nn.Conv2d(Cin, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
    blk = Bottleneck(Cin, planes, stride, 1, down)
    blk.conv1 = MVF(blk.conv1, T, Cin, 0.125, True, False, "THW")
    sd = blk.state_dict(); pre = "block/%s/" % name
    vals = synth.synth_state_dict({pre + k: tuple(v.shape) for k, v in sd.items()})
    blk.load_state_dict({k: torch.from_numpy(vals[pre + k]) for k in sd}); blk = blk.cuda().train()
    tr = BlockTrainer(blk)
    x = torch.from_numpy(synth.synth_tensor("block_x/" + name, (N * T, Cin, H, W))).cuda()
    y = tr.forward(x)
    dy = torch.from_numpy(synth.synth_tensor("block_dy/" + name, tuple(y.shape))).cuda()
    dx = tr.backward(dy)
    ge = max((rel_err(tr.grad_of(p).cpu().numpy(), g[name + "/train/grad/" + pn]), pn) for pn, p in blk.named_parameters())
    print("X3=%s %-28s y %.2e  dx %.2e  worst grad %.2e (%s)" % (os.environ.get("MVF_F32_X3", "1"), name, rel_err(y.cpu().numpy(), g[name + "/train/y"]),
                                                                rel_err(dx.cpu().numpy(), g[name + "/train/dx"]), ge[0], ge[1]))
