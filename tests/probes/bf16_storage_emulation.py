"""CPU: how far does bf16 STORAGE alone move the oracle's stage outputs?  (conv inputs/weights/outputs and BN outputs rounded
to bf16, fp32 arithmetic otherwise) -- the yardstick for the engine's bf16 deviation on the same synthetic network."""
import sys, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, os.path.join(R, "tests")]
from helpers import rel_l2
from mvfnet_amd import synth, arch
from oracle import net_torch
import torch.nn.functional as F
r = lambda t: t.bfloat16().float()
real_conv, real_bn, real_relu = F.conv2d, F.batch_norm, F.relu
def run(shape, emulate):
    b,t,h,w = shape
    shp = arch.state_dict_shapes(50)
    vals = synth.synth_state_dict({"r50/" + k: v for k, v in shp.items()})
    sd = {k: torch.from_numpy(vals["r50/" + k]) for k in shp}
    if emulate:
        F.conv2d = lambda x, wt, *a, **k: r(real_conv(r(x), r(wt), *a, **k))
        F.batch_norm = lambda x, *a, **k: real_bn(x, *a, **k)
        F.relu = lambda x: r(real_relu(x))
    try:
        st = {}
        with torch.no_grad():
            loss = net_torch.forward_train(torch.from_numpy(synth.synth_clip_batch(b,t,h,w)), torch.from_numpy(synth.synth_labels(b)), sd, depth=50, T=t, new_buffers={}, stages=st)
    finally:
        F.conv2d, F.batch_norm, F.relu = real_conv, real_bn, real_relu
    return float(loss), st
for shape in [(3,3,80,112)]:
    l0, s0 = run(shape, False)
    l1, s1 = run(shape, True)
    print(shape, "loss fp32", l0, "bf16-storage emulation", l1, {k: round(rel_l2(s1[k].numpy(), s0[k].numpy()), 4) for k in s0})
