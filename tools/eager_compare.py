#!/usr/bin/env python3
"""The ">= 1.5x PyTorch-ROCm" comparator of BASELINE.json: the SAME train step (oracle/net_torch.py's restatement of the reference
graph: batch-stat BN, MVF closed form, head, CE, clip + SGD-nesterov) executed by PyTorch-ROCm eager on this GPU.

    python tools/eager_compare.py [--dtype f32|bf16] [--clips 32] [--steps 5] [--benchmark 0|1]

bf16 = torch.autocast(bfloat16) + channels_last activations / weights (the like-for-like comparator of the bf16 engine);
f32 = the reference's shipped precision.  cudnn.benchmark=False (MIOpen immediate mode: no exhaustive kernel search, which takes
3-25 minutes on a fresh box) unless --benchmark 1.  Prints one JSON line.  Test/bench infrastructure: imports oracle/."""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--benchmark", type=int, default=0)
    a = ap.parse_args()
    from mvfnet_amd import synth
    from mvfnet_amd.arch import state_dict_shapes
    from oracle import net_torch
    torch.backends.cudnn.benchmark = bool(a.benchmark)
    shp = state_dict_shapes(a.depth)
    pre = "r%d/" % a.depth
    vals = synth.synth_state_dict({pre + k: v for k, v in shp.items()})
    bf16 = a.dtype == "bf16"
    sd = {}
    for k in shp:
        t = torch.from_numpy(vals[pre + k]).cuda()
        if bf16 and t.dim() == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        if t.dtype == torch.float32 and "running" not in k:
            t.requires_grad_(True)
        sd[k] = t
    imgs = torch.randn(a.clips, a.frames, 3, a.size, a.size, device="cuda")
    if bf16:
        imgs = imgs.reshape(-1, 3, a.size, a.size).contiguous(memory_format=torch.channels_last).reshape(a.clips, a.frames, 3, a.size, a.size)
    labels = torch.randint(0, 400, (a.clips, 1), device="cuda")
    mom = {}
    params = {k: v for k, v in sd.items() if v.requires_grad}

    def step():
        for p in params.values():
            p.grad = None
        nb = {}
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            loss = net_torch.forward_train(imgs, labels, sd, a.depth, T=a.frames, new_buffers=nb, dropout_ratio=0.5)
        loss.backward()
        with torch.no_grad():
            net_torch.sgd_nesterov_step(params, {k: v.grad for k, v in params.items()}, mom)
            for k, v in nb.items():
                sd[k] = v
        return loss

    t0 = time.perf_counter()
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t_warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"eager_clips_per_s": round(a.clips * a.steps / el, 2), "ms_per_step": round(el / a.steps * 1e3, 2), "dtype": a.dtype,
                      "mode": "autocast(bf16)+channels_last" if bf16 else "fp32", "cudnn_benchmark": bool(a.benchmark), "warmup_s": round(t_warm, 1),
                      "loss": float(loss), "clips": a.clips, "torch": torch.__version__}))


if __name__ == "__main__":
    main()
