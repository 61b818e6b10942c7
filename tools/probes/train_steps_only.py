"""8 plain bf16 train steps (no event brackets, no roofline pass) for timeline analysis under rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mvfnet_amd
from mvfnet_amd import synth
m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, 8), None, dict(average_clips=None)).cuda().train()
eng = m.train_engine(dtype=torch.bfloat16)
eng.dropout = 0.5
imgs = torch.randn(32, 8, 3, 224, 224, device="cuda")
labels = torch.randint(0, 400, (32, 1), device="cuda")
for _ in range(8):
    eng.train_step(imgs, labels)
torch.cuda.synchronize()
