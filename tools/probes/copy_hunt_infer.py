"""Device copies per bf16 inference step, with the Python stacks that launch them (torch.profiler)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mvfnet_amd
m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, 8), None, dict(average_clips=None))
m.backbone.engine_dtype = torch.bfloat16
m = m.cuda().eval()
m.backbone.engine().streams = 2
imgs = torch.randn(32, 8, 3, 224, 224, device="cuda")
for _ in range(3): m(imgs, None, return_loss=False, return_numpy=False)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m(imgs, None, return_loss=False, return_numpy=False)
    torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    n = e.name
    if "copy" in n.lower() or "Memcpy" in n or "memset" in n.lower():
        c[(n[:50], tuple(s[-60:] for s in (e.stack[:2] if e.stack else ())))] += 1
for k, v in c.most_common(10): print(v, k)
