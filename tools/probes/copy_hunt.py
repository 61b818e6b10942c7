"""Who launches the ~60 tiny device copies per training step?  torch.profiler with stacks around one train_step."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mvfnet_amd
from mvfnet_amd import synth
m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, 8), None, dict(average_clips=None)).cuda().train()
eng = m.train_engine(dtype=torch.bfloat16)
imgs = torch.from_numpy(synth.synth_clip_batch(8, 8, 224, 224)).cuda()
labels = torch.randint(0, 400, (8, 1), device="cuda")
for _ in range(2): eng.train_step(imgs, labels)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    eng.train_step(imgs, labels)
    torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    n = e.name
    if "copy" in n.lower() or "Memcpy" in n:
        c[(n[:60], tuple(e.stack[:3]) if e.stack else ())] += 1
for k, v in c.most_common(12): print(v, k)
