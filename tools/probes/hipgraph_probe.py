import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvfnet_amd import synth
import mvfnet_amd
m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, 8, dropout_ratio=0.5), None, dict(average_clips=None))
sd = m.state_dict()
vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True)
m = m.cuda().train()
eng = m.train_engine(dtype=torch.bfloat16)
CLIPS = int(sys.argv[1]) if len(sys.argv) > 1 else 32
imgs = torch.from_numpy(synth.synth_clip_batch(CLIPS, 8, 224, 224)).cuda()
labels = torch.from_numpy(synth.synth_labels(CLIPS)).cuda()
print('clips', CLIPS)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager  ms/step", timeit(lambda: eng.train_step(imgs, labels)))
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): eng.train_step(imgs, labels)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss = eng.train_step(imgs, labels)
    torch.cuda.synchronize()
    print("graph  ms/step", timeit(lambda: g.replay()), "loss", float(loss))
except Exception as e:
    print("capture failed:", repr(e)[:400])
