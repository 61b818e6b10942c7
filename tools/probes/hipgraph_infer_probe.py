"""Is bf16 inference host-bound?  Capture the eval forward in a HIP graph and compare with eager launches."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvfnet_amd import synth
import mvfnet_amd
for dt in (torch.bfloat16, torch.float32):
    m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, 8), None, dict(average_clips=None))
    sd = m.state_dict()
    vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True)
    m.backbone.engine_dtype = dt
    m = m.cuda().eval()
    imgs = torch.from_numpy(synth.synth_clip_batch(32, 8, 224, 224)).cuda()
    fwd = lambda: m(imgs, None, return_loss=False, return_numpy=False)
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter(); fwd(); host = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
    print(str(dt)[6:], "eager ms/step %.3f (host enqueue of one step %.3f ms)" % (timeit(fwd), host))
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fwd()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            out = fwd()
        torch.cuda.synchronize()
        print(str(dt)[6:], "graph ms/step %.3f" % timeit(lambda: g.replay()))
    except Exception as e:
        print("capture failed:", repr(e)[:300])
