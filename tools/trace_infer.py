"""Only inference passes, for a clean rocprofv3 kernel trace:  python tools/trace_infer.py [bf16|f32] [passes] [streams]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.T_FRAMES = 8
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m = bench.build_model(50, dtype, False)
m.backbone.engine().streams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
imgs = torch.randn(32, 8, 3, 224, 224, device="cuda")
with torch.no_grad():
    for _ in range(4): m(imgs, None, return_loss=False, return_numpy=False)
    torch.cuda.synchronize()
    for _ in range(n): m(imgs, None, return_loss=False, return_numpy=False)
torch.cuda.synchronize()
