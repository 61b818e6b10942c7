"""Only training steps (no roofline pass), for a clean rocprofv3 kernel trace:  python tools/trace_steps.py [bf16|f32] [steps] [clips]"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.T_FRAMES = 8
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
clips = int(sys.argv[3]) if len(sys.argv) > 3 else 32
m = bench.build_model(50, dtype, True)
eng = m.train_engine(dtype=torch.bfloat16 if dtype == "bf16" else torch.float32)
imgs = torch.randn(clips, 8, 3, 224, 224, device="cuda"); labels = torch.randint(0, 400, (clips, 1), device="cuda")
for _ in range(6): eng.train_step(imgs, labels)          # (the launch plan replays from the fifth step on)
torch.cuda.synchronize()
print("TRACE_BEGIN"); sys.stdout.flush()
for _ in range(steps): eng.train_step(imgs, labels)
torch.cuda.synchronize()
