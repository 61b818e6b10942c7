"""How long does the HOST need to enqueue one training step (no GPU sync inside)? If close to the GPU step time, the
engine is launch-bound and needs a C++ plan / hipGraph."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
bench.T_FRAMES = 8
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
clips = int(sys.argv[2]) if len(sys.argv) > 2 else 32          # [r5] 12 = the reference's videos_per_gpu (configs/MVFNet/K400/*_r50_dense.py:121-123)
m = bench.build_model(50, dtype, True)
eng = m.train_engine(dtype=torch.bfloat16 if dtype == "bf16" else torch.float32)
imgs = torch.randn(clips, 8, 3, 224, 224, device="cuda"); labels = torch.randint(0, 400, (clips, 1), device="cuda")
for _ in range(8): eng.train_step(imgs, labels)          # ([r6] past the launch plan's two eager + two recorded steps)
torch.cuda.synchronize()
N = 4            # few enough steps that the launch queues never fill (with 20 the host blocks on the queue and measures the GPU)
enq, tot = [], []
for _ in range(7):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): eng.train_step(imgs, labels)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) / N * 1e3); tot.append((t2 - t0) / N * 1e3)
enq.sort(); tot.sort()
t0, t1, t2 = 0.0, enq[len(enq) // 2] * N / 1e3, tot[len(tot) // 2] * N / 1e3          # medians of seven bursts
plans = [s_["plan"] is not None for s_ in getattr(eng, "_plans", {}).values()]
print("clips %d " % clips, end=""); print("dtype %s: host enqueue %.2f ms/step, total %.2f ms/step (launch plan: %s)" % (dtype, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3, plans))
if len(sys.argv) > 3 and sys.argv[3] == "noprofile":
    sys.exit(0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): eng.train_step(imgs, labels)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
