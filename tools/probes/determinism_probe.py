"""Race detector: the training step has no atomics, so N steps with the side-stream overlap must be BIT-identical to N steps
with every kernel on one stream.  python tools/probes/determinism_probe.py [steps] [dtype]"""
import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, os.path.join(R, "tests")]
from mvfnet_amd import synth
import mvfnet_amd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
def model():
    m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, 8, dropout_ratio=0.5), None, dict(average_clips=None))
    sd = m.state_dict()
    vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True)
    return m.cuda().train()
imgs = torch.from_numpy(synth.synth_clip_batch(8, 8, 224, 224)).cuda()
labels = torch.from_numpy(synth.synth_labels(8)).cuda()
res = []
for overlap in (True, False, True):
    torch.manual_seed(0)
    m = model(); eng = m.train_engine(dtype=dt); eng.overlap_wgrad = overlap
    losses = [float(eng.train_step(imgs, labels)) for _ in range(steps)]
    torch.cuda.synchronize()
    res.append((losses, eng.flat_params.clone(), torch.cat([b.flatten().float() for b in m.buffers()])))
    print("overlap", overlap, "loss[0]", losses[0], "loss[-1]", losses[-1], "finite", all(map(lambda v: v == v, losses)))
for i in (1, 2):
    print("run0 vs run%d: losses equal %s, params equal %s, buffers equal %s" % (i, res[0][0] == res[i][0], torch.equal(res[0][1], res[i][1]), torch.equal(res[0][2], res[i][2])))
