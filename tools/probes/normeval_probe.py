import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, mvfnet_amd
from mvfnet_amd import synth
from helpers import golden, rel_err
g = golden("normeval_cases.npz")
cfg = mvfnet_amd.mvfnet_config(50, 4, dropout_ratio=0.0); cfg["backbone"]["norm_eval"] = True
m = mvfnet_amd.build_recognizer(cfg, None, dict(average_clips=None))
sd = m.state_dict(); vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True); m = m.cuda().train()
eng = m.train_engine()
imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 96, 96, seed=77)).cuda(); labels = torch.from_numpy(synth.synth_labels(2)).cuda()
loss = eng.forward(imgs, labels); eng.backward()
params = dict(m.named_parameters())
errs = sorted(((rel_err(eng.grad_of(params[k[5:]]).cpu().numpy(), g[k]), k) for k in g.files if k.startswith("grad/")), reverse=True)
print("X3=%s loss rel %.2e" % (os.environ.get("MVF_F32_X3", "1"), abs(float(loss) - float(g["loss/0"])) / float(g["loss/0"])), [(round(e, 5), k[5:]) for e, k in errs[:6]])
g64 = golden("normeval_fp64.npz")
e64 = sorted(((rel_err(eng.grad_of(params[k[5:]]).cpu().numpy().astype(np.float64), g64[k]), k) for k in g64.files if k.startswith("grad/")), reverse=True)
print("  vs the reference in fp64:", [(round(float(e), 6), k[5:]) for e, k in e64])
nerr = sorted(((abs(float(eng.grad_of(params[n]).double().norm()) - r) / max(r, 1e-6), n) for n, r in zip(list(g["grad_names"]), g["grad_norms"])), reverse=True)
print("  norms", [(round(e, 5), n) for e, n in nerr[:5]])
