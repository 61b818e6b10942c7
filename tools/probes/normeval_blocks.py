"""norm_eval train step (fp32): save every block's outgoing gradient dx; run once per MVF_F32_X3 setting, then `diff a.npz b.npz`."""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
import numpy as np, torch, mvfnet_amd
from mvfnet_amd import synth
if sys.argv[1] == "diff":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        if len(sys.argv) > 4 and not k.startswith(sys.argv[4]):
            continue
        d = np.abs(a[k].astype(np.float64) - b[k]).max() / max(np.abs(b[k]).max(), 1e-30)
        l2 = np.linalg.norm(a[k].astype(np.float64) - b[k]) / max(np.linalg.norm(b[k]), 1e-30)
        print("%-14s max/max %.2e  rel L2 %.2e  shape %s" % (k, d, l2, a[k].shape))
    sys.exit(0)
cfg = mvfnet_amd.mvfnet_config(50, 4, dropout_ratio=0.0); cfg["backbone"]["norm_eval"] = True
m = mvfnet_amd.build_recognizer(cfg, None, dict(average_clips=None))
sd = m.state_dict(); vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True); m = m.cuda().train()
eng = m.train_engine(); eng.keep_io = True
imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 96, 96, seed=77)).cuda(); labels = torch.from_numpy(synth.synth_labels(2)).cuda()
eng.forward(imgs, labels); eng.backward(); torch.cuda.synchronize()
out = {}
for i, blk in enumerate(eng.blocks):
    out["b%02d_out" % i] = blk.io["out"].float().cpu().numpy()
    out["b%02d_g" % i] = blk.io["g"].float().cpu().numpy()
    out["b%02d_dx" % i] = blk.io["dx"].float().cpu().numpy()
blk = eng.blocks[13]
def find(obj, tag):
    for (key, shape, dt), t in eng._bufs.items():
        if key == (id(obj), tag):
            return t.float().cpu().numpy()
    return np.zeros(1, dtype=np.float32)
for nm, obj in (("c1", blk.c1), ("c2", blk.c2), ("c3", blk.c3), ("cd", blk.cd)):
    out["x13_%s_z" % nm] = find(obj, "z"); out["x13_%s_dx" % nm] = find(obj, "dx")
for nm, obj in (("b1", blk.b1), ("b2", blk.b2), ("b3", blk.b3), ("bd", blk.bd)):
    out["x13_%s_dz" % nm] = find(obj, "dz"); out["x13_%s_apply" % nm] = find(obj, "apply")
np.savez(sys.argv[1], **out)
