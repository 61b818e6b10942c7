import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import mvfnet_amd
from mvfnet_amd import synth
from oracle import net_torch
T = 4
torch.set_num_threads(16)
m0 = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, T), None, dict(average_clips=None))
sd = m0.state_dict()
vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
cpu_sd = {k: torch.from_numpy(vals["r50/" + k]) for k in sd}
for seed in (3, 4, 5, 6):
    for clips in (2, 4):
        imgs = torch.from_numpy(synth.synth_clip_batch(clips, T, 64, 64, seed=seed))
        labels = torch.from_numpy(synth.synth_labels(clips, seed=seed))
        with torch.no_grad():
            ref = float(net_torch.forward_train(imgs, labels, {k: v.clone() for k, v in cpu_sd.items()}, 50))
        m2 = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, T), None, dict(average_clips=None))
        m2.load_state_dict(cpu_sd)
        m2 = m2.cuda().train()
        e = m2.train_engine(dtype=torch.bfloat16)
        e.dropout = 0.0
        l = float(e.train_step(imgs.cuda(), labels.cuda()))
        print("seed %d clips %d: ref %.5f bf16 %.5f rel %.2e" % (seed, clips, ref, l, abs(l - ref) / abs(ref)), flush=True)
