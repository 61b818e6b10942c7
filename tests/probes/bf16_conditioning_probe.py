import sys, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
from helpers import rel_l2
from mvfnet_amd import synth
import mvfnet_amd
from oracle import net_torch
def model(depth, T):
    m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(depth, T, dropout_ratio=0.0), None, dict(average_clips=None))
    sd = m.state_dict(); pre = "r%d/" % depth
    vals = synth.synth_state_dict({pre + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals[pre + k]) for k in sd}, strict=True)
    return m.cuda().train()
for shape in [(3,3,80,112),(12,3,80,112),(24,4,80,112)]:
    b,t,h,w = shape
    for dt in (torch.float32, torch.bfloat16):
        m = model(50, t); eng = m.train_engine(dtype=dt)
        imgs, labels = synth.synth_clip_batch(b,t,h,w), synth.synth_labels(b)
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        ref_st = {}
        with torch.no_grad():
            rl = net_torch.forward_train(torch.from_numpy(imgs), torch.from_numpy(labels), sd, depth=50, T=t, new_buffers={}, stages=ref_st)
        st = {}
        loss = eng.forward(torch.from_numpy(imgs).cuda(), torch.from_numpy(labels).cuda(), stages=st)
        print(shape, str(dt)[6:], "loss", float(loss), float(rl), {k: round(rel_l2(v.float().cpu().permute(0,3,1,2).numpy(), ref_st[k].numpy()),4) for k,v in st.items() if k in ref_st})
