"""12 plain bf16 inference steps (32 clips, 2 launch chains) for timeline analysis under rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mvfnet_amd
m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, 8), None, dict(average_clips=None))
m.backbone.engine_dtype = torch.bfloat16
m = m.cuda().eval()
m.backbone.engine().streams = int(os.environ.get("CHAINS", "2"))
imgs = torch.randn(32, 8, 3, 224, 224, device="cuda")
for _ in range(12):
    m(imgs, None, return_loss=False, return_numpy=False)
torch.cuda.synchronize()
