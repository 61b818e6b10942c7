"""fp32 conv error against an fp64 convolution for small maps / strides (the shapes of a 96 x 96 R50 step): MVF_F32_X3=0 vs 1."""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, torch.nn.functional as F
from mvfnet_amd import _lib
lib, check = _lib.lib, _lib.check
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
for (n, h, cin, cout, k, s) in [(8, 24, 64, 64, 3, 1), (8, 24, 128, 128, 3, 2), (8, 12, 128, 128, 3, 1), (8, 12, 256, 256, 3, 2), (8, 6, 256, 256, 3, 1), (8, 6, 512, 512, 3, 2),
                                (8, 3, 512, 512, 3, 1), (8, 14, 256, 256, 3, 1), (8, 6, 1024, 256, 1, 1)]:
    gen = torch.Generator().manual_seed(h * cin + k)
    x = torch.randn(n, h, h, cin, generator=gen)
    w = torch.randn(cout, cin, k, k, generator=gen) / (cin * k * k) ** 0.5
    pad = k // 2
    ho = (h + 2 * pad - k) // s + 1
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), stride=s, padding=pad).permute(0, 2, 3, 1)
    xc, wc = x.cuda(), w.cuda()
    wp = torch.empty(cout, k, k, cin, device="cuda")
    check(lib.mvf_pack_conv_weight(p(wc), cout, cin, k, k, k, cin, None, p(wp), 0, None))
    d = _lib.ConvDesc(n, h, h, cin, cout, k, k, s, pad, ho, ho, cin, 0, 0, 0, 0, 0)
    y = torch.empty(n, ho, ho, cout, device="cuda")
    check(lib.mvf_conv2d_nhwc_fwd(C.byref(d), p(xc), None, p(wp), None, None, p(y), None))
    torch.cuda.synchronize()
    e = (y.cpu().double() - ref)
    print("X3=%s n%d h%-2d cin%-4d cout%-4d k%d s%d  rel L2 %.2e  max/max %.2e" % (os.environ.get("MVF_F32_X3", "1"), n, h, cin, cout, k, s,
                                                                                 float(e.norm() / ref.norm()), float(e.abs().max() / ref.abs().max())))
