import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from mvfnet_amd import _lib as L
lib, check = L.lib, L.check
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
BF = torch.bfloat16
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
for (nt, h, k, c) in [(256, 56, 64, 256), (256, 28, 128, 512)]:
    m = nt * h * h
    gen = torch.Generator().manual_seed(3)
    a = (torch.relu(torch.randn(m, k, generator=gen) + 0.3) * (torch.rand(k, generator=gen) + 0.5)).cuda().to(BF)
    w3 = (torch.randn(c, k, generator=gen) * 0.08).cuda().to(BF)
    gamma, beta = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    perm = torch.randperm(nt // 8, generator=gen).cuda()
    ap = a.view(nt // 8, 8 * h * h, k)[perm].reshape(m, k).contiguous()
    ws = torch.empty(max(lib.mvf_bn_workspace_bytes(m, max(c, k)), 300 * k * k * 4 + 4096, 1 << 28), dtype=torch.uint8, device="cuda")
    out = {}
    for tag, src in (("id", a), ("perm", ap)):
        # pass over the stored z
        z = (src.float() @ w3.float().t()).to(BF)
        o1 = [torch.empty(c, device="cuda") for _ in range(4)]
        check(lib.mvf_bn_train_stats(P(z), m, c, P(gamma), P(beta), C.c_float(1e-5), C.c_float(0.1), None, None, P(o1[0]), P(o1[1]), P(o1[2]), P(o1[3]), P(ws), ws.numel(), L.MVF_BF16, None))
        gram, amean = torch.empty(k, k, device="cuda"), torch.empty(4, k, device="cuda")
        one, zero = torch.ones(k, device="cuda"), torch.zeros(k, device="cuda")
        d = L.ConvDesc(nt, h, h, k, k, 1, 1, 1, 0, h, h, k, 1, 0, 0, 0, 0)
        check(lib.mvf_conv2d_nhwc_wgrad_wgs(C.byref(d), P(src), P(src), None, 1, k, 1, k, P(gram), P(ws), ws.numel(), 256, None))
        check(lib.mvf_bn_train_stats(P(src), m, k, P(one), P(zero), C.c_float(1e-5), C.c_float(0.1), None, None, P(amean[0]), P(amean[1]), P(amean[2]), P(amean[3]), P(ws), ws.numel(), L.MVF_BF16, None))
        o2 = [torch.empty(c, device="cuda") for _ in range(4)]
        check(lib.mvf_bn_train_stats_gram(P(gram), P(amean[0]), P(w3), m, c, k, P(gamma), P(beta), C.c_float(1e-5), C.c_float(0.1), None, None, P(o2[0]), P(o2[1]), P(o2[2]), P(o2[3]), L.MVF_BF16, None))
        torch.cuda.synchronize()
        out[tag] = ([t.cpu().numpy() for t in o1[:2]], [t.cpu().numpy() for t in o2[:2]])
    zx = a.double() @ w3.double().t()
    m64, i64 = zx.mean(0).cpu().numpy(), (1 / torch.sqrt(zx.var(0, unbiased=False) + 1e-5)).cpu().numpy()
    print("M%d k%d c%d | clip permutation changes: pass mean %.1e invstd %.1e ; Gram mean %.1e invstd %.1e | vs fp64: pass %.1e %.1e ; Gram %.1e %.1e" % (
        m, k, c, rel(out["perm"][0][0], out["id"][0][0]), rel(out["perm"][0][1], out["id"][0][1]), rel(out["perm"][1][0], out["id"][1][0]), rel(out["perm"][1][1], out["id"][1][1]),
        rel(out["id"][0][0], m64), rel(out["id"][0][1], i64), rel(out["id"][1][0], m64), rel(out["id"][1][1], i64)))
