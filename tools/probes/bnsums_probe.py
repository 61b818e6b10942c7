import sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from mvfnet_amd import _lib
from mvfnet_amd._lib import ConvDesc
lib, check = _lib.lib, _lib.check
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
for dtype, dt in ((torch.float32, 0), (torch.bfloat16, 1)):
    for (n, h, w, cout, cin, k) in ((2, 9, 7, 256, 64, 1), (3, 10, 10, 64, 64, 3), (1, 20, 20, 128, 128, 3)):
        g = torch.Generator().manual_seed(1)
        pad = k // 2
        m = n * h * w
        dz = (torch.randn(m, cout, generator=g) * 0.5).to(dtype).cuda()
        wgt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).cuda()
        wd = torch.empty(cin, k, k, cout, dtype=dtype, device="cuda")
        check(lib.mvf_pack_conv_weight_dgrad(P(wgt), cout, cin, k, k, P(wd), dt, None))
        z = torch.randn(m, cin, generator=g).to(dtype).cuda()
        mean, invstd = torch.randn(cin, generator=g).cuda() * 0.1, (torch.rand(cin, generator=g) + 0.5).cuda()
        scale, shift = (torch.rand(cin, generator=g) + 0.5).cuda(), torch.randn(cin, generator=g).cuda() * 0.3
        d = ConvDesc(n, h, w, cout, cin, k, k, 1, k - 1 - pad, h, w, cout, dt, 0, 0, 0, 0, 0)
        ws = torch.zeros(max(lib.mvf_conv2d_workspace_bytes(C.byref(d)), 1), dtype=torch.uint8, device="cuda")
        y1, y2 = torch.empty(m, cin, dtype=dtype, device="cuda"), torch.empty(m, cin, dtype=dtype, device="cuda")
        rows = lib.mvf_conv2d_stats_rows(C.byref(d))
        part = torch.zeros(rows, cin, 2, device="cuda")
        check(lib.mvf_conv2d_nhwc_dgrad_bnsums(C.byref(d), P(dz), P(wd), P(y1), P(z), P(mean), P(invstd), P(scale), P(shift), P(part), P(ws), ws.numel(), None))
        dg, db = torch.empty(cin, device="cuda"), torch.empty(cin, device="cuda")
        check(lib.mvf_bn_bwd_finalize(P(part), rows, cin, P(dg), P(db), None))
        torch.cuda.synchronize()
        check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), P(dz), None, P(wd), None, None, P(y2), P(ws), ws.numel(), None))
        ws2 = torch.empty(lib.mvf_bn_workspace_bytes(m, cin), dtype=torch.uint8, device="cuda")
        dg2, db2 = torch.empty(cin, device="cuda"), torch.empty(cin, device="cuda")
        check(lib.mvf_bn_bwd_reduce(P(y2), cin, P(z), None, m, cin, P(mean), P(invstd), P(scale), P(shift), 2, None, P(dg2), P(db2), P(ws2), ws2.numel(), dt, None))
        torch.cuda.synchronize()
        print(dtype, (n, h, w, cout, cin, k), "y equal", torch.equal(y1, y2), "dgamma rel", float((dg - dg2).abs().max() / dg2.abs().max()), "dbeta rel", float((db - db2).abs().max() / db2.abs().max()))
