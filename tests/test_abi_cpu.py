"""CPU: the C-ABI library loads and exports every symbol include/mvfnet_hip.h declares (no compute calls)."""
import ctypes

import pytest


def test_library_loads_and_exports_declared_symbols():
    from mvfnet_amd import _lib
    names = _lib.declared_symbols()
    assert "mvf_fwd_infer" in names and "mvf_bwd" in names
    for n in names:
        assert hasattr(_lib.lib, n), "libmvfnet_hip.so does not export %s" % n
    assert _lib.lib.mvf_abi_version() == 2        # [r5] 2: mvf_conv_desc_t.x_c0


def test_argument_validation_needs_no_gpu():
    from mvfnet_amd import _lib
    d = _lib.MvfDesc(10, 16, 4, 4, 4, 4, 7, 0, 0)
    assert _lib.lib.mvf_fwd_infer(ctypes.byref(d), None, None, None, None, None, None, None, None) == -2
    assert b"multiple of n_segment" in _lib.lib.mvf_last_error()
    assert _lib.lib.mvf_fwd_train_workspace_bytes(ctypes.byref(d)) == 0
    d = _lib.MvfDesc(8, 16, 4, 4, 4, 4, 7, 0, 0)
    assert _lib.lib.mvf_fwd_train_workspace_bytes(ctypes.byref(d)) > 0
    assert _lib.lib.mvf_fwd_infer(ctypes.byref(d), None, None, None, None, None, None, None, None) == -1   # NULL x


def test_product_modules_refuse_cpu_tensors():
    import torch
    import torch.nn as nn
    from mvfnet_amd.modules import MVF
    m = MVF(nn.Conv2d(32, 16, 1, bias=False), 4, 32, 0.125)
    assert sorted(m.state_dict()) == sorted(
        ["net.weight", "shift_conv.weight", "h_conv.weight", "w_conv.weight", "bn.weight", "bn.bias",
         "bn.running_mean", "bn.running_var", "bn.num_batches_tracked"])
    assert tuple(m.shift_conv.weight.shape) == (4, 1, 3, 1, 1)
    assert tuple(m.h_conv.weight.shape) == (4, 1, 1, 3, 1)
    assert tuple(m.w_conv.weight.shape) == (4, 1, 1, 1, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(8, 32, 5, 5))


def test_stencil_partial_row_plan_follows_the_kernel_choice():
    """[r5] mvf_nhwc_stencil_stats_rows names the partial rows of the launch that WILL run (no launch, no GPU): the LDS-tiled bf16 kernel
    (clips x bands of rows) where its plan exists and makes >= 150 workgroups, the register-chunked kernel's (clips x pixel bands) otherwise."""
    import os
    if "stencil_lds" in os.environ.get("MVF_POLICY", ""):
        pytest.skip("the plan switches are set in the environment")
    from mvfnet_amd import _lib
    L = _lib
    rows = lambda nt, t, hw, c, cs, dt: L.lib.mvf_nhwc_stencil_stats_rows(ctypes.byref(L.MvfDesc(nt, c, hw, hw, t, cs, 7, L.MVF_NHWC, dt)), c, cs)  # noqa: E731
    assert rows(256, 8, 14, 1024, 128, L.MVF_BF16) == 32 * 7          # layer3 at 32 clips: 64 channels x 2 rows x 8 frames in 57 KB -> 448 workgroups
    assert rows(256, 16, 14, 1024, 128, L.MVF_BF16) == 16 * 7         # C4: 32 channels x 2 rows x 16 frames
    assert rows(256, 8, 7, 2048, 256, L.MVF_BF16) == 32 * 4           # layer4: whole planes would make 128 workgroups -> the halved budget, 2-row bands
    assert rows(96, 8, 14, 1024, 128, L.MVF_BF16) == 12 * 7           # 12 clips: 168 workgroups, still tiled
    assert rows(32, 8, 14, 1024, 128, L.MVF_BF16) == 4 * 25           # 4 clips: 56 workgroups -> the chunked kernel (8 pixels per workgroup)
    assert rows(256, 8, 14, 1024, 128, L.MVF_F32) == 32 * 25          # fp32: chunked


def test_plan_run_generic_call_marshals_integers_and_floats_in_order():
    """[r6] mvf_plan_run (csrc/launch_plan.hip) calls a recorded entry point as int (*)(uint64 x NI, float x NF): on x86-64 System V the two argument classes are
    assigned independently, so a prototype that INTERLEAVES them (the library's own entry points do: pointers, sizes, an eps, a stream) receives each argument where
    it expects it.  Checked here with host callbacks in place of the library's entry points -- no GPU, no launch: every integer word (64-bit pointers, 32-bit ints in
    the low half of their slot), every float, more integers than registers (the stack), and the stop-at-first-failure contract with the failing op's index."""
    import ctypes as C
    from mvfnet_amd import _lib
    seen = []
    proto_a = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float)

    def fa(p, n, eps, big, mom, q, a, b, c, d, s):
        seen.append(("a", p, n, round(eps, 6), big, round(mom, 6), q, a, b, c, d, round(s, 6)))
        return 0
    proto_b = C.CFUNCTYPE(C.c_int, C.c_int)

    def fb(code):
        seen.append(("b", code))
        return code
    cba, cbb = proto_a(fa), proto_b(fb)
    ops = (_lib.PlanOp * 4)()
    words = (C.c_ulonglong * 16)(0x7F00DEADBEE0, 12, 1 << 40, 0x7F00CAFEF000, 1, 2, 3, 4,   # op 0: 8 integer-class words (two past the six registers)
                                 0,                                                              # op 1
                                 0xFFFFFFFF00000000 | 5,                                         # op 2: an int argument reads the LOW half of its slot -> 5 -> failure code
                                 0)                                                              # op 3 (never reached)
    floats = (C.c_float * 4)(1e-5, 0.9, 2.5, 0.0)
    for i, (fn, ni, nf, w0, f0) in enumerate(((cba, 8, 3, 0, 0), (cbb, 1, 0, 8, 3), (cbb, 1, 0, 9, 3), (cbb, 1, 0, 10, 3))):
        ops[i].kind, ops[i].n_int, ops[i].n_flt, ops[i].fn, ops[i].word0, ops[i].float0 = 0, ni, nf, C.cast(fn, C.c_void_p).value, w0, f0
    failed = C.c_int(-1)
    rc = _lib.lib.mvf_plan_run(ops, 4, words, floats, C.byref(failed))
    assert rc == 5 and failed.value == 2
    assert seen == [("a", 0x7F00DEADBEE0, 12, 1e-5, 1 << 40, 0.9, 0x7F00CAFEF000, 1, 2, 3, 4, 2.5), ("b", 0), ("b", 5)]
    # a malformed record is refused before anything is called
    ops[0].n_int = 41
    assert _lib.lib.mvf_plan_run(ops, 1, words, floats, C.byref(failed)) == -1 and failed.value == 0          # MVF_EINVAL
    assert b"bad call record" in _lib.lib.mvf_last_error()
