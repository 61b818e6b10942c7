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
