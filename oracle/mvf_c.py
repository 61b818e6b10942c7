"""ORACLE loader (test infrastructure): ctypes view of oracle/libmvf_oracle.so (oracle/mvf_ref.c, built by oracle/Makefile)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmvf_oracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "mvf_ref.c")):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _LIB = C.CDLL(path)
    return _LIB


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def forward_backward(x, T, cs, mode_bits, wt, wh, ww, use_hs, training, gamma, beta, rm, rv, g=None):
    """Runs mvf_ref_forward (and mvf_ref_backward when g is given). Returns dict of numpy arrays."""
    L = lib()
    x = _f(x)
    NT, Cc, H, W = x.shape
    wt, wh, ww = _f(wt), _f(wh if wh is not None else wt), _f(ww if ww is not None else wt)
    gamma, beta = _f(gamma if gamma is not None else np.ones(cs)), _f(beta if beta is not None else np.zeros(cs))
    rm = _f(rm if rm is not None else np.zeros(cs)).copy()
    rv = _f(rv if rv is not None else np.ones(cs)).copy()
    out = np.empty_like(x)
    m = NT * H * W
    mean, invstd, ypre = np.zeros(cs), np.ones(cs), np.zeros((cs, m))
    rc = L.mvf_ref_forward(_pf(x), NT, Cc, H, W, T, cs, mode_bits, _pf(wt), _pf(wh), _pf(ww), int(use_hs), int(training), _pf(gamma),
                           _pf(beta), _pf(rm), _pf(rv), C.c_double(1e-5), C.c_double(0.1), _pf(out), _pd(mean), _pd(invstd), _pd(ypre))
    assert rc == 0
    res = dict(out=out, running_mean=rm, running_var=rv)
    if g is not None:
        g = _f(g)
        dx = np.empty_like(x)
        dwt, dwh, dww = np.zeros((cs, 3)), np.zeros((cs, 3)), np.zeros((cs, 3))
        dg, db = np.zeros(cs), np.zeros(cs)
        rc = L.mvf_ref_backward(_pf(g), _pf(x), _pd(ypre), NT, Cc, H, W, T, cs, mode_bits, _pf(wt), _pf(wh), _pf(ww), int(use_hs),
                                int(training), _pf(gamma), _pf(beta), _pd(mean), _pd(invstd), _pf(dx), _pd(dwt), _pd(dwh), _pd(dww), _pd(dg), _pd(db))
        assert rc == 0
        res.update(dx=dx, dwt=dwt, dwh=dwh, dww=dww, dgamma=dg, dbeta=db)
    return res
