"""ctypes wrapper of oracle/conv_ref.c (oracle — test infrastructure only): builds libconvref.so with gcc on
demand and exposes numpy-in / numpy-out functions with the same TF semantics as oracle/tfops.py."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "conv_ref.c")
SO = os.path.join(_HERE, "libconvref.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", SRC, "-o", SO])
    return SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _same(n, k, s, d=1):
    o = -(-n // s)
    return o


def conv2d_fwd(x, w, b=None, stride=1, dil=1):
    x, w = _f(x), _f(w)
    b = None if b is None else _f(b)
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    y = np.empty((n, _same(h, kh, stride), _same(wd, kw, stride), cout), np.float32)
    lib().ref_conv2d_fwd(_p(x), _p(w), _p(b), _p(y), n, h, wd, cin, cout, kh, kw, stride, dil)
    return y


def conv2d_dgrad(dy, w, xshape, stride=1, dil=1):
    dy, w = _f(dy), _f(w)
    n, h, wd, cin = xshape
    kh, kw, _, cout = w.shape
    dx = np.empty(xshape, np.float32)
    lib().ref_conv2d_dgrad(_p(dy), _p(w), _p(dx), n, h, wd, cin, cout, kh, kw, stride, dil)
    return dx


def conv2d_wgrad(x, dy, kshape, stride=1, dil=1):
    x, dy = _f(x), _f(dy)
    n, h, wd, cin = x.shape
    kh, kw, _, cout = kshape
    dw = np.empty(kshape, np.float32)
    db = np.empty((cout,), np.float32)
    lib().ref_conv2d_wgrad(_p(x), _p(dy), _p(dw), _p(db), n, h, wd, cin, cout, kh, kw, stride, dil)
    return dw, db


def dwconv2d_fwd(x, w, stride=1):
    x, w = _f(x), _f(w)
    n, h, wd, c = x.shape
    kh, kw = w.shape[:2]
    y = np.empty((n, _same(h, kh, stride), _same(wd, kw, stride), c), np.float32)
    lib().ref_dwconv2d_fwd(_p(x), _p(w), _p(y), n, h, wd, c, kh, kw, stride)
    return y


def conv2d_transpose(x, w, b=None, stride=2):
    x, w = _f(x), _f(w)
    b = None if b is None else _f(b)
    n, h, wd, cin = x.shape
    kh, kw, cout, _ = w.shape
    y = np.empty((n, h * stride, wd * stride, cout), np.float32)
    lib().ref_conv2d_transpose(_p(x), _p(w), _p(b), _p(y), n, h, wd, cin, cout, kh, kw, stride)
    return y


def maxpool_fwd(x, k, stride, same):
    x = _f(x)
    n, h, wd, c = x.shape
    ho = _same(h, k, stride) if same else (h - k) // stride + 1
    wo = _same(wd, k, stride) if same else (wd - k) // stride + 1
    y = np.empty((n, ho, wo, c), np.float32)
    lib().ref_maxpool_fwd(_p(x), _p(y), n, h, wd, c, k, stride, int(same))
    return y
