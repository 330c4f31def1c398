"""oracle -- CPU restatement of the reference CTC-CRF path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
See oracle/crf_oracle.c for what is restated (reference file:line) and how it is pinned.
"""
import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

from .fst_io import read_fst  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """gcc-compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    args = ["make", "-s", "-C", _HERE]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args + ["all"])
    if os.path.isdir("/root/reference/src/ctc_crf"):
        subprocess.check_call(args + ["ref"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "crf_oracle.c")
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _graph_args(g: Dict):
    src = np.ascontiguousarray(g["src"], dtype=np.int32)
    dst = np.ascontiguousarray(g["dst"], dtype=np.int32)
    lab = np.ascontiguousarray(g["lab"], dtype=np.int32)
    w = np.ascontiguousarray(g["w"], dtype=np.float32)
    sw = np.ascontiguousarray(g["start_w"], dtype=np.float32)
    ew = np.ascontiguousarray(g["end_w"], dtype=np.float32)
    keep = (src, dst, lab, w, sw, ew)
    return keep, [ctypes.c_int(int(g["S"])), ctypes.c_int(int(g["A"])), _p(src, ctypes.c_int),
                  _p(dst, ctypes.c_int), _p(lab, ctypes.c_int), _p(w, ctypes.c_float),
                  _p(sw, ctypes.c_float), _p(ew, ctypes.c_float)]


def den(g: Dict, logits: np.ndarray, lx: np.ndarray, precision: str = "f64"):
    """gpu_den restatement. logits [B,T,V] f32 -> (grad_den [B,T,V] f32, costs_alpha[B], costs_beta[B])."""
    lib = _load()
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    B, T, V = logits.shape
    lx = np.ascontiguousarray(lx, dtype=np.int32)
    grad = np.zeros((B, T, V), dtype=np.float32)
    ca = np.zeros(B, dtype=np.float64)
    cb = np.zeros(B, dtype=np.float64)
    keep, ga = _graph_args(g)
    fn = getattr(lib, f"oracle_den_{precision}")
    rc = fn(*ga, _p(logits, ctypes.c_float), B, T, V, _p(lx, ctypes.c_int), _p(grad, ctypes.c_float),
            _p(ca, ctypes.c_double), _p(cb, ctypes.c_double))
    assert rc == 0
    return grad, ca, cb


def den_alpha(g: Dict, logits: np.ndarray, lx: np.ndarray, precision: str = "f64"):
    """Debug entry: gpu_den restatement + its forward table in the layout of the reference's `alpha` buffer, [B, T+1, S] float64
    (row lx[b] with end_weight added, as alpha_last_kernel leaves it; rows beyond lx[b] NaN)."""
    lib = _load()
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    B, T, V = logits.shape
    lx = np.ascontiguousarray(lx, dtype=np.int32)
    grad = np.zeros((B, T, V), dtype=np.float32)
    ca = np.zeros(B, dtype=np.float64)
    cb = np.zeros(B, dtype=np.float64)
    alpha = np.full((B, T + 1, int(g["S"])), np.nan, dtype=np.float64)
    keep, ga = _graph_args(g)
    fn = getattr(lib, f"oracle_den_alpha_{precision}")
    rc = fn(*ga, _p(logits, ctypes.c_float), B, T, V, _p(lx, ctypes.c_int), _p(grad, ctypes.c_float),
            _p(ca, ctypes.c_double), _p(cb, ctypes.c_double), _p(alpha, ctypes.c_double))
    assert rc == 0
    return grad, ca, cb, alpha


def ctc(logits: np.ndarray, labels: np.ndarray, lx: np.ndarray, ly: np.ndarray, precision: str = "f64"):
    """gpu_ctc restatement on [B,T,V] log-probs -> (grad_ctc [B,T,V], costs_ctc[B] (+loglike), valid[B])."""
    lib = _load()
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    B, T, V = logits.shape
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    lx = np.ascontiguousarray(lx, dtype=np.int32)
    ly = np.ascontiguousarray(ly, dtype=np.int32)
    grad = np.zeros((B, T, V), dtype=np.float32)
    cc = np.zeros(B, dtype=np.float64)
    valid = np.zeros(B, dtype=np.int32)
    fn = getattr(lib, f"oracle_ctc_{precision}")
    rc = fn(_p(logits, ctypes.c_float), B, T, V, _p(labels, ctypes.c_int), _p(lx, ctypes.c_int),
            _p(ly, ctypes.c_int), _p(grad, ctypes.c_float), _p(cc, ctypes.c_double), _p(valid, ctypes.c_int))
    assert rc == 0
    return grad, cc, valid


def ctc_alpha(logits: np.ndarray, labels: np.ndarray, lx: np.ndarray, ly: np.ndarray, precision: str = "f64"):
    """ctc() plus the forward table alpha [B,T,Smax] (Smax = 2 max(ly) + 1; NaN where an utterance has no entry), in the reference's
    workspace layout per utterance (gpu_ctc_kernels.h:134-196)."""
    lib = _load()
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    B, T, V = logits.shape
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    lx = np.ascontiguousarray(lx, dtype=np.int32)
    ly = np.ascontiguousarray(ly, dtype=np.int32)
    grad = np.zeros((B, T, V), dtype=np.float32)
    cc = np.zeros(B, dtype=np.float64)
    valid = np.zeros(B, dtype=np.int32)
    smax = int(2 * ly.max() + 1) if B else 1
    alpha = np.full((B, T, smax), np.nan, dtype=np.float64)
    fn = getattr(lib, f"oracle_ctc_alpha_{precision}")
    rc = fn(_p(logits, ctypes.c_float), B, T, V, _p(labels, ctypes.c_int), _p(lx, ctypes.c_int),
            _p(ly, ctypes.c_int), _p(grad, ctypes.c_float), _p(cc, ctypes.c_double), _p(valid, ctypes.c_int), _p(alpha, ctypes.c_double), smax)
    assert rc == 0
    return grad, cc, valid, alpha


def ctc_crf(g: Dict, logits: np.ndarray, labels: np.ndarray, lx: np.ndarray, ly: np.ndarray,
            lamb: float = 0.1, size_average: bool = True, precision: str = "f64",
            threads: Optional[int] = None):
    """_CTC_CRF.forward restatement -> dict(loss, grad [B,T,V], costs_den[B], costs_ctc[B])."""
    lib = _load()
    if threads is not None:
        # (the environment variable is read once, when libgomp is loaded -- i.e. before this line: set the ICV through the runtime's
        # own entry point, which liboracle.so's dependency on libgomp makes visible through its handle)
        lib.omp_set_num_threads(ctypes.c_int(int(threads)))
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    B, T, V = logits.shape
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    lx = np.ascontiguousarray(lx, dtype=np.int32)
    ly = np.ascontiguousarray(ly, dtype=np.int32)
    grad = np.zeros((B, T, V), dtype=np.float32)
    loss = ctypes.c_double(0.0)
    cd = np.zeros(B, dtype=np.float64)
    cc = np.zeros(B, dtype=np.float64)
    keep, ga = _graph_args(g)
    fn = getattr(lib, f"oracle_ctc_crf_{precision}")
    rc = fn(*ga, _p(logits, ctypes.c_float), B, T, V, _p(labels, ctypes.c_int), _p(lx, ctypes.c_int),
            _p(ly, ctypes.c_int), ctypes.c_double(lamb), int(bool(size_average)), _p(grad, ctypes.c_float),
            ctypes.byref(loss), _p(cd, ctypes.c_double), _p(cc, ctypes.c_double))
    assert rc == 0
    return dict(loss=loss.value, grad=grad, costs_den=cd, costs_ctc=cc)
