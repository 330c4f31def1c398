"""cat_amd/ctc_crf/_C.py -- ctypes binding of libctc_crf_hip.so, mirroring the reference's pybind module
``ctc_crf._C`` (src/ctc_crf/binding.cpp:120-126: gpu_den, gpu_ctc, init_env, release_env).

There is NO CPU fallback and no PyTorch fallback: if the HIP library is missing the import fails,
and every call goes through the C ABI of include/ctc_crf_hip.h.
"""
import ctypes
import os
import sys
import threading
import warnings
from typing import Dict, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRF_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libctc_crf_hip.so")  # CRF_LIB: A/B builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the gfx950 HIP library has not been built. "
        "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `python -m cat_amd.build`) first; "
        "there is no CPU fallback for the CTC-CRF loss.")

_lib = ctypes.CDLL(LIB_PATH)

_i64, _i32, _f32, _vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p
_lib.crf_graph_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(_vp)]
_lib.crf_graph_create.restype = ctypes.c_int
_lib.crf_graph_create_from_arcs.argtypes = [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.POINTER(_vp)]
_lib.crf_graph_create_from_arcs.restype = ctypes.c_int
_lib.crf_graph_destroy.argtypes = [_vp]
_lib.crf_graph_destroy.restype = None
_lib.crf_graph_dims.argtypes = [_vp] + [ctypes.POINTER(_i64)] * 4
_lib.crf_graph_dims.restype = ctypes.c_int
_lib.crf_graph_stats.argtypes = [_vp, ctypes.POINTER(_i64), ctypes.c_int]
_lib.crf_graph_stats.restype = ctypes.c_int
_lib.crf_workspace_bytes.argtypes = [_vp, _i64, _i64, _i64, _i64]
_lib.crf_workspace_bytes.restype = _i64
_lib.crf_den_kernels.argtypes = [_vp, _i64, _i64, _i64]
_lib.crf_den_kernels.restype = ctypes.c_int
_lib.crf_loss_fwd_bwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _f32,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]
_lib.crf_loss_fwd_bwd.restype = ctypes.c_int
_lib.crf_loss_fwd_bwd_logits.argtypes = [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _f32,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]
_lib.crf_loss_fwd_bwd_logits.restype = ctypes.c_int
_lib.crf_profile_enable.argtypes = [ctypes.c_int]
_lib.crf_profile_enable.restype = None
_lib.crf_profile_read.argtypes = [ctypes.POINTER(_f32), ctypes.c_int]
_lib.crf_profile_read.restype = ctypes.c_int
_lib.crf_timing_read.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
_lib.crf_timing_read.restype = ctypes.c_int
_lib.crf_stage_i32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
_lib.crf_stage_i32.restype = ctypes.c_int
_lib.crf_debug_set.argtypes = [ctypes.c_char_p, ctypes.c_int]
_lib.crf_debug_set.restype = ctypes.c_int
_lib.crf_debug_unset.argtypes = [ctypes.c_char_p]
_lib.crf_debug_unset.restype = ctypes.c_int
_lib.crf_debug_list.restype = ctypes.c_char_p
_lib.crf_last_error.restype = ctypes.c_char_p
_lib.crf_last_den_kernel.restype = ctypes.c_char_p
_lib.crf_last_call_streams.restype = ctypes.c_int
_lib.crf_last_side_stream.restype = ctypes.c_char_p
_lib.crf_last_fallback_counts.argtypes = [ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p]
_lib.crf_last_fallback_counts.restype = ctypes.c_int
_lib.crf_version.restype = ctypes.c_char_p
_lib.crf_build_switches.restype = ctypes.c_char_p

EXPORTED_SYMBOLS = (
    "crf_graph_create", "crf_graph_create_from_arcs", "crf_graph_destroy", "crf_graph_dims", "crf_graph_stats",
    "crf_workspace_bytes", "crf_den_kernels", "crf_debug_stream_check", "crf_debug_decode_check", "crf_debug_facbatch_check", "crf_debug_fac_emulate", "crf_debug_res_emulate", "crf_debug_stage_plan", "crf_loss_fwd_bwd", "crf_loss_fwd_bwd_logits", "crf_profile_enable", "crf_profile_read", "crf_timing_read", "crf_stage_i32",
    "crf_debug_set", "crf_debug_unset", "crf_debug_list", "crf_last_den_kernel", "crf_last_call_streams", "crf_last_side_stream", "crf_last_fallback_counts", "crf_build_switches", "crf_last_error", "crf_version",
)

PROFILE_SLOTS = ("prep", "den_fwd_chain", "den_bwd_chain", "ctc_fwd_chain", "ctc_bwd_chain", "grad",
                 "finalize", "call")


def profile_enable(on: bool) -> None:
    _lib.crf_profile_enable(1 if on else 0)


def profile_read() -> Dict[str, float]:
    """Per-kernel HIP-event durations (ms) of the last loss_fwd_bwd call on this thread."""
    buf = (_f32 * 8)()
    n = _lib.crf_profile_read(buf, 8)
    return {PROFILE_SLOTS[i]: float(buf[i]) for i in range(n)}


def last_den_kernel() -> str:
    """Template instantiation of the denominator recursions' kernel in this thread's last call (include/ctc_crf_hip.h)."""
    return _lib.crf_last_den_kernel().decode()


def last_call_streams() -> int:
    """1 / 2 / 3: streams this thread's last call put work on (caller's, + side stream, + third stream)."""
    return int(_lib.crf_last_call_streams())


def last_side_stream() -> str:
    """Kind of the side stream of the last call's context and how many candidates were probed (include/ctc_crf_hip.h)."""
    return _lib.crf_last_side_stream().decode()


def last_fallback_counts(stream: int = 0):
    """(denominator, numerator): utterances of this thread's last call that a fallback redid.  Synchronises `stream` -- diagnostics only."""
    buf = (ctypes.c_int32 * 2)()
    _check(_lib.crf_last_fallback_counts(buf, _vp(stream)))
    return int(buf[0]), int(buf[1])


_SERIAL_WARNED = False
_SIDE_STREAM_SEEN = set()      # (device, stream) contexts whose side stream has been looked at: the check costs one set lookup per step afterwards


def _warn_if_serial(key=None) -> None:
    """The library found no stream that runs beside the caller's: every kernel of the loss runs one after the other on the caller's
    stream -- same results, about 1.6 x the step time at the benchmark shape.  The C library says so on stderr (once per context); a
    training script sees it here, once per process, where Python's warning filters and loggers can pick it up."""
    global _SERIAL_WARNED
    if _SERIAL_WARNED or key in _SIDE_STREAM_SEEN:
        return
    if key is not None:
        _SIDE_STREAM_SEEN.add(key)
    desc = _lib.crf_last_side_stream().decode()
    if desc.startswith("none"):
        _SERIAL_WARNED = True
        warnings.warn(f"ctc_crf: no HIP stream of this process runs beside the caller's stream (side stream: {desc}); the CTC-CRF loss runs its "
                      "kernels one after the other -- correct, but about 1.6x slower than the two-stream schedule. Usual cause: every hardware "
                      "queue of the process is shared with the caller's stream (GPU_MAX_HW_QUEUES too small).", RuntimeWarning, stacklevel=3)


def version() -> str:
    return _lib.crf_version().decode()


def build_switches() -> Dict[str, int]:
    """The CRF_X_* build-time switches of the loaded library (include/ctc_crf_hip.h crf_build_switches)."""
    return {k: int(v) for k, v in (kv.split("=") for kv in _lib.crf_build_switches().decode().split())}


def _check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"ctc_crf_hip error {rc}: {_lib.crf_last_error().decode()}")


# Debug / experiment switches of tests and tools (include/ctc_crf_hip.h crf_debug_set): the LIBRARY never reads the
# environment.  For the tools' convenience this binding applies CRF_DEBUG="name=value,name=value" once, at import.
_DEBUG_SET: Dict[str, int] = {}


def debug_set(key: str, value: Optional[int]) -> None:
    """Set (value None: return to the default) one switch of crf_debug_list()."""
    if value is None:
        _check(_lib.crf_debug_unset(key.encode()))
        _DEBUG_SET.pop(key, None)
    else:
        _check(_lib.crf_debug_set(key.encode(), int(value)))
        _DEBUG_SET[key] = int(value)


def debug_list() -> str:
    return _lib.crf_debug_list().decode()


class debug_opts:
    """``with debug_opts(no_factored=1, bat_ul=16): ...`` -- switches set for the block, previous values restored after."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: _DEBUG_SET.get(k) for k in self.kw}
        for k, v in self.kw.items():
            debug_set(k, v)
        return self

    def __exit__(self, *a):
        for k, v in self.old.items():
            debug_set(k, v)


# process-global graphs, one per device -- the reference keeps one graph per process in globals
# indexed by DEVICE_HASH[cudaGetDevice()] (den_calculate.cu:263-273, 436-438).  Identified by a GENERATION number, not by
# the handle's value: a later graph can be allocated at the address of one that was destroyed, and a context that compared
# pointers would then take the newcomer for its own (and destroy it).
_GRAPHS: Dict[int, int] = {}
_GRAPH_GEN: Dict[int, int] = {}
_gen_counter = 0


def graph_generation(dev: int) -> Optional[int]:
    """Generation number of the graph currently loaded on `dev` (None: none)."""
    return _GRAPH_GEN.get(dev)


def graph_dims(handle: int):
    S, A, P, ML = _i64(), _i64(), _i64(), _i64()
    _check(_lib.crf_graph_dims(_vp(handle), S, A, P, ML))
    return dict(S=S.value, A=A.value, P=P.value, max_label=ML.value)


def graph_stats(handle: int):
    buf = (_i64 * 27)()
    _check(_lib.crf_graph_stats(_vp(handle), buf, 27))
    k = ("S", "A", "P", "Pr", "Sr", "fwd_ell_arcs", "bwd_ell_arcs", "fwd_bank_conflicts", "bwd_bank_conflicts", "deg",
         "res_K", "res_fwd_slots", "res_bwd_slots", "res_fwd_bank_conflicts", "res_bwd_bank_conflicts", "res_rows",
         "fac", "fac_matched_pairs", "regauged", "fac_tail_rows", "fac_fwd_slots", "fac_bwd_slots", "fac_fused_rows", "fac_G", "fac_geom", "fac_chunks", "facp")
    d = dict(zip(k, [int(x) for x in buf]))
    d["max_in_deg"], d["max_out_deg"] = d["deg"] // 100000, d.pop("deg") % 100000
    d["res_fwd_rows"], d["res_bwd_rows"] = d["res_rows"] // 100000, d.pop("res_rows") % 100000
    d["fac_Gf"], d["fac_Gb"] = d["fac_G"] // 100000, d.pop("fac_G") % 100000
    return d


def den_kernels(handle: int, B: int, T: int, V: int) -> str:
    """Which denominator kernels a call of this shape takes (include/ctc_crf_hip.h crf_den_kernels)."""
    k = _lib.crf_den_kernels(_vp(handle), B, T, V)
    if k < 0:
        _check(1)
    return ("streaming", "resident", "factored", "batch")[k]


def debug_stream_check(handle: int, UL: int, want: int):
    """Host-side construction + self-check of the utterance-minor arc streams (include/ctc_crf_hip.h)."""
    out = (_i64 * 4)()
    _lib.crf_debug_stream_check.argtypes = [_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_i64)]
    _lib.crf_debug_stream_check.restype = ctypes.c_int
    _check(_lib.crf_debug_stream_check(_vp(handle), UL, want, out))
    return {"tasks": int(out[0]), "rest_rows": int(out[1]), "steps": int(out[2]), "arc_records": int(out[3])}


def debug_facbatch_check(handle: int):
    """Host-side check of the factored rows of the utterance-minor kernels (include/ctc_crf_hip.h)."""
    out = (_i64 * 4)()
    _lib.crf_debug_facbatch_check.argtypes = [_vp, ctypes.POINTER(_i64)]
    _lib.crf_debug_facbatch_check.restype = ctypes.c_int
    _check(_lib.crf_debug_facbatch_check(_vp(handle), out))
    return {"NU": int(out[0]), "fwd_records": int(out[1]), "bwd_records": int(out[2]), "arcs": int(out[3])}


def debug_fac_emulate(handle: int, T: int = 6, seed: int = 1):
    """CPU emulation of the factored recursion kernels on the layout tables (include/ctc_crf_hip.h): (plain, forward, backward)."""
    out = (ctypes.c_double * 3)()
    _lib.crf_debug_fac_emulate.argtypes = [_vp, ctypes.c_int, ctypes.c_uint, ctypes.POINTER(ctypes.c_double)]
    _lib.crf_debug_fac_emulate.restype = ctypes.c_int
    _check(_lib.crf_debug_fac_emulate(_vp(handle), T, seed, out))
    return float(out[0]), float(out[1]), float(out[2])


def debug_stage_plan(T: int, B: int):
    """The stage plan of the staged grad pass (include/ctc_crf_hip.h: crf_debug_stage_plan; no GPU) under the current debug switches."""
    n = 18
    out = (ctypes.c_int32 * (4 + 4 * n))()
    _lib.crf_debug_stage_plan.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), ctypes.c_int]
    _lib.crf_debug_stage_plan.restype = ctypes.c_int
    _check(_lib.crf_debug_stage_plan(T, B, out, len(out)))
    ns = int(out[0])
    arr = lambda j: [int(out[4 + j * n + k]) for k in range(n)]
    return {"nstage": ns, "piece": int(out[1]), "one_launch": bool(out[2]), "workgroups": int(out[3]),
            "bound": arr(0)[:ns + 1], "poff": arr(1)[:ns + 2], "fpb": arr(2)[:ns + 1], "nf": arr(3)[:ns + 1]}


def debug_res_emulate(handle: int, T: int = 6, seed: int = 1):
    """CPU emulation of the generic register-resident kernels on the layout tables: (plain, forward, backward)."""
    out = (ctypes.c_double * 3)()
    _lib.crf_debug_res_emulate.argtypes = [_vp, ctypes.c_int, ctypes.c_uint, ctypes.POINTER(ctypes.c_double)]
    _lib.crf_debug_res_emulate.restype = ctypes.c_int
    _check(_lib.crf_debug_res_emulate(_vp(handle), T, seed, out))
    return float(out[0]), float(out[1]), float(out[2])


def timing_read(n: int = 16384):
    """In-kernel phase stamps of the last call (timing builds only; [] in a product build)."""
    buf = (ctypes.c_uint64 * n)()
    k = _lib.crf_timing_read(buf, n)
    return [int(buf[i]) for i in range(k)]


def compile_graph_host_only(fst_name: str) -> int:
    """Compile the tables on the host only (device = -1): diagnostics and CPU-side tests."""
    out = _vp()
    _check(_lib.crf_graph_create(os.fsencode(fst_name), -1, ctypes.byref(out)))
    return out.value


def init_env(fst_name: str, gpus: torch.Tensor) -> None:
    """binding.cpp:51-56 ``init_env`` -> Init(): load den_lm once per listed GPU."""
    for dev in [int(i) for i in gpus.tolist()]:
        out = _vp()
        _check(_lib.crf_graph_create(os.fsencode(fst_name), dev, ctypes.byref(out)))
        if dev in _GRAPHS:  # the reference leaks on a second Init; we replace
            _lib.crf_graph_destroy(_vp(_GRAPHS.pop(dev)))
        global _gen_counter
        _gen_counter += 1
        _GRAPHS[dev] = out.value
        _GRAPH_GEN[dev] = _gen_counter


def release_env(gpus: torch.Tensor) -> None:
    """binding.cpp:58-63 ``release_env`` -> Release()."""
    for dev in [int(i) for i in gpus.tolist()]:
        h = _GRAPHS.pop(dev, None)
        _GRAPH_GEN.pop(dev, None)
        if h is not None:
            _lib.crf_graph_destroy(_vp(h))


def release_handles(handles: Dict[int, int]) -> None:
    """Release exactly the graphs a CRFContext created ({device: generation}): a device whose graph has been replaced
    since (a second CRFContext on the same device) is left alone -- the replacement already destroyed the old tables."""
    for dev, gen in handles.items():
        if _GRAPH_GEN.get(dev) == gen:
            _GRAPH_GEN.pop(dev)
            _lib.crf_graph_destroy(_vp(_GRAPHS.pop(dev)))


def graph_for(device: torch.device) -> int:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    h = _GRAPHS.get(idx)
    if h is None:
        raise RuntimeError(
            f"no denominator graph loaded on cuda:{idx}: create ctc_crf.CRFContext(den_lm, {idx}) first "
            "(reference: cat/ctc/train.py:137-141)")
    return h


def _ptr(t: Optional[torch.Tensor]):
    return _vp(0) if t is None else _vp(t.data_ptr())


class _PinnedRing:
    """A few pinned int32 staging buffers per device, reused round-robin: a copy from PAGEABLE memory makes the
    host wait for the stream (the next call could not be enqueued while this one runs).  The device copy is
    made by a KERNEL that reads the pinned buffer (crf_stage_i32): a DMA copy in the caller's stream left the
    GPU idle for ~0.15 ms between two calls while the copy engine started (rocprofv3 timeline), and a copy
    stream of its own would be the process's fifth stream -- HIP maps streams onto four hardware queues, two of
    the loss's own side streams then share one and the numerator recursions run one after the other.
    A slot is reused only after the event of its last copy has completed; host threads take slots under a lock."""

    SLOTS = 8

    def __init__(self):
        self.buf = [None] * self.SLOTS
        self.ev = [None] * self.SLOTS
        self.i = 0
        self.lock = threading.Lock()

    def stage(self, src: torch.Tensor, dev: torch.device) -> torch.Tensor:
        with self.lock:
            return self._stage(src, dev)

    def _stage(self, src: torch.Tensor, dev: torch.device) -> torch.Tensor:
        k = self.i
        self.i = (k + 1) % self.SLOTS
        n = src.numel()
        if self.ev[k] is not None:
            self.ev[k].synchronize()
        if self.buf[k] is None or self.buf[k].numel() < n:
            self.buf[k] = torch.empty(max(n, 4096), dtype=torch.int32).pin_memory()
        self.buf[k][:n].copy_(src)
        out = torch.empty(n, dtype=torch.int32, device=dev)
        cur = torch.cuda.current_stream(dev)
        with torch.cuda.device(dev):
            _check(_lib.crf_stage_i32(_vp(out.data_ptr()), _vp(self.buf[k].data_ptr()), n, _vp(cur.cuda_stream)))
        ev = torch.cuda.Event()
        ev.record(cur)
        self.ev[k] = ev
        return out


_RINGS = {}


def _h2d_async(src: torch.Tensor, dev: torch.device) -> torch.Tensor:
    if src.is_cuda:
        return src
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ring = _RINGS.get(key)
    if ring is None:
        ring = _RINGS.setdefault(key, _PinnedRing())
    return ring.stage(src, dev)


_FUSED_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}

# Debug aid for the parity tests (set_debug_poison): fill the workspace with NaN bit patterns before every call, so
# that a kernel reading a row before its producer has written it cannot pass by finding the previous call's values in the
# block the caching allocator hands back.
_POISON_WS = False


def set_debug_poison(on: bool) -> None:
    global _POISON_WS
    _POISON_WS = bool(on)


def _validate_meta(lx_cpu, ly_cpu, lab_cpu, T: int, V: int) -> None:
    """Host-resident metadata is checked before the launch (no device sync: these tensors are on the CPU already).
    The reference does not check: lx > T walks past the buffers, a label >= V indexes the logits out of bounds
    (gpu_ctc_kernels.h:146-152 reads probs[... + label])."""
    if lx_cpu.numel() and (int(lx_cpu.min()) < 0 or int(lx_cpu.max()) > T):
        raise RuntimeError(f"frame lengths must lie in [0, T={T}], got [{int(lx_cpu.min())}, {int(lx_cpu.max())}]")
    if ly_cpu is not None:
        if ly_cpu.numel() and int(ly_cpu.min()) < 0:
            raise RuntimeError("negative label length")
        if int(ly_cpu.sum()) > lab_cpu.numel():
            raise RuntimeError(f"sum(label_lengths)={int(ly_cpu.sum())} exceeds len(labels)={lab_cpu.numel()}")
        n = int(ly_cpu.sum())
        if n and (int(lab_cpu[:n].min()) <= 0 or int(lab_cpu[:n].max()) >= V):
            raise RuntimeError(f"labels must lie in [1, V-1={V - 1}] (0 is the blank), got [{int(lab_cpu[:n].min())}, {int(lab_cpu[:n].max())}]")


def loss_fwd_bwd(logits: torch.Tensor, labels: Optional[torch.Tensor], lx: torch.Tensor,
                 ly: Optional[torch.Tensor], c_den: float, c_ctc: float, graph: Optional[int],
                 want_costs: bool = False, fused: bool = False):
    """One call of the hot path (include/ctc_crf_hip.h ``crf_loss_fwd_bwd``).

    logits [N,T,V] f32 on the GPU, contiguous; labels/lx/ly int32 on CPU (as CAT passes them,
    cat/ctc/train.py:176-190) or on the GPU.  Returns (loss[1], grad[N,T,V], extras dict).
    """
    assert logits.is_cuda and logits.is_contiguous() and logits.dim() == 3
    if fused:   # raw network output, log_softmax fused in (crf_loss_fwd_bwd_logits)
        if logits.dtype not in _FUSED_DTYPES:
            raise RuntimeError(f"fused log_softmax: expect float32, bfloat16 or float16 network output, got {logits.dtype}")
    else:
        assert logits.dtype == torch.float32
    dev = logits.device
    N, T, V = logits.shape
    lx32 = lx.to(torch.int32)
    if c_ctc != 0.0:
        ly32 = ly.to(torch.int32)
        ly_cpu = ly32.cpu()
        max_l = int(ly_cpu.max()) if N > 0 else 0
        off = (torch.cumsum(ly_cpu, 0, dtype=torch.int32) - ly_cpu).to(torch.int32)
        lab32 = labels.to(torch.int32).reshape(-1)
        if not lx32.is_cuda and not lab32.is_cuda:
            _validate_meta(lx32, ly_cpu, lab32, T, V)
        if lab32.numel() == 0:
            lab32 = torch.zeros(1, dtype=torch.int32)
        # one H2D copy for all integer metadata (the reference issues ~5, gpu_ctc.h:143-229), through pinned
        # staging so that it does not serialise the host with the stream
        meta = _h2d_async(torch.cat([lx32.cpu().reshape(-1), ly_cpu.reshape(-1), off.reshape(-1), lab32.cpu()]), dev)
        lx_d, ly_d, off_d, lab_d = meta[:N], meta[N:2 * N], meta[2 * N:3 * N], meta[3 * N:]
    else:
        max_l = 0
        if not lx32.is_cuda:
            _validate_meta(lx32, None, None, T, V)
        lx_d = _h2d_async(lx32.cpu().reshape(-1), dev) if not lx32.is_cuda else lx32
        ly_d = off_d = lab_d = None
        meta = lx_d
    gh = _vp(graph) if (graph and c_den != 0.0) else _vp(0)
    ws_bytes = _lib.crf_workspace_bytes(gh, N, T, V, max_l)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    if _POISON_WS:
        ws.fill_(0xFF)   # every float / double / int32 of the workspace reads as NaN / -1
    grad = torch.empty(logits.shape, dtype=torch.float32, device=dev)
    out = torch.empty(1 + 3 * N, dtype=torch.float32, device=dev)
    invalid = torch.empty(N, dtype=torch.int32, device=dev) if c_ctc != 0.0 else None
    loss, c_alpha, c_beta, c_ctc_t = out[:1], out[1:1 + N], out[1 + N:1 + 2 * N], out[1 + 2 * N:]
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        if fused:
            rc = _lib.crf_loss_fwd_bwd_logits(gh, _ptr(logits), _FUSED_DTYPES[logits.dtype], _ptr(lab_d), _ptr(off_d), _ptr(lx_d),
                                              _ptr(ly_d), N, T, V, max_l, c_den, c_ctc, _ptr(grad), _ptr(loss), _ptr(c_alpha),
                                              _ptr(c_beta), _ptr(c_ctc_t), _ptr(invalid), _ptr(ws), ws_bytes, _vp(stream))
        else:
            rc = _lib.crf_loss_fwd_bwd(gh, _ptr(logits), _ptr(lab_d), _ptr(off_d), _ptr(lx_d), _ptr(ly_d),
                                       N, T, V, max_l, c_den, c_ctc, _ptr(grad), _ptr(loss), _ptr(c_alpha),
                                       _ptr(c_beta), _ptr(c_ctc_t), _ptr(invalid), _ptr(ws), ws_bytes, _vp(stream))
    _check(rc)
    if c_den != 0.0 and c_ctc != 0.0:
        _warn_if_serial((dev.index, int(stream)))
    del meta
    extras = dict(costs_alpha=c_alpha, costs_beta=c_beta, costs_ctc=c_ctc_t, invalid=invalid) if want_costs else {}
    return loss, grad, extras


def gpu_den(logits: torch.Tensor, grad_net: torch.Tensor, input_lengths: torch.Tensor,
            costs_alpha: torch.Tensor, costs_beta: torch.Tensor) -> None:
    """Same signature and in-place outputs as the reference's ``_C.gpu_den`` (binding.cpp:65-84)."""
    _, g, ex = loss_fwd_bwd(logits.contiguous(), None, input_lengths, None, 1.0, 0.0, graph_for(logits.device), True)
    grad_net.copy_(g)
    costs_alpha.copy_(ex["costs_alpha"])
    costs_beta.copy_(ex["costs_beta"])


def gpu_ctc(probs: torch.Tensor, grads: torch.Tensor, labels: torch.Tensor, label_sizes: torch.Tensor,
            sizes: torch.Tensor, minibatch_size: int, costs: torch.Tensor, blank_label: int = 0) -> None:
    """Same signature as the reference's ``_C.gpu_ctc`` (binding.cpp:86-117): probs/grads are
    [T,N,V] (the reference's transposed layout, __init__.py:70), costs is a CPU tensor receiving
    +loglike.  Our kernels work on [N,T,V] directly, so this mirror transposes at the edge."""
    assert blank_label == 0 and probs.size(1) == minibatch_size
    # c_ctc = -1  ->  grad = +gamma_ctc, exactly what the reference's kernel writes (:431-435)
    _, g, ex = loss_fwd_bwd(probs.transpose(0, 1).contiguous(), labels, sizes, label_sizes, 0.0, -1.0, None, True)
    grads.copy_(g.transpose(0, 1))
    costs.copy_(ex["costs_ctc"].to(costs.device))


for _kv in filter(None, os.environ.get("CRF_DEBUG", "").split(",")):   # tools only (see debug_set)
    _k, _, _v = _kv.partition("=")
    try:                                                                # a stray or stale CRF_DEBUG must not break `import ctc_crf`
        debug_set(_k.strip(), int(_v) if _v.strip() else 1)
        print(f"[ctc_crf] CRF_DEBUG: switch {_k.strip()} = {_DEBUG_SET[_k.strip()]} (tools / tests only: changes product behaviour)", file=sys.stderr)
    except (ValueError, RuntimeError) as _e:
        print(f"[ctc_crf] CRF_DEBUG: ignored {_kv!r}: {_e}", file=sys.stderr)
