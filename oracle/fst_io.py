"""oracle/fst_io.py -- independent OpenFst `vector`/`standard` binary reader for the ORACLE side.

TEST INFRASTRUCTURE ONLY (see oracle/crf_oracle.c header).  The product has its own reader in
cat_amd/csrc/fst_graph.cpp; tests cross-check the two.

Restates what the reference obtains through OpenFst 1.6.7 (un-vendored, fetched by
src/ctc_crf/Makefile:10-14) at its call sites src/ctc_crf/gpu_den/fst_read.cc:23-60:
``StdVectorFst::Read``, ``NumStates``, ``Start``, ``Final``, ``StateIterator``, ``ArcIterator``.
Binary layout (OpenFst FstHeader + VectorFst body, little endian):
  int32 magic 0x7eb2fdd6 | string fsttype | string arctype | int32 version | int32 flags |
  uint64 properties | int64 start | int64 numstates | int64 numarcs |
  [symbol tables if flags&1 / flags&2] |
  per state: float32 final | int64 narcs | narcs x {int32 ilabel, int32 olabel, float32 w, int32 next}
Conventions applied exactly as fst_read.cc:40-60 does:
  label = ilabel - 1, weight = -cost, start_weight[start] = 0 else -inf,
  end_weight[s] = -Final(s) if Final(s) != Zero (=+inf) else -inf.
"""
import struct
from typing import Dict

import numpy as np

FST_MAGIC = 0x7EB2FDD6
SYMTAB_MAGIC = 2125658996


def _read_string(buf: bytes, off: int):
    (n,) = struct.unpack_from("<i", buf, off)
    off += 4
    return buf[off:off + n], off + n


def _skip_symbol_table(buf: bytes, off: int) -> int:
    (magic,) = struct.unpack_from("<i", buf, off)
    if magic != SYMTAB_MAGIC:
        raise ValueError("bad symbol table magic")
    off += 4
    _, off = _read_string(buf, off)
    _avail, size = struct.unpack_from("<qq", buf, off)
    off += 16
    for _ in range(size):
        _, off = _read_string(buf, off)
        off += 8
    return off


def read_fst(path: str) -> Dict[str, np.ndarray]:
    """Return dict(S, A, src, dst, lab, w, start_w, end_w) in the reference's conventions.

    Arcs are ordered by (source state ascending, file order) = the iteration order of
    fst_read.cc:42-60."""
    with open(path, "rb") as f:
        buf = f.read()
    off = 0
    (magic,) = struct.unpack_from("<I", buf, off)
    off += 4
    if magic != FST_MAGIC:
        raise ValueError(f"{path}: not an OpenFst binary (magic {magic:#x})")
    fsttype, off = _read_string(buf, off)
    arctype, off = _read_string(buf, off)
    if fsttype != b"vector" or arctype != b"standard":
        raise ValueError(f"{path}: need vector/standard, got {fsttype!r}/{arctype!r}")
    version, flags = struct.unpack_from("<ii", buf, off)
    off += 8
    off += 8  # properties
    start, nstates, _narcs_hdr = struct.unpack_from("<qqq", buf, off)
    off += 24
    if flags & 1:
        off = _skip_symbol_table(buf, off)
    if flags & 2:
        off = _skip_symbol_table(buf, off)
    src, dst, lab, w = [], [], [], []
    start_w = np.full(nstates, -np.inf, dtype=np.float32)
    end_w = np.full(nstates, -np.inf, dtype=np.float32)
    if 0 <= start < nstates:
        start_w[start] = 0.0
    for s in range(nstates):
        final, narcs = struct.unpack_from("<fq", buf, off)
        off += 12
        if final != float("inf"):
            end_w[s] = np.float32(-final)
        for _ in range(narcs):
            il, _ol, cost, nxt = struct.unpack_from("<iifi", buf, off)
            off += 16
            src.append(s)
            dst.append(nxt)
            lab.append(il - 1)
            w.append(-cost)
    return dict(
        S=int(nstates), A=len(src), start=int(start),
        src=np.asarray(src, dtype=np.int32), dst=np.asarray(dst, dtype=np.int32),
        lab=np.asarray(lab, dtype=np.int32), w=np.asarray(w, dtype=np.float32),
        start_w=start_w, end_w=end_w,
    )
