"""CPU tests (no GPU): the C-ABI library loads, exports every symbol include/*.h declares, and its
host-side logic (FST reader, graph compiler, error paths) behaves -- no compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import fst_io
from cat_amd.den_lm import synth_den_lm
from tests.conftest import ROOT
from tests.util import crf_env


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "ctc_crf_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(crf_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import ctc_crf
    core = ctc_crf._C
    lib = ctypes.CDLL(core.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 11
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ctc_crf_hip.h but not exported"
    assert set(syms) == set(core.EXPORTED_SYMBOLS)
    assert "gfx950" in core.version()


def test_python_surface_matches_reference_names():
    """The names CAT imports (cat/ctc/train.py:118,137) and the _C-level mirrors (binding.cpp:120-126)."""
    import ctc_crf
    for n in ("CTC_CRF_LOSS", "WARP_CTC_LOSS", "CRFContext", "_CTC_CRF", "_WARP_CTC_GPU", "ctc_crf_loss", "__version__"):
        assert hasattr(ctc_crf, n)
    for n in ("gpu_den", "gpu_ctc", "init_env", "release_env"):
        assert hasattr(ctc_crf._C, n)
    crit = ctc_crf.CTC_CRF_LOSS()
    assert crit.lamb == 0.1 and crit.size_average is True


def test_error_paths_without_gpu(tmp_path):
    import ctc_crf
    core = ctc_crf._C
    out = ctypes.c_void_p()
    rc = core._lib.crf_graph_create(os.fsencode(os.path.join(str(tmp_path), "nope.fst")), 0, ctypes.byref(out))
    assert rc == 1 and b"cannot open" in core._lib.crf_last_error()
    bad = os.path.join(str(tmp_path), "bad.fst")
    open(bad, "wb").write(b"\x00" * 64)
    assert core._lib.crf_graph_create(os.fsencode(bad), 0, ctypes.byref(out)) == 2
    with pytest.raises(RuntimeError):
        ctc_crf.CRFContext(os.path.join(str(tmp_path), "nope.fst"), 0)  # same message path as the reference (:154-156)
    # a truncated but otherwise valid file
    p = os.path.join(str(tmp_path), "t.fst")
    synth_den_lm(8, 10, 3, seed=0, path=p)
    data = open(p, "rb").read()
    open(p, "wb").write(data[: len(data) // 2])
    assert core._lib.crf_graph_create(os.fsencode(p), -1, ctypes.byref(out)) == 2


def test_graph_compiler_host_only(tmp_path, golden_dir):
    """crf_graph_create(device=-1) builds all tables on the host: check them against the oracle's reader."""
    import ctc_crf
    core = ctc_crf._C
    h = core.compile_graph_host_only(os.path.join(golden_dir, "den_lm_fixture.fst"))
    d = core.graph_dims(h)
    assert (d["S"], d["A"], d["P"], d["max_label"]) == (9, 24, 9, 4)  # every state has one in-label -> P == S
    core._lib.crf_graph_destroy(ctypes.c_void_p(h))
    p = os.path.join(str(tmp_path), "big.fst")
    g = synth_den_lm(72, 2048, 24, seed=0, path=p)
    r = fst_io.read_fst(p)
    h = core.compile_graph_host_only(p)
    st = core.graph_stats(h)
    assert st["S"] == r["S"] == 4097 and st["A"] == r["A"] == g["A"]
    pairs = len(set(zip(r["dst"].tolist(), r["lab"].tolist())))
    assert st["P"] == pairs
    assert st["max_in_deg"] == np.bincount(r["dst"]).max() and st["max_out_deg"] == np.bincount(r["src"]).max()
    assert st["A"] <= st["fwd_ell_arcs"] <= 1.35 * st["A"] and st["A"] <= st["bwd_ell_arcs"] <= 1.35 * st["A"]
    assert st["fwd_bank_conflicts"] < 0.1 * st["A"]  # the column assignment leaves < 10% of gathers on a shared bank
    core._lib.crf_graph_destroy(ctypes.c_void_p(h))
    ws = core._lib.crf_workspace_bytes(ctypes.c_void_p(0), 4, 100, 72, 10)
    assert ws > 4 * 100 * 72 * 4


def test_register_resident_layouts_host_only(tmp_path, golden_dir):
    """The two register-resident layouts, built on the host (device = -1): the generic one fits the metric graph
    into K = 2 CUs per recursion; the factored one recognises the T o LM structure without being told (every LM
    history appears as (g, blank) and (g, token): H matched pairs, one fused backward row per pair), halves the
    arc slots and fits ONE CU per recursion; graphs without that structure keep the generic layout."""
    import ctc_crf
    core = ctc_crf._C
    p = os.path.join(str(tmp_path), "m.fst")
    H, d = 2048, 24
    synth_den_lm(72, H, d, seed=0, path=p)
    h = core.compile_graph_host_only(p)
    st = core.graph_stats(h)
    core._lib.crf_graph_destroy(ctypes.c_void_p(h))
    cap = 512 * 30 * 4                                   # arc slots of one workgroup: 512 threads x 30 chunks x 4
    assert st["res_K"] == 2 and st["A"] <= st["res_fwd_slots"] <= 2 * cap and st["A"] <= st["res_bwd_slots"] <= 2 * cap
    assert st["fac"] == 1
    assert H - 2 <= st["fac_matched_pairs"] <= H and H - 2 <= st["fac_fused_rows"] <= H
    assert st["fac_fwd_slots"] <= cap and st["fac_bwd_slots"] <= cap            # K = 1
    assert st["fac_fwd_slots"] < 0.6 * st["res_fwd_slots"] and st["fac_bwd_slots"] < 0.6 * st["res_bwd_slots"]
    assert st["fac_Gf"] * 4 <= 65536 and st["fac_Gb"] * 4 <= 65536              # 16-bit LDS byte offsets
    # the reference's own 9-state test graph and random graphs: no (tail, main) structure -> generic layout only
    for name in ["den_lm_fixture.fst"] + [f"rand{i}.fst" for i in range(6)]:
        h = core.compile_graph_host_only(os.path.join(golden_dir, name))
        st = core.graph_stats(h)
        core._lib.crf_graph_destroy(ctypes.c_void_p(h))
        assert st["res_K"] >= 1
        assert st["fac"] in (0, 1)
    # CRF_NO_FACTORED keeps the generic layout (read at graph creation)
    with crf_env(CRF_NO_FACTORED=1):
        h = core.compile_graph_host_only(p)
        assert core.graph_stats(h)["fac"] == 0
        core._lib.crf_graph_destroy(ctypes.c_void_p(h))


def test_factored_layout_takes_an_estimated_ngram_graph(tmp_path):
    """A den_lm estimated from text (cat_amd.den_lm.prep_den_lm) has rows far longer than one lane's 80 arcs -- the
    low-order history states are entered from hundreds of states.  The factored layout must still take it (rows cut
    into pieces on adjacent lanes) instead of falling back to the streaming kernels; host only."""
    import numpy as np
    import ctc_crf
    from cat_amd import den_lm
    core = ctc_crf._C
    V = 40
    rng = np.random.default_rng(3)
    trans = rng.dirichlet(np.ones(V - 1) * 0.05, size=(V, V))
    seqs = []
    for _ in range(1200):
        L, sq, a, b = int(rng.integers(8, 30)), [], 0, 0
        for _ in range(L):
            c = 1 + int(rng.choice(V - 1, p=trans[a, b])); sq.append(c); a, b = b, c
        seqs.append(sq)
    p = os.path.join(str(tmp_path), "est.fst")
    g = den_lm.prep_den_lm(seqs, V, p, 4, 3, 150, selection="count")
    h = core.compile_graph_host_only(p)
    st = core.graph_stats(h)
    core._lib.crf_graph_destroy(ctypes.c_void_p(h))
    assert st["S"] == g["S"] and st["A"] == g["A"]
    assert st["max_in_deg"] > 80                          # longer than a lane
    assert st["fac"] == 1 and st["fac_fwd_slots"] > 0 and st["fac_bwd_slots"] > 0
    assert st["fac_matched_pairs"] >= (st["S"] - 1) // 2 - 2


def test_utterance_minor_block_decode():
    """crf_batch_frame_kernel finds its (utterance group, direction) combo and its chunk from the block id (crf_internal.h:
    bat_decode, shared by the kernel and this check): for every combo count and grid the host launches -- nslot is at least
    ceil(#combos / 8) -- every (combo, chunk) must be taken by exactly one workgroup."""
    import ctc_crf
    core = ctc_crf._C
    core._lib.crf_debug_decode_check.argtypes = [ctypes.c_int, ctypes.c_int]
    for ncombo in range(2, 66, 2):                              # 2 directions x 1..32 groups
        for nslot in {(ncombo + 7) // 8, (ncombo + 7) // 8 + 1, 7, 16, 33, 128}:
            if nslot >= (ncombo + 7) // 8:
                rc = core._lib.crf_debug_decode_check(nslot, ncombo)
                assert rc == 0, (ncombo, nslot, core._lib.crf_last_error().decode())


def test_which_kernels_take_which_graph(tmp_path):
    """crf_den_kernels on host-only graphs: the benchmark graph -> factored (one CU per recursion); 1.5 x - 2 x its size -> factored
    over two CUs per recursion (without that: generic register-resident over K = 4 CUs); beyond four CUs' registers -> utterance-minor, or round 1's streaming kernels with
    CRF_NO_BATCH=1; with CRF_NO_FACTORED=1 the benchmark graph takes the generic layout (K = 2)."""
    import ctc_crf
    core = ctc_crf._C

    def which(H, d, **env):
        with crf_env(**env):
            p = os.path.join(str(tmp_path), f"g{H}_{d}.fst")
            if not os.path.exists(p):
                synth_den_lm(72, H, d, 0, path=p)
            h = core.compile_graph_host_only(p)
            st = core.graph_stats(h)
            k = core.den_kernels(h, 64, 1500, 72)
            core._lib.crf_graph_destroy(ctypes.c_void_p(h))
            return k, st

    k, st = which(2048, 24)
    assert k == "factored" and st["fac"] == 1
    k, st = which(2048, 24, CRF_NO_FACTORED=1)
    assert k == "resident" and st["res_K"] == 2
    k, st = which(3072, 24)                                       # 1.5 x: the factored layout over TWO CUs per recursion
    assert k == "factored" and st["fac"] == 1 and st["fac_geom"] == 3
    k, st = which(3072, 24, CRF_FAC_NO_K2=1)                      # ... without it: the generic layout, K = 4
    assert k == "resident" and st["fac"] == 0 and st["res_K"] == 4
    k, st = which(4096, 24)                                       # 8193 states, 208 k arcs
    assert k == "factored" and st["fac_geom"] == 3 and st["S"] == 8193
    k, st = which(4096, 24, CRF_FAC_NO_K2=1)                      # (needs the 64 KiB state-vector buffers)
    assert k == "resident" and st["res_K"] == 4


def test_factored_layouts_emulated_on_the_host(tmp_path, golden_dir):
    """crf_debug_fac_emulate walks the factored register-resident layout's tables the way the kernels do -- packed arc words of
    every CU / wave / lane, slice ends, butterfly over multi-lane rows, row constants (registers and LDS table), implicit and
    tabulated entries, second copy, rowless states -- for a few frames of random emissions in fp64: forward and backward sums
    through the layout = the sum through the graph's own row tables, and for every frame the grad pass's label-sorted pair lists
    applied to the stored rows give that path mass again.  Every geometry (768 threads with the constants in registers
    / in the table, 512 threads, two CUs per recursion), with and without the second copy; graphs: small T o LM, the reference's
    9-state fixture, an estimated n-gram graph with multi-lane rows, the benchmark graph, and a graph of 1.5 x its size (which
    takes two CUs by itself).  With two CUs each has a private vector that receives only what the kernel fetches; the negative
    control (the list of L / A entries dropped) must poison the sums."""
    import math
    import ctc_crf
    from cat_amd import den_lm
    core = ctc_crf._C

    def emu(path, T=5, **env):
        with crf_env(**env):
            h = core.compile_graph_host_only(path)
            st = core.graph_stats(h)
            r = core.debug_fac_emulate(h, T, 7)
            core._lib.crf_graph_destroy(ctypes.c_void_p(h))
            return st["fac_geom"], r

    def agree(r):
        # (backward: start weight x arc weight of the rowless states is ONE fp32 table constant -- a rounding of 6e-8 on those terms
        # when the start weight is not 1, i.e. in re-gauged graphs)
        return all(math.isfinite(x) and x > 0 for x in r) and abs(r[1] - r[0]) <= 1e-9 * r[0] and abs(r[2] - r[0]) <= 1e-6 * r[0]

    small = os.path.join(str(tmp_path), "small.fst")
    synth_den_lm(24, 96, 8, seed=13, path=small)
    V = 40
    rng = np.random.default_rng(3)
    trans = rng.dirichlet(np.ones(V - 1) * 0.05, size=(V, V))
    seqs = []
    for _ in range(1200):
        L, sq, a, b = int(rng.integers(8, 30)), [], 0, 0
        for _ in range(L):
            c = 1 + int(rng.choice(V - 1, p=trans[a, b])); sq.append(c); a, b = b, c
        seqs.append(sq)
    est = os.path.join(str(tmp_path), "est.fst")
    den_lm.prep_den_lm(seqs, V, est, 4, 3, 150, selection="count")
    # the planner's own choice: 1024 threads x 15 chunks (geometry 4) unless MOST of a LARGE graph's arcs sit in rows longer than a
    # lane (more than a fifth and more than 20 000 of them: a den_lm estimated from a large corpus), which keeps the 768-thread table
    # geometry -- measured in round 4: S = 3 006 (14.9 k such arcs) faster on 1024 threads, S = 6 836 (27.3 k) slower
    assert emu(small)[0] == 4 and emu(est)[0] == 4

    def corpus(V, n, seed):                                       # the same second-order source, sampled in bulk (40 000 sentences in a second)
        rng = np.random.default_rng(seed)
        cum = np.cumsum(rng.dirichlet(np.ones(V - 1) * 0.05, size=(V, V)), axis=-1)
        out = []
        for L, u in zip(rng.integers(10, 40, size=n), rng.random((n, 40))):
            sq, a, b = [], 0, 0
            for k in range(int(L)):
                c = 1 + min(int(np.searchsorted(cum[a, b], u[k])), V - 2); sq.append(c); a, b = b, c
            out.append(sq)
        return out

    big = os.path.join(str(tmp_path), "est_big.fst")
    den_lm.prep_den_lm(corpus(72, 40000, 0), 72, big, 4, 3, 2000, selection="count")
    g, r = emu(big, T=3)
    assert g == 1 and agree(r)                                    # tens of thousands of arcs in multi-lane rows: 768 threads, table geometry
    for path in (small, os.path.join(golden_dir, "den_lm_fixture.fst"), est):
        for env, geom in (({}, (4, 1)), ({"CRF_FAC_THREADS": 1024}, (4,)), ({"CRF_FAC_THREADS": 768}, (0, 1)), ({"CRF_FAC_NO_RCL": 1}, (0,)), ({"CRF_FAC_RCL": 1}, (1,)), ({"CRF_FAC_K2": 1}, (3,)), ({"CRF_FAC_K2": 1, "CRF_FAC_THREADS": 1024}, (5,)),
                          ({"CRF_FAC_THREADS": 512}, (2,)), ({"CRF_FAC_NO_DUP": 1}, (4, 1)), ({"CRF_FAC_NO_DUP": 1, "CRF_FAC_THREADS": 768}, (0, 1))):
            g, r = emu(path, **env)
            assert g in geom and agree(r), (path, env, g, r)
    try:                                                          # negative control: NaN sums, or the emulator's own checks object
        g, r = emu(small, CRF_FAC_K2=1, CRF_EMU_DROP_LIST=1)
        assert g == 3 and not agree(r)
    except RuntimeError as ex:
        assert "emulation" in str(ex)
    bench = os.path.join(str(tmp_path), "bench.fst")
    synth_den_lm(72, 2048, 24, seed=0, path=bench)
    g, r = emu(bench, T=3)
    assert g == 4 and agree(r)                                    # 1024 threads x 15 chunks: the planner's first choice
    g, r = emu(bench, T=3, CRF_FAC_THREADS=768)
    assert g == 0 and agree(r)                                    # 768 threads, row constants in registers (round 2's default)
    mid = os.path.join(str(tmp_path), "mid.fst")
    synth_den_lm(72, 3072, 24, seed=0, path=mid)
    g, r = emu(mid, T=3)
    assert g == 3 and agree(r)                                    # two CUs per recursion, 768 threads each (the planner's choice)
    g, r = emu(mid, T=3, CRF_FAC_THREADS=1024, CRF_FAC_K2=1)
    assert g == 5 and agree(r)                                    # ... 1024 threads each (round 4: built, measured slower, on request only)
    v217 = os.path.join(str(tmp_path), "v217.fst")             # the benchmark LM over 217 classes: rows of up to ~500 arcs on several lanes;
    synth_den_lm(217, 2048, 24, seed=0, path=v217)              # at 768 threads it fits ONE CU only with all 21 chunk slots holding arcs
    g, r = emu(v217, T=3)
    assert g == 4 and agree(r)                                    # (the 1024-thread geometry takes it too)
    with crf_env(CRF_FAC_THREADS=768):
        h = core.compile_graph_host_only(v217)
        st = core.graph_stats(h)
        r = core.debug_fac_emulate(h, 3, 7)
        core._lib.crf_graph_destroy(ctypes.c_void_p(h))
    assert st["fac_geom"] == 1 and st["fac_chunks"] == 21 and agree(r), (st["fac_geom"], st["fac_chunks"], r)
    from tests.util import transform_graph                       # renumbered, reordered, weight-pushed (re-gauged by the compiler), long rows
    from oracle import fst_io
    wide = os.path.join(str(tmp_path), "wide.fst")
    synth_den_lm(150, 300, 120, seed=7, path=wide)
    pushed = os.path.join(str(tmp_path), "pushed.fst")
    transform_graph(fst_io.read_fst(wide), pushed, seed=7, renumber=True, reorder=True, push=True)
    for env in ({}, {"CRF_FAC_K2": 1}, {"CRF_FAC_THREADS": 512}):
        g, r = emu(pushed, **env)
        assert g >= 0 and agree(r), (env, g, r)


def test_second_layout_for_two_utterances_emulated_on_the_host(tmp_path, golden_dir):
    """Round 5: beside a 1024-thread main layout the graph compiler builds a SECOND factored layout -- 512 threads x 30 chunks, row
    constants in the LDS table, implicit entries (HostGraph::facp) -- for the two-utterance kernel (crf_fac_pair2_kernel<.., 512, 30, ..>:
    256 registers per wave instead of the 168 its 768-thread version spills at).  It has its own rows, entries and grad-pass lists;
    the same emulation (switch emu_facp) walks it: forward / backward sums = the plain recursion, pair lists = the path mass.  Graphs
    whose main layout is a 768-thread one or lies on two CUs get none (the two-utterance kernel takes 768-thread layouts as they are)."""
    import math
    import ctc_crf
    from cat_amd import den_lm
    core = ctc_crf._C

    def emu(path, T=4, **env):
        with crf_env(**env):
            h = core.compile_graph_host_only(path)
            st = core.graph_stats(h)
            r = None
            if st["facp"]:
                with crf_env(CRF_EMU_FACP=1):
                    r = core.debug_fac_emulate(h, T, 11)
            core._lib.crf_graph_destroy(ctypes.c_void_p(h))
            return st, r

    def agree(r):
        return all(math.isfinite(x) and x > 0 for x in r) and abs(r[1] - r[0]) <= 1e-9 * r[0] and abs(r[2] - r[0]) <= 1e-6 * r[0]

    small = os.path.join(str(tmp_path), "small.fst")
    synth_den_lm(24, 96, 8, seed=13, path=small)
    bench = os.path.join(str(tmp_path), "bench.fst")
    synth_den_lm(72, 2048, 24, seed=0, path=bench)
    v217 = os.path.join(str(tmp_path), "v217.fst")
    synth_den_lm(217, 2048, 24, seed=0, path=v217)
    rng = np.random.default_rng(5)
    V = 40
    trans = rng.dirichlet(np.ones(V - 1) * 0.05, size=(V, V))
    seqs = []
    for _ in range(1200):
        L, sq, a, b = int(rng.integers(8, 30)), [], 0, 0
        for _ in range(L):
            c = 1 + int(rng.choice(V - 1, p=trans[a, b])); sq.append(c); a, b = b, c
        seqs.append(sq)
    est = os.path.join(str(tmp_path), "est.fst")
    den_lm.prep_den_lm(seqs, V, est, 4, 3, 150, selection="count")                   # multi-lane rows
    for path in (small, os.path.join(golden_dir, "den_lm_fixture.fst"), bench, v217, est):
        st, r = emu(path, T=3 if path in (bench, v217) else 4)
        assert st["fac_geom"] == 4 and st["facp"] == 1 and agree(r), (path, st["fac_geom"], st["facp"], r)
        st, r = emu(path, T=3, CRF_FAC_NO_DUP=1)
        assert st["facp"] == 1 and agree(r), (path, r)
    for env in ({"CRF_FAC_THREADS": 768}, {"CRF_FAC_K2": 1}, {"CRF_NO_FACP": 1}, {"CRF_FAC_THREADS": 512}):
        st, r = emu(small, **env)
        assert st["facp"] == 0 and r is None, env
    mid = os.path.join(str(tmp_path), "mid.fst")
    synth_den_lm(72, 3072, 24, seed=0, path=mid)                  # two CUs per recursion: no second layout
    st, r = emu(mid)
    assert st["fac_geom"] == 3 and st["facp"] == 0


def test_generic_resident_layout_emulated_on_the_host(tmp_path, golden_dir):
    """crf_debug_res_emulate: the generic register-resident layout (any graph that fits K <= 4 compute units: pair rows forward,
    state copies backward, long rows split into sub-rows with virtual copies of their entry, one produced entry per row) walked
    on the host like the kernels walk it, fp64: forward and backward sums = the recursion over the graph's arcs, every entry
    has one producer, the grad pass's pair lists give the path mass in every frame.  Graphs: random general graphs (states
    entered with several labels), the reference's fixture, T o LM small (K = 1, forced 2 and 4), an estimated n-gram graph
    (rows of hundreds of arcs), the benchmark graph (K = 2) and one of 1.5 x its size (K = 4)."""
    import math
    import ctc_crf
    from cat_amd import den_lm
    core = ctc_crf._C

    def emu(path, T=4, **env):
        with crf_env(**env):
            h = core.compile_graph_host_only(path)
            st = core.graph_stats(h)
            r = core.debug_res_emulate(h, T, 11)
            core._lib.crf_graph_destroy(ctypes.c_void_p(h))
            return st["res_K"], r

    def agree(r):
        # (backward: start weight x arc weight of the rowless states is ONE fp32 table constant -- a rounding of 6e-8 on those terms
        # when the start weight is not 1, i.e. in re-gauged graphs)
        return all(math.isfinite(x) and x > 0 for x in r) and abs(r[1] - r[0]) <= 1e-9 * r[0] and abs(r[2] - r[0]) <= 1e-6 * r[0]

    for name in ["den_lm_fixture.fst"] + [f"rand{i}.fst" for i in range(6)]:
        K, r = emu(os.path.join(golden_dir, name))
        assert K == 1 and agree(r), (name, K, r)
    small = os.path.join(str(tmp_path), "small.fst")
    synth_den_lm(24, 96, 8, seed=13, path=small)
    for mink in (1, 2, 4):
        K, r = emu(small, CRF_RES_MINK=mink)
        assert K == mink and agree(r), (mink, K, r)
    V = 40
    rng = np.random.default_rng(3)
    trans = rng.dirichlet(np.ones(V - 1) * 0.05, size=(V, V))
    seqs = []
    for _ in range(1200):
        L, sq, a, b = int(rng.integers(8, 30)), [], 0, 0
        for _ in range(L):
            c = 1 + int(rng.choice(V - 1, p=trans[a, b])); sq.append(c); a, b = b, c
        seqs.append(sq)
    est = os.path.join(str(tmp_path), "est.fst")
    den_lm.prep_den_lm(seqs, V, est, 4, 3, 150, selection="count")
    K, r = emu(est)
    assert K >= 1 and agree(r), (K, r)
    bench = os.path.join(str(tmp_path), "bench.fst")
    synth_den_lm(72, 2048, 24, seed=0, path=bench)
    K, r = emu(bench, T=3)
    assert K == 2 and agree(r)
    mid = os.path.join(str(tmp_path), "mid.fst")
    synth_den_lm(72, 3072, 24, seed=0, path=mid)
    K, r = emu(mid, T=3)
    assert K == 4 and agree(r)


def test_generic_layout_that_does_not_fit_the_lds_takes_another_family(tmp_path):
    """The generic register-resident kernels reserve two fixed 64 KiB state-vector buffers; a K = 1 layout with ~6.6 k rows and
    V = 1000 classes needs more than the CU's 160 KiB with them.  Such a call must take the next kernel family (crf_den_kernels
    != resident) instead of failing with 'graph too large' (round-2 advisor finding); a smaller V on the same graph still fits."""
    import ctc_crf
    core = ctc_crf._C
    p = os.path.join(str(tmp_path), "v1000.fst")
    synth_den_lm(1000, 3300, 5, 0, path=p)
    with crf_env(CRF_NO_FACTORED=1):
        h = core.compile_graph_host_only(p)
        st = core.graph_stats(h)
        assert st["S"] == 6601 and st["res_K"] == 1 and st["fac"] == 0
        rows = max(st["res_fwd_rows"], st["res_bwd_rows"])
        need = 2 * 65536 + 4 * (rows + 2 * ((1000 + 1 + 63) // 64 * 64) + 48)
        assert need > 160 * 1024                                  # (the layout the finding was about)
        assert core.den_kernels(h, 8, 100, 1000) != "resident"
        core._lib.crf_graph_destroy(ctypes.c_void_p(h))


def test_debug_switches_are_set_by_name_not_from_the_environment(tmp_path):
    """crf_debug_set / crf_debug_unset / crf_debug_list: the library's only switches; the process environment is never read."""
    import ctc_crf
    core = ctc_crf._C
    names = [ln.split(":")[0] for ln in core.debug_list().splitlines() if ln]
    assert "no_factored" in names and "bat_ul" in names and len(names) == len(set(names)) >= 30
    with pytest.raises(RuntimeError):
        core.debug_set("no_such_switch", 1)
    p = os.path.join(str(tmp_path), "g.fst")
    synth_den_lm(24, 64, 6, 0, path=p)

    def fac():
        h = core.compile_graph_host_only(p)
        f = core.graph_stats(h)["fac"]
        core._lib.crf_graph_destroy(ctypes.c_void_p(h))
        return f
    os.environ["CRF_NO_FACTORED"] = "1"                            # (the round-2 spelling: must have no effect any more)
    try:
        assert fac() == 1
    finally:
        del os.environ["CRF_NO_FACTORED"]
    core.debug_set("no_factored", 1)
    try:
        assert fac() == 0
    finally:
        core.debug_set("no_factored", None)
    assert fac() == 1
    csrc = os.path.join(ROOT, "cat_amd", "csrc")
    src = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h", ".cpp")))
    assert "getenv" not in src
