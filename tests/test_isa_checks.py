"""CPU tests of what can be read off the compiled gfx950 code without a GPU (hipcc cross-compiles here).

`tools/isa_check_lds_issue.py`: the inline-asm LDS reads of the numerator chains (crf_device.h lds_issue_* / lds_landed) are
asynchronous loads the compiler knows nothing about; the check proves on the ISA of THIS build that nothing touches their destination
registers between the read and its wait (round-4 advisor: checked by eye for one compiler version only).  The assembly (~90 s of hipcc)
is cached under the system temp directory, keyed by a hash of the sources and the compiler version."""
import hashlib
import importlib.util
import os
import shutil
import subprocess
import tempfile

import pytest

from tests.conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _tool():
    spec = importlib.util.spec_from_file_location("isa_check_lds_issue", os.path.join(ROOT, "tools", "isa_check_lds_issue.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _assembly(mod):
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "cat_amd", "csrc")
    for f in sorted(os.listdir(csrc)) + ["../../include/ctc_crf_hip.h"]:
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(subprocess.run([HIPCC, "--version"], capture_output=True).stdout)
    path = os.path.join(tempfile.gettempdir(), f"crf_isa_{h.hexdigest()[:16]}.s")
    if not os.path.exists(path):
        tmp = path + f".{os.getpid()}.tmp"
        mod.assemble(tmp)
        os.replace(tmp, path)
    return path


GOOD = """_Zkernel:
	;;#ASMSTART
	ds_read_b64 v[4:5], v9
	;;#ASMEND
	s_and_saveexec_b64 s[0:1], s[2:3]
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_exp_f32_e32 v7, v6
.LBB0_2:                                ;   in Loop
	s_or_b64 exec, exec, s[0:1]
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_add_f64 v[4:5], v[4:5], v[10:11]
	s_endpgm
"""


def test_checker_on_hand_made_assembly():
    mod = _tool()
    n, bad = mod.check(GOOD.split("\n"))
    assert n == 1 and not bad, bad
    # the compiler "re-uses" half of the destination for something else before the wait
    n, bad = mod.check(GOOD.replace("v_exp_f32_e32 v7, v6", "v_mov_b32_e32 v5, v6").split("\n"))
    assert n == 1 and len(bad) == 1 and "destination touched" in bad[0]
    # ... or reads it (a copy / a spill of the not-yet-landed value)
    n, bad = mod.check(GOOD.replace("v_exp_f32_e32 v7, v6", "v_mov_b32_e32 v7, v4").split("\n"))
    assert len(bad) == 1
    # control leaves the region before the wait
    n, bad = mod.check(GOOD.replace("s_cbranch_execz .LBB0_2", "s_cbranch_execz .LBB0_9").split("\n"))
    assert len(bad) == 1 and "leaves the region" in bad[0]
    # no wait at all
    n, bad = mod.check(GOOD.replace("s_waitcnt lgkmcnt(0)", "s_nop 0").split("\n"))
    assert len(bad) == 1 and "no `s_waitcnt" in bad[0]


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
@pytest.mark.timeout(900)
def test_inline_lds_reads_of_this_build():
    """Every inline `ds_read` of every kernel of the product source, as compiled by the hipcc of this image."""
    mod = _tool()
    n, bad = mod.check(open(_assembly(mod)).read().split("\n"))
    assert n >= 100, n          # (the numerator chains of four label-length classes, forward and backward: 176 on ROCm 7.2)
    assert not bad, "\n".join(bad[:10])


def _kernel_meta(path):
    """{mangled kernel name: {vgpr_count, vgpr_spill_count, ...}} from the amdhsa.kernels metadata of the assembly."""
    import re
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.match(r"\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return out


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
@pytest.mark.timeout(900)
def test_register_budgets_the_schedule_relies_on():
    """Occupancy is part of the schedule, and ONE register changes it (VGPRs are granted in eights, 512 per SIMD lane):
    - the metric graph's grad den kernel (256 threads: one wave per SIMD and workgroup) at <= 160 and the numerator chains of short label
      sequences (512 threads: two waves per SIMD) at <= 96: two grad workgroups and a chain workgroup share a CU, 2 x 160 + 2 x 96 = 512.
      At 153 (-> 160) and 97 (-> 104) the chains of B >= 128 ran BEHIND the grad pass: 4.8 -> 5.6 ms per step, found by bisecting
      (profiles/round5_ab_one_register.txt);
    - the 1024-thread recursions at <= 128 (4 waves per SIMD), the 768-thread ones at <= 168 (3), the 512 x 30 two-utterance geometry at <= 256,
      none of them spilling inside the frame loop (the few spilled prologue constants are counted and capped)."""
    mod = _tool()
    meta = _kernel_meta(_assembly(mod))
    gd = meta["_ZN3crf19crf_grad_den_kernelILi1ELi1ELi256ELi32ELi5ELi1EEEvNS_10LossParamsE"]
    assert gd["vgpr_count"] <= 160 and gd["vgpr_spill_count"] == 0, gd
    ch = meta["_ZN3crf19crf_ctc_pair_kernelILi1EEEvNS_10LossParamsE"]          # (label sequences of the metric shape: one register set per thread)
    assert ch["vgpr_count"] <= 96 and ch["vgpr_spill_count"] == 0, ch
    pairs = {k: v for k, v in meta.items() if "crf_fac_pair_kernelILb" in k}
    assert len(pairs) >= 12
    for k, v in pairs.items():
        cap = 128 if "ELi1024ELi15E" in k else 168 if "ELi768E" in k else 256
        assert v["vgpr_count"] <= cap and v["vgpr_spill_count"] <= 8, (k, v)     # (a handful of spills outside the frame loop: the prologue's constants)
