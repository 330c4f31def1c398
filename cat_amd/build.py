"""cat_amd/build.py -- compile the gfx950 HIP library in-tree (cat_amd/lib/libctc_crf_hip.so).

`python -m cat_amd.build [--force] [-v] [--only k_fac.hip,...]`.  hipcc cross-compiles without a GPU.  One translation unit per kernel
family (csrc/k_*.hip) plus the host side (crf_host.hip) and the graph compiler (fst_graph.cpp, res_layout.cpp), compiled IN PARALLEL into
objects under cat_amd/lib/obj[-<tag>]/ and linked with -z defs (a kernel instantiation the host launches but no family instantiates is a
link error).  Objects are rebuilt when their source or any header is newer.  The .so is git-ignored but travels with gpurun snapshots.

  CRF_BUILD_OUT   another output path (A/B builds: tools/build_ab.sh); objects then go to <out>.obj/
  CRF_BUILD_DEFS  extra -D switches; with -DCRF_TIMING the families are compiled as ONE unit (csrc/crf_unity.hip: the stamp buffer is one
                  device global)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
FAMILIES = ["k_fac_768.hip", "k_fac_pair2.hip", "k_fac_1024.hip", "k_batch.hip", "k_grad.hip", "k_res.hip", "k_chain.hip", "k_robust.hip"]   # slowest first
UNITS = FAMILIES + ["crf_host.hip", "res_layout.cpp", "fst_graph.cpp"]
HEADERS = [os.path.join(CSRC, h) for h in ("crf_internal.h", "crf_device.h", "crf_kernels_decl.h", "k_res_common.h", "k_fac_body.h")] + \
          [os.path.join(os.path.dirname(HERE), "include", "ctc_crf_hip.h")]
OUT = os.environ.get("CRF_BUILD_OUT") or os.path.join(HERE, "lib", "libctc_crf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SRCS = [os.path.join(CSRC, u) for u in UNITS]
DEPS = SRCS + HEADERS


def _objdir() -> str:
    return os.path.join(HERE, "lib", "obj") if not os.environ.get("CRF_BUILD_OUT") else OUT + ".obj"


def _compile(unit: str, defs, verbose: bool, force: bool) -> str:
    src = os.path.join(CSRC, unit)
    obj = os.path.join(_objdir(), os.path.splitext(unit)[0] + ".o")
    stamp = obj + ".defs"
    want = " ".join(defs)
    fresh = (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want and
             all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in [src] + HEADERS))
    if fresh:
        return obj
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-pass-failed", *defs, "-c", src, "-o", obj]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    open(stamp, "w").write(want)
    return obj


def build(force: bool = False, verbose: bool = False, jobs: int = 0) -> str:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    os.makedirs(_objdir(), exist_ok=True)
    defs = os.environ.get("CRF_BUILD_DEFS", "").split()
    units = list(UNITS)
    if any(d.startswith("-DCRF_TIMING") for d in defs):
        units = ["crf_unity.hip", "res_layout.cpp", "fst_graph.cpp"]
    jobs = jobs or int(os.environ.get("CRF_BUILD_JOBS", "0")) or min(len(units), os.cpu_count() or 4)
    with ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda u: _compile(u, defs, verbose, force), units))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs", *objs, "-o", OUT])
    return OUT


def assemble(path: str, units=None) -> str:
    """gfx950 assembly (device side only) of the kernel families, concatenated into `path` -- what tools/isa_*.py and
    tests/test_isa_checks.py read; the units are assembled in parallel."""
    units = list(units or FAMILIES)
    defs = os.environ.get("CRF_BUILD_DEFS", "").split()

    def one(unit):
        tmp = f"{path}.{os.path.splitext(unit)[0]}.s"
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-pass-failed", *defs,
                        os.path.join(CSRC, unit), "-o", tmp], check=True, stderr=subprocess.DEVNULL)
        return tmp

    with ThreadPoolExecutor(min(len(units), os.cpu_count() or 4)) as ex:
        parts = list(ex.map(one, units))
    with open(path, "w") as out:
        for p in parts:
            out.write(open(p).read())
            os.remove(p)
    return path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
