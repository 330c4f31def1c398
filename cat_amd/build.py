"""cat_amd/build.py -- compile the gfx950 HIP library in-tree (cat_amd/lib/libctc_crf_hip.so).

`python -m cat_amd.build [--force]`.  hipcc cross-compiles without a GPU.  The .so is git-ignored
but travels with gpurun snapshots."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, "csrc", f) for f in ("fst_graph.cpp", "res_layout.cpp", "crf_kernels.hip")]
DEPS = SRCS + [os.path.join(HERE, "csrc", "crf_internal.h"), os.path.join(os.path.dirname(HERE), "include", "ctc_crf_hip.h")]
OUT = os.environ.get("CRF_BUILD_OUT") or os.path.join(HERE, "lib", "libctc_crf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", *os.environ.get("CRF_BUILD_DEFS", "").split(), *SRCS, "-o", OUT]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
