"""tools/pmc_summary.py TAG [ROUND] -- condense the rocprofv3 outputs of tools/gpu_round.sh (gpu_prof.sh)
(gpurun_out/) into the small summaries kept under profiles/ (ROUND: file name prefix, default "round4"):

    profiles/<ROUND>_<TAG>_bench.json          the bench line (with cpu_baseline)
    profiles/<ROUND>_<TAG>_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary
    profiles/<ROUND>_<TAG>_pmc.json            per-kernel FETCH_SIZE / WRITE_SIZE (KB per launch) and SQ counters
    profiles/pmc_traffic.json                 HBM bytes per launch of every loss kernel + the whole-path ratio, KEYED BY
                                              WORKLOAD (read by bench.py; another workload prints traffic: null)
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
ROUND = sys.argv[2] if len(sys.argv) > 2 else "round4"


def counters(tag, what):
    f = glob.glob(os.path.join(OUT, f"pmc_{what}_{tag}", "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    if not f:
        return acc
    per_dispatch = defaultdict(lambda: defaultdict(float))
    names = {}
    for r in csv.DictReader(open(f[0])):
        key = r["Dispatch_Id"]
        names[key] = r["Kernel_Name"]
        per_dispatch[key][r["Counter_Name"]] += float(r["Counter_Value"])
    for key, cs in per_dispatch.items():
        for c, v in cs.items():
            acc[names[key]][c].append(v)
    return acc


def main():
    tag = sys.argv[1]
    os.makedirs(PROF, exist_ok=True)
    shutil.copy(os.path.join(OUT, f"bench_{tag}.json"), os.path.join(PROF, f"{ROUND}_{tag}_bench.json"))
    ks = glob.glob(os.path.join(OUT, f"prof_{tag}", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(PROF, f"{ROUND}_{tag}_kernel_stats.csv"))
    traffic, sq = {}, {}
    fe, wr, s = counters(tag, "fetch"), counters(tag, "write"), counters(tag, "sq")
    def nsteps(acc, what):   # calls of the loss in that profiling pass = launches of the finalize kernel (round 5: folded into the robust grad launch)
        for name in ("crf_robust_grad_kernel", "crf_finalize_kernel"):   # (one robust grad launch per call with a denominator; finalize only where it is not folded in)
            for k, cs in acc.items():
                if name in k and what in cs:
                    return max(1, len(cs[what]))
        return 1
    for k in sorted(set(fe) | set(wr)):
        if "crf" not in k:
            continue
        traffic[k] = {}
        if "FETCH_SIZE" in fe.get(k, {}):
            v = fe[k]["FETCH_SIZE"]
            traffic[k]["FETCH_SIZE_KB_avg_per_launch"] = round(sum(v) / len(v), 1)
            traffic[k]["FETCH_SIZE_KB_per_call"] = round(sum(v) / nsteps(fe, "FETCH_SIZE"), 1)
            traffic[k]["launches_per_call"] = round(len(v) / nsteps(fe, "FETCH_SIZE"), 2)
        if "WRITE_SIZE" in wr.get(k, {}):
            v = wr[k]["WRITE_SIZE"]
            traffic[k]["WRITE_SIZE_KB_avg_per_launch"] = round(sum(v) / len(v), 1)
            traffic[k]["WRITE_SIZE_KB_per_call"] = round(sum(v) / nsteps(wr, "WRITE_SIZE"), 1)
    for k in sorted(s):
        if "crf" in k:
            sq[k] = {c: int(sum(v) / len(v)) for c, v in sorted(s[k].items())}
    doc = {
        "command": "rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline "
                   "(tools/gpu_prof.sh; FETCH_SIZE, WRITE_SIZE and the SQ counters in three separate passes)",
        "note": "FETCH_SIZE/WRITE_SIZE in KB per launch. On gfx950 FETCH_SIZE reports 1/2 of a wide (16 B/lane) coalesced "
                "stream (MI355X_MICROARCH.md HBM section): the grad kernels' row reads are such streams.",
        "traffic": traffic, "sq": sq,
    }
    json.dump(doc, open(os.path.join(PROF, f"{ROUND}_{tag}_pmc.json"), "w"), indent=1)
    # bytes per call of the loss, per kernel.  FETCH_SIZE is doubled for the kernels whose reads are wide (16 B / lane)
    # coalesced streams (MI355X_MICROARCH.md, HBM section): the grad pass reading the Q / BP / CA / CB rows.
    bench = json.load(open(os.path.join(OUT, f"bench_{tag}.json")))
    wl = bench["config"]["workload"]
    import re
    m = re.search(r"B=(\d+) per GPU, T=(\d+), V=(\d+).*S=(\d+) states, A=(\d+) arcs.*?(ragged lx|lx = T)", wl)
    key = f"B{m.group(1)}_T{m.group(2)}_V{m.group(3)}_S{m.group(4)}_A{m.group(5)}_{'ragged' if m.group(6) == 'ragged lx' else 'full'}" if m else wl
    kernels, total = {}, 0
    for k, v in traffic.items():
        wide = "crf_grad_den_kernel" in k or "crf_grad_ctc_kernel" in k
        byts = int((v.get("FETCH_SIZE_KB_per_call", 0) * (2 if wide else 1) + v.get("WRITE_SIZE_KB_per_call", 0)) * 1024)
        short = k.replace("void ", "").replace("crf::", "").split("<")[0].split("(")[0]
        kernels[short] = kernels.get(short, 0) + byts
        total += byts
    alg = bench["roofline"]["den_fwd_bwd"]["bytes"] + bench["roofline"]["den_fwd_bwd"]["bytes_num"]
    # the denominator recursions' kernel as the library names it (crf_last_den_kernel): rocprofv3's demangled name without
    # "void crf::", blanks and the argument list -- bench.py prints `traffic` only for a call that ran THIS instantiation
    den = [k for k in traffic if any(n in k for n in ("crf_fac_pair_kernel", "crf_fac2_pair_kernel", "crf_res_pair_kernel", "crf_batch_frame_kernel", "crf_batch_persist_kernel", "crf_den_pair_kernel"))]
    den_kernel = den[0].replace("void ", "").replace("crf::", "").split("(")[0].replace(" ", "") if len(den) == 1 else None
    tr = {"workload": key, "den_kernel": den_kernel, "kernels": kernels,
          "whole_path": {"pmc_bytes_per_call": total, "algorithmic_bytes": alg, "ratio": round(total / max(1, alg), 3)},
          "source": f"profiles/{ROUND}_{tag}_pmc.json: (FETCH_SIZE [x2 for the grad kernels' 16-byte row streams] + WRITE_SIZE) * 1024 bytes "
                    "per call of the loss, every launch of a kernel summed"}
    json.dump(tr, open(os.path.join(PROF, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(tr, indent=1))
    for k, v in traffic.items():
        print(k[:60], v)
    # the rest of the evidence run: points, the GPU suite's summary, soak, estimated graphs, the 1-rank torchrun line
    for f in sorted(glob.glob(os.path.join(OUT, f"pt_{tag}_*.json"))):
        if os.path.getsize(f) > 0:
            shutil.copy(f, os.path.join(PROF, f"{ROUND}_{tag}_point_" + os.path.basename(f)[len(f"pt_{tag}_"):]))
    for src, dst in ((f"pt_{tag}_estimated.txt", f"{ROUND}_{tag}_point_estimated.txt"), (f"soak_{tag}.txt", f"{ROUND}_{tag}_soak.txt"),
                     (f"torchrun1_{tag}.json", f"{ROUND}_{tag}_torchrun_1rank.json")):
        if os.path.exists(os.path.join(OUT, src)):
            shutil.copy(os.path.join(OUT, src), os.path.join(PROF, dst))
    log = os.path.join(OUT, f"pytest_{tag}.log")
    if os.path.exists(log):
        open(os.path.join(PROF, f"{ROUND}_{tag}_pytest_gpu.txt"), "w").write("".join(open(log).readlines()[-12:]))


if __name__ == "__main__":
    main()
