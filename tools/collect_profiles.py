#!/usr/bin/env python3
"""Copy what tools/gpu_profiles.sh measured (gpurun_out/profiles/) into profiles/ (tracked).

Usage: python tools/collect_profiles.py [tag]        (tag defaults to r02)
Writes
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command
  profiles/<tag>_bench_profiled.json the bench line printed by that profiled run
  profiles/<tag>_trimmed_kernel_stats.csv  the same trace of a batch with 20 % of its mates trimmed (tools/ragged_rate.py 10000000 100 0.8)
  profiles/pmc_counters.json        per-launch counters of the classify kernels, stamped with the commit and with
                                    the hash of the kernel sources they were taken on; bench.py uses the file only
                                    when that hash equals the hash of the sources it runs (no stale traffic figures)
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "profiles")
KERNEL_SOURCES = ["classify_uni.hpp", "classify_uni_plan.inc", "classify_uni_loads.inc", "classify_uni_staging.inc", "classify_uni_tiles.inc", "classify_uni_rounds.inc",
                  "classify_uni_vote.inc", "classify_uni_anchored.inc", "classify_uni_sparse.inc", "classify_common.hpp", "classify.hip", "anchor_verdict.hip", "kmer_device.hpp", "shark_internal.hpp", "lds_table.hpp"]    # same list as bench.py


def kernel_src_sha():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "shark_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def newest(pattern):
    fs = glob.glob(pattern)
    if not fs:
        raise SystemExit("missing " + pattern)
    return max(fs, key=os.path.getmtime)


raw = json.load(open(os.path.join(src, "counters_raw.json")))
stats = newest(src + "/kt/*/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag))
if glob.glob(src + "/kt_trimmed/*/*_kernel_stats.csv"):     # tools/ragged_rate.py 10000000 100 0.8: rows of the classify and pre-pass kernels
    rows = list(csv.reader(open(newest(src + "/kt_trimmed/*/*_kernel_stats.csv"))))
    with open(os.path.join(ROOT, "profiles", "%s_trimmed_kernel_stats.csv" % tag), "w") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if any(x in r[0] for x in ("classify", "uniform_check", "class_", "anchor_verdict")):
                w.writerow(r)
bench = json.loads([l for l in open(os.path.join(src, "kt.json")) if l.startswith("{")][-1])
json.dump(bench, open(os.path.join(ROOT, "profiles", "%s_bench_profiled.json" % tag), "w"), indent=1)
avg_ms = {}
for r in csv.DictReader(open(stats)):
    if "classify" in r["Name"] or "anchor_verdict" in r["Name"]:
        avg_ms[r["Name"].split("(")[0].replace("void shk::", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)

commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "shark_amd/csrc"], capture_output=True, text=True).stdout.strip()
out = {
    "commit": commit + ("+dirty" if dirty else ""), "kernel_src_sha": kernel_src_sha(), "k": 17, "bf_log2": 33,
    "how": "tools/gpu_profiles.sh -> bench.py --profile-passes: rocprofv3 --pmc <one counter set per child run> --kernel-trace; per launch of the kernel that "
           "does the work (10 M pairs 2x150 bp; the last of three launches); FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (uncorrected)",
    "kernel_stats": {k: {"calls": c, "avg_ms": round(ms, 4)} for k, (c, ms) in avg_ms.items()},
    "workloads": {},
}
for name, c in raw.items():
    # bench.py --profile-passes: {workload: {"kernel": ..., counter: value per launch, ...}}
    if not c:
        continue
    if name == "valu_mix":      # shk_measure_valu_mix's kernel: VALU instructions per wave-iteration by waves per SIMD
        out["valu_mix_instructions_per_wave_iteration"] = c
        continue
    if name == "lookup_calibration":   # the FETCH_SIZE calibration on a known number of random lookups (bench.py fetch_calibration)
        out["lookup_calibration"] = c
        continue
    e = {"kernel": c.get("kernel"), "kernel_reported_by_library": c.get("kernel_reported_by_library"), "pairs": 10_000_000, "n_assoc": c.get("n_assoc")}
    if c.get("kernels"):       # a launch whose work is split over two kernels: the counters below are their sums, this is the split
        e["kernels"] = c["kernels"]
    for key in ("FETCH_SIZE", "WRITE_SIZE"):
        if key in c:
            e[key + "_KB"] = c[key]
    for key in ("TCC_HIT_sum", "TCC_MISS_sum", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_THREAD_CYCLES_VALU", "SQ_INST_CYCLES_VMEM"):
        if key in c:
            e[key] = c[key]
    if "SQ_INSTS_VALU" in e:
        e["valu_per_pair"] = round(e["SQ_INSTS_VALU"] / e["pairs"], 1)
        e["salu_per_pair"] = round(e.get("SQ_INSTS_SALU", 0) / e["pairs"], 1)
        if "SQ_THREAD_CYCLES_VALU" in e:
            e["active_lanes_per_valu"] = round(e["SQ_THREAD_CYCLES_VALU"] / e["SQ_INSTS_VALU"], 1)
    if "TCC_MISS_sum" in e:
        e["memory_side_requests_per_pair"] = round(e["TCC_MISS_sum"] / e["pairs"], 1)
    out["workloads"][name] = e
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
