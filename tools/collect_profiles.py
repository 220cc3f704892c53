#!/usr/bin/env python3
"""Copy the record run (tools/gpu_final_r1.sh -> gpurun_out/final/) into profiles/.

Usage: python tools/collect_profiles.py [tag]        (tag defaults to r01_final)
Takes the newest rocprofv3 output directory of each pass, keeps the kernel-stats CSV, and writes
pmc_traffic.json (per launch of the classify kernel; FETCH_SIZE correction as in profiles/README.md).
"""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
src = "gpurun_out/final"

def newest(pattern):
    fs = glob.glob(pattern)
    if not fs:
        raise SystemExit("missing " + pattern)
    return max(fs, key=os.path.getmtime)

def counters(path):
    acc, name = {}, ""
    for r in csv.DictReader(open(path)):
        if "classify_fast" in r["Kernel_Name"]:
            name = r["Kernel_Name"]
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, name

bench = json.loads(open(src + "/bench.json").read().strip().splitlines()[-1])
json.dump(bench, open("profiles/%s_bench.json" % tag, "w"), indent=1)
shutil.copy(newest(src + "/kt/*/*_kernel_stats.csv"), "profiles/%s_kernel_stats.csv" % tag)
host = [json.loads(l) for l in open(src + "/host_path.json") if l.strip().startswith("{")]
json.dump(host, open("profiles/%s_host_path.json" % tag, "w"), indent=1)

f, kname = counters(newest(src + "/pmc_fetch/*/*_counter_collection.csv"))
w, _ = counters(newest(src + "/pmc_write/*/*_counter_collection.csv"))
cfg = bench["config"]
pairs = cfg["pairs_per_step_per_gpu"]
known_in = 2 * 150 * pairs + 2 * 8 * (pairs + 1)          # bases of both mates + both offset arrays
fetch_kb, write_kb = f["FETCH_SIZE"], w["WRITE_SIZE"]
out = {
    "pairs": pairs, "k": 17, "bf_log2": 33, "on_target": cfg["on_target"],
    "kernel": kname.split("(")[0].replace("void shk::", "") + " (" + cfg["probe_mode"] + ")",
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "TCC_HIT_sum": w.get("TCC_HIT_sum"), "TCC_MISS_sum": w.get("TCC_MISS_sum"),
    "correction": "FETCH_SIZE x2 (MI355X_MICROARCH.md: gfx950 tallies 128-B requests of coalesced streams at 64 B; "
                  "the fetched bytes here are the coalesced input stream); WRITE_SIZE exact",
    "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
    "hbm_bytes_per_launch_uncorrected": int(fetch_kb * 1024 + write_kb * 1024),
    "known_input_bytes": known_in,
}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
for r in csv.DictReader(open("profiles/%s_kernel_stats.csv" % tag)):
    if "classify_fast" in r["Name"]:
        print("kernel stats:", r["Name"][:60], "calls", r["Calls"], "avg_ms %.3f" % (float(r["AverageNs"]) / 1e6))
print("bench kernel_ms", bench["roofline"]["kernel_ms"], "kt run kernel_ms", json.loads(open(src + "/kt.json").read().strip().splitlines()[-1])["roofline"]["kernel_ms"])
