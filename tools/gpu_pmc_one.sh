#!/bin/bash
# one SQ counter pass of the bench workload at on-target fraction $1 (environment passes through, e.g. SHK_LIB_PATH)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_one; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --steps 2 --warmup 1 --total-pairs 10000000 --no-configs --no-cpu-baseline --no-boundary --no-cli --on-target $1 > $OUT/a.json 2> $OUT/a.err
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_one/a/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "classify_uni_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if max(v) > 1e6: print(k, c, round(max(v) / 1e7, 1), "per pair")
PY
