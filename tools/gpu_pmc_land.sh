#!/bin/bash
# counter passes of tools/landscape.py: tools/gpu_pmc_land.sh <tag> <genes> <on-target> [extra landscape args]
# (each --pmc set in its own run, --kernel-trace only beside it)
export TMPDIR=/tmp
tag=$1; genes=$2; ot=$3; shift 3
OUT=gpurun_out/$tag; mkdir -p $OUT
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$n -- python3 tools/landscape.py --genes $genes --ot $ot --reps 1 "$@" > $OUT/pmc_$n.jsonl 2> $OUT/pmc_$n.err || exit 1
done
python3 - $OUT $genes $ot <<'PY'
import csv, glob, collections, sys, json
out, genes, ot = sys.argv[1:4]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "classify_uni_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"genes": int(genes), "on_target": float(ot)}
for c, v in sorted(acc.items()):
    res[c + "_per_pair"] = round(max(v) / 1e7, 2)
print(json.dumps(res))
open(out + "/pmc_summary.jsonl", "a").write(json.dumps(res) + "\n")
for f in glob.glob(out + "/pmc_*/*/*.csv"):
    import os; os.remove(f)
PY
