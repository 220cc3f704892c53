#!/bin/bash
# SQ instruction counters for the classify kernel at on-target fraction $1, tag $2
export TMPDIR=/tmp
OT=${1:-0.5}; TAG=${2:-x}
mkdir -p gpurun_out/pmcsq_$TAG
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmcsq_$TAG/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --on-target $OT > gpurun_out/pmcsq_$TAG/sq.json 2> gpurun_out/pmcsq_$TAG/sq.err
python3 - <<PY
import csv,glob,collections
fs=glob.glob("gpurun_out/pmcsq_$TAG/sq/*/*_counter_collection.csv")
acc=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "classify_fast" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$TAG", " ".join("%s=%.4g"%(k,sum(v)/len(v)) for k,v in sorted(acc.items())))
PY
