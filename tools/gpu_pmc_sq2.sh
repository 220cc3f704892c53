#!/bin/bash
# SQ busy/issue counters for the classify kernel at on-target fraction $1, tag $2 (two passes)
export TMPDIR=/tmp
OT=${1:-0.5}; TAG=${2:-x}
mkdir -p gpurun_out/pmcsq2_$TAG
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/pmcsq2_$TAG/sq_names.txt
run() {
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmcsq2_$TAG/$1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --on-target $OT > gpurun_out/pmcsq2_$TAG/$1.json 2> gpurun_out/pmcsq2_$TAG/$1.err
}
run a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
run b "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS"
python3 - <<PY
import csv,glob,collections
for p in "ab":
    fs=glob.glob("gpurun_out/pmcsq2_$TAG/%s/*/*_counter_collection.csv"%p)
    if not fs: print(p,"no output", open("gpurun_out/pmcsq2_$TAG/%s.err"%p).read()[-600:]); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "classify_fast" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("$TAG", p, " ".join("%s=%.4g"%(k,sum(v)/len(v)) for k,v in sorted(acc.items())))
PY
