#!/bin/bash
# PMC passes for the classify kernel at a given on-target fraction: $1
export TMPDIR=/tmp
OT=${1:-0.5}
TAG=${2:-x}
mkdir -p gpurun_out/pmc_$TAG
run() { # name counters...
  n=$1; shift
  timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG/$n -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --on-target $OT > gpurun_out/pmc_$TAG/$n.json 2> gpurun_out/pmc_$TAG/$n.err
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
# NOTE: a pass with TA_*/GRBM_* counters aborted inside rocprofv3 (signal 6) and hung until the gpurun limit: not collected
run tcc FETCH_SIZE TCC_REQ_sum
python3 - <<PY
import csv,glob,collections
for n in ("sq","sq2","tcc"):
    fs=glob.glob("gpurun_out/pmc_$TAG/%s/*/*_counter_collection.csv"%n)
    if not fs: print(n,"no csv"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "classify_fast" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print("$TAG",n,k,"%.4g"%(sum(v)/len(v)),len(v))
PY
