#!/bin/bash
# counters for the classify kernel on the GENCODE-shaped index (tests/scale_check.py, no oracle)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_scale
for set in "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_scale/$tag -- python3 tests/scale_check.py --skip-bitvector --oracle-pairs 0 --pairs 10000000 > gpurun_out/pmc_scale/$tag.out 2> gpurun_out/pmc_scale/$tag.err
done
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_scale/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "classify_fast" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("scale", " ".join("%s=%.4g"%(k,sum(v)/len(v)) for k,v in sorted(acc.items())))
PY
tail -3 gpurun_out/pmc_scale/FETCH_SIZE.out
