#!/bin/bash
# counters of the classify kernels for one landscape case: tools/pmc_case.sh <out-dir> <counter-set> -- <landscape args>
#   counter sets: sq (instructions / waits), tcc (L2 requests, hits, misses), tcp (vector L1), ic (instruction cache), if (instruction fetch)
export TMPDIR=/tmp
out=$1; set_=$2; shift 3
case $set_ in
  sq)  C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" ;;
  tcc) C="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" ;;
  ic)  C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" ;;
  if)  C="SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM" ;;
  tcp) C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" ;;
esac
rm -rf $out/pmc_$set_; mkdir -p $out
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $out/pmc_$set_ -- python3 tools/landscape.py "$@" > $out/pmc_$set_.out 2> $out/pmc_$set_.err || { tail -5 $out/pmc_$set_.err; exit 1; }
python3 - $out/pmc_$set_ <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "classify_uni" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0].replace("void shk::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if max(v) > 1e5: print(k, c, round(max(v) / 1e7, 2), "per pair (of 10 M)")
PY
rm -rf $out/pmc_$set_
