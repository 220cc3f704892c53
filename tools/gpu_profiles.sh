#!/bin/bash
# Profiles of the classify kernels for profiles/ (run on the GPU box through gpurun; then `python tools/collect_profiles.py rNN` here).
#   kt     : rocprofv3 --kernel-trace --stats of the bench command (headline workload + configs[2] index)
#   pmc_*  : counter passes, each in its own run, --kernel-trace only beside --pmc (gpurun refuses anything else):
#            FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum | SQ set a | SQ set b
#            for the headline workload at 0 / 50 / 100 % on-target pairs and for the configs[2] index
export TMPDIR=/tmp
OUT=gpurun_out/profiles
rm -rf $OUT; mkdir -p $OUT
BENCH1="bench.py --steps 2 --warmup 1 --total-pairs 10000000 --no-configs --no-cpu-baseline --no-boundary --no-cli"
SQA="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
SQB="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS"
pass() { # tag name counters -- program args
  tag=$1; name=$2; ctr=$3; shift 3
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$tag/$name -- python3 "$@" > $OUT/$tag/$name.json 2> $OUT/$tag/$name.err
}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-boundary --no-cli > $OUT/kt.json 2> $OUT/kt.err
for ot in 0.0 0.5 1.0; do
  tag=configs1_ot$ot; mkdir -p $OUT/$tag
  pass $tag fetch "FETCH_SIZE" $BENCH1 --on-target $ot
  pass $tag write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $BENCH1 --on-target $ot
  pass $tag sqa "$SQA" $BENCH1 --on-target $ot
  pass $tag sqb "$SQB" $BENCH1 --on-target $ot
done
tag=configs2; mkdir -p $OUT/$tag
SC="tests/scale_check.py --skip-bitvector --oracle-pairs 0"
pass $tag fetch "FETCH_SIZE" $SC
pass $tag write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $SC
pass $tag sqa "$SQA" $SC
pass $tag sqb "$SQB" $SC
# keep only what the collector reads (counter CSVs are large)
python3 - <<'PY'
import csv, glob, json, os, collections
out = "gpurun_out/profiles"
res = {}
for tag in sorted(os.listdir(out)):
    d = os.path.join(out, tag)
    if not os.path.isdir(d) or tag == "kt":
        continue
    acc = collections.defaultdict(list)
    kname = ""
    for f in glob.glob(d + "/*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "classify_uni_kernel" in r["Kernel_Name"] or "classify_fast_kernel" in r["Kernel_Name"]:
                # (a launch that returns at once -- the kernel that is not responsible for the batch -- has ~no instructions)
                acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    best = {}
    for (kn, cn), v in acc.items():
        best.setdefault(kn, {})[cn] = max(v) if cn != "dummy" else 0   # per launch; identical launches, take the largest (skips the no-op)
    res[tag] = best
json.dump(res, open(out + "/counters_raw.json", "w"), indent=1)
for f in glob.glob(out + "/*/*/*/*_counter_collection.csv") + glob.glob(out + "/*/*/*/*_kernel_trace.csv") + glob.glob(out + "/*/*/*/*agent_info.csv"):
    os.remove(f)
print(json.dumps(res)[:3000])
PY
ls -R $OUT | head -60
