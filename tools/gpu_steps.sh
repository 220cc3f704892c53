#!/bin/bash
# GPU-box driver: tools/gpu_steps.sh <tag> <step> [<step> ...]   (through gpurun; see tools/gpu.sh)
# Every step logs to gpurun_out/<tag>/<step>.log and the run stops at the first step that fails (no GPU step is started behind
# one that timed out or was killed).  Steps:
#   suite            the whole `pytest -m gpu` suite          scale      tests/test_gpu_scale.py (configs[2] / configs[4] shapes)
#   t:<expr>         pytest tests/test_gpu_parity.py -k <expr>           fuzz       tests/test_gpu_fuzz.py
#   bench            the default bench line (what the driver runs)       bench_quick  headline only, no side legs
#   bench:<args>     bench.py with the given arguments (separated by @)
#   profiles         tools/gpu_profiles.sh (kernel stats + counter passes for profiles/)
#   land:<args>      tools/landscape.py <args> (separated by @)         ragged:<genes>  tools/ragged_rate.py (GENES=<genes>)
#   ab:<ot>:<lib>... bench workload at on-target <ot> on library variants (base = in-tree, else tools/variants/<lib>.so), interleaved
#   abpy:<libs>:<script+args>  a script on several library variants (tools/variants/<lib>.so, base = in-tree; libs joined by +), interleaved
#   kt:<script+args>  rocprofv3 --kernel-trace --stats of a python script: the classify kernels' rows
#   pmc:<script+args> one SQ counter pass (rocprofv3 --pmc) of a python script: instructions per pair of the classify kernels
#   py:<script+args> python3 <script> <args> (separated by @)           mem        memory / tmpfs / cpu figures of the box
#   env:VAR=VALUE    exported for the steps behind it
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
i=0
for step in "$@"; do
  i=$((i + 1))
  echo "== $step $(date +%T)"
  log=$out/$(printf %02d $i)_$(echo "$step" | tr ':/ ,' '____').log
  case $step in
    suite)       timeout -k 10 1150 python3 -m pytest tests -m gpu -x -q --durations=60 > $log 2>&1 ;;
    scale)       timeout -k 10 1150 python3 -m pytest tests/test_gpu_scale.py -x -q --durations=5 > $log 2>&1 ;;
    fuzz)        timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz.py -x -q -s > $log 2>&1 ;;
    t:*)         timeout -k 10 1150 python3 -m pytest tests/test_gpu_parity.py -k "${step#t:}" -x -q --durations=8 > $log 2>&1 ;;
    bench)       timeout -k 10 900 python3 bench.py > $out/bench.json 2> $log ;;
    bench_quick) timeout -k 10 600 python3 bench.py --no-configs --no-boundary --no-cpu-baseline --no-cli --no-live-counters > $out/bench_quick.json 2> $log ;;
    bench:*)     timeout -k 10 900 python3 bench.py $(echo "${step#bench:}" | tr '@' ' ') > $log.json 2> $log ;;
    profiles)    bash tools/gpu_profiles.sh > $log 2>&1 ;;
    land:*)      timeout -k 10 900 python3 tools/landscape.py $(echo "${step#land:}" | tr '@' ' ') > $log.jsonl 2> $log ;;
    ragged:*)    GENES=${step#ragged:} timeout -k 10 600 python3 tools/ragged_rate.py > $log.jsonl 2> $log ;;
    py:*)        timeout -k 10 900 python3 $(echo "${step#py:}" | tr '@' ' ') > $log.out 2> $log ;;
    ab:*)        IFS=: read -r _ ot libs <<< "$step"
                 for rep in 1 2; do for v in $(echo $libs | tr ':' ' '); do
                   if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
                   timeout -k 10 300 python3 bench.py --no-configs --no-cpu-baseline --no-boundary --no-cli --no-live-counters --no-async --no-trimmed --steps 2 --reps-per-step 2 --total-pairs 20000000 --on-target $ot 2>> $log \
                     | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['config']['on_target'], d['roofline']['kernel_ms'], d['config']['assoc_per_step'], d['roofline']['kernel_reported_by_library'])" | tee -a $log.txt
                 done; done; unset SHK_LIB_PATH ;;
    abpy:*)      # abpy:<lib>+<lib>..:<script@args>  the script on library variants (base = in-tree), interleaved, twice; one output line each
                 IFS=: read -r _ libs cmdl <<< "$step"
                 for rep in 1 2; do for v in $(echo $libs | tr '+' ' '); do
                   if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
                   timeout -k 10 400 python3 $(echo "$cmdl" | tr '@' ' ') 2>> $log | sed "s/^/$v /" | tee -a $log.txt
                 done; done; unset SHK_LIB_PATH ;;
    pmc:*)       # one SQ counter pass of a script: per-pair instruction counts of the classify kernels (10 M pairs assumed)
                 rm -rf $out/pmc_tmp; GENES=${GENES:-1} timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
                   --kernel-trace --output-format csv -d $out/pmc_tmp -- python3 $(echo "${step#pmc:}" | tr '@' ' ') > $log.out 2> $log \
                 && python3 - $out/pmc_tmp >> $log.out <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "classify_" in r["Kernel_Name"] or "anchor_verdict" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0].replace("void shk::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if max(v) > 1e6: print(k, c, round(max(v) / 1e7, 1), "per pair (of 10 M)")
PY
                 rm -rf $out/pmc_tmp ;;
    kt:*)        # kernel trace with per-kernel statistics of a python script (rocprofv3 --kernel-trace --stats): the classify kernels' rows
                 rm -rf $out/kt_tmp; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_tmp -- python3 $(echo "${step#kt:}" | tr '@' ' ') > $log.out 2> $log \
                 && { f=$(find $out/kt_tmp -name "*_kernel_stats.csv" | head -1); head -1 $f >> $log.out; grep -E "shk::|rocclr" $f >> $log.out; }
                 rm -rf $out/kt_tmp ;;
    env:*)       export "${step#env:}"; echo "exported ${step#env:}" > $log ;;   # for the steps behind it
    mem)         { nproc; free -g; df -h /dev/shm /tmp; cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/cpu.max 2>/dev/null; grep -c processor /proc/cpuinfo; which rocprofv3; } > $log 2>&1 ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
  rc=$?
  echo "rc=$rc" >> $log
  tail -4 $log
  if [ $rc -ne 0 ]; then echo "step $step failed (rc=$rc): stopping"; exit $rc; fi
done
