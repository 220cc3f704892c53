#!/bin/bash
# A/B library variants on the bench workload at three on-target fractions, REPS interleaved repetitions:
#   [REPS=2] tools/gpu_ab2.sh base NAME...   ("base" = the in-tree library, NAME = tools/variants/NAME.so from tools/build_variant.sh)
export TMPDIR=/tmp
run() { python bench.py --no-configs --no-cpu-baseline --no-boundary --steps 4 --total-pairs 20000000 --on-target $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['config']['on_target'], round(d['value']/1e6,1), d['roofline']['kernel_ms'], d['config']['assoc_per_step'], d['config']['probe_mode'])"; }
for rep in $(seq ${REPS:-1}); do
for ot in 0.0 0.5 1.0; do
  for v in "$@"; do
    if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
    run $ot $v
  done
done
done
