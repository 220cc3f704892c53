#!/bin/bash
# A/B a kernel variant against the committed library on the bench workload: tools/gpu_ab.sh VARIANT_NAME
export TMPDIR=/tmp
run() { python bench.py --no-configs --no-cpu-baseline --no-boundary --steps 4 --total-pairs 20000000 --on-target $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['config']['on_target'], round(d['value']/1e6,1), d['roofline']['kernel_ms'], d['config']['assoc_per_step'])"; }
for rep in 1 2; do
for ot in 0.0 0.5 1.0; do run $ot base; SHK_LIB_PATH=$PWD/tools/variants/$1.so run $ot $1; done
done
