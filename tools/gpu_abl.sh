#!/bin/bash
# timing-only ablations of classify_uni_kernel (tools/variants/abl.so, built with -DSHK_ABLATION): tools/gpu_abl.sh ON_TARGET BITS...
export TMPDIR=/tmp
export SHK_LIB_PATH=$PWD/tools/variants/abl.so
ot=$1; shift
for a in "$@"; do
  SHK_ABLATE=$a python bench.py --no-configs --no-cpu-baseline --no-boundary --steps 4 --total-pairs 20000000 --on-target $ot 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate $a', d['config']['on_target'], d['roofline']['kernel_ms'], d['config']['assoc_per_step'])"
done
