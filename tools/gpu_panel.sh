#!/bin/bash
# panel-sized indices (GENES genes of 2400 bases) on the bench's reads: tools/gpu_panel.sh "GENES..." base NAME...
export TMPDIR=/tmp
genes=$1; shift
for g in $genes; do for v in "$@"; do
  if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
  python bench.py --no-configs --no-cpu-baseline --no-boundary --steps 3 --total-pairs 10000000 --genes $g --gene-len 2400 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', $g, d['roofline']['kernel_ms'], d['config']['assoc_per_step'], d['config']['probe_mode'], d['config'].get('n_set_bits'))"
done; done
