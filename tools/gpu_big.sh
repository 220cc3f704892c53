#!/bin/bash
# large-table workloads (uniform: 250 / 1000 / 60000 genes; trimmed reads: 100 / 60000 genes): tools/gpu_big.sh base NAME...
export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
  for g in 250 1000; do
    python tests/scale_check.py --genes $g --bf-log2 33 --skip-bitvector --oracle-pairs 0 2>/dev/null | grep kernel_ms | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v uniform', $g, d['mode'], d['kernel_ms'])"
  done
  python tests/scale_check.py --skip-bitvector --oracle-pairs 0 2>/dev/null | grep kernel_ms | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v uniform 60000', d['mode'], d['kernel_ms'])"
  for g in 100 60000; do GENES=$g python tools/ragged_rate.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v trimmed', d['genes'], d['mode'], d['kernel_ms'])"; done
done
