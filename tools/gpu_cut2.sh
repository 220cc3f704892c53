#!/bin/bash
# configs[2] index at 0 / 100 % on-target pairs, in-tree library against tools/variants/old.so
export TMPDIR=/tmp
mkdir -p gpurun_out/cut2
for ot in 0.0 1.0; do
for v in base old; do
  if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
  python tests/scale_check.py --skip-bitvector --oracle-pairs 0 --on-target $ot 2>/dev/null | grep kernel_ms | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v uniform 60000 ot=$ot', d['mode'], d['kernel_ms'], d['n_assoc'])" | tee -a gpurun_out/cut2/ab.txt
done
done
