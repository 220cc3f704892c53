#!/bin/bash
# table-mode instantiations at 6 instead of 8 waves per SIMD (80 VGPRs: no spills in the quality-mask kernels): tools/variants/tw6.so
export TMPDIR=/tmp
for rep in 1 2; do for v in base tw6; do
  if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
  python tests/scale_check.py --k 31 --q 20 --single --bf-log2 37 --skip-bitvector --oracle-pairs 0 2>/dev/null | grep kernel_ms | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v configs4 shape', d['mode'], d['kernel_ms'], d['n_assoc'])"
  python tests/scale_check.py --skip-bitvector --oracle-pairs 0 2>/dev/null | grep kernel_ms | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v configs2 index', d['mode'], d['kernel_ms'], d['n_assoc'])"
  python tests/scale_check.py --genes 1000 --bf-log2 33 --skip-bitvector --oracle-pairs 0 2>/dev/null | grep kernel_ms | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v 1000 genes', d['mode'], d['kernel_ms'], d['n_assoc'])"
done; done
