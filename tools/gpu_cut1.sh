#!/bin/bash
# first GPU check of the bound cut: targeted parity tests, a fuzz run, then A/B against the previous library (tools/variants/old.so)
export TMPDIR=/tmp
mkdir -p gpurun_out/cut3
( time timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_table or dense_table or panel or bound_cut or synthetic_parity or handworked or example_bit_exact or quality_mask or ragged or long" ) > gpurun_out/cut3/tests.log 2>&1 || { tail -30 gpurun_out/cut3/tests.log; exit 1; }
tail -5 gpurun_out/cut3/tests.log
( FUZZ_UNI=1 timeout -k 10 200 python tests/fuzz_parity.py 400 777001 ) > gpurun_out/cut3/fuzz.log 2>&1 || { tail -15 gpurun_out/cut3/fuzz.log; exit 1; }
tail -3 gpurun_out/cut3/fuzz.log
REPS=2 bash tools/gpu_ab2.sh base old 2>&1 | tee gpurun_out/cut3/ab.txt
for v in; do
  if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
  python tests/scale_check.py --skip-bitvector --oracle-pairs 0 2>/dev/null | grep kernel_ms | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v uniform 60000', d['mode'], d['kernel_ms'], d['n_assoc'])" | tee -a gpurun_out/cut3/ab.txt
done
