#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/cut6
( time timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_table or dense_table or panel or bound_cut or synthetic_parity or handworked or example_bit_exact or quality_mask or ragged or long" ) > gpurun_out/cut6/tests.log 2>&1 || { tail -30 gpurun_out/cut6/tests.log; exit 1; }
tail -3 gpurun_out/cut6/tests.log
( timeout -k 10 100 python tests/fuzz_parity.py 250 555001 ) > gpurun_out/cut6/fuzz.log 2>&1 || { tail -15 gpurun_out/cut6/fuzz.log; exit 1; }
tail -1 gpurun_out/cut6/fuzz.log
( timeout -k 10 400 python tests/scale_check.py --skip-bitvector --oracle-pairs 100000 ) > gpurun_out/cut6/scale.log 2>&1 || { tail -15 gpurun_out/cut6/scale.log; exit 1; }
grep "kernel_ms\|parity\|OK" gpurun_out/cut6/scale.log | cut -c1-300
bash tools/gpu_big.sh base 2>&1 | tee gpurun_out/cut6/big.txt
