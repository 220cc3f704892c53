#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/cut5
( time timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_table or dense_table or panel or bound_cut or synthetic_parity or handworked or example_bit_exact or quality_mask or ragged or long" ) > gpurun_out/cut5/tests.log 2>&1 || { tail -30 gpurun_out/cut5/tests.log; exit 1; }
tail -3 gpurun_out/cut5/tests.log
( timeout -k 10 100 python tests/fuzz_parity.py 250 555001 ) > gpurun_out/cut5/fuzz.log 2>&1 || { tail -15 gpurun_out/cut5/fuzz.log; exit 1; }
tail -1 gpurun_out/cut5/fuzz.log
bash tools/gpu_big.sh base old 2>&1 | tee gpurun_out/cut5/big.txt
