#!/bin/bash
# the cut tried at the early decision's stop in every mode: parity subset, then A/B against tools/variants/acc0.so (commit aab87eb)
export TMPDIR=/tmp
mkdir -p gpurun_out/acc2
( time timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_table or dense_table or panel or bound_cut or synthetic_parity or handworked or quality_mask" ) > gpurun_out/acc2/tests.log 2>&1 || { tail -30 gpurun_out/acc2/tests.log; exit 1; }
tail -3 gpurun_out/acc2/tests.log
( timeout -k 10 100 python tests/fuzz_parity.py 300 737001 ) > gpurun_out/acc2/fuzz.log 2>&1 || { tail -15 gpurun_out/acc2/fuzz.log; exit 1; }
tail -1 gpurun_out/acc2/fuzz.log
bash tools/gpu_big.sh base acc0 2>&1 | tee gpurun_out/acc2/big.txt
bash tools/gpu_panel.sh "30 100" base acc0 2>&1 | tee gpurun_out/acc2/panel.txt
