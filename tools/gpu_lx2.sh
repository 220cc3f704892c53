#!/bin/bash
# lds-table mode for trimmed reads + the occupancy experiment (tools/variants/lxw8.so: 8 instead of 16 waves in the one workgroup per CU)
export TMPDIR=/tmp
mkdir -p gpurun_out/lx2
( time timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_table or bound_cut or synthetic_parity or ragged or quality_mask or long or handworked" ) > gpurun_out/lx2/tests.log 2>&1 || { tail -30 gpurun_out/lx2/tests.log; exit 1; }
tail -3 gpurun_out/lx2/tests.log
( FUZZ_UNI=1 timeout -k 10 100 python tests/fuzz_parity.py 150 616001 ) > gpurun_out/lx2/fuzz.log 2>&1 || { tail -15 gpurun_out/lx2/fuzz.log; exit 1; }
tail -1 gpurun_out/lx2/fuzz.log
GENES=1 python tools/ragged_rate.py 2>/dev/null | tail -1
bash tools/gpu_ab2.sh base lxw8 2>&1 | tee gpurun_out/lx2/ab.txt
