#!/bin/bash
# after the lds-table slot fix: the tiny-index tests (incl. every reference k-mer as a read), then the profile passes
export TMPDIR=/tmp
mkdir -p gpurun_out/fix1
( time timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_table or bound_cut or synthetic_parity or example_bit_exact or handworked" ) > gpurun_out/fix1/tests.log 2>&1 || { tail -30 gpurun_out/fix1/tests.log; exit 1; }
tail -3 gpurun_out/fix1/tests.log
bash tools/gpu_profiles.sh > gpurun_out/profiles_run.log 2>&1
tail -2 gpurun_out/profiles_run.log
bash tools/gpu_ab2.sh base 2>&1 | tee gpurun_out/fix1/ontarget.txt
