#!/bin/bash
# the early decision (vote<J>, J < U): parity tests, fuzz, configs[2] with the oracle sample, A/B against tools/variants/noacc.so (-DSHK_NO_ACCEPT=1)
export TMPDIR=/tmp
mkdir -p gpurun_out/acc1
( time timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_table or dense_table or panel or bound_cut or synthetic_parity or handworked or example_bit_exact or quality_mask or ragged or long or tie or single" ) > gpurun_out/acc1/tests.log 2>&1 || { tail -30 gpurun_out/acc1/tests.log; exit 1; }
tail -3 gpurun_out/acc1/tests.log
( timeout -k 10 100 python tests/fuzz_parity.py 400 717001 ) > gpurun_out/acc1/fuzz.log 2>&1 || { tail -15 gpurun_out/acc1/fuzz.log; exit 1; }
tail -1 gpurun_out/acc1/fuzz.log
( FUZZ_UNI=1 timeout -k 10 120 python tests/fuzz_parity.py 200 727001 ) > gpurun_out/acc1/fuzz_uni.log 2>&1 || { tail -15 gpurun_out/acc1/fuzz_uni.log; exit 1; }
tail -1 gpurun_out/acc1/fuzz_uni.log
( timeout -k 10 400 python tests/scale_check.py --skip-bitvector --oracle-pairs 100000 ) > gpurun_out/acc1/scale.log 2>&1 || { tail -15 gpurun_out/acc1/scale.log; exit 1; }
grep "kernel_ms\|parity\|OK" gpurun_out/acc1/scale.log | cut -c1-300
bash tools/gpu_ab2.sh base noacc 2>&1 | tee gpurun_out/acc1/ab.txt
bash tools/gpu_big.sh base noacc 2>&1 | tee gpurun_out/acc1/big.txt
