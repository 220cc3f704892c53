#!/bin/bash
# long differential fuzz against the oracle: plain bias (all probe modes) and FUZZ_UNI=1 (uniform kernel, small indices)
export TMPDIR=/tmp
mkdir -p gpurun_out/fuzz
( timeout -k 10 420 python tests/fuzz_parity.py ${1:-3000} ${2:-4240001} ) > gpurun_out/fuzz/plain.log 2>&1 || { tail -15 gpurun_out/fuzz/plain.log; exit 1; }
tail -1 gpurun_out/fuzz/plain.log
( FUZZ_UNI=1 timeout -k 10 650 python tests/fuzz_parity.py ${3:-1300} ${4:-5250001} ) > gpurun_out/fuzz/uni.log 2>&1 || { tail -15 gpurun_out/fuzz/uni.log; exit 1; }
tail -1 gpurun_out/fuzz/uni.log
