#!/bin/bash
# round-3 GPU session driver: tools/gpu_r3.sh <tag> <step> [<step> ...]; every step logs to gpurun_out/<tag>/<step>.log
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for step in "$@"; do
  echo "== $step $(date +%T)"
  case $step in
    bench_tests) timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -k "bench" -x -q > $out/$step.log 2>&1 ;;
    cli_tests)   timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -k "cli" -x -q > $out/$step.log 2>&1 ;;
    fuzz)        timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -x -q -s > $out/$step.log 2>&1 ;;
    anch_tests)  timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -k "anchored or chimeric or synthetic_parity or every_kernel or long_and_ragged" -x -q > $out/$step.log 2>&1 ;;
    sparse_tests) timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -k "sparse_first or lds_table or example_bit or bound_cut or uniformity or device_api or device_resident" -x -q > $out/$step.log 2>&1 ;;
    long_tests)  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -k "long_pairs or every_kernel or long_and_ragged or packed_position or anchored" -x -q > $out/$step.log 2>&1 ;;
    land_ab)     timeout -k 10 900 python tools/landscape.py --genes 250,1000,10000,60000 --ab > $out/landscape_ab.jsonl 2> $out/$step.log ;;
    land_ot)     timeout -k 10 900 python tools/landscape.py --genes 1000,60000 --ot 0,1 --ab > $out/landscape_ot.jsonl 2> $out/$step.log ;;
    land_all)    timeout -k 10 900 python tools/landscape.py > $out/landscape.jsonl 2> $out/$step.log ;;
    scale)       timeout -k 10 1100 python -m pytest tests/test_gpu_scale.py -x -q > $out/$step.log 2>&1 ;;
    suite)       timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=15 > $out/$step.log 2>&1 ;;
    bench)       timeout -k 10 600 python3 bench.py --steps 10 --warmup 2 > $out/bench.json 2> $out/$step.log ;;
    bench_quick) timeout -k 10 600 python3 bench.py --steps 5 --warmup 1 --no-configs --no-boundary --no-cpu-baseline --no-cli > $out/bench_quick.json 2> $out/$step.log ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
  rc=$?
  echo "rc=$rc" >> $out/$step.log
  tail -4 $out/$step.log
  if [ $rc -ne 0 ]; then echo "step $step failed (rc=$rc): stopping"; exit $rc; fi
done
