#!/bin/bash
# scratch: geometry / plans kept across reads of equal lengths (ragged kernels)
for v in head base; do
  if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
  echo "== $v"
  for full in 0 0.8; do
    for g in 1 100 60000; do
      GENES=$g timeout -k 10 300 python tools/ragged_rate.py 10000000 100 $full 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['genes'], d['untrimmed'], d['kernel_ms'], d['n_assoc'])" || exit 1
    done
  done
done
