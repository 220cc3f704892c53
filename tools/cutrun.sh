#!/bin/bash
# scratch: running fetch pointers against the committed kernels
for v in head base head base; do
  if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
  echo "== $v"
  timeout -k 10 300 python tools/landscape.py --genes 1,100,60000 --ot 0,0.5 --reps 3 2>/dev/null | python3 -c "import sys,json; [print(d['genes'], d['on_target'], d['kernel_ms'], d['n_assoc']) for d in map(json.loads, sys.stdin)]" || exit 1
  timeout -k 10 300 python tools/landscape.py --genes 60000 --ot 0.5 --k 31 --q 20 --reps 3 2>/dev/null | python3 -c "import sys,json; [print(d['genes'], 'k31q20', d['kernel_ms'], d['n_assoc']) for d in map(json.loads, sys.stdin)]" || exit 1
done
