#!/bin/bash
# scratch: the second acceptance behind round B, at 1 % and 4 % substitutions
for v in head base head base; do
  if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
  echo "== $v"
  for sr in 0.01 0.04; do
    timeout -k 10 300 python tools/landscape.py --genes 1 --ot 0.5,1 --reps 3 --sub-rate $sr 2>/dev/null | python3 -c "import sys,json; [print(d['sub_rate'], d['on_target'], d['kernel_ms'], d['n_assoc']) for d in map(json.loads, sys.stdin)]" || exit 1
  done
done
