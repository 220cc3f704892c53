#!/bin/bash
# scratch: A/B against the committed kernels
mkdir -p gpurun_out/part
for v in head base head base; do
  if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
  echo "== $v"
  timeout -k 10 300 python tools/landscape.py --genes 1 --ot 0,0.5,1 --reps 3 2> gpurun_out/part/land_$v.log | python3 -c "import sys,json; [print(d[\"on_target\"], d[\"kernel_ms\"]) for d in map(json.loads, sys.stdin)]" || exit 1
done
