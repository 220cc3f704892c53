#!/bin/bash
# scratch pad for one-off A/B runs on the GPU box (tools/gpu.sh 1200 'bash tools/cutrun.sh'); the pattern used throughout round 3:
# build a variant of the library (tools/build_variant.sh NAME -DSHK_...), then time both in ONE run -- boxes differ by up to 8 %
for v in base; do
  if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
  echo "== $v"
  timeout -k 10 300 python tools/landscape.py --genes 1 --ot 0,0.5,1 --reps 3 2>/dev/null | python3 -c "import sys,json; [print(d['genes'], d['on_target'], d['kernel_ms'], d['n_assoc']) for d in map(json.loads, sys.stdin)]" || exit 1
done
