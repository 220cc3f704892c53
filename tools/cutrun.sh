#!/bin/bash
# scratch: the one-gene kernel without its base loads (timing only)
export SHK_LIB_PATH=$PWD/tools/variants/abl.so
for a in 0 4; do
  echo "== ablate $a"
  SHK_ABLATE=$a timeout -k 10 300 python tools/landscape.py --genes 1 --ot 0,1 --reps 3 2> /dev/null | python3 -c "import sys,json; [print(d[\"on_target\"], d[\"kernel_ms\"], d[\"n_assoc\"]) for d in map(json.loads, sys.stdin)]" || exit 1
done
