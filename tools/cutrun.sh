#!/bin/bash
# scratch: panels follow the assigned fraction
for v in 1 0; do
  if [ $v = 1 ]; then export SHK_BIG_LDS_ALWAYS=1; else unset SHK_BIG_LDS_ALWAYS; fi
  echo "== always big: $v"
  timeout -k 10 500 python tools/landscape.py --genes 60,100,150 --ot 0,0.2,0.5,1 --reps 3 2>/dev/null | python3 -c "import sys,json; [print(d['genes'], d['on_target'], d['kernel_ms'], d['n_assoc']) for d in map(json.loads, sys.stdin)]" || exit 1
done
