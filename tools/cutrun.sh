#!/bin/bash
# scratch: margin of the partial round
mkdir -p gpurun_out/part
for v in pk1 base pk3; do
  if [ $v = base ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/tools/variants/$v.so; fi
  echo "== $v"
  timeout -k 10 300 python tools/landscape.py --genes 60000 --ot 0,0.5 --reps 3 2> gpurun_out/part/land_$v.log | cut -c1-20,150-190 || exit 1
done
