#!/bin/bash
# scratch: A/B of the sparse first round on the one-gene index
mkdir -p gpurun_out/sparse
for L in 150 100; do
  timeout -k 10 300 python tools/landscape.py --genes 1 --ot 0,0.5,1 --ab --ab-var SHK_NO_SPARSE --reps 3 --read-len $L >> gpurun_out/sparse/ab.jsonl 2>> gpurun_out/sparse/ab.log || exit 1
done
cat gpurun_out/sparse/ab.jsonl
