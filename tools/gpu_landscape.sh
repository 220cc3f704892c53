#!/bin/bash
# the landscape tables of profiles/README.md: kernel time per 10 M pairs over index sizes (2^33-bit filter; 60 000 genes: 2^36), on-target rates,
# with and without the anchored extension, read lengths, trimmed reads
export TMPDIR=/tmp
out=${1:-gpurun_out/landscape}; mkdir -p $out
python tools/landscape.py --genes 1,10,60,100,150,250,1000,10000,60000 --ot 0.5 > $out/sizes.jsonl 2>/dev/null
python tools/landscape.py --genes 1,100,1000,60000 --ot 0.0,1.0 > $out/on_target.jsonl 2>/dev/null
python tools/landscape.py --genes 100,1000,10000,60000 --ot 0.25,0.5,1 --ab --ab-var SHK_NO_PRE_VERDICT > $out/pre_ab.jsonl 2>/dev/null
python tools/landscape.py --genes 60000 --ot 0,0.5 --ab --ab-var SHK_NO_KTAB > $out/ktab_ab.jsonl 2>/dev/null
python tools/landscape.py --genes 1000,60000 --ot 0.5 --ab > $out/anchor_ab.jsonl 2>/dev/null
python tools/landscape.py --genes 1 --ot 0,0.5,1 --ab --ab-var SHK_NO_SPARSE > $out/sparse_ab.jsonl 2>/dev/null
python tools/landscape.py --genes 9,10 --ot 0.5,1 --ab --ab-var SHK_NO_SPARSE > $out/sparse_multi_ab.jsonl 2>/dev/null
python tools/landscape.py --genes 1,1000,60000 --ot 0.5 --read-len 300 --pairs 5000000 > $out/len300.jsonl 2>/dev/null
python tools/landscape.py --genes 1,1000,60000 --ot 0.5 --read-len 250 --pairs 5000000 > $out/len250.jsonl 2>/dev/null
python tools/landscape.py --genes 1,1000,60000 --ot 0.5 --read-len 100 > $out/len100.jsonl 2>/dev/null
python tools/landscape.py --genes 60000 --ot 0.5 --k 31 --q 20 --ab --ab-var SHK_NO_PRE_VERDICT > $out/k31q20.jsonl 2>/dev/null
for g in 1 100 1000 60000; do GENES=$g python tools/ragged_rate.py 2>/dev/null | tail -1; done > $out/ragged.jsonl
cat $out/*.jsonl > $out/all.txt; wc -l $out/all.txt   # (tools/format_landscape.py $out turns them into the tables)
