#!/bin/bash
# Profiles of the classify kernels for profiles/ (run on the GPU box through gpurun; then `python tools/collect_profiles.py rNN` here).
#   kt       : rocprofv3 --kernel-trace --stats of the bench command (headline workload + configs[2] index + configs[4] shape)
#   kt_trimmed: the same of tools/ragged_rate.py 10000000 100 0.8 (a batch with 20 % of its mates trimmed)
#   counters : `bench.py --profile-passes` = the counter passes bench.py itself makes live (one rocprofv3 --pmc child per counter set,
#              --kernel-trace only beside --pmc: gpurun refuses anything else), plus a fourth SQ set, for the headline workload at
#              0 / 50 / 100 % on-target pairs, the configs[2] index and both quality models of the configs[4] shape
export TMPDIR=/tmp
OUT=gpurun_out/profiles
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 2 --warmup 1 --reps-per-step 4 --no-cpu-baseline --no-boundary --no-cli --no-live-counters --no-trimmed > $OUT/kt.json 2> $OUT/kt.err || { tail -5 $OUT/kt.err; exit 1; }
# keep only the stats summary of the trace (the per-dispatch CSVs are large)
find $OUT/kt -name "*_kernel_trace.csv" -delete
# a trimmed batch (20 % of the mates cut to 100-150 bases): the passes over the offsets and the class-by-class instantiation
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_trimmed -- python3 tools/ragged_rate.py 10000000 100 0.8 > $OUT/kt_trimmed.json 2> $OUT/kt_trimmed.err || { tail -5 $OUT/kt_trimmed.err; exit 1; }
find $OUT/kt_trimmed -name "*_kernel_trace.csv" -delete
timeout -k 10 1100 python3 bench.py --profile-passes $OUT/counters_raw.json 2> $OUT/counters.err || { tail -5 $OUT/counters.err; exit 1; }
tail -3 $OUT/counters.err
ls -R $OUT | head -40
