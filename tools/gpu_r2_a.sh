#!/bin/bash
# round-2 first pass: whole GPU suite (incl. the configs[2]/[4] scale tests), then the bench line
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
( time python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/r2a/tests.log 2>&1
tail -25 gpurun_out/r2a/tests.log
( time python bench.py ) > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
tail -5 gpurun_out/r2a/bench.err
cat gpurun_out/r2a/bench.json
