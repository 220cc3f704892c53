#!/bin/bash
# the whole GPU suite (incl. the configs[2]/[4] scale tests), then the bench line -- what the driver runs at round end
export TMPDIR=/tmp
mkdir -p gpurun_out/suite
( time python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/suite/tests.log 2>&1
tail -25 gpurun_out/suite/tests.log
( time python bench.py ) > gpurun_out/suite/bench.json 2> gpurun_out/suite/bench.err
tail -5 gpurun_out/suite/bench.err
cat gpurun_out/suite/bench.json
