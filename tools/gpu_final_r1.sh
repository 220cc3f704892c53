#!/bin/bash
# round-1 record run: tests, official bench line, rocprofv3 kernel stats, PMC traffic passes, host-path rate
export TMPDIR=/tmp
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/final/tests.log
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/final/kt.json 2> gpurun_out/final/kt.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final/pmc_fetch.json 2> gpurun_out/final/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/final/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final/pmc_write.json 2> gpurun_out/final/pmc_write.err
python tools/host_path_rate.py > gpurun_out/final/host_path.json 2> gpurun_out/final/host_path.err
cat gpurun_out/final/tests.log gpurun_out/final/bench.json gpurun_out/final/host_path.json
