#!/bin/bash
# one gpurun call: parity tests, bench variants, rocprofv3 kernel trace + PMC passes
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/t3.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --on-target 0.0 > gpurun_out/b_off.json 2> gpurun_out/b_off.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --on-target 1.0 > gpurun_out/b_on.json 2> gpurun_out/b_on.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_kt.json 2> gpurun_out/prof_kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch.json 2> gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write.json 2> gpurun_out/pmc_write.err
cat gpurun_out/t3.log; cat gpurun_out/b_off.json gpurun_out/b_on.json; find gpurun_out -name '*.csv' | head -30
