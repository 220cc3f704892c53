#!/bin/bash
# kernel experiments: 8 vs 6 waves per SIMD for classify_uni_kernel; landscape over index sizes
export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
run() { python bench.py --no-configs --no-cpu-baseline --no-boundary --steps 3 --total-pairs 20000000 --on-target $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['config']['on_target'], round(d['value']/1e6,1), d['roofline']['kernel_ms'], d['ms_per_step'])"; }
for ot in 0.0 0.5 1.0; do run $ot w8; done
for ot in 0.0 0.5 1.0; do SHK_LIB_PATH=$PWD/tools/variants/w6.so run $ot w6; done
for g in 10 100 1000 10000; do
  python tests/scale_check.py --genes $g --bf-log2 33 --oracle-pairs 0 --skip-bitvector 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('genes', $g, d['mode'], d['kernel_ms'], d['reads_per_s'], d['info']['n_set_bits'])"
done
