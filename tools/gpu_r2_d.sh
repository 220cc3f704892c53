#!/bin/bash
# after generalising classify_uni_kernel to the table modes: parity, fuzz, landscape over index sizes
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python tests/fuzz_parity.py 800 31337 2>&1 | tail -3
FUZZ_UNI=1 timeout 600 python tests/fuzz_parity.py 150 4242 2>&1 | tail -2
for g in 10 100 1000 10000 60000; do
  bl=33; if [ $g = 60000 ]; then bl=36; fi
  python tests/scale_check.py --genes $g --bf-log2 $bl --oracle-pairs 0 --skip-bitvector 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('genes', $g, d['mode'], d['kernel_ms'], d['reads_per_s'], d['info']['n_set_bits'])"
done
python tests/scale_check.py --genes 60000 --bf-log2 37 --k 31 --q 20 --single --oracle-pairs 0 --skip-bitvector 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4', d['mode'], d['kernel_ms'], d['reads_per_s'])"
