#!/bin/bash
# the landscape table of profiles/README.md: kernel time per 10 M pairs over index sizes (2^33-bit filter), uniform and trimmed reads
export TMPDIR=/tmp
for g in 10 60 100 150 250 1000 10000; do
  python tests/scale_check.py --genes $g --bf-log2 33 --skip-bitvector --oracle-pairs 0 2>/dev/null | grep kernel_ms | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('uniform', $g, d['info']['n_set_bits'], d['mode'], d['kernel_ms'])"
done
for g in 1 100 60000; do GENES=$g python tools/ragged_rate.py 2>/dev/null | tail -1; done
