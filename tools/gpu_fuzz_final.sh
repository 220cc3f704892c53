#!/bin/bash
# smoke() + a long fuzz of the final library
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/gpu_fuzz_long.sh 3000 6260001 1200 7270001
