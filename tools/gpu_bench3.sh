#!/bin/bash
# three bench points, no tests; env passes through
export TMPDIR=/tmp
for ot in 0.5 0.0 1.0; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --on-target $ot 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); r=j['roofline']; print('$1', j['config']['probe_mode'], j['config']['on_target'], 'kern_ms=%.2f'%r['kernel_ms'], 'reads/s=%.0fM'%(j['value']/1e6))"
done
