export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
python - <<'PY'
import time, numpy as np, torch
from shark_amd import SharkHip, synth
genes = synth.make_reference(1, 20000)
for env in (None, "1"):
    import os
    if env: os.environ["SHK_NO_KMER_TABLE"] = "1"
    t0=time.time(); h = SharkHip(k=17, c=0.6, bf_bits=1<<33); info=h.build([g.tobytes() for g in genes]); print("build", round(time.time()-t0,3), h.probe_mode(), info)
PY
for ot in 0.0 0.5 1.0; do python bench.py --no-configs --no-cpu-baseline --no-boundary --steps 3 --total-pairs 20000000 --on-target $ot 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('km', d['config']['on_target'], round(d['value']/1e6,1), d['roofline']['kernel_ms'], d['config']['assoc_per_step'], d['config']['probe_mode'], d['config']['index_build_s'])"; done
for ot in 0.0 0.5 1.0; do SHK_NO_KMER_TABLE=1 python bench.py --no-configs --no-cpu-baseline --no-boundary --steps 3 --total-pairs 20000000 --on-target $ot 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pos', d['config']['on_target'], round(d['value']/1e6,1), d['roofline']['kernel_ms'], d['config']['assoc_per_step'], d['config']['probe_mode'], d['config']['index_build_s'])"; done
