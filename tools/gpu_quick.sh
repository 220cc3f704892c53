#!/bin/bash
# parity tests + three bench points (50 %, 0 %, 100 % on-target)
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/tq.log
for ot in 0.5 0.0 1.0; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --on-target $ot > gpurun_out/bq_$ot.json 2> gpurun_out/bq_$ot.err
done
cat gpurun_out/tq.log
python - <<'PY'
import json
for ot in ("0.5","0.0","1.0"):
    try:
        j=json.load(open("gpurun_out/bq_%s.json"%ot))
        r=j["roofline"]
        print(ot, "reads/s=%.1fM ms=%.2f kern_ms=%.2f alg=%.0fGB/s frac=%.4f"%(j["value"]/1e6,j["ms_per_step"],r["kernel_ms"],r["achieved"],r["frac"]))
    except Exception as e:
        print(ot,"ERR",e, open("gpurun_out/bq_%s.err"%ot).read()[-800:])
PY
