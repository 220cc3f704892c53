#!/bin/bash
# scratch: per-launch tail of the headline bench (kernel trace)
export TMPDIR=/tmp
rm -rf gpurun_out/tail; mkdir -p gpurun_out/tail
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tail/kt -- python3 bench.py --steps 5 --warmup 1 --no-configs --no-boundary --no-cpu-baseline --no-cli > gpurun_out/tail/b.json 2> gpurun_out/tail/b.err
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tail/kt/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if ("shk::" in n or "rocclr" in n) and int(r["Calls"]) >= 40:
        print(r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), n[:90])
PY
rm -rf gpurun_out/tail/kt
