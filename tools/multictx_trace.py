#!/usr/bin/env python3
"""`shark --gpus N --devices 0,0,...` under rocprofv3 --kernel-trace: do the workers' kernels really run side by side?
usage: python3 tools/multictx_trace.py <out-dir> [workers] [pairs]
Writes <out-dir>/multictx_trace.json: per HIP stream (= per worker pipeline) the classify kernels it ran and their busy time,
and how much of that time kernels of DIFFERENT workers overlapped.  The command is the binary itself behind `--`."""
import csv, glob, json, os, pathlib, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import synth
from tests.test_gpu_multictx import _write_pairs, _write_fasta

out = pathlib.Path(sys.argv[1]); out.mkdir(parents=True, exist_ok=True)
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
td = pathlib.Path(tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"))
rng = np.random.default_rng(20260502)
genes = synth.make_genes(rng, 40, 600, 3000, share_every=3)
_write_fasta(td / "g.fa", genes)
f1, f2 = _write_pairs(td, rng, genes, pairs, 100, 0.4)
exe = os.path.join(root, "shark_amd", "bin", "shark")
common = ["-r", str(td / "g.fa"), "-1", f1, "-2", f2, "-t", "8", "--batch", os.environ.get("TRACE_BATCH", "20000")]
one = subprocess.run([exe] + common + ["-o", str(td / "a1"), "-p", str(td / "a2"), "--gpus", "1"], capture_output=True)
assert one.returncode == 0, one.stderr.decode()[-2000:]
prof = out / "kt"
cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", str(prof), "--", exe] + common + \
      ["-o", str(td / "b1"), "-p", str(td / "b2"), "--gpus", str(workers), "--devices", ",".join(["0"] * workers)]
r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, TMPDIR="/tmp"))
assert r.returncode == 0, r.stderr.decode()[-3000:]
same = r.stdout == one.stdout and (td / "b1").read_bytes() == (td / "a1").read_bytes() and (td / "b2").read_bytes() == (td / "a2").read_bytes()
rows = []
for f in glob.glob(str(prof / "**" / "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        rows.append((row.get("Stream_Id") or row.get("Queue_Id"), row["Kernel_Name"].split("(")[0].replace("void shk::", ""),
                     int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row.get("Queue_Id")))
cls = [x for x in rows if x[1].startswith("classify_")]
streams = {}
for s, k, a, b, q in cls:
    e = streams.setdefault(s, {"queue": q, "kernels": 0, "busy_ns": 0, "names": {}})
    e["kernels"] += 1; e["busy_ns"] += b - a; e["names"][k] = e["names"].get(k, 0) + 1
# time during which classify kernels of at least two different streams were running at once (sweep over start / end events)
ev = sorted([(a, 1, s) for s, k, a, b, q in cls] + [(b, -1, s) for s, k, a, b, q in cls])
live, last, overlap_ns, any_ns = {}, None, 0, 0
for t, d, s in ev:
    if last is not None:
        n = sum(1 for v in live.values() if v > 0)
        if n >= 1: any_ns += t - last
        if n >= 2: overlap_ns += t - last
    live[s] = live.get(s, 0) + d
    last = t
# the workers' pipelines in time: each stream's span from its first to its last classify kernel, and how often the stream changes
# from one classify kernel to the next in start order (one worker after the other would change once)
spans = {s: [min(a for s2, k, a, b, q in cls if s2 == s), max(b for s2, k, a, b, q in cls if s2 == s)] for s in streams}
t0 = min(v[0] for v in spans.values()) if spans else 0
order = [s for s, k, a, b, q in sorted(cls, key=lambda x: x[2])]
changes = sum(1 for i in range(1, len(order)) if order[i] != order[i - 1])
for s in streams:
    streams[s]["span_ms"] = [(spans[s][0] - t0) / 1e6, (spans[s][1] - t0) / 1e6]
res = {"stream_changes_in_start_order": changes, "command": " ".join(cmd[6:]).replace(str(td), "<tmp>"), "workers": workers, "pairs": pairs, "outputs_equal_to_one_worker": bool(same),
       "classify_kernel_streams": streams, "classify_busy_any_ms": any_ns / 1e6, "classify_overlap_of_two_or_more_streams_ms": overlap_ns / 1e6,
       "overlap_fraction": (overlap_ns / any_ns) if any_ns else None, "kernel_rows": len(rows)}
(out / "multictx_trace.json").write_text(json.dumps(res, indent=1))
print(json.dumps(res))
for f in glob.glob(str(prof / "**" / "*"), recursive=True):
    if os.path.isfile(f): os.remove(f)
import shutil; shutil.rmtree(td)
sys.exit(0 if same and len(streams) >= workers else 1)
