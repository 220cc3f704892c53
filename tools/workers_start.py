#!/usr/bin/env python3
"""what N workers cost a `shark` process before its first batch: the bundled example (5 000 pairs) with --devices 0 / 0,0 / 0,0,0,0, several
runs each, the command's own timeline ("contexts created", "index built", "contexts destroyed") and the wall time from outside"""
import os, subprocess, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ex = os.path.join(ROOT, "tests", "golden", "example")
exe = os.path.join(ROOT, "shark_amd", "bin", "shark")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for devs in ("0", "0,0", "0,0,0,0"):
    rows = []
    for rep in range(reps):
        t0 = time.time()
        r = subprocess.run([exe, "-v", "-r", ex + "/ENSG00000277117.fa", "-1", ex + "/sample_1.fq", "-2", ex + "/sample_2.fq", "-o", "/tmp/o1.fq", "-p", "/tmp/o2.fq",
                            "--devices", devs], capture_output=True)
        dt = time.time() - t0
        st = {}
        for l in r.stderr.decode().splitlines():
            if l.startswith("[shark/ms] "):
                nm, ms = l[11:].split(" (epoch")[0].rsplit(" ", 1)
                st[nm] = float(ms) / 1e3
        rows.append((dt, st.get("contexts created", 0.0), st.get("index built", 0.0), st.get("contexts destroyed", 0.0), r.returncode))
    med = lambda i: statistics.median(x[i] for x in rows[1:])       # (the first run of a shape warms the file cache)
    print("--devices %-8s wall %.3f s | contexts created %.3f | index built %.3f | contexts destroyed %.3f | rc %s | all: %s"
          % (devs, med(0), med(1), med(2), med(3), set(x[4] for x in rows), " ".join("%.3f" % x[1] for x in rows)), flush=True)
