#!/bin/bash
# fixed costs of a `shark` process: the bundled example (5 000 pairs) end to end against its own timeline
cd "$(dirname "$0")/.."
ex=tests/golden/example
python3 - <<'PY'
import subprocess, time, os
ex = "tests/golden/example"
for rep in range(4):
    t0 = time.time()
    r = subprocess.run(["shark_amd/bin/shark", "-v", "-r", ex + "/ENSG00000277117.fa", "-1", ex + "/sample_1.fq", "-2", ex + "/sample_2.fq", "-o", "/tmp/o1.fq", "-p", "/tmp/o2.fq"],
                       capture_output=True)
    dt = time.time() - t0
    tl = [l[11:] for l in r.stderr.decode().splitlines() if l.startswith("[shark/ms]")]
    print("wall %.3f s | %s" % (dt, " | ".join(tl)))
for var in ({"HSA_ENABLE_SDMA": "0"}, {"HIP_VISIBLE_DEVICES": "0"}, {"GPU_MAX_HW_QUEUES": "1"}, {"HSA_ENABLE_INTERRUPT": "0"}):
    t0 = time.time()
    r = subprocess.run(["shark_amd/bin/shark", "-v", "-r", ex + "/ENSG00000277117.fa", "-1", ex + "/sample_1.fq", "-2", ex + "/sample_2.fq", "-o", "/tmp/o1.fq", "-p", "/tmp/o2.fq"],
                       capture_output=True, env=dict(os.environ, **var))
    dt = time.time() - t0
    tl = [l[11:] for l in r.stderr.decode().splitlines() if l.startswith("[shark/ms]")]
    print(var, "wall %.3f s | %s" % (dt, " | ".join(tl[-3:])))
PY
