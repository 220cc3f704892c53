#!/usr/bin/env python3
"""End-to-end rate of the `shark` CLI (FASTQ files in, ssv + FASTQ files out) on synthetic 2x150 bp pairs.
usage: python tools/cli_rate.py [pairs] [extra shark args...]     env: ON_TARGET (default 0.02), CLI_T (e.g. "16,64,128"), HEADERS=var, GZ=1 (the sample files gzip -1 compressed: the inflate-bound path), RUN_ENVS="V=1;V=3" (the same files once per setting), TRIM=0.2 (that share of each mate file's reads cut to 100-150 bases)"""
import json, os, subprocess, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shark_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
on_target = float(os.environ.get('ON_TARGET', '0.02'))
extra = sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(5)
gene = synth.make_reference(1, 20000)[0]
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
L = 150
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
open(os.path.join(td, "g.fa"), "wb").write(b">gene0\n" + gene.tobytes() + b"\n")


VAR = os.environ.get("HEADERS", "fixed") == "var"   # read names without zero padding: records of several widths (the general reader)
TRIM = float(os.environ.get("TRIM", "0"))           # share of the pairs cut to 100-150 bases (both mates alike)


def write_fastq(path, mate):
    # runs of records of one width as (m, W) byte matrices: "@r<digits>/m\n" + seq + "\n+\n" + qual + "\n"
    chunk = 1_000_000
    with open(path, "wb") as f:
        b0 = 0
        while b0 < n:
            nd = len(str(b0)) if VAR else 9
            m = min(chunk, n - b0, (10 ** nd - b0) if VAR else n)
            H = 2 + nd + 3                                   # "@r" digits "/m\n"
            W = H + L + 3 + L + 1
            rec = np.empty((m, W), dtype=np.uint8)
            rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
            idx = np.arange(b0, b0 + m, dtype=np.int64)
            for d in range(nd):
                rec[:, 2 + d] = ord("0") + (idx // 10 ** (nd - 1 - d)) % 10
            rec[:, H - 3] = ord("/"); rec[:, H - 2] = ord("0") + mate; rec[:, H - 1] = 10
            prng = np.random.default_rng(1000 + b0)          # the same pairs are on-target in both mate files
            on = prng.random(m) < on_target
            st = prng.integers(0, len(gene) - 400, size=m)
            seqs = np.where(on[:, None], gene[st[:, None] + np.arange(L)[None, :]], acgt[rng.integers(0, 4, size=(m, L))])
            rec[:, H:H + L] = seqs
            rec[:, H + L] = 10; rec[:, H + L + 1] = ord("+"); rec[:, H + L + 2] = 10
            rec[:, H + L + 3:H + 2 * L + 3] = ord("I")
            rec[:, H + 2 * L + 3] = 10
            if TRIM > 0:
                # a share of the reads cut to a random length in [100, 150]: records of mixed widths (a trimmed sample)
                ln = np.where(prng.random(m) < TRIM, prng.integers(100, L + 1, size=m), L)
                col = np.arange(W)[None, :]
                keep = (col < H + ln[:, None]) | ((col >= H + L) & (col < H + L + 3 + ln[:, None])) | (col == W - 1)
                rec[keep].tofile(f)
                b0 += m
                continue
            rec.tofile(f)
            b0 += m


t0 = time.time()
write_fastq(os.path.join(td, "r1.fq"), 1)
write_fastq(os.path.join(td, "r2.fq"), 2)
gen_s = time.time() - t0
SUFFIX = ""
if os.environ.get("GZ", "0") == "1":
    ps = [subprocess.Popen(["gzip", "-1", os.path.join(td, f)]) for f in ("r1.fq", "r2.fq")]
    assert all(p.wait() == 0 for p in ps)
    SUFFIX = ".gz"
if os.environ.get('WARM', '1') == '1':      # read the files once, untimed: the first read of freshly written tmpfs pages pays for their LRU activation
    for f in ('r1.fq', 'r2.fq'):
        subprocess.run(['cat', os.path.join(td, f + SUFFIX)], stdout=subprocess.DEVNULL)
# CLI_T=16,64 runs the same files once per thread count
# RUN_ENVS="A=1;A=3;A=1" repeats every run once per entry with that variable set (A/B on the same files)
for t, run_env in [(t, e) for t in ([x for x in os.environ.get("CLI_T", "").split(",") if x] or [None]) for e in (os.environ.get("RUN_ENVS", "").split(";") if os.environ.get("RUN_ENVS") else [""])]:
    args = extra + (["-t", t] if t else [])
    env = dict(os.environ)
    if run_env:
        env[run_env.split("=")[0]] = run_env.split("=", 1)[1]
    for f in ("o1.fq", "o2.fq", "out.ssv"):      # (a run that has to truncate the 10 GB files of the run before pays for freeing their pages)
        if os.path.exists(os.path.join(td, f)):
            os.unlink(os.path.join(td, f))
    t0 = time.time()
    with open(os.path.join(td, "out.ssv"), "wb") as so:
        r = subprocess.run([os.path.join(root, "shark_amd", "bin", "shark"), "-r", os.path.join(td, "g.fa"), "-1", os.path.join(td, "r1.fq" + SUFFIX),
                            "-2", os.path.join(td, "r2.fq" + SUFFIX), "-o", os.path.join(td, "o1.fq"), "-p", os.path.join(td, "o2.fq"), "-v"] + args,
                           stdout=so, stderr=subprocess.PIPE, env=env)
    t1 = time.time()
    dt = t1 - t0
    ep = [float(l.split("(epoch ")[1].rstrip(")")) for l in r.stderr.decode().splitlines() if l.startswith("[shark/ms]") and "(epoch " in l]
    gaps = {"before_main_ms": round((ep[0] - t0) * 1e3, 1), "after_last_ms": round((t1 - ep[-1]) * 1e3, 1)} if ep else {}
    print(json.dumps({"pairs": n, "gaps": gaps, "cli_s": round(dt, 2), "reads_per_s_M": round(2 * n / dt / 1e6, 2), "rc": r.returncode, "gen_s": round(gen_s, 1),
                      "ssv_lines": sum(1 for _ in open(os.path.join(td, "out.ssv"), "rb")), "args": args, "run_env": run_env, "headers": "var" if VAR else "fixed", "gz": SUFFIX == ".gz",
                      "rss": [l[12:].replace("\t", " ") for l in r.stderr.decode().splitlines() if l.startswith("[shark/rss]")], "mem": [l[12:] for l in r.stderr.decode().splitlines() if l.startswith("[shark/mem]")], "writers": [l for l in r.stderr.decode().splitlines() if l.startswith("[shark/writers]")], "stderr_tail": r.stderr.decode()[-500:], "timeline": [l[11:].split(" (epoch")[0] for l in r.stderr.decode().splitlines() if l.startswith("[shark/ms]")]}), flush=True)
subprocess.run(["rm", "-rf", td])
