#!/usr/bin/env python3
"""bench.py -- throughput of the k-mer classification hot path on MI355X.

A "step" is one pass of the hot path (shk_classify_device: FastqSplitter
join/mask semantics + ReadAnalyzer + BF::get_index)
over synthetic read pairs that are already resident in HBM.

Headline workload (BASELINE.json configs[1], the configuration the metric is
quoted on): 1 gene x 20 kb uniform-ACGT reference, synthetic 2x150 bp pairs
(50 % on-target, 1 % substitutions, 0.2 % N), k=17, c=0.6, 2^33-bit filter,
classified in launches of 10 M pairs.

Scaling (--scaling, default strong): ONE fixed read set of --total-pairs pairs
per step (default 80 M = 8 launches of 10 M pairs; chunk c is generated from
seed SEED+1+c whatever the number of GPUs) is split over the ranks, so
N GPUs classify the same reads 1 GPU does and `gene_count_checksum` /
`assoc_per_step` must not change with N.  Every rank holds its own replica of
the index (rebuilt deterministically; build time reported separately, not
timed); there is no data-path collective; the per-gene assigned-read counts
are all-reduced over RCCL once, inside the timed region, by the library's own
shk_dist_gene_counts_allreduce.  `--scaling weak` gives every rank its own
10 M-pair batch instead (round-1 behaviour).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
  roofline       `achieved`/`peak`/`frac`: algorithmic bytes of the classify
                 kernel / its HIP-event duration against the 8 TB/s HBM peak
                 (the contract's figure); `bound` names the resource that
                 really binds the kernel and `frac_of_binding` how much of THAT
                 is used.  `valu_from_profile` (instruction-issue occupancy, SQ
                 counters), `hbm_actual_from_profile` and `traffic`
                 (FETCH/WRITE counter bytes) come from the committed
                 profiles/pmc_counters.json -- a separate rocprofv3 --pmc pass,
                 not this run -- and only when that file was taken on the same
                 kernel sources (sha256-stamped)
  cpu_baseline   the CPU oracle (a port of the reference path) on this host
  configs        the same measurement on BASELINE configs[2]'s index
                 (60 000 genes, 2^36-bit filter) and on configs[4]'s shape
                 (k=31, -q 20, --single, 2^37 bits), 10 M-pair steps
  batch_boundary PCIe-inclusive rate of the host-buffer entry points (never
                 `value`)
"""
import argparse
import hashlib
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver (already exported on the pool)
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD = 256 * 4            # 256 CUs x 4 SIMD-32
CLK_GHZ = 2.4               # max clock; a wave64 VALU instruction issues over 2 cycles
LAUNCH_PAIRS = 10_000_000
L2_LINE_BYTES = 128         # a memory-side request of gfx950's L2 is a 128-byte line (MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B)
KERNEL_SOURCES = ["classify_uni.hpp", "classify_common.hpp", "classify.hip", "kmer_device.hpp", "shark_internal.hpp", "lds_table.hpp"]


def kernel_src_sha():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "shark_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def load_pmc(k, bf_log2):
    """profiles/pmc_counters.json, only when it was taken on the kernel sources being run -> (dict or None, note)"""
    pfile = os.path.join(ROOT, "profiles", "pmc_counters.json")
    sha = kernel_src_sha()
    if not os.path.exists(pfile):
        return None, "profiles/pmc_counters.json absent"
    try:
        pj = json.load(open(pfile))
    except Exception as ex:   # a broken profile file must not break the bench line
        return None, "profiles/pmc_counters.json unreadable: %r" % (ex,)
    if pj.get("kernel_src_sha") != sha:
        return None, "profiles/pmc_counters.json was taken on other kernel sources (%s, now %s): not used" % (pj.get("kernel_src_sha"), sha)
    if pj.get("k") != k or pj.get("bf_log2") != bf_log2:
        return None, "profiles/pmc_counters.json has no entry for this workload"
    return pj, "counters from profiles/pmc_counters.json (commit %s, same kernel sources)" % pj.get("commit")


def cpu_info():
    model, phys = "", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return model, len(phys)


def spawn_ranks(n):
    """`python3 bench.py --gpus N` without a launcher: this process starts the N ranks itself, one process per GPU -- as the
    reference starts its N workers from the one command line (main.cpp:219-223) -- forwards rank 0's JSON line and exits
    non-zero when any rank does.  It never imports torch and never touches a GPU (a process that has initialised the GPU
    must not be replaced or forked on this pool); the ranks are ordinary child processes of it."""
    import socket
    import subprocess
    import threading
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0:
                    rc = c if c > 0 else 1
                    print("bench.py: rank %d of %d exited with code %d" % (r, n, c), file=sys.stderr)
        time.sleep(0.05)
    for r in live:          # a rank failed: the others would wait for it at the next barrier -- stop exactly the processes started here
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    reader.join(timeout=20)
    sys.stdout.write(b"".join(lines).decode(errors="replace"))
    sys.stdout.flush()
    sys.exit(rc)


def cli_end_to_end(args, genes, dev, h, L):
    """`shark` (shark_amd/bin/shark, the reference's command line: main.cpp:83-240, README.md:47-52) on FASTQ files: synthetic pairs of
    the headline shape are written to /dev/shm (untimed), the command is run as a user would run it and its wall time taken from
    outside; the number of ssv lines must equal the number of associations the library returns for the same pairs resident in HBM.
    On-target rate 2 %: one gene against a whole sample is what a one-gene reference is used for (at 50 % the run would measure
    the writing of 5 GB of output FASTQ)."""
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    import torch
    from shark_amd import synth
    n, ot = args.cli_pairs, 0.02
    exe = os.path.join(ROOT, "shark_amd", "bin", "shark")
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    if shutil.disk_usage(base).free < 2.6 * n * 2 * (2 * L + 20):
        return {"skipped": "not enough room in %s for %d pairs" % (base, n)}
    td = tempfile.mkdtemp(dir=base)
    try:
        t0 = time.time()
        b = synth.make_pairs_device(n, genes, dev, seed=synth.SEED + 99, read_len=L, on_target=ot)
        r = h.classify_device(n, b["seq1"].data_ptr(), b["off1"].data_ptr(), b["seq2"].data_ptr(), b["off2"].data_ptr(), max_read_len=L)
        want_lines = int(r.n_assoc)
        nd = 9
        H = 2 + nd + 3
        W = H + L + 3 + L + 1
        idx = torch.arange(n, device=dev, dtype=torch.int64)
        for mate, key in ((1, "seq1"), (2, "seq2")):
            with open(os.path.join(td, "r%d.fq" % mate), "wb") as f:
                for c0 in range(0, n, 2_000_000):            # "@r<9 digits>/<mate>\n" + bases + "\n+\n" + qualities + "\n", 2 M records at a time
                    m = min(2_000_000, n - c0)
                    rec = torch.empty((m, W), dtype=torch.uint8, device=dev)
                    rec[:, 0] = ord("@")
                    rec[:, 1] = ord("r")
                    ii = idx[c0:c0 + m]
                    for d in range(nd):
                        rec[:, 2 + d] = (ord("0") + (ii // 10 ** (nd - 1 - d)) % 10).to(torch.uint8)
                    rec[:, H - 3] = ord("/")
                    rec[:, H - 2] = ord("0") + mate
                    rec[:, H - 1] = 10
                    rec[:, H:H + L] = b[key][c0 * L:(c0 + m) * L].view(m, L)
                    rec[:, H + L] = 10
                    rec[:, H + L + 1] = ord("+")
                    rec[:, H + L + 2] = 10
                    rec[:, H + L + 3:H + 2 * L + 3] = ord("I")
                    rec[:, H + 2 * L + 3] = 10
                    rec.cpu().numpy().tofile(f)
                    del rec
        del b
        with open(os.path.join(td, "g.fa"), "wb") as f:
            for gi, g in enumerate(genes):
                f.write(b">gene%d\n" % gi + g.tobytes() + b"\n")
        for mate in (1, 2):   # one untimed read: the first read of freshly written tmpfs pages pays for their LRU activation, four times a later one
            subprocess.run(["cat", os.path.join(td, "r%d.fq" % mate)], stdout=subprocess.DEVNULL)
        gen_s = time.time() - t0
        runs = []
        for threads in (12, 32):
            t0 = time.time()
            with open(os.path.join(td, "out.ssv"), "wb") as so:
                pr = subprocess.run([exe, "-r", os.path.join(td, "g.fa"), "-1", os.path.join(td, "r1.fq"), "-2", os.path.join(td, "r2.fq"),
                                     "-o", os.path.join(td, "o1.fq"), "-p", os.path.join(td, "o2.fq"), "-k", str(args.k), "-v", "-t", str(threads)],
                                    stdout=so, stderr=subprocess.PIPE)
            dt = time.time() - t0
            lines = 0
            with open(os.path.join(td, "out.ssv"), "rb") as so:
                for blk in iter(lambda: so.read(1 << 24), b""):
                    lines += blk.count(b"\n")
            stages = {}
            for ln in pr.stderr.decode(errors="replace").splitlines():
                if ln.startswith("[shark/ms] "):
                    nm, ms = ln[11:].split(" (epoch")[0].rsplit(" ", 1)
                    stages[nm] = round(float(ms) / 1e3, 3)
            runs.append({"threads": threads, "wall_s": round(dt, 3), "value": round(2 * n / dt, 1), "unit": "reads/s", "rc": pr.returncode,
                         "ssv_lines": lines, "ssv_lines_equal_device_result": lines == want_lines, "stage_s_since_start": stages,
                         "fastq_out_bytes": os.path.getsize(os.path.join(td, "o1.fq")) + os.path.getsize(os.path.join(td, "o2.fq"))})
        best = max(runs, key=lambda x: x["value"] if (x["rc"] == 0 and x["ssv_lines_equal_device_result"]) else 0.0)
        out = {"what": "shark_amd/bin/shark -r g.fa -1 r1.fq -2 r2.fq -o o1.fq -p o2.fq -t T > out.ssv on %d pairs 2x%d bp in %s (files written and read once "
                       "beforehand, untimed; on-target rate %.2f); wall time of the whole process from outside" % (n, L, base, ot),
               "pairs": n, "on_target": ot, "input_bytes": 2 * n * W, "expected_ssv_lines": want_lines, "generate_s": round(gen_s, 1), "runs": runs}
        out.update({k: best[k] for k in ("threads", "wall_s", "value", "unit", "ssv_lines", "ssv_lines_equal_device_result")})
        return out
    finally:
        shutil.rmtree(td, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--total-pairs", type=int, default=80_000_000, help="strong scaling: pairs per step over ALL GPUs")
    ap.add_argument("--pairs", type=int, default=LAUNCH_PAIRS, help="pairs per launch (and per GPU per step with --scaling weak)")
    ap.add_argument("--k", type=int, default=17)
    ap.add_argument("--bf-log2", type=int, default=33)
    ap.add_argument("--genes", type=int, default=1)
    ap.add_argument("--gene-len", type=int, default=20000)
    ap.add_argument("--on-target", type=float, default=0.5, help="fraction of pairs drawn from a gene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs[2] workload")
    ap.add_argument("--no-boundary", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurement")
    ap.add_argument("--no-cli", action="store_true", help="skip the end-to-end run of the shark command line")
    ap.add_argument("--cli-pairs", type=int, default=16_000_000)
    ap.add_argument("--cpu-sample-pairs", type=int, default=0, help="0 = threads x 50 000 (one reference chunk per thread)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)           # (does not return)

    import numpy as np
    import torch
    from shark_amd import SharkHip
    from shark_amd import dist as sdist
    from shark_amd import synth
    from shark_amd.capi import hip_memcpy_dtoh

    rank, local_rank, world = sdist.env_rank()
    if world != args.gpus:
        # the line's n_gpus must be what was asked for: a launcher that started another number of ranks is an error, not a warning
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python3 bench.py --gpus N`, or with a launcher whose "
                 "--nproc-per-node equals --gpus)" % (args.gpus, world))
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        sys.exit("bench.py: no GPU (there is no CPU path)")
    if world > n_dev and sdist.backend_name() == "nccl":
        sys.exit("bench.py: --gpus %d but this node has %d GPU(s); one process per GPU over RCCL needs a GPU per rank "
                 "(SHARK_DIST_BACKEND=gloo rehearses the N-rank path on fewer)" % (world, n_dev))
    local_rank %= n_dev                    # (a gloo dry run may put several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sdist.init(sdist.backend_name(), dev)

    k, c, bf_bits = args.k, 0.6, 1 << args.bf_log2
    L = 150
    lp = args.pairs

    # ---- which chunks of the read set this rank owns -----------------------------
    if args.scaling == "strong":
        n_chunks = max(world, (args.total_pairs + lp - 1) // lp)
        n_chunks = (n_chunks + world - 1) // world * world           # equal shards
        chunk_pairs = args.total_pairs // n_chunks
        my_chunks = list(range(rank * (n_chunks // world), (rank + 1) * (n_chunks // world)))
        pairs_per_step_all = chunk_pairs * n_chunks
    else:
        chunk_pairs = lp
        my_chunks = [rank]
        pairs_per_step_all = lp * world

    def barrier():
        torch.cuda.synchronize()
        sdist.barrier()
        torch.cuda.synchronize()

    # ---- index: replicated by deterministic rebuild on every GPU (not timed) -------
    genes = synth.make_reference(args.genes, args.gene_len)
    t0 = time.time()
    h = SharkHip(k=k, c=c, bf_bits=bf_bits, device=local_rank)
    info = h.build([g.tobytes() for g in genes])
    t_build = time.time() - t0
    h.dist_init(sdist)                                 # RCCL communicator inside the library (no-op for world 1)
    comm_rank, ranks_seen = h.dist_info()              # what that communicator says (ncclCommCount), not what the environment said
    if ranks_seen != world or comm_rank != rank:
        sys.exit("bench.py: rank %d of %d, but the communicator reports rank %d of %d" % (rank, world, comm_rank, ranks_seen))

    # ---- this rank's shard of the read set, generated in HBM ------------------------
    batches = [synth.make_pairs_device(chunk_pairs, genes, dev, seed=synth.SEED + 1 + cidx, read_len=L, on_target=args.on_target)
               for cidx in my_chunks]
    torch.cuda.synchronize()
    ptrs = [{kk: (v.data_ptr() if v is not None else 0) for kk, v in b.items()} for b in batches]

    def step(hh, pp, n):
        res = None
        for p in pp:
            res = hh.classify_device(n, p["seq1"], p["off1"], p["seq2"], p["off2"], p["qual1"], p["qual2"], max_read_len=L)
        return res

    def timed(hh, pp, n, steps, warmup, reduce_counts):
        for _ in range(warmup):
            step(hh, pp, n)
        hh.gene_counts_reset()
        hh.timing_enable(True)
        barrier()
        t0 = time.perf_counter()
        res = None
        n_assoc = 0
        for _ in range(steps):
            for p in pp:
                res = hh.classify_device(n, p["seq1"], p["off1"], p["seq2"], p["off2"], p["qual1"], p["qual2"], max_read_len=L)
                n_assoc += int(res.n_assoc)
        counts = hh.dist_gene_counts_allreduce() if reduce_counts else None    # the path's one exchange step (RCCL)
        barrier()
        dt = time.perf_counter() - t0
        tm = hh.timing()
        hh.timing_enable(False)
        return sdist.max_over_ranks(dt, dev), tm, res, n_assoc, counts

    dt, tm, res, n_assoc_local, counts = timed(h, ptrs, chunk_pairs, args.steps, args.warmup, True)
    n_assoc = torch.tensor([n_assoc_local], dtype=torch.int64, device=dev)
    sdist.allreduce_sum_(n_assoc)
    reads_per_step = 2 * pairs_per_step_all
    value = reads_per_step * args.steps / dt

    # ---- BASELINE configs[2] index (60 000 genes, 2^36 bits): every rank, 10 M-pair steps -----
    cfg2 = cfg4 = None
    if not args.no_configs:
        h.close()
        for b in batches[1:]:
            b.clear()
        g2 = synth.make_gencode_like_reference(60000)
        t0 = time.time()
        h2 = SharkHip(k=17, c=0.6, bf_bits=1 << 36, device=local_rank)
        info2 = h2.build([g.tobytes() for g in g2])
        t_build2 = time.time() - t0
        b2 = synth.make_pairs_device(lp, g2, dev, seed=synth.SEED + 7 + rank, read_len=L, on_target=0.5)
        torch.cuda.synchronize()
        p2 = [{kk: (v.data_ptr() if v is not None else 0) for kk, v in b2.items()}]
        steps2 = max(2, min(args.steps, 10))
        dt2, tm2, _, n_assoc2, _ = timed(h2, p2, lp, steps2, 1, False)
        k2 = tm2["total_ms"] / max(tm2["n_launches"], 1)
        w2 = h2.count_work(lp, p2[0]["seq1"], p2[0]["off1"], p2[0]["seq2"], p2[0]["off2"]) if rank == 0 else None
        tab2_bytes = 16 << 29     # the configs[2] index's position table: 2^29 buckets of 16 bytes = 8 GiB (1.69e8 set bits at load <= 0.3)
        cfg2 = {"workload": "configs[2] index: 60000 genes (1.78e8 bases, lognormal lengths, every 10th gene shares half of its predecessor), "
                            "%d pairs 2x150 bp per GPU per step, k=17 c=0.6 bf=2^36 bits" % lp,
                "value": round(2 * lp * world * steps2 / dt2, 1), "unit": "reads/s", "n_gpus": world, "steps": steps2,
                "ms_per_step": round(dt2 / steps2 * 1e3, 3), "kernel_ms": round(k2, 4), "probe_mode": h2.probe_mode(),
                "index_build_s": round(t_build2, 3), "n_set_bits": int(info2["n_set_bits"]), "tot_idx": int(info2["tot_idx"]),
                "assoc_per_step": n_assoc2 // steps2, "tie_reads": int(tm2["last_n_tie"])}
        if rank == 0:
            # This index is bound by the RATE of random memory-side requests: one 16-byte bucket per k-mer that is looked up, each
            # a 128-byte line of an 8 GiB table.  The ceiling is measured here, in this run, with the library's own measurement
            # entry point (same device, same table size, plain and streaming loads, the better of the two); the requests the
            # classify kernel makes per launch are the L2 misses of the counter pass on the same kernel sources (the bound cut,
            # the early decision and the anchored extension make fewer lookups than the reads hold k-mers).
            ceil_plain = h2.measure_random_lookups(tab2_bytes, 1 << 31, False)
            ceil_nt = h2.measure_random_lookups(tab2_bytes, 1 << 31, True)
            ceiling = max(ceil_plain, ceil_nt)
            alg2 = w2["n_bases"] + 8 * w2["n_kmers"] + 16 * w2["n_hits"] + 2 * w2["n_list_ids"] + 8 * (2 * lp)
            rl = {"bound": "memory-side request rate (random 128-B lines behind L2)", "kmers_in_reads": int(w2["n_kmers"]), "hits": int(w2["n_hits"]),
                  "algorithmic_bytes_per_launch": int(alg2), "achieved": round(alg2 / (k2 * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                  "frac": round(alg2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                  "G_kmers_per_s": round(w2["n_kmers"] / (k2 * 1e-3) / 1e9, 1),
                  "ceiling_G_lookups_per_s": round(ceiling, 1), "ceiling_measured": {"table_bytes": tab2_bytes, "plain": round(ceil_plain, 1),
                                                                                    "streaming": round(ceil_nt, 1), "how": "shk_measure_random_lookups, this run"}}
            pj2, note2 = load_pmc(17, 33)
            e2 = pj2.get("workloads", {}).get("configs2") if pj2 else None
            if e2 and e2.get("TCC_MISS_sum"):
                req = e2["TCC_MISS_sum"]
                rl["memory_side_requests_per_launch_from_profile"] = int(req)
                rl["G_requests_per_s"] = round(req / (k2 * 1e-3) / 1e9, 1)
                rl["frac_of_binding"] = round(min(1.0, req / (k2 * 1e-3) / 1e9 / ceiling), 3)
                rl["line_GBps"] = round(L2_LINE_BYTES * req / (k2 * 1e-3) / 1e9, 1)
                rl["frac_of_hbm_peak"] = round(L2_LINE_BYTES * req / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 3)
                rl["frac_of_hbm_peak_at_64B_sectors"] = round(64 * req / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 3)
            rl["counters"] = note2
            cfg2["roofline"] = rl
        h2.close()
        del b2
        # BASELINE configs[4] shape on this GPU's shard: k=31, -q 20, --single, 2^37-bit filter (the quality-mask path at max k)
        t0 = time.time()
        h4 = SharkHip(k=31, c=0.6, bf_bits=1 << 37, min_quality=20, single=True, device=local_rank)
        info4 = h4.build([g.tobytes() for g in g2])
        t_build4 = time.time() - t0
        cfg4 = None
        for qual_model in ("uniform", "ends"):
            # "ends": low qualities at the 3' end of the reads (1.2 % of the bases below Q20); "uniform": rounds 1-2's model, 10 % of the
            # bases Q2-29 anywhere, under which a 31-mer survives -q 20 with 0.13 and the hit path is barely exercised (shark_amd/synth.py)
            b4 = synth.make_pairs_device(lp, g2, dev, seed=synth.SEED + 7 + rank, read_len=L, on_target=0.5, with_qual=True, qual_model=qual_model)
            torch.cuda.synchronize()
            p4 = [{kk: (v.data_ptr() if v is not None else 0) for kk, v in b4.items()}]
            dt4, tm4, _, n_assoc4, _ = timed(h4, p4, lp, steps2, 1, False)
            k4 = tm4["total_ms"] / max(tm4["n_launches"], 1)
            below_q20 = float((b4["qual1"][:1 << 24] < 53).float().mean().item())
            e4 = {"qualities": qual_model, "frac_bases_below_q20": round(below_q20, 4),
                  "value": round(2 * lp * world * steps2 / dt4, 1), "unit": "reads/s", "ms_per_step": round(dt4 / steps2 * 1e3, 3),
                  "kernel_ms": round(k4, 4), "assoc_per_step": n_assoc4 // steps2, "frac_pairs_assigned": round(n_assoc4 / steps2 / lp, 4)}
            if qual_model == "uniform":
                old4 = e4
            else:
                cfg4 = {"workload": "configs[4] shape: the same 60000 genes, %d pairs 2x150 bp with qualities (Q30-41, low 3' tails: %.1f %% of the bases "
                                    "below Q20) per GPU per step, k=31 c=0.6 -q 20 --single bf=2^37 bits" % (lp, 100 * below_q20),
                        "n_gpus": world, "steps": steps2, "probe_mode": h4.probe_mode(), "index_build_s": round(t_build4, 3),
                        "n_set_bits": int(info4["n_set_bits"])}
                cfg4.update(e4)
                cfg4["rounds_1_2_workload"] = old4
            del b4, p4
        h4.close()
        # the headline context again for the roofline counters / cpu sample below
        h = SharkHip(k=k, c=c, bf_bits=bf_bits, device=local_rank)
        h.build([g.tobytes() for g in genes])
        res = step(h, ptrs[:1], chunk_pairs)

    if rank != 0:
        sdist.finalize()
        return

    n = chunk_pairs
    ptr, batch = ptrs[0], batches[0]
    # ---- roofline of the dominant kernel (classify_fast_kernel) ------------------
    # exact algorithmic bytes of one launch:
    #   bases (+quals) + 8 B per probed k-mer (one 64-bit filter word)
    #   + per hit 8 B rank word + 8 B CSR offsets + 2 B per gene id + 8 B result per read
    w = h.count_work(n, ptr["seq1"], ptr["off1"], ptr["seq2"], ptr["off2"])
    alg_bytes = w["n_bases"] + 8 * w["n_kmers"] + 16 * w["n_hits"] + 2 * w["n_list_ids"] + 8 * (2 * n)
    kern_ms = tm["total_ms"] / max(tm["n_launches"], 1)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    sha = kernel_src_sha()
    traffic, valu, hbm_actual = None, None, None
    pj, prof_note = load_pmc(k, args.bf_log2)
    e = pj.get("workloads", {}).get("configs1_ot%.2f" % args.on_target) if pj else None
    if pj and (not e or e.get("pairs") != n):
        prof_note = "profiles/pmc_counters.json has no entry for this workload"
    elif pj:
        prof_note += ", per launch of %d pairs" % n
        fetch, write = e["FETCH_SIZE_KB"] * 1024.0, e["WRITE_SIZE_KB"] * 1024.0
        traffic = int(2 * fetch + write)          # the guide's gfx950 correction (FETCH_SIZE x2)
        hbm_actual = {"fetch_bytes_counter": int(fetch), "write_bytes_counter": int(write),
                      "bytes_uncorrected": int(fetch + write), "bytes_fetch_x2": traffic,
                      "GBps_uncorrected": round((fetch + write) / (kern_ms * 1e-3) / 1e9, 1),
                      "GBps_fetch_x2": round(traffic / (kern_ms * 1e-3) / 1e9, 1),
                      "frac_of_peak_uncorrected": round((fetch + write) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                      "frac_of_peak_fetch_x2": round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                      "known_input_bytes": int(w["n_bases"] + 16 * (n + 1)),
                      "load_width": "input bases are fetched as aligned dwords (4 B per lane, 3 per lane per read), offsets as 8-B loads; "
                                    "the guide calibrates the x2 only for 16 B-per-lane streams, so both figures are given"}
        iv = e["SQ_INSTS_VALU"]
        valu = {"insts_per_pair": round(iv / n, 1), "salu_per_pair": round(e.get("SQ_INSTS_SALU", 0) / n, 1),
                "lds_per_pair": round(e.get("SQ_INSTS_LDS", 0) / n, 1),
                "cycles_per_valu_inst": round(kern_ms * 1e-3 * CLK_GHZ * 1e9 * N_SIMD / iv, 2),
                "frac_of_issue_ceiling": round(iv * 2.0 / (kern_ms * 1e-3 * CLK_GHZ * 1e9 * N_SIMD), 4),
                "ceiling": "1024 SIMD-32 x %.1f GHz / 2 cycles per wave64 instruction" % CLK_GHZ}
    lds_mode = h.probe_mode().startswith("lds-")
    roofline = {"bound": "valu-issue" if lds_mode else "memory-side request rate (random 128-B lines behind L2)",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                "frac_of_binding": valu["frac_of_issue_ceiling"] if (valu and lds_mode) else None,
                "what_frac_is": "the contract's figure: SURVEY 8(d) algorithmic bytes (every k-mer of every read, as the reference visits them) / kernel time / 8 TB/s "
                                "HBM peak -- a model figure, not HBM use, and it may exceed 1: on this index the exact table sits in LDS, and the bound cut, the early "
                                "decision and the sparse first round (DESIGN.md 3) prove most of those probes irrelevant to the read's result and never make them "
                                "(an off-target pair needs 128 of its 268 k-mers, a pair from the gene 64).  `bound` names what does bind the kernel, `frac_of_binding` its use "
                                "(VALU wave-instructions x 2 cycles / (1024 SIMDs x 2.4 GHz x kernel time))",
                "kernel": "classify_uni_kernel" if "table" in h.probe_mode() else "classify_fast_kernel",
                "kernel_ms": round(kern_ms, 4), "launches": int(tm["n_launches"]),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "bytes_per_read": round(alg_bytes / (2 * n), 1),
                "kmers": int(w["n_kmers"]), "hits": int(w["n_hits"]),
                "valu_from_profile": valu, "hbm_actual_from_profile": hbm_actual, "traffic_source": "profiles/pmc_counters.json (a separate rocprofv3 --pmc pass "
                "on the same kernel sources, not this run)" if traffic is not None else None,
                "kernel_src_sha": sha, "counters": prof_note}

    # ---- PCIe-inclusive rate of the host-buffer entry point (never `value`) ---------
    boundary = None
    if not args.no_boundary and world == 1:
        nb = min(n, 4_000_000)
        hbp = synth.to_host_sample(batch, nb, L)
        from shark_amd.capi import SHK_PIPE_DEPTH
        boundary = {"pairs_per_batch": nb, "batches": 12, "in_flight": SHK_PIPE_DEPTH,
                    "what": "shk_classify_submit / shk_classify_wait over host buffers: H2D + kernels + D2H, results on the host; "
                            "the H2D of the next batches overlaps the kernels of the current one"}
        for kind in ("pageable", "pinned"):
            bufs = []
            for rep in range(2):                      # two sets of host buffers, used alternately
                arrs = {}
                for kk in ("seq1", "seq2"):
                    t = torch.from_numpy(hbp[kk]).clone()
                    arrs[kk] = (t.pin_memory() if kind == "pinned" else t).numpy()
                for kk in ("off1", "off2"):
                    t = torch.from_numpy(hbp[kk].view(np.int64)).clone()
                    arrs[kk] = (t.pin_memory() if kind == "pinned" else t).numpy().view(np.uint64)
                bufs.append(arrs)

            def stream(nbatches):
                tickets, assoc = [], 0
                for i in range(nbatches):
                    if len(tickets) == SHK_PIPE_DEPTH:
                        assoc += int(h.wait(tickets.pop(0), copy=False)[0][-1])
                    a = bufs[i % 2]
                    tickets.append(h.submit(a["seq1"], a["off1"], a["seq2"], a["off2"]))
                while tickets:
                    assoc += int(h.wait(tickets.pop(0), copy=False)[0][-1])
                return assoc
            stream(3)
            t0 = time.perf_counter()
            assoc = stream(boundary["batches"])
            tb = (time.perf_counter() - t0) / boundary["batches"]
            boundary[kind] = {"value": round(2 * nb / tb, 1), "unit": "reads/s", "ms_per_batch": round(tb * 1e3, 2),
                              "GBps_h2d": round((hbp["seq1"].nbytes + hbp["seq2"].nbytes) / tb / 1e9, 1), "assoc_per_batch": assoc // boundary["batches"]}

    # ---- CPU baseline: the oracle (port of the reference path) on this host -------
    cpu = None
    if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only (bounded sample)
        from oracle import pyoracle
        threads = os.cpu_count() or 1
        try:
            threads = len(os.sched_getaffinity(0))
        except Exception:
            pass
        model, phys = cpu_info()
        ns = args.cpu_sample_pairs or min(n, threads * 50000)
        # the sample: half from the head of the batch, half from its middle (the generator works in chunks of 2^20 pairs)
        n_head = ns - ns // 2 if n >= 2 * ns else ns
        n_mid, first_mid = ns - n_head, n // 2
        hb = synth.to_host_sample(batch, n_head, L)
        if n_mid:
            hm = synth.to_host_sample(batch, n_mid, L, first=first_mid)
            for kk in ("seq1", "seq2"):
                hb[kk] = np.concatenate([hb[kk], hm[kk]])
            hb["off1"] = np.arange(0, (ns + 1) * L, L, dtype=np.uint64)
            hb["off2"] = hb["off1"].copy()
        o = pyoracle.Shark(k=k, c=c, bf_bits=bf_bits)
        o.build([g.tobytes() for g in genes])
        t0 = time.perf_counter()
        ogoff, ogids = o.classify(hb["seq1"], hb["off1"], hb["seq2"], hb["off2"], nthreads=threads)
        tc = time.perf_counter() - t0
        # the sample doubles as an end-of-run parity check against the GPU result (`res`: the last launch on this batch)
        parity = True
        for first, cnt, o_first in ((0, n_head, 0), (first_mid, n_mid, n_head)):
            if not cnt:
                continue
            goff = np.empty(cnt + 1, dtype=np.uint32)
            hip_memcpy_dtoh(goff, res.gene_off + 4 * first, (cnt + 1) * 4)
            n_ids = int(goff[cnt]) - int(goff[0])
            gids = np.empty(max(n_ids, 1), dtype=np.uint16)
            hip_memcpy_dtoh(gids, res.gene_ids + 2 * int(goff[0]), n_ids * 2)
            want_off = ogoff[o_first:o_first + cnt + 1].astype(np.int64) - int(ogoff[o_first])
            parity = parity and bool(np.array_equal(goff.astype(np.int64) - int(goff[0]), want_off)
                                     and np.array_equal(gids[:n_ids], ogids[int(ogoff[o_first]):int(ogoff[o_first + cnt])]))
        # one thread on one reference chunk (SURVEY 8d asks for -t 1 next to all cores)
        n1 = min(n_head, 50000)
        t0 = time.perf_counter()
        o.classify(hb["seq1"][:int(hb["off1"][n1])], hb["off1"][:n1 + 1], hb["seq2"][:int(hb["off2"][n1])], hb["off2"][:n1 + 1], nthreads=1)
        t1 = time.perf_counter() - t0
        cpu = {"value": round(2 * ns / tc, 1), "unit": "reads/s", "cores": threads, "kind": "port",
               "physical_cores": phys or None, "cpu_model": model,
               "sample": ("%d pairs of the same batch (%s), %d threads x 50 000-read chunks (main.cpp:215), %.1f s"
                          % (ns, ("the whole launch: head, middle and tail" if ns == n else
                                  "the first %d and %d from its middle" % (n_head, n_mid) if n_mid else "its first %d" % n_head), threads, tc)),
               "parity_with_gpu": parity,
               "one_thread": {"value": round(2 * n1 / t1, 1), "unit": "reads/s", "sample": "first %d pairs, %.1f s" % (n1, t1)}}
        o.close()

    # ---- the drop-in command itself, end to end (never `value`): FASTQ files in, ssv + FASTQ files out ----------------
    cli = None   # (last: it classifies other reads with the same context, which invalidates `res`)
    if not args.no_cli and world == 1:
        cli = cli_end_to_end(args, genes, dev, h, L)

    out = {
        "metric": "reads/s (paired 2x150 bp, k=%d)" % k,
        "value": round(value, 1),
        "unit": "reads/s",
        "n_gpus": world,
        "ranks_seen": ranks_seen,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "configs[1]: %d gene(s) x %d bp, 2x150 bp pairs in launches of %d, k=%d c=%.1f bf=2^%d bits; "
                               "%d pairs per step over %d GPU(s) (%s scaling)"
                               % (args.genes, args.gene_len, chunk_pairs, k, c, args.bf_log2, pairs_per_step_all, world, args.scaling),
                   "pairs_per_launch": chunk_pairs, "launches_per_step_per_gpu": len(my_chunks), "pairs_per_step": pairs_per_step_all,
                   "reads_per_step": reads_per_step, "on_target": args.on_target,
                   "seed": synth.SEED, "index_build_s": round(t_build, 3), "n_set_bits": int(info["n_set_bits"]),
                   "assoc_per_step": int(n_assoc.item()) // args.steps,
                   "gene_count_checksum": int(np.asarray(counts, dtype=np.uint64).sum()) // args.steps,
                   "gene_counts_allreduce": ("single GPU: no collective" if world == 1 else
                                             "shk_dist_gene_counts_allreduce (RCCL inside libsharkhip; ranks_seen = ncclCommCount)" if sdist.backend_name() == "nccl"
                                             else "gloo dry run (SHARK_DIST_BACKEND): torch.distributed all-reduce of the library's local counters"),
                   "long_reads": int(tm["last_n_long"]), "tie_reads": int(tm["last_n_tie"]), "probe_mode": h.probe_mode()},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "configs": [cfg2, cfg4] if cfg2 else [],
        "batch_boundary": boundary,
        "cli_end_to_end": cli,
    }
    print(json.dumps(out), flush=True)
    sdist.finalize()


if __name__ == "__main__":
    main()
