#!/usr/bin/env python3
"""bench.py -- throughput of the k-mer classification hot path on MI355X.

A "step" is one pass of the hot path (shk_classify_device: FastqSplitter
join/mask semantics + ReadAnalyzer + BF::get_index)
over one batch of synthetic read pairs that is already resident in HBM.

Workload (BASELINE.json configs[1], the configuration the metric is quoted
on): 1 gene x 20 kb uniform-ACGT reference, 10 M synthetic 2x150 bp pairs
(50 % on-target, 1 % substitutions, 0.2 % N), k=17, c=0.6, 2^33-bit filter.
Weak scaling: every rank classifies its own 10 M-pair batch against its own
replica of the index (rebuilt deterministically per GPU; no data-path
collective); the per-gene assigned-read counts are all-reduced over RCCL once
after the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
`roofline` (algorithmic HBM bytes of the classify kernel / its HIP-event
duration, against the 8 TB/s peak) and `cpu_baseline` (the CPU oracle, a port
of the reference path, timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver (already exported on the pool)
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=10_000_000, help="read pairs per step per GPU")
    ap.add_argument("--k", type=int, default=17)
    ap.add_argument("--bf-log2", type=int, default=33)
    ap.add_argument("--genes", type=int, default=1)
    ap.add_argument("--gene-len", type=int, default=20000)
    ap.add_argument("--on-target", type=float, default=0.5, help="fraction of pairs drawn from a gene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-pairs", type=int, default=0, help="0 = cores x 50 000 (one reference chunk per thread)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from shark_amd import SharkHip
    from shark_amd import dist as sdist
    from shark_amd import synth

    rank, local_rank, world = sdist.env_rank()
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    local_rank %= max(torch.cuda.device_count(), 1)   # (a gloo dry run may put several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sdist.init(sdist.backend_name(), dev)

    k, c, bf_bits = args.k, 0.6, 1 << args.bf_log2
    n = args.pairs
    L = 150

    # ---- index: replicated by deterministic rebuild on every GPU ---------------
    genes = synth.make_reference(args.genes, args.gene_len)
    t0 = time.time()
    h = SharkHip(k=k, c=c, bf_bits=bf_bits, device=local_rank)
    info = h.build([g.tobytes() for g in genes])
    t_build = time.time() - t0

    # ---- one batch per rank, generated in HBM ---------------------------------
    batch = synth.make_pairs_device(n, genes, dev, seed=synth.SEED + 1 + rank, read_len=L, on_target=args.on_target)
    torch.cuda.synchronize()
    ptr = {kk: (v.data_ptr() if v is not None else 0) for kk, v in batch.items()}

    def step():
        return h.classify_device(n, ptr["seq1"], ptr["off1"], ptr["seq2"], ptr["off2"], max_read_len=L)

    def barrier():
        torch.cuda.synchronize()
        sdist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    h.gene_counts_reset()
    h.timing_enable(True)
    barrier()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        res = step()          # returns after the context's stream has drained
    barrier()
    dt = time.perf_counter() - t0
    tm = h.timing()
    h.timing_enable(False)
    dt = sdist.max_over_ranks(dt, dev)

    # ---- per-gene counts: the one exchange step, RCCL all-reduce ----------------
    counts = torch.from_numpy(h.gene_counts(max(info["nidx"], 1)).astype(np.int64)).to(dev)
    n_assoc = torch.tensor([int(res.n_assoc)], dtype=torch.int64, device=dev)
    sdist.allreduce_sum_(counts)
    sdist.allreduce_sum_(n_assoc)

    reads_per_step = 2 * n * world
    value = reads_per_step * args.steps / dt

    if rank != 0:
        sdist.finalize()
        return

    # ---- roofline of the dominant kernel (classify_fast_kernel) ------------------
    # exact algorithmic bytes of one launch:
    #   bases (+quals) + 8 B per probed k-mer (one 64-bit filter word)
    #   + per hit 8 B rank word + 8 B CSR offsets + 2 B per gene id + 8 B result per read
    w = h.count_work(n, ptr["seq1"], ptr["off1"], ptr["seq2"], ptr["off2"])
    alg_bytes = w["n_bases"] + 8 * w["n_kmers"] + 16 * w["n_hits"] + 2 * w["n_list_ids"] + 8 * (2 * n)
    kern_ms = tm["total_ms"] / max(tm["n_launches"], 1)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            if tj.get("pairs") == n and tj.get("k") == k and tj.get("bf_log2") == args.bf_log2:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                "kernel": "classify_fast_kernel", "kernel_ms": round(kern_ms, 4), "launches": int(tm["n_launches"]),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "bytes_per_read": round(alg_bytes / (2 * n), 1),
                "sector_granular_GBps": round((w["n_bases"] + 64 * (w["n_kmers"] + 3 * w["n_hits"])) / (kern_ms * 1e-3) / 1e9, 1) if kern_ms > 0 else None,
                "kmers": int(w["n_kmers"]), "hits": int(w["n_hits"])}

    # ---- CPU baseline: the oracle (port of the reference path) on this host -------
    cpu = None
    if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only (bounded sample)
        from oracle import pyoracle
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        ns = args.cpu_sample_pairs or min(n, cores * 50000)
        hb = synth.to_host_sample(batch, ns, L)
        o = pyoracle.Shark(k=k, c=c, bf_bits=bf_bits)
        o.build([g.tobytes() for g in genes])
        t0 = time.perf_counter()
        ogoff, ogids = o.classify(hb["seq1"], hb["off1"], hb["seq2"], hb["off2"], nthreads=cores)
        tc = time.perf_counter() - t0
        # the sample doubles as an end-of-run parity check against the GPU result
        goff = np.empty(ns + 1, dtype=np.uint32)
        from shark_amd.capi import hip_memcpy_dtoh
        hip_memcpy_dtoh(goff, res.gene_off, (ns + 1) * 4)
        parity = bool(np.array_equal(goff, ogoff))
        # one thread on one reference chunk (SURVEY 8d asks for -t 1 next to all cores)
        n1 = min(ns, 50000)
        t0 = time.perf_counter()
        o.classify(hb["seq1"][:int(hb["off1"][n1])], hb["off1"][:n1 + 1], hb["seq2"][:int(hb["off2"][n1])], hb["off2"][:n1 + 1], nthreads=1)
        t1 = time.perf_counter() - t0
        cpu = {"value": round(2 * ns / tc, 1), "unit": "reads/s", "cores": cores, "kind": "port",
               "sample": "first %d pairs of the same batch, %d threads x 50 000-read chunks (main.cpp:215), %.1f s" % (ns, cores, tc),
               "parity_with_gpu": parity,
               "one_thread": {"value": round(2 * n1 / t1, 1), "unit": "reads/s", "sample": "first %d pairs, %.1f s" % (n1, t1)}}
        o.close()

    out = {
        "metric": "reads/s (paired 2x150 bp, k=%d)" % k,
        "value": round(value, 1),
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "configs[1]: %d gene(s) x %d bp, %d pairs 2x150 bp per GPU per step, k=%d c=%.1f bf=2^%d bits"
                               % (args.genes, args.gene_len, n, k, c, args.bf_log2),
                   "pairs_per_step_per_gpu": n, "reads_per_step": reads_per_step, "on_target": args.on_target,
                   "seed": synth.SEED, "index_build_s": round(t_build, 3), "n_set_bits": int(info["n_set_bits"]),
                   "assoc_per_step": int(n_assoc.item()), "gene_count_checksum": int(counts.sum().item()),
                   "long_reads": int(tm["last_n_long"]), "tie_reads": int(tm["last_n_tie"]), "probe_mode": h.probe_mode()},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    sdist.finalize()


if __name__ == "__main__":
    main()
