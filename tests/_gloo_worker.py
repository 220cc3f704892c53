"""worker for tests/test_dist_gloo.py: one rank of a world_size-N gloo job.

mode "batches": each rank classifies the in-memory batches it owns (batch i -> rank i mod N, shark_amd/dist.py) with the
CPU oracle standing in for the GPU, then the per-gene counts are all-reduced exactly as bench.py does.

mode "files": the sharded run over FASTQ FILES, as `shark --gpus N` feeds its GPUs: the record-aligned byte ranges come
from the CLI's own partition code (shark_amd/bin/shark-fastq-parts = fastq_partition.hpp); rank r parses and classifies
the byte ranges of batches r, r+N, ... and renders their ssv lines; rank 0 gathers the pieces and merges them in batch
order.  The merged ssv must equal the single-process oracle CLI's byte for byte."""
import json
import os
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from shark_amd import dist as sdist  # noqa: E402
from tests import synth  # noqa: E402


def mode_batches(out_path, rank, world):
    rng = np.random.default_rng(42)                 # identical data on every rank
    genes = synth.make_genes(rng, 12, 300, 1200, share_every=4)
    b = synth.make_reads(rng, genes, 2000, read_len=100, paired=True, on_target=0.7)
    o = pyoracle.Shark(k=15, c=0.5, bf_bits=1 << 20)
    o.build([bytes(g) for g in genes])               # index replicated by deterministic rebuild
    counts = torch.zeros(16, dtype=torch.int64)
    lines = 0
    t_local = 0.1 * (rank + 1)
    for first, last in sdist.shard_batches(2000, 300, rank, world):
        o1 = b["off1"][first:last + 1]
        o2 = b["off2"][first:last + 1]
        goff, gids = o.classify(b["seq1"][int(o1[0]):int(o1[-1])], o1 - o1[0], b["seq2"][int(o2[0]):int(o2[-1])], o2 - o2[0])
        counts += torch.from_numpy(np.bincount(gids, minlength=16)[:16].astype(np.int64))
        lines += int(goff[-1])
    tot = torch.tensor([lines], dtype=torch.int64)
    sdist.allreduce_sum_(counts)
    sdist.allreduce_sum_(tot)
    tmax = sdist.max_over_ranks(t_local, torch.device("cpu"))
    if rank == 0:
        json.dump({"counts": counts.tolist(), "lines": int(tot.item()), "tmax": tmax, "world": world}, open(out_path, "w"))


def parse_range(path, b, e):
    with open(path, "rb") as f:
        f.seek(b)
        lines = f.read(e - b).split(b"\n")[:-1]
    return [(lines[i][1:].split()[0], lines[i + 1], lines[i + 3]) for i in range(0, len(lines), 4)]


def mode_files(out_path, rank, world, fasta, fq1, fq2, batch):
    names, seqs = zip(*synth.read_fasta(fasta))
    o = pyoracle.Shark(k=15, c=0.5, bf_bits=1 << 33)      # the CLI's -b 1
    o.build(list(seqs))
    tool = os.path.join(ROOT, "shark_amd", "bin", "shark-fastq-parts")
    table = json.loads(subprocess.run([tool, str(batch), "2", fq1, fq2], capture_output=True, text=True, check=True).stdout)
    assert table["ok"]
    mine = {}
    counts = torch.zeros(64, dtype=torch.int64)
    for i, (b1, e1, b2, e2, n, regular) in enumerate(table["batches"]):
        if sdist.batch_owner(i, world) != rank:
            continue
        assert regular
        r1, r2 = parse_range(fq1, b1, e1), parse_range(fq2, b2, e2)
        assert len(r1) == len(r2) == n
        bt = synth.batch_from_lists([s for _, s, _ in r1], [s for _, s, _ in r2])
        goff, gids = o.classify(bt["seq1"], bt["off1"], bt["seq2"], bt["off2"])
        ssv = b"".join(r1[k][0] + b" " + names[gids[j]] + b"\n" for k in range(n) for j in range(goff[k], goff[k + 1]))
        mine[i] = ssv
        counts += torch.from_numpy(np.bincount(gids, minlength=64)[:64].astype(np.int64))
    sdist.allreduce_sum_(counts)
    pieces = [None] * world
    if world > 1:
        td.gather_object(mine, pieces if rank == 0 else None, dst=0)
    else:
        pieces = [mine]
    if rank == 0:
        merged = {}
        for p in pieces:
            merged.update(p)
        assert sorted(merged) == list(range(len(table["batches"])))
        with open(out_path, "wb") as f:
            for i in sorted(merged):
                f.write(merged[i])
        json.dump({"counts": counts.tolist(), "batches": len(table["batches"])}, open(out_path + ".json", "w"))


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    rank, world = sdist.init("gloo")
    if mode == "batches":
        mode_batches(out_path, rank, world)
    else:
        mode_files(out_path, rank, world, sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]))
    sdist.finalize()


if __name__ == "__main__":
    main()
