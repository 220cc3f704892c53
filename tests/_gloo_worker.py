"""worker for tests/test_dist_gloo.py: one rank of a world_size-N gloo job.
Each rank classifies the batches it owns (batch i -> rank i mod N, the rule of
`shark --gpus N` and shark_amd/dist.py) with the CPU oracle standing in for
the GPU, then the per-gene counts are all-reduced exactly as bench.py does."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from shark_amd import dist as sdist  # noqa: E402
from tests import synth  # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, world = sdist.init("gloo")
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
    sdist.finalize()


if __name__ == "__main__":
    main()
