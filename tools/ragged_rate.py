#!/usr/bin/env python3
"""Kernel time on TRIMMED reads (mixed lengths): the bench workload with every mate cut to a random length in [lo, 150].
usage: python tools/ragged_rate.py [pairs] [lo] [full]    full = fraction of the mates left untrimmed (default 0: every length uniform in [lo, 150]);
env SHK_FORCE_GENERIC=1 -> classify_fast_kernel for comparison"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shark_amd import SharkHip, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 100
full = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
genes_n = int(os.environ.get("GENES", "1"))
dev = torch.device("cuda:0")
genes = synth.make_reference(1, 20000) if genes_n == 1 else synth.make_gencode_like_reference(genes_n)
h = SharkHip(k=17, c=0.6, bf_bits=1 << (33 if genes_n < 20000 else 36))
h.build([g.tobytes() for g in genes])
b = synth.make_pairs_device(n, genes, dev, seed=synth.SEED + 1)
g = torch.Generator(device=dev); g.manual_seed(7)
out = {}
for key in ("seq1", "seq2"):
    L = torch.randint(lo, 151, (n,), generator=g, device=dev)
    if full > 0:
        L = torch.where(torch.rand(n, generator=g, device=dev) < full, torch.full_like(L, 150), L)
    keep = torch.arange(150, device=dev)[None, :] < L[:, None]
    out[key] = b[key].view(n, 150)[keep].contiguous()
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(L, 0)
    out["off" + key[-1]] = off
torch.cuda.synchronize()
# (one untimed call: buffers, code objects)
h.classify_device(n, out["seq1"].data_ptr(), out["off1"].data_ptr(), out["seq2"].data_ptr(), out["off2"].data_ptr(), max_read_len=150)
h.timing_enable(True)
for _ in range(4):
    r = h.classify_device(n, out["seq1"].data_ptr(), out["off1"].data_ptr(), out["seq2"].data_ptr(), out["off2"].data_ptr(), max_read_len=150)
tm = h.timing()
print(json.dumps({"pairs": n, "lengths": [lo, 150], "untrimmed": full, "genes": genes_n, "mode": h.probe_mode(), "generic": os.environ.get("SHK_FORCE_GENERIC", "0"),
                  "kernel_ms": round((tm["total_ms"] + tm["prepass_ms"]) / tm["n_launches"], 3), "of_which_prepass_ms": round(tm["prepass_ms"] / tm["n_launches"], 3), "n_assoc": int(r.n_assoc), "long": tm["last_n_long"]}))
