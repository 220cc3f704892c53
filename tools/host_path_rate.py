#!/usr/bin/env python3
"""PCIe-inclusive rate of shk_classify (host buffers in, host results out): H2D + kernels + D2H, one batch at a time.
Reported in DESIGN.md next to the HBM-resident `value` of bench.py; never used as `value`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shark_amd import SharkHip, synth

n = 4_000_000
genes = synth.make_reference(1, 20000)
h = SharkHip(k=17, c=0.6, bf_bits=1 << 33)
h.build([g.tobytes() for g in genes])
dev = torch.device("cuda:0")
b = synth.make_pairs_device(n, genes, dev, seed=synth.SEED + 1)
hb = synth.to_host_sample(b, n)
for pinned in (False, True):
    arrs = {}
    for k in ("seq1", "seq2"):
        t = torch.from_numpy(hb[k])
        arrs[k] = (t.pin_memory() if pinned else t).numpy()
    for k in ("off1", "off2"):
        t = torch.from_numpy(hb[k].view(np.int64))
        arrs[k] = (t.pin_memory() if pinned else t).numpy().view(np.uint64)
    h.classify(arrs["seq1"], arrs["off1"], arrs["seq2"], arrs["off2"])
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        goff, gids = h.classify(arrs["seq1"], arrs["off1"], arrs["seq2"], arrs["off2"])
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"host_path": "pinned" if pinned else "pageable", "pairs": n, "ms_per_batch": round(dt * 1e3, 2),
                      "reads_per_s": round(2 * n / dt / 1e6, 1), "bytes_h2d": int(2 * n * 150 + 2 * (n + 1) * 8), "assoc": int(goff[-1])}))
