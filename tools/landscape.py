#!/usr/bin/env python3
"""Kernel time per 10 M pairs (2 x 150 bp, k = 17) over index sizes and on-target rates, with and without the anchored extension
(SHK_NO_ANCHOR=1 at index build time); the two must return identical associations.  One JSON line per measurement.
usage: python tools/landscape.py [--genes 250,1000,10000,60000] [--ot 0.5] [--pairs 10000000] [--ab [--ab-var SHK_NO_SPARSE]] [--k 17] [--q 0] [--read-len 150]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shark_amd import SharkHip, synth
from shark_amd.capi import hip_memcpy_dtoh

ap = argparse.ArgumentParser()
ap.add_argument("--genes", default="10,60,100,150,250,1000,10000,60000")
ap.add_argument("--ot", default="0.5")
ap.add_argument("--pairs", type=int, default=10_000_000)
ap.add_argument("--k", type=int, default=17)
ap.add_argument("--q", type=int, default=0)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--ab", action="store_true", help="also measure without the anchored extension and compare the results")
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--sub-rate", type=float, default=0.01, help="substitutions per base of the on-target reads")
ap.add_argument("--ab-var", default="SHK_NO_ANCHOR", help="the build-time switch --ab compares (SHK_NO_ANCHOR, SHK_NO_SPARSE)")
a = ap.parse_args()
dev = torch.device("cuda:0")
for ng in [int(x) for x in a.genes.split(",")]:
    genes = synth.make_reference(1, 20000) if ng == 1 else synth.make_gencode_like_reference(ng)
    bf_log2 = 33 if ng < 20000 else (36 if a.k <= 17 else 37)
    for ot in [float(x) for x in a.ot.split(",")]:
        b = synth.make_pairs_device(a.pairs, genes, dev, seed=synth.SEED + 7, read_len=a.read_len, on_target=ot, with_qual=a.q > 0, sub_rate=a.sub_rate)
        torch.cuda.synchronize()
        ptr = {k: (v.data_ptr() if v is not None else 0) for k, v in b.items()}
        res = {}
        for anchor in ([True, False] if a.ab else [True]):
            if anchor:
                os.environ.pop(a.ab_var, None)
            else:
                os.environ[a.ab_var] = "1"
            h = SharkHip(k=a.k, c=0.6, bf_bits=1 << bf_log2, min_quality=a.q, single=a.q > 0)
            info = h.build([g.tobytes() for g in genes])
            h.classify_device(a.pairs, ptr["seq1"], ptr["off1"], ptr["seq2"], ptr["off2"], ptr["qual1"], ptr["qual2"], max_read_len=a.read_len)
            h.timing_enable(True)
            for _ in range(a.reps):
                r = h.classify_device(a.pairs, ptr["seq1"], ptr["off1"], ptr["seq2"], ptr["off2"], ptr["qual1"], ptr["qual2"], max_read_len=a.read_len)
            tm = h.timing()
            goff = np.empty(a.pairs + 1, np.uint32); hip_memcpy_dtoh(goff, r.gene_off, goff.nbytes)
            gids = np.empty(max(int(r.n_assoc), 1), np.uint16); hip_memcpy_dtoh(gids, r.gene_ids, int(r.n_assoc) * 2)
            res[anchor] = (goff, gids[:int(r.n_assoc)])
            print(json.dumps({"genes": ng, "bf_log2": bf_log2, "k": a.k, "q": a.q, "on_target": ot, "pairs": a.pairs, "anchored" if a.ab_var == "SHK_NO_ANCHOR" else "with": anchor, "read_len": a.read_len, "sub_rate": a.sub_rate, "mode": h.probe_mode(),
                              "n_set_bits": info["n_set_bits"], "kernel_ms": round(tm["total_ms"] / tm["n_launches"], 3), "n_assoc": int(r.n_assoc),
                              "last_n_long": tm["last_n_long"], "last_n_tie": tm["last_n_tie"]}), flush=True)
            h.close()
        if a.ab:
            same = np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1])
            print(json.dumps({"genes": ng, "on_target": ot, "anchored_equals_plain": bool(same)}), flush=True)
            assert same
        del b
