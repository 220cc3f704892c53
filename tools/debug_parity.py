import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import synth
from oracle import pyoracle
from shark_amd import SharkHip
k, bf_bits, paired, read_len = 5, 1 << 12, True, 60
rng = np.random.default_rng(1234 + k)
genes = synth.make_genes(rng, 40, 100, 1500, share_every=4)
o = pyoracle.Shark(k=k, c=0.6, bf_bits=bf_bits); o.build([bytes(g) for g in genes])
h = SharkHip(k=k, c=0.6, bf_bits=bf_bits); print(h.build([bytes(g) for g in genes]))
b = synth.make_reads(rng, genes, 3000, read_len=read_len, paired=paired, on_target=0.6, n_rate=0.01, lower_rate=0.05, var_len=True)
og, oi = o.classify(b["seq1"], b["off1"], b["seq2"], b["off2"])
hg, hi = h.classify(b["seq1"], b["off1"], b["seq2"], b["off2"])
print("goff equal", np.array_equal(og, hg), "ids equal", np.array_equal(oi, hi), "tie", h.timing()["last_n_tie"])
bad = 0
for i in range(3000):
    a, c = tuple(oi[og[i]:og[i+1]]), tuple(hi[hg[i]:hg[i+1]])
    if a != c:
        s1 = bytes(b["seq1"][int(b["off1"][i]):int(b["off1"][i+1])]); s2 = bytes(b["seq2"][int(b["off2"][i]):int(b["off2"][i+1])])
        print(i, "oracle", a, "hip", c, "analyze", o.analyze(s1 + b"N" + s2)[1:], len(s1), len(s2))
        bad += 1
        if bad > 12: break
print("bad", bad)
