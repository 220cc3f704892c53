#!/usr/bin/env python3
"""(test infrastructure; run by hand on a GPU box: `python tests/scale_check.py`)
GENCODE-scale run (BASELINE configs[2] shape): ~60 k genes / ~1.8e8 reference k-mers, 2^36-bit filter,
2x150 bp pairs.  Checks (a) the index build at scale, (b) that both probe structures give identical
results, (c) bit-exact parity with the CPU oracle on a sample, and reports kernel time."""
import argparse, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import numpy as np
import torch
from shark_amd import SharkHip, synth
from shark_amd.capi import hip_memcpy_dtoh

ap = argparse.ArgumentParser()
ap.add_argument("--genes", type=int, default=60000)
ap.add_argument("--bf-log2", type=int, default=36)
ap.add_argument("--bf-bits", type=int, default=0, help="exact filter size in bits (overrides --bf-log2; e.g. 3<<33 for `-b 3`)")
ap.add_argument("--pairs", type=int, default=10_000_000)
ap.add_argument("--k", type=int, default=17)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--q", type=int, default=0)
ap.add_argument("--single", action="store_true")
ap.add_argument("--oracle-pairs", type=int, default=200000)
ap.add_argument("--skip-bitvector", action="store_true")
ap.add_argument("--on-target", type=float, default=0.5)
a = ap.parse_args()

BF_BITS = a.bf_bits if a.bf_bits else 1 << a.bf_log2
rng = np.random.default_rng(synth.SEED)
lens = np.clip(np.exp(rng.normal(np.log(2000), 0.9, size=a.genes)), 200, 20000).astype(np.int64)
genes = synth.make_reference(a.genes, lens)
for g in range(9, a.genes, 10):          # every 10th gene shares its first half with its predecessor
    h = min(len(genes[g - 1]) // 2, len(genes[g]))
    genes[g][:h] = genes[g - 1][:h]
print("reference: %d genes, %.3e bases" % (a.genes, float(lens.sum())), flush=True)
dev = torch.device("cuda:0")
batch = synth.make_pairs_device(a.pairs, genes, dev, seed=synth.SEED + 7, read_len=a.read_len, with_qual=a.q > 0, on_target=a.on_target)
torch.cuda.synchronize()
ptr = {k: (v.data_ptr() if v is not None else 0) for k, v in batch.items()}
res = {}
for mode in (["auto"] if a.skip_bitvector else ["auto", "bitvector"]):
    if mode == "bitvector":
        os.environ["SHK_PROBE"] = "bitvector"
    else:
        os.environ.pop("SHK_PROBE", None)
    t0 = time.time()
    h = SharkHip(k=a.k, c=0.6, bf_bits=BF_BITS, min_quality=a.q, single=a.single)
    info = h.build([g.tobytes() for g in genes])
    tb = time.time() - t0
    h.timing_enable(True)
    for _ in range(3):
        r = h.classify_device(a.pairs, ptr["seq1"], ptr["off1"], ptr["seq2"], ptr["off2"], ptr["qual1"], ptr["qual2"], max_read_len=a.read_len)
    tm = h.timing()
    goff = np.empty(a.pairs + 1, np.uint32); hip_memcpy_dtoh(goff, r.gene_off, goff.nbytes)
    gids = np.empty(int(r.n_assoc), np.uint16); hip_memcpy_dtoh(gids, r.gene_ids, gids.nbytes)
    res[mode] = (goff, gids)
    print(json.dumps({"mode": h.probe_mode(), "build_s": round(tb, 2), "info": info, "kernel_ms": round(tm["total_ms"] / tm["n_launches"], 3),
                      "reads_per_s": round(2 * a.pairs / (tm["total_ms"] / tm["n_launches"] * 1e-3) / 1e6, 1), "n_assoc": int(r.n_assoc),
                      "tie_reads": tm["last_n_tie"], "multi_assoc_reads": int((np.diff(goff.astype(np.int64)) > 1).sum())}), flush=True)
    h.close()
if len(res) == 2:
    same = np.array_equal(res["auto"][0], res["bitvector"][0]) and np.array_equal(res["auto"][1], res["bitvector"][1])
    print("probe structures agree:", same, flush=True)
    assert same
if a.oracle_pairs:
    from oracle import pyoracle
    t0 = time.time()
    o = pyoracle.Shark(k=a.k, c=0.6, bf_bits=BF_BITS, min_quality=a.q, single=a.single)
    o.build([g.tobytes() for g in genes])
    print("oracle index built in %.1f s (%d set bits)" % (time.time() - t0, o.num_kmer()), flush=True)
    hb = synth.to_host_sample(batch, a.oracle_pairs, a.read_len)
    t0 = time.time()
    og, oi = o.classify(hb["seq1"], hb["off1"], hb["seq2"], hb["off2"], hb["qual1"], hb["qual2"], nthreads=os.cpu_count())
    goff, gids = res["auto"]
    ok = np.array_equal(og, goff[:a.oracle_pairs + 1]) and np.array_equal(oi, gids[:int(goff[a.oracle_pairs])])
    print("oracle sample (%d pairs, %.1f s): parity %s" % (a.oracle_pairs, time.time() - t0, ok), flush=True)
    assert ok
print("SCALE TEST OK")
