#!/usr/bin/env python3
"""(test infrastructure)  Randomised differential test of the HIP path against the CPU oracle: random k, filter size
(power of two or not, sparse to almost full), gene sets with shared halves, read lengths from empty to beyond the LDS
specialisations, N / lower case / quality masks, -s, confidence, single-end or paired, every probe structure; every case
also probes the reference's own k-mers one by one (a key that the structure the index is probed through has lost shows
there; among whole reads it hides behind its neighbours' coverage).

tests/test_gpu_fuzz.py runs a fixed-seed schedule of these cases in the `-m gpu` suite and checks that every value of
shk_probe_mode() was drawn.  By hand on a GPU box: `python tests/fuzz_parity.py [iterations] [seed] [bias]` prints one line
per case, stops at the first mismatch and prints the seed to replay."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime for both)

from oracle import pyoracle
from shark_amd import SharkHip
from shark_amd.capi import hip_memcpy_dtoh
from tests import synth

ALL_MODES = ["bitvector-mod", "bitvector", "summary+bitvector", "table", "summary+table", "lds-summary+table", "table-mod",
             "lds-summary+table-mod", "lds-table", "minimiser-table"]
ENV_KEYS = ("SHK_PROBE", "SHK_FORCE_GENERIC", "SHK_NO_LDS_TABLE", "SHK_KTAB", "SHK_NO_LDS_SUMMARY", "SHK_NO_SUMMARY", "SHK_TILE_FIRST")


def run_case(seed, bias=""):
    """one random configuration -> (ok, probe mode, one-line description).
    bias: "" = the broad distribution; "uni" = what classify_uni_kernel takes (sparse filters, one length per mate);
    "mid" = indices of 10^5 .. 10^6 k-mers in filters of 2^26 .. 2^30 bits (L2-summary + table, big LDS summary, plain table);
    "mod" = filter sizes that are not a power of two with a table; "ktab" = the minimiser-bucketed table (k = 15 ... 17)."""
    rng = np.random.default_rng(seed)
    env = {}
    k = int(rng.choice([1, 2, 5, 11, 16, 17, 18, 21, 25, 31, int(rng.integers(1, 32))]))
    u = rng.random()
    if u < 0.69:
        bf_bits = 1 << int(rng.integers(8, 30))
    elif u < 0.99:
        bf_bits = int(rng.integers(300, 1 << 22))
    else:
        bf_bits = int(rng.choice([3, 5, 6, 7])) << 32      # direct-remainder positions (1.5 - 3.5 GiB filters)
    n_genes = int(rng.choice([1, 2, 7, 40, 300]))
    gl = int(rng.choice([60, 400, 2500]))
    q = int(rng.choice([0, 0, 2, 20, 35]))
    single = bool(rng.random() < 0.3)
    c = float(rng.choice([0.0, 0.3, 0.6, 0.9, 1.0]))
    paired = bool(rng.random() < 0.75)
    read_len = int(rng.choice([20, 50, 76, 100, 150, 151, 250, 300, 700]))
    if rng.random() < 0.5:
        env["SHK_PROBE"] = "bitvector"
    var_len = bool(rng.random() < 0.6)
    if bias == "uni":
        env.pop("SHK_PROBE", None)
        bf_bits = (1 << int(rng.integers(24, 34))) if rng.random() < 0.7 else int(rng.choice([3, 5, 6, 7])) << 32
        n_genes = int(rng.choice([1, 2, 7]))
        read_len = int(rng.choice([1, 8, 16, 17, 31, 33, 50, 76, 100, 125, 150, 151, 200, 250, 256, 259, 300]))
        var_len = bool(rng.random() < 0.15)
        env["SHK_TILE_FIRST"] = "1" if rng.random() < 0.5 else "0"     # (one-gene indices, three pairs per pass: the tiles' round in front or not)
    elif bias == "mid":
        env.pop("SHK_PROBE", None)
        k = int(rng.choice([13, 17, 21, 31]))
        bf_bits = 1 << int(rng.integers(26, 31))
        n_genes = int(rng.choice([120, 300, 600]))
        gl = 2500
        read_len = int(rng.choice([76, 100, 150, 250]))
        var_len = bool(rng.random() < 0.3)
    elif bias == "mod":
        env.pop("SHK_PROBE", None)
        k = int(rng.choice([11, 17, 31]))
        bf_bits = int(rng.integers(1 << 20, 1 << 27)) | 1
        n_genes = int(rng.choice([2, 40, 300]))
        gl = int(rng.choice([400, 2500]))
        var_len = bool(rng.random() < 0.3)
    elif bias == "trim":
        # trimmed samples on indices held in LDS: the three-pairs kernel by offsets (one length class or many, mates shorter than k, single-end),
        # the class-by-class path where the longest mates do not fit three pairs per pass
        env.pop("SHK_PROBE", None)
        k = int(rng.choice([5, 12, 17, 17, 21, 31]))
        bf_bits = 1 << int(rng.integers(24, 34))
        n_genes = int(rng.choice([1, 1, 2, 7]))
        q = 0 if rng.random() < 0.8 else q
        read_len = int(rng.choice([76, 100, 125, 140, 150, 151, 160, 200, 250]))
        var_len = bool(rng.random() < 0.9)
        env["SHK_TILE_FIRST"] = "1" if rng.random() < 0.5 else "0"
    elif bias == "pre":
        # references of many genes behind a position table, streams from the genes: anchor_verdict_kernel in front of the table kernels
        # (uniform and trimmed batches), shared halves, every threshold
        env.pop("SHK_PROBE", None)
        k = int(rng.choice([11, 17, 17, 21, 31]))
        bf_bits = (1 << int(rng.integers(24, 33))) if rng.random() < 0.8 else (int(rng.integers(1 << 24, 1 << 28)) | 1)
        n_genes = int(rng.choice([40, 300]))
        gl = int(rng.choice([400, 2500]))
        read_len = int(rng.choice([76, 100, 150, 151, 250, 300]))
        var_len = bool(rng.random() < 0.4)
    elif bias == "ktab":
        # the k-mer keyed, minimiser-bucketed table: k = 15 ... 17, power-of-two filters from sparse to a quarter full (dense ones
        # give the filter's false positives, which the table holds as keys of their own, a share of every read's k-mers)
        env = {"SHK_KTAB": "1", "SHK_NO_LDS_SUMMARY": "1", "SHK_NO_SUMMARY": "1"}
        k = int(rng.choice([15, 16, 17, 17]))
        bf_bits = 1 << int(rng.integers(20, 34))
        n_genes = int(rng.choice([1, 7, 40, 300]))
        read_len = int(rng.choice([17, 33, 76, 100, 150, 151, 250, 300, 700]))
        var_len = bool(rng.random() < 0.3)
    if rng.random() < 0.15 and "SHK_PROBE" not in env and bias != "ktab":
        env["SHK_FORCE_GENERIC"] = "1"          # classify_fast_kernel's table instantiations instead of classify_uni_kernel
    if rng.random() < 0.2:
        env["SHK_NO_LDS_TABLE"] = "1"
    for kk in ENV_KEYS:
        os.environ.pop(kk, None)
    os.environ.update(env)
    try:
        genes = synth.make_genes(rng, n_genes, max(20, gl // 3), gl, share_every=int(rng.choice([0, 2, 5])))
        kw = dict(k=k, c=c, bf_bits=bf_bits, min_quality=q, single=single)
        o = pyoracle.Shark(**kw)
        nidx = o.build([bytes(g) for g in genes])
        h = SharkHip(**kw)
        info = h.build([bytes(g) for g in genes])
        ok = info["nidx"] == nidx and info["n_set_bits"] == o.num_kmer() and np.array_equal(o.bf_words(), h.copy_bf())
        feats = []
        n_reads = int(rng.choice([1, 63, 64, 65, 1000, 4000]))
        b = synth.make_reads(rng, genes, n_reads, read_len=read_len, paired=paired, on_target=float(rng.choice([0.5, 1.0, 1.0]) if bias == "pre" else rng.choice([0.0, 0.5, 1.0])),
                             n_rate=float(rng.choice([0.0, 0.002, 0.05])), lower_rate=float(rng.choice([0.0, 0.1])),
                             var_len=var_len, qual=q > 0)
        og, oi = o.classify(b["seq1"], b["off1"], b["seq2"], b["off2"], b["qual1"], b["qual2"], nthreads=4)
        hg, hi = h.classify(b["seq1"], b["off1"], b["seq2"], b["off2"], b["qual1"], b["qual2"])
        ok = ok and np.array_equal(og, hg) and np.array_equal(oi, hi)
        # the same batch resident in HBM (uniformity is then decided on the device), with a true, an unknown or a wrong length bound
        dev = torch.device("cuda:0")
        t = {kk: (torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) if v is not None else None) for kk, v in b.items()}
        pt = {kk: (v.data_ptr() if v is not None else 0) for kk, v in t.items()}
        bound = int(rng.choice([0, read_len, max(1, read_len // 3)]))
        if n_reads:
            r = h.classify_device(n_reads, pt["seq1"], pt["off1"], pt["seq2"], pt["off2"], pt["qual1"], pt["qual2"], max_read_len=bound)
            dg = np.empty(n_reads + 1, np.uint32)
            hip_memcpy_dtoh(dg, r.gene_off, dg.nbytes)
            di = np.empty(int(r.n_assoc), np.uint16)
            if len(di):
                hip_memcpy_dtoh(di, r.gene_ids, di.nbytes)
            ok = ok and np.array_equal(og, dg) and np.array_equal(oi, di)
            lk = h.last_kernel()
            feats = [f for f, tag in (("offsets", "offsets"), ("pre", "+pre-verdict"), ("classes", "classes"), ("tiles", "+tiles-first"), ("anch", "+anchored-extension"),
                                      ("tri", "+three-pairs")) if tag in lk]
        # key by key: (up to 20 000 of) the reference's k-mers, each as a read of its own
        km = [g[i:i + k] for g in genes for i in range(0, len(g) - k + 1)]
        if km:
            km = km[::max(1, len(km) // 20000)]
            kb = synth.batch_from_lists(km, None, [b"I" * k] * len(km) if q > 0 else None)
            kog, koi = o.classify(kb["seq1"], kb["off1"], kb["seq2"], kb["off2"], kb["qual1"], kb["qual2"], nthreads=4)
            khg, khi = h.classify(kb["seq1"], kb["off1"], kb["seq2"], kb["off2"], kb["qual1"], kb["qual2"])
            ok = ok and np.array_equal(kog, khg) and np.array_equal(koi, khi)
        mode = h.probe_mode()
        desc = "seed=%d bias=%s k=%d bf=%d genes=%d len=%d%s%s q=%d s=%d c=%.1f reads=%d%s mode=%s set=%d assoc=%d feat=%s" % (
            seed, bias or "-", k, bf_bits, n_genes, read_len, "x2" if paired else "", "~" if var_len else "", q, single, c, n_reads,
            "".join(" %s=%s" % kv for kv in sorted(env.items())), mode, info["n_set_bits"], int(og[-1]), "+".join(feats) or "-")
        h.close()
        o.close()
        return ok, mode, desc
    finally:
        for kk in ENV_KEYS:
            os.environ.pop(kk, None)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 20261003
    bias = sys.argv[3] if len(sys.argv) > 3 else ("uni" if os.environ.get("FUZZ_UNI") == "1" else "")
    t_start = time.time()
    modes, feats = {}, {}
    for it in range(iters):
        ok, mode, desc = run_case(seed0 + it, bias)
        modes[mode] = modes.get(mode, 0) + 1
        for f in desc.rsplit("feat=", 1)[1].split("+"):      # (what the device-resident call's kernels were: offsets, pre-verdict, classes ...)
            feats[f] = feats.get(f, 0) + 1
        print("%4d %s %s" % (it, desc, "ok" if ok else "MISMATCH"), flush=True)
        if not ok:
            print("replay: python tests/fuzz_parity.py 1 %d %s" % (seed0 + it, bias))
            sys.exit(1)
    print("FUZZ OK: %d cases in %.0f s, probe modes %s, kernels of the resident call %s" % (iters, time.time() - t_start, modes, feats))


if __name__ == "__main__":
    main()
