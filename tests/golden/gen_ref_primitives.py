#!/usr/bin/env python3
"""Generates tests/golden/ref_primitives.json by running the REAL reference code
(the sdsl-free headers compiled in place into oracle/_ref/libsharkref.so:
kmer_utils.hpp, xxhash.hpp, FastqSplitter.hpp, FastaSplitter.hpp, kseq.h,
small_vector.hpp).  Only runs where /root/reference exists (this container);
the JSON it writes is data (inputs + the reference's outputs) and travels.

Covers what example/*.truth.* does not exercise: N restarts, lower case,
len < k, k in {1,5,17,31}, quality masking, single-end, gz input, CRLF."""
import ctypes as C
import gzip
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402

R = pyoracle.ref()
assert R is not None, "oracle/_ref/libsharkref.so missing: run `make -C oracle ref`"
rng = random.Random(1234)


def rand_seq(n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


out = {"generator": "oracle/_ref (reference headers compiled in place)"}

# to_int over all 7-bit characters
out["to_int"] = [R.ref_to_int(c) for c in range(128)]

# revcompl / lsappend / rsprepend / _get_hash on random k-mers
prim = []
for k in (1, 2, 5, 16, 17, 31):
    for _ in range(25):
        v = rng.getrandbits(2 * k)
        c = rng.getrandbits(2)
        prim.append({"k": k, "kmer": v, "c": c, "revcompl": R.ref_revcompl(v, k), "lsappend": R.ref_lsappend(v, c, k),
                     "rsprepend": R.ref_rsprepend(v, c, k), "hash": R.ref_get_hash(v)})
out["kmer_prims"] = prim

# build_kmer from every start position of strings with N / lower case / other characters
strings = ["", "A", "ACGT", "N" * 40, "acgtACGTnNacgtACGT" * 3, rand_seq(60), rand_seq(80, "ACGTN"), rand_seq(120, "ACGTacgtNn-*"),
           "ACGTNACGTNACGTNACGTNACGTN", rand_seq(33) + "N" + rand_seq(33), "N" + rand_seq(50) + "N"]
bk = []
for s in strings:
    for k in (1, 5, 17, 31):
        res = []
        for p0 in range(0, len(s) + 1):
            p = C.c_int(p0)
            v = R.ref_build_kmer(s.encode(), len(s), C.byref(p), k)
            res.append([v, p.value])
        bk.append({"seq": s, "k": k, "results": res})
out["build_kmer"] = bk

# FastqSplitter: joined / masked strings for small FASTQ files (paired, single, -q)
def write_fastq(path, recs, gz=False, crlf=False):
    nl = "\r\n" if crlf else "\n"
    data = "".join("@%s%s%s+%s%s%s" % (i, nl, s + nl, nl, q, nl) for i, s, q in recs)
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(data.encode())
    else:
        with open(path, "wb") as f:
            f.write(data.encode())


def rand_qual(n):
    return "".join(chr(33 + (rng.randint(2, 41) if rng.random() < 0.8 else rng.randint(0, 10))) for _ in range(n))


recs1, recs2 = [], []
for i in range(40):
    l1, l2 = rng.randint(0 if i % 9 == 0 else 20, 90), rng.randint(1, 90)
    recs1.append(("r%d/1 comment here" % i if i % 5 == 0 else "r%d/1" % i, rand_seq(l1, "ACGTacgtN"), None))
    recs2.append(("r%d/2" % i, rand_seq(l2, "ACGTN"), None))
recs1 = [(i, s, rand_qual(len(s))) for i, s, _ in recs1]
recs2 = [(i, s, rand_qual(len(s))) for i, s, _ in recs2]
fq = []
with tempfile.TemporaryDirectory() as td:
    for variant in ("plain", "gz", "crlf"):
        p1, p2 = os.path.join(td, "a_%s.fq" % variant), os.path.join(td, "b_%s.fq" % variant)
        write_fastq(p1, recs1, gz=variant == "gz", crlf=variant == "crlf")
        write_fastq(p2, recs2, gz=variant == "gz", crlf=variant == "crlf")
        for paired in (True, False):
            for q in (0, 20, 38):
                h = R.ref_fastq_read(p1.encode(), p2.encode() if paired else None, q)
                n = R.ref_fastq_count(h)
                rows = []
                for i in range(n):
                    ln = C.c_size_t()
                    ptr = R.ref_fastq_joined(h, i, C.byref(ln))
                    joined = C.string_at(ptr, ln.value)
                    rows.append({"joined_hex": joined.hex(), "id1": R.ref_fastq_id(h, i, 0).decode(),
                                 "seq1": R.ref_fastq_seq(h, i, 0).decode(), "qual1": R.ref_fastq_qual(h, i, 0).decode(),
                                 "id2": R.ref_fastq_id(h, i, 1).decode() if paired else "",
                                 "seq2": R.ref_fastq_seq(h, i, 1).decode() if paired else "",
                                 "qual2": R.ref_fastq_qual(h, i, 1).decode() if paired else ""})
                R.ref_fastq_free(h)
                fq.append({"variant": variant, "paired": paired, "q": q, "reads": rows})
out["fastq_records"] = {"mate1": recs1, "mate2": recs2}
out["fastq_splitter"] = fq

# FastaSplitter on the bundled example + a hand-made multi-record FASTA
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "x.fa")
    fa_text = ">g1 first gene\nACGTACGTAC\nGTACGT\n\n>g2\n" + rand_seq(70) + "\n" + rand_seq(30) + "\n>empty\n>g4\tdesc\nNNNNACGT\n"
    open(p, "w").write(fa_text)
    h = R.ref_fasta_read(p.encode())
    out["fasta"] = {"text": fa_text, "records": [[R.ref_fasta_name(h, i).decode(), R.ref_fasta_seq(h, i).decode()]
                                                  for i in range(R.ref_fasta_count(h))]}
    R.ref_fasta_free(h)

# small_vector: appended lists read back
sv = []
for n in (0, 1, 3, 4, 9):
    vals = [rng.getrandbits(16) for _ in range(n)]
    arr = (C.c_uint16 * max(n, 1))(*vals)
    o = (C.c_uint16 * max(n, 1))()
    last = C.c_uint16()
    m = R.ref_small_vector(arr, n, o, C.byref(last))
    sv.append({"in": vals, "out": list(o[:m]), "last": last.value if m else None})
out["small_vector"] = sv

json.dump(out, open(os.path.join(HERE, "ref_primitives.json"), "w"))
print("wrote ref_primitives.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})
