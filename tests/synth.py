"""Seeded synthetic references / reads and small FASTA/FASTQ helpers for tests.

Everything is numpy so the same bytes can be fed to the oracle and to the HIP
library through the identical SoA batch layout (include/shark_hip.h)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTacgtNn", b"TGCAtgcaNn"):
    COMP[a] = b


def random_seq(rng, n):
    return ACGT[rng.integers(0, 4, size=n)].copy()


def revcomp(a):
    return COMP[a[::-1]]


def make_genes(rng, n_genes, min_len=200, max_len=3000, share_every=0):
    """list of uint8 arrays; with share_every>0 every such gene copies the first
    half of its predecessor (forces multi-gene lists and ties)."""
    genes = []
    for g in range(n_genes):
        L = int(rng.integers(min_len, max_len + 1))
        s = random_seq(rng, L)
        if share_every and g % share_every == share_every - 1 and genes:
            p = genes[-1]
            h = min(len(p) // 2, L)
            s[:h] = p[:h]
        genes.append(s)
    return genes


def make_reads(rng, genes, n, read_len=150, paired=True, on_target=0.5, sub_rate=0.01, n_rate=0.002,
               lower_rate=0.0, var_len=False, qual=False):
    """returns dict(seq1, off1, seq2, off2, qual1, qual2) of numpy arrays"""
    s1, s2, q1, q2 = [], [], [], []
    for _ in range(n):
        L1 = int(rng.integers(max(1, read_len // 2), read_len + 1)) if var_len else read_len
        L2 = int(rng.integers(max(1, read_len // 2), read_len + 1)) if var_len else read_len
        if genes and rng.random() < on_target:
            g = genes[int(rng.integers(0, len(genes)))]
            frag = int(rng.integers(min(len(g), max(L1, L2)), min(len(g), max(L1, L2) * 3) + 1))
            st = int(rng.integers(0, len(g) - frag + 1))
            f = g[st:st + frag]
            m1 = f[:L1].copy()
            m2 = revcomp(f)[:L2].copy()
            if len(m1) < L1:
                m1 = np.concatenate([m1, random_seq(rng, L1 - len(m1))])
            if len(m2) < L2:
                m2 = np.concatenate([m2, random_seq(rng, L2 - len(m2))])
        else:
            m1, m2 = random_seq(rng, L1), random_seq(rng, L2)
        for m in (m1, m2):
            sub = rng.random(len(m)) < sub_rate
            m[sub] = ACGT[rng.integers(0, 4, size=int(sub.sum()))]
            m[rng.random(len(m)) < n_rate] = ord("N")
            if lower_rate:
                lo = rng.random(len(m)) < lower_rate
                m[lo] = m[lo] | 0x20
        s1.append(m1)
        s2.append(m2)
        if qual:
            for lst, m in ((q1, m1), (q2, m2)):
                q = np.where(rng.random(len(m)) < 0.9, rng.integers(30, 42, size=len(m)), rng.integers(2, 30, size=len(m)))
                lst.append((q + 33).astype(np.uint8))
    out = {"seq1": _cat(s1), "off1": _off(s1)}
    if paired:
        out["seq2"], out["off2"] = _cat(s2), _off(s2)
    else:
        out["seq2"], out["off2"] = None, None
    out["qual1"] = _cat(q1) if qual else None
    out["qual2"] = _cat(q2) if (qual and paired) else None
    return out


def _cat(lst):
    return np.concatenate(lst).astype(np.uint8) if lst else np.zeros(0, np.uint8)


def _off(lst):
    off = np.zeros(len(lst) + 1, dtype=np.uint64)
    if lst:
        off[1:] = np.cumsum([len(x) for x in lst])
    return off


def batch_from_lists(m1, m2=None, q1=None, q2=None):
    m1 = [np.frombuffer(bytes(x), dtype=np.uint8) for x in m1]
    out = {"seq1": _cat(m1), "off1": _off(m1), "seq2": None, "off2": None, "qual1": None, "qual2": None}
    if m2 is not None:
        m2 = [np.frombuffer(bytes(x), dtype=np.uint8) for x in m2]
        out["seq2"], out["off2"] = _cat(m2), _off(m2)
    if q1 is not None:
        out["qual1"] = _cat([np.frombuffer(bytes(x), dtype=np.uint8) for x in q1])
    if q2 is not None:
        out["qual2"] = _cat([np.frombuffer(bytes(x), dtype=np.uint8) for x in q2])
    return out


def read_fasta(path):
    """[(name, seq bytes)] -- plain multi-line FASTA (test fixtures only)"""
    recs, name, parts = [], None, []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    recs.append((name, b"".join(parts)))
                name, parts = line[1:].split()[0] if line[1:].split() else b"", []
            elif name is not None:
                parts.append(line)
    if name is not None:
        recs.append((name, b"".join(parts)))
    return recs


def read_fastq(path):
    """[(id, seq, qual)] -- 4-line FASTQ (test fixtures only)"""
    recs = []
    with open(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().rstrip(b"\r\n")
            f.readline()
            q = f.readline().rstrip(b"\r\n")
            recs.append((h[1:].split()[0], s, q))
    return recs


def assoc_lists(gene_off, gene_ids):
    return [tuple(int(x) for x in gene_ids[gene_off[i]:gene_off[i + 1]]) for i in range(len(gene_off) - 1)]
