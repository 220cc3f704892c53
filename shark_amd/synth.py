"""Seeded synthetic workloads generated directly in HBM (torch is plumbing:
device memory + RNG).  Shapes follow SURVEY.md 8(d): uniform ACGT references,
2x150 bp pairs, 50 % drawn from a gene (fragment 300-500 bp, mate 2 reverse
complemented), 50 % uniform random, 1 % substitutions, 0.2 % N."""
import numpy as np
import torch

SEED = 0x5A4B2020


def make_reference(n_genes, gene_len, seed=SEED):
    """list of numpy uint8 arrays (host; they go through shk_ref_add)"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    if np.isscalar(gene_len):
        lens = [int(gene_len)] * n_genes
    else:
        lens = [int(x) for x in gene_len]
    return [acgt[rng.integers(0, 4, size=L)] for L in lens]


def make_gencode_like_reference(n_genes=60000, seed=SEED):
    """BASELINE configs[2..4] reference: gene lengths lognormal(median 2 kb) clipped to
    [200, 20 000] (60 000 genes = 1.78e8 bases), uniform ACGT, every 10th gene shares its first half
    with its predecessor (forces multi-gene lists and ties).  Returns a list of numpy uint8 arrays."""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.exp(rng.normal(np.log(2000), 0.9, size=n_genes)), 200, 20000).astype(np.int64)
    genes = make_reference(n_genes, lens, seed=seed)
    for g in range(9, n_genes, 10):
        h = min(len(genes[g - 1]) // 2, len(genes[g]))
        genes[g][:h] = genes[g - 1][:h]
    return genes


def _qualities(m, L, g, device, model):
    """Phred+33 qualities of m reads of length L.
    "ends" (skewed high, so that -q 20 leaves valid k-mers): Q30-41 everywhere except a low-quality tail at the
      3' end -- its length floor(Exp(mean 2)), at most 30 bases -- and 0.2 % isolated low bases, both Q2-19: about 1.2 % of the
      bases fall below Q20, and a 31-mer away from the tail survives -q 20 with 0.94;
    "uniform" (rounds 1-2): 10 % of the bases Q2-29 anywhere -- 6.4 % below Q20, a 31-mer survives with 0.13, which left the
      k = 31 quality-mask HIT path almost unexercised (1.3 % of the pairs assigned)."""
    hi = torch.randint(30, 42, (m, L), generator=g, device=device)
    if model == "uniform":
        lo = torch.randint(2, 30, (m, L), generator=g, device=device)
        q = torch.where(torch.rand(m, L, generator=g, device=device) < 0.9, hi, lo)
    else:
        lo = torch.randint(2, 20, (m, L), generator=g, device=device)
        tail = torch.clamp((-2.0 * torch.log(torch.rand(m, generator=g, device=device).clamp_min(1e-12))).floor(), max=30).to(torch.int64)
        low = torch.arange(L, device=device)[None, :] >= (L - tail)[:, None]
        low |= torch.rand(m, L, generator=g, device=device) < 0.002
        q = torch.where(low, lo, hi)
    return (q + 33).to(torch.uint8)


def make_pairs_device(n, genes, device, seed=SEED, read_len=150, on_target=0.5, sub_rate=0.01, n_rate=0.002,
                      with_qual=False, chunk=1 << 20, qual_model="ends"):
    """n pairs of fixed-length mates resident on `device`.
    returns dict(seq1, off1, seq2, off2, qual1, qual2) of torch tensors (uint8 / int64)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    # all genes concatenated; a read is drawn from one gene
    cat = torch.from_numpy(np.concatenate(genes)).to(device)
    glen = torch.tensor([len(x) for x in genes], dtype=torch.int64, device=device)
    gstart = torch.cumsum(glen, 0) - glen
    L = read_len
    seq1 = torch.empty(n * L, dtype=torch.uint8, device=device)
    seq2 = torch.empty(n * L, dtype=torch.uint8, device=device)
    qual1 = torch.empty(n * L, dtype=torch.uint8, device=device) if with_qual else None
    qual2 = torch.empty(n * L, dtype=torch.uint8, device=device) if with_qual else None
    ar = torch.arange(L, device=device)
    for b0 in range(0, n, chunk):
        m = min(chunk, n - b0)
        gi = torch.randint(0, len(genes), (m,), generator=g, device=device)
        gl = glen[gi]
        frag = torch.minimum(torch.randint(300, 501, (m,), generator=g, device=device), gl)
        frag = torch.maximum(frag, torch.full_like(frag, 1))
        st = (torch.rand(m, generator=g, device=device) * (gl - frag + 1).to(torch.float32)).to(torch.int64)
        st = torch.minimum(st, gl - frag) + gstart[gi]
        idx1 = torch.minimum(st[:, None] + ar[None, :], (gstart[gi] + gl - 1)[:, None])
        idx2 = torch.maximum((st + frag - 1)[:, None] - ar[None, :], gstart[gi][:, None])
        t1 = cat[idx1]
        t2 = comp[cat[idx2].to(torch.int64)]
        r1 = acgt[torch.randint(0, 4, (m, L), generator=g, device=device)]
        r2 = acgt[torch.randint(0, 4, (m, L), generator=g, device=device)]
        on = (torch.rand(m, generator=g, device=device) < on_target)[:, None]
        m1 = torch.where(on, t1, r1)
        m2 = torch.where(on, t2, r2)
        for mm in (m1, m2):
            sub = torch.rand(m, L, generator=g, device=device) < sub_rate
            mm[sub] = acgt[torch.randint(0, 4, (int(sub.sum().item()),), generator=g, device=device)]
            mm[torch.rand(m, L, generator=g, device=device) < n_rate] = ord("N")
        seq1[b0 * L:(b0 + m) * L] = m1.reshape(-1)
        seq2[b0 * L:(b0 + m) * L] = m2.reshape(-1)
        if with_qual:
            for qq in (qual1, qual2):
                qq[b0 * L:(b0 + m) * L] = _qualities(m, L, g, device, qual_model).reshape(-1)
    off = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=device)
    out = {"seq1": seq1, "off1": off, "seq2": seq2, "off2": off.clone(), "qual1": qual1, "qual2": qual2}
    # libsharkhip works on its own NON-BLOCKING streams: they do not wait for torch's stream, so the batch has to be complete before
    # its pointers are handed over (a caller that classified right away read offsets that were still being written -- found as a
    # memory fault in bench.py's CLI leg, round 4)
    torch.cuda.synchronize(device)
    return out


def to_host_sample(batch, n_sample, read_len=150, first=0):
    """pairs [first, first + n_sample) as numpy arrays (same bytes the GPU classified)"""
    L = read_len
    out = {}
    for key in ("seq1", "seq2", "qual1", "qual2"):
        t = batch.get(key)
        out[key] = t[first * L:(first + n_sample) * L].cpu().numpy() if t is not None else None
    off = np.arange(0, (n_sample + 1) * L, L, dtype=np.uint64)
    out["off1"] = off
    out["off2"] = off.copy() if batch.get("seq2") is not None else None
    return out
