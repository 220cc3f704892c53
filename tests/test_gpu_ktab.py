"""The k-mer keyed, minimiser-bucketed table (`shk_probe_mode()` = "minimiser-table"): a third statement of the
index for k = 15 ... 17, keyed by the canonical k-mer itself -- every canonical k-mer whose filter bit is set, the reference's own
and the filter's false positives alike, enumerated over all 4^k / 2 of them when the index is built -- and laid out so that the
consecutive k-mers of a read, which share their minimiser, are looked up in one 128-byte line.  It must answer exactly what
`BF::get_index` answers (bloomfilter.h:78-102): parity with the oracle on whole reads, on every reference k-mer as a read of its own,
and -- the false-positive enumeration has no other witness -- on uniform-random k-mers against the plain filter words.

Run on the GPU box with `pytest -m gpu`."""
import numpy as np
import pytest

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

from tests import synth

pytestmark = pytest.mark.gpu


def _ktab_env(monkeypatch, on=True):
    for v in ("SHK_PROBE", "SHK_KTAB", "SHK_NO_KTAB", "SHK_NO_LDS_SUMMARY", "SHK_NO_SUMMARY", "SHK_NO_LDS_TABLE", "SHK_FORCE_GENERIC", "SHK_KTAB_NT"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("SHK_NO_LDS_SUMMARY", "1")
    monkeypatch.setenv("SHK_NO_SUMMARY", "1")
    if on:
        monkeypatch.setenv("SHK_KTAB", "1")


def _hip(**kw):
    from shark_amd import SharkHip
    return SharkHip(**kw)


def _same(o, h, b, nthreads=4):
    og, oi = o.classify(b["seq1"], b["off1"], b["seq2"], b["off2"], b.get("qual1"), b.get("qual2"), nthreads=nthreads)
    hg, hi = h.classify(b["seq1"], b["off1"], b["seq2"], b["off2"], b.get("qual1"), b.get("qual2"))
    assert np.array_equal(og, hg) and np.array_equal(oi, hi)
    return int(og[-1])


@pytest.mark.parametrize("k,bf_bits,n_genes,read_len,paired,q", [
    (17, 1 << 33, 40, 150, True, 0), (17, 1 << 28, 300, 150, True, 0), (17, 1 << 26, 300, 100, True, 20), (16, 1 << 30, 40, 125, True, 0),
    (15, 1 << 27, 40, 76, False, 0), (17, 1 << 24, 40, 250, True, 0), (17, 1 << 30, 7, 300, True, 30), (16, 1 << 22, 7, 150, True, 0),
])
def test_whole_reads_and_every_reference_kmer(oracle, monkeypatch, k, bf_bits, n_genes, read_len, paired, q):
    """whole reads (on- and off-target, N, lower case, substitutions, uniform and trimmed batches, host and device-resident) and every
    reference k-mer as a read of its own: the oracle's result through the minimiser-bucketed table -- asserted to be the chain that
    ran -- on filters from sparse (2^33 bits) to dense (2^22 bits for 10^4 k-mers: the filter's false positives are then a quarter
    per cent of ALL k-mers, each a key of the table with the list of the reference k-mer it collides with)"""
    _ktab_env(monkeypatch)
    rng = np.random.default_rng(k * 1000 + n_genes)
    genes = synth.make_genes(rng, n_genes, 300, 2500, share_every=3)
    kw = dict(k=k, c=0.5, bf_bits=bf_bits, min_quality=q)
    o = oracle.Shark(**kw)
    o.build([bytes(g) for g in genes])
    h = _hip(**kw)
    info = h.build([bytes(g) for g in genes])
    assert info["n_set_bits"] == o.num_kmer()
    assert h.probe_mode() == "minimiser-table"
    total = 0
    for var_len in (False, True):
        for on_target in (0.0, 0.6, 1.0):
            b = synth.make_reads(rng, genes, 3000, read_len=read_len, paired=paired, on_target=on_target, n_rate=0.004, lower_rate=0.02,
                                 var_len=var_len, qual=q > 0)
            total += _same(o, h, b)
            if read_len <= 300:
                assert ", 8, " in h.last_kernel(), h.last_kernel()          # PM_KTAB instantiation of classify_uni_kernel
            # the same batch resident in HBM
            dev = torch.device("cuda:0")
            t = {kk: (torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) if v is not None else None) for kk, v in b.items()}
            pt = {kk: (v.data_ptr() if v is not None else 0) for kk, v in t.items()}
            n = len(b["off1"]) - 1
            from shark_amd.capi import hip_memcpy_dtoh
            r = h.classify_device(n, pt["seq1"], pt["off1"], pt["seq2"], pt["off2"], pt["qual1"], pt["qual2"], max_read_len=read_len)
            og, oi = o.classify(b["seq1"], b["off1"], b["seq2"], b["off2"], b.get("qual1"), b.get("qual2"), nthreads=4)
            dg = np.empty(n + 1, np.uint32)
            hip_memcpy_dtoh(dg, r.gene_off, dg.nbytes)
            di = np.empty(int(r.n_assoc), np.uint16)
            if len(di):
                hip_memcpy_dtoh(di, r.gene_ids, di.nbytes)
            assert np.array_equal(og, dg) and np.array_equal(oi, di)
    assert total > 1000
    km = [g[i:i + k] for g in genes for i in range(0, len(g) - k + 1)]
    km = km[::max(1, len(km) // 60000)]
    kb = synth.batch_from_lists(km, None, [b"I" * k] * len(km) if q > 0 else None)
    assert _same(o, h, kb) >= len(km)           # every reference k-mer is found (c = 0.5: one k-mer covers itself)
    assert ", 8, " in h.last_kernel()


def _random_kmers(gen, n, k, dev):
    """n uniform-random k-mers as single-end reads of k bases, resident in HBM: (seq, off)"""
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    seq = acgt[torch.randint(0, 4, (n * k,), generator=gen, device=dev)]
    off = torch.arange(0, n + 1, device=dev, dtype=torch.int64) * k
    return seq, off


@pytest.mark.parametrize("k,bf_bits,n_genes", [(17, 1 << 26, 300), (16, 1 << 24, 40), (15, 1 << 22, 40), (17, 1 << 30, 600)])
def test_uniform_random_kmers_agree_with_the_filter_words(oracle, monkeypatch, k, bf_bits, n_genes):
    """the witness of the false-positive enumeration: uniform-random k-mers, each a read of its own, classified through the
    minimiser-bucketed table and through the plain filter words + rank directory (`SHK_PROBE=bitvector`, which shares nothing
    with it but the hash).  On a dense filter a random k-mer hits a set bit often enough (0.5-25 %) for tens of thousands of
    false positives per batch -- each must come back with the list of the bit it collides with.  A 200 000-k-mer sample also
    against the oracle."""
    from shark_amd.capi import hip_memcpy_dtoh
    rng = np.random.default_rng(k + n_genes)
    genes = synth.make_genes(rng, n_genes, 300, 2500, share_every=4)
    kw = dict(k=k, c=0.0, bf_bits=bf_bits)
    _ktab_env(monkeypatch)
    h = _hip(**kw)
    info = h.build([bytes(g) for g in genes])
    assert h.probe_mode() == "minimiser-table"
    monkeypatch.setenv("SHK_PROBE", "bitvector")
    hb = _hip(**kw)
    hb.build([bytes(g) for g in genes])
    assert hb.probe_mode().startswith("bitvector")
    monkeypatch.delenv("SHK_PROBE")
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + k)
    n = 5_000_000
    hits = 0
    for rep in range(4):
        seq, off = _random_kmers(gen, n, k, dev)
        torch.cuda.synchronize()
        res = []
        for ctx in (h, hb):
            r = ctx.classify_device(n, seq.data_ptr(), off.data_ptr(), max_read_len=k)
            g = np.empty(n + 1, np.uint32)
            hip_memcpy_dtoh(g, r.gene_off, g.nbytes)
            ids = np.empty(int(r.n_assoc), np.uint16)
            if len(ids):
                hip_memcpy_dtoh(ids, r.gene_ids, ids.nbytes)
            res.append((g, ids))
        assert ", 8, " in h.last_kernel(), h.last_kernel()
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]), rep
        hits += int((np.diff(res[0][0].astype(np.int64)) > 0).sum())
        if rep == 0:
            o = oracle.Shark(**kw)
            o.build([bytes(g) for g in genes])
            m = 200_000
            hs, ho = seq[:m * k].cpu().numpy(), off[:m + 1].cpu().numpy().astype(np.uint64)
            og, oi = o.classify(hs, ho, None, None, nthreads=8)
            assert np.array_equal(og, res[0][0][:m + 1]) and np.array_equal(oi, res[0][1][:int(og[-1])])
            o.close()
    # a random k-mer is in the filter with probability (set bits) / (filter bits): the batches held that many hits, give or take
    expect = 4 * n * info["n_set_bits"] / bf_bits
    assert 0.9 * expect - 100 < hits < 1.1 * expect + 100, (hits, expect)
    h.close()
    hb.close()


def test_switches_and_fallbacks(oracle, monkeypatch):
    """without SHK_KTAB a small index does not get the table (its position table sits in the caches); SHK_NO_KTAB=1 wins over
    SHK_KTAB; k outside 15 ... 17, a filter size that is not a power of two and wrap mode keep the position table; SHK_KTAB_NT=1
    (streaming loads) gives the same results"""
    rng = np.random.default_rng(5)
    genes = synth.make_genes(rng, 40, 300, 2000, share_every=3)
    b = synth.make_reads(rng, genes, 2000, read_len=150, on_target=0.5)

    def mode(k=17, bf_bits=1 << 30, **env):
        _ktab_env(monkeypatch, on=False)
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        o = oracle.Shark(k=k, c=0.6, bf_bits=bf_bits)
        o.build([bytes(g) for g in genes])
        h = _hip(k=k, c=0.6, bf_bits=bf_bits)
        h.build([bytes(g) for g in genes])
        _same(o, h, b)
        m = h.probe_mode()
        h.close()
        o.close()
        return m
    assert mode() == "table"
    assert mode(SHK_KTAB="1") == "minimiser-table"
    assert mode(SHK_KTAB="1", SHK_KTAB_NT="1") == "minimiser-table"
    assert mode(SHK_KTAB="1", SHK_NO_KTAB="1") == "table"
    assert mode(k=18, SHK_KTAB="1") == "table"
    assert mode(k=14, SHK_KTAB="1") == "table"
    assert mode(bf_bits=3 << 28, SHK_KTAB="1") == "table-mod"
