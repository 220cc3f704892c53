"""GPU parity at the index sizes shark is used at: BASELINE configs[2] and configs[4] shapes.

  configs[2]/[3]: GENCODE-shaped reference (60 000 genes, 1.78e8 bases), k=17, 2^36-bit filter (-b 8)
  configs[4]    : same reference, k=31, -q 20, --single, 2^37-bit filter (-b 16)

10 M device-resident 2x150 bp pairs each (one launch of the 100 M / 200 M-pair streams the configs
name; the per-GPU shard of the 8-GPU configs).  Checked per config:
  * the index built on the device equals the oracle's: set bits, list total, every gene list
    (and, for the 8 GiB filter, every filter word)
  * both probe chains (position table vs filter words + rank directory) give identical results
  * a 2 M-pair sample of the batch is bit-equal to the oracle's associations
  * per-gene counters equal the histogram of the per-read results
  * configs[2] AND configs[4] at their full length: 100 M pairs / 200 M pairs with qualities streamed through
    shk_classify_submit / _wait in 4 M-pair batches (three in flight), every batch equal to the resident classification of
    the same pairs, counters = histogram, oracle sample of the last batch
  * membership at scale: 50 M reference k-mers, each classified as a read of its own, all come back assigned (a key
    the position table had lost, or a search that ends too early, would show here and nowhere in whole-read parity),
    and both probe chains return the same associations for them
  * configs[2]'s index is probed through the minimiser-bucketed table (k = 17): 10^9 uniform-random 17-mers, each a read of its
    own, must come back as the plain filter words answer them -- the only witness of the table's enumeration of the filter's
    false positives (2.5 x 10^6 of them among 10^9 random k-mers)
The oracle's serial build takes ~80 s per index (pass 2 of the reference is single-threaded, main.cpp:154-189); the tests use
so_shark_build_mt (same index, tests/test_oracle.py), a few seconds."""
import os

import numpy as np
import pytest

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

pytestmark = pytest.mark.gpu

PAIRS = 10_000_000
SAMPLE = 2_000_000
STREAM_SAMPLE = 500_000         # the oracle's sample of the LAST streamed batch (the resident batch it is cut from has its own 2 M-pair sample)
L = 150
KMER_READS = 50_000_000
KMER_READS_BITVECTOR = 5_000_000   # the filter-word chain re-classifies a tenth of the reference k-mers (both chains must agree on them)


def _reference_kmers_as_reads(genes, k, dev, with_qual, n_reads=KMER_READS):
    """the first n_reads windows of k bases that lie inside one gene, as a single-end batch of reads of length k"""
    KMER_READS = n_reads
    lens = np.array([len(g) for g in genes], dtype=np.int64)
    n_bases = int(np.searchsorted(np.cumsum(lens), KMER_READS + 100 * k)) + 1      # genes that hold that many windows
    cat = torch.from_numpy(np.concatenate(genes[:n_bases])).to(dev)
    gid = torch.repeat_interleave(torch.arange(n_bases, device=dev, dtype=torch.int32), torch.from_numpy(lens[:n_bases]).to(dev))
    inside = gid[:len(gid) - k + 1] == gid[k - 1:]
    reads = cat.unfold(0, k, 1)[inside][:KMER_READS].contiguous()
    n = reads.shape[0]
    off = torch.arange(0, (n + 1) * k, k, dtype=torch.int64, device=dev)
    qual = torch.full((n * k,), ord("I"), dtype=torch.uint8, device=dev) if with_qual else None
    return n, reads.reshape(-1), off, qual


STREAM_PAIRS = 100_000_000
STREAM_BATCH = 4_000_000


def _stream_full_length(h, o, batch, goff, gids, stream_pairs=STREAM_PAIRS, min_assigned=0.45):
    """BASELINE configs[2]'s full stream -- 100 M pairs -- (configs[4]: 200 M pairs = 400 M reads, WITH their qualities: twice the
    bytes over the link) through shk_classify_submit / _wait in batches of 4 M pairs, three in
    flight, as the reference streams a sample of any length in chunks (main.cpp:66-77, FastqSplitter.hpp:47-93).  The pool of
    10 M device-generated pairs is copied to pinned host memory once; batch i is the window of 4 M pairs that starts at a
    rolling offset, so consecutive batches differ.  Every batch's associations must equal the resident classification of the
    same pairs (`goff`/`gids`, already checked against the oracle sample), the per-gene counters the histogram of everything
    that was returned, and a 500 000-pair sample of the LAST batch the oracle's answer."""
    from shark_amd.capi import SHK_PIPE_DEPTH
    pool1 = batch["seq1"].cpu().pin_memory().numpy()
    pool2 = batch["seq2"].cpu().pin_memory().numpy()
    hasq = batch.get("qual1") is not None
    qool1 = batch["qual1"].cpu().pin_memory().numpy() if hasq else None
    qool2 = batch["qual2"].cpu().pin_memory().numpy() if hasq else None
    off = np.arange(0, (STREAM_BATCH + 1) * L, L, dtype=np.uint64)
    n_batches = stream_pairs // STREAM_BATCH
    firsts = [(i * 1_000_003) % (PAIRS - STREAM_BATCH) for i in range(n_batches)]
    cnt = np.diff(goff.astype(np.int64))
    h.gene_counts_reset()
    hist = np.zeros(65536, np.uint64)
    tickets, total, last = [], 0, None

    def drain():
        nonlocal total, last
        first, t = tickets.pop(0)
        bo, bi = h.wait(t, copy=False)
        assert np.array_equal(np.diff(bo.astype(np.int64)), cnt[first:first + STREAM_BATCH]), "batch at pair %d: per-read counts differ" % first
        want = gids[int(goff[first]):int(goff[first + STREAM_BATCH])]
        assert np.array_equal(bi, want), "batch at pair %d: gene ids differ" % first
        hist[:] += np.bincount(bi, minlength=65536).astype(np.uint64)
        total += len(bi)
        last = (first, bo.copy(), bi.copy())

    for first in firsts:
        if len(tickets) == SHK_PIPE_DEPTH:
            drain()
        sl = slice(first * L, (first + STREAM_BATCH) * L)
        tickets.append((first, h.submit(pool1[sl], off, pool2[sl], off, qool1[sl] if hasq else None, qool2[sl] if hasq else None)))
    while tickets:
        drain()
    assert total > min_assigned * stream_pairs
    assert np.array_equal(h.gene_counts(65536), hist), "per-gene counters differ from the histogram of the streamed results"
    first, bo, bi = last
    lo = STREAM_BATCH - STREAM_SAMPLE                            # the tail of the last batch
    s1 = pool1[(first + lo) * L:(first + STREAM_BATCH) * L]
    s2 = pool2[(first + lo) * L:(first + STREAM_BATCH) * L]
    so = np.arange(0, (STREAM_SAMPLE + 1) * L, L, dtype=np.uint64)
    q1 = qool1[(first + lo) * L:(first + STREAM_BATCH) * L] if hasq else None
    q2 = qool2[(first + lo) * L:(first + STREAM_BATCH) * L] if hasq else None
    og, oi = o.classify(s1, so, s2, so, q1, q2, nthreads=min(os.cpu_count() or 1, 64))
    assert np.array_equal(og.astype(np.int64), bo[lo:].astype(np.int64) - int(bo[lo]))
    assert np.array_equal(oi, bi[int(bo[lo]):])


def _scale_case(oracle, monkeypatch, k, bf_log2, q, single, compare_words, stream=0, min_assigned=0.45):
    from shark_amd import SharkHip, synth
    from shark_amd.capi import hip_memcpy_dtoh
    genes = synth.make_gencode_like_reference(60000)
    gbytes = [g.tobytes() for g in genes]
    dev = torch.device("cuda:0")
    batch = synth.make_pairs_device(PAIRS, genes, dev, seed=synth.SEED + 7, read_len=L, with_qual=q > 0)
    torch.cuda.synchronize()
    ptr = {kk: (v.data_ptr() if v is not None else 0) for kk, v in batch.items()}

    o = oracle.Shark(k=k, c=0.6, bf_bits=1 << bf_log2, min_quality=q, single=single)
    nidx = o.build(gbytes, nthreads=min(os.cpu_count() or 1, 32))

    res, kres = {}, {}
    for mode in ("auto", "bitvector"):
        if mode == "bitvector":
            monkeypatch.setenv("SHK_PROBE", "bitvector")
        else:
            monkeypatch.delenv("SHK_PROBE", raising=False)
        h = SharkHip(k=k, c=0.6, bf_bits=1 << bf_log2, min_quality=q, single=single)
        info = h.build(gbytes)
        assert ("table" in h.probe_mode()) == (mode == "auto"), h.probe_mode()
        assert info["nidx"] == nidx == 60000
        assert info["n_set_bits"] == o.num_kmer()
        if mode == "auto":
            off, ids = h.copy_lists()
            oi = o.index_kmer()
            assert info["tot_idx"] == len(oi) == int(off[-1])
            assert np.array_equal(ids, oi), "gene lists differ from the oracle's _index_kmer"
            del off, ids, oi
            if compare_words:
                hw = h.copy_bf()
                assert np.array_equal(hw, o.bf_words()), "filter words differ"
                del hw
        h.gene_counts_reset()
        r = h.classify_device(PAIRS, ptr["seq1"], ptr["off1"], ptr["seq2"], ptr["off2"], ptr["qual1"], ptr["qual2"],
                              max_read_len=L)
        goff = np.empty(PAIRS + 1, np.uint32)
        hip_memcpy_dtoh(goff, r.gene_off, goff.nbytes)
        gids = np.empty(int(r.n_assoc), np.uint16)
        if len(gids):
            hip_memcpy_dtoh(gids, r.gene_ids, gids.nbytes)
        counts = h.gene_counts(65536)
        assert np.array_equal(counts, np.bincount(gids, minlength=65536).astype(np.uint64))
        res[mode] = (goff, gids)
        if mode == "auto":
            # which chain ran: the minimiser-bucketed table for k = 17 (PM_KTAB = 8), the position table for k = 31 (PM_TAB = 3)
            assert h.probe_mode() == ("minimiser-table" if k <= 17 else "table"), h.probe_mode()
            assert (", 8, " if k <= 17 else ", 3, ") in h.last_kernel(), h.last_kernel()
        # every reference k-mer as a read of its own
        nk, kseq, koff, kqual = _reference_kmers_as_reads(genes, k, dev, q > 0, KMER_READS if mode == "auto" else KMER_READS_BITVECTOR)
        torch.cuda.synchronize()
        rk = h.classify_device(nk, kseq.data_ptr(), koff.data_ptr(), 0, 0, kqual.data_ptr() if kqual is not None else 0, 0, max_read_len=k)
        koffs = np.empty(nk + 1, np.uint32)
        hip_memcpy_dtoh(koffs, rk.gene_off, koffs.nbytes)
        kids = np.empty(int(rk.n_assoc), np.uint16)
        hip_memcpy_dtoh(kids, rk.gene_ids, kids.nbytes)
        kcnt = np.diff(koffs.astype(np.int64))
        if not single:
            assert kcnt.min() >= 1, "reference k-mer %d is not found in the index" % int(np.argmin(kcnt))
        else:
            assert (kcnt == 1).mean() > 0.9      # --single drops the k-mers that several genes share
        kres[mode] = (koffs, kids)
        del kseq, koff, kqual
        if stream and mode == "auto":
            _stream_full_length(h, o, batch, goff, gids, stream, min_assigned)
        h.close()
    assert np.array_equal(res["auto"][0], res["bitvector"][0]) and np.array_equal(res["auto"][1], res["bitvector"][1]), \
        "the two probe chains disagree"
    nb = len(kres["bitvector"][0]) - 1       # (the filter-word chain took the first KMER_READS_BITVECTOR of them)
    assert np.array_equal(kres["auto"][0][:nb + 1], kres["bitvector"][0]) and np.array_equal(kres["auto"][1][:int(kres["auto"][0][nb])], kres["bitvector"][1]), \
        "the two probe chains disagree on the reference's own k-mers"

    hb = synth.to_host_sample(batch, SAMPLE, L)
    og, oi = o.classify(hb["seq1"], hb["off1"], hb["seq2"], hb["off2"], hb["qual1"], hb["qual2"],
                        nthreads=min(os.cpu_count() or 1, 64))
    goff, gids = res["auto"]
    assert np.array_equal(og, goff[:SAMPLE + 1]), "gene_off differs at read %d" % int(np.argmax(og != goff[:SAMPLE + 1]))
    assert np.array_equal(oi, gids[:int(goff[SAMPLE])])
    o.close()
    return goff, gids


def test_config2_gencode_scale_k17_8gb(oracle, monkeypatch):
    """BASELINE configs[2] (and [3]'s per-GPU shard): 60 000 genes, k=17, c=0.6, bf = 2^36 bits."""
    goff, gids = _scale_case(oracle, monkeypatch, k=17, bf_log2=36, q=0, single=False, compare_words=True, stream=STREAM_PAIRS)
    cnt = np.diff(goff.astype(np.int64))
    assert 0.45 * PAIRS < (cnt > 0).sum() < 0.60 * PAIRS     # half the pairs are drawn from genes
    assert (cnt > 1).sum() > 100_000                          # shared gene halves: genuine ties


def test_config4_gencode_scale_k31_q20_single_16gb(oracle, monkeypatch):
    """BASELINE configs[4]: k=31 (max k), -q 20 (quality-mask path), --single, bf = 2^37 bits -- and its full stream: 200 M pairs
    (400 M reads) with their qualities through submit / wait, as the reference streams a sample of any length in 50 000-read
    chunks (FastqSplitter.hpp:47-93; the qualities are consumed at :70,:84,:104-109)."""
    goff, gids = _scale_case(oracle, monkeypatch, k=31, bf_log2=37, q=20, single=True, compare_words=False, stream=200_000_000, min_assigned=0.30)
    cnt = np.diff(goff.astype(np.int64))
    assert cnt.max() == 1                                     # --single: never more than one gene per read
    # qualities low at the read ends only (shark_amd/synth.py "ends"): most on-target pairs keep enough valid 31-mers, so the
    # quality-mask HIT path is exercised at scale (the rounds 1-2 model assigned 1.3 % of the pairs)
    assert cnt.sum() > 0.30 * PAIRS


def test_config2_index_random_kmers_agree_with_the_filter_words(monkeypatch):
    """10^9 uniform-random 17-mers against the configs[2] index, each a read of its own (c = 0: whatever list the k-mer's filter
    bit has is the answer, bloomfilter.h:78-102), through the minimiser-bucketed table and through the plain filter words + rank
    directory.  A random 17-mer is a reference k-mer with probability 2 %, and one of the filter's FALSE positives -- a key the
    table holds only because the build enumerated all 4^17 / 2 canonical k-mers -- with probability 0.25 %: 2.5 x 10^6 of those
    in this test, every one of which must come back with the list of the bit it collides with."""
    from shark_amd import SharkHip, synth
    from shark_amd.capi import hip_memcpy_dtoh
    k, n, chunks = 17, 100_000_000, 10
    genes = synth.make_gencode_like_reference(60000)
    gbytes = [g.tobytes() for g in genes]
    monkeypatch.delenv("SHK_PROBE", raising=False)
    monkeypatch.setenv("SHK_KTAB", "1")              # (no switching by what the last batch looked like)
    h = SharkHip(k=k, c=0.0, bf_bits=1 << 36)
    info = h.build(gbytes)
    assert h.probe_mode() == "minimiser-table"
    monkeypatch.setenv("SHK_PROBE", "bitvector")
    hb = SharkHip(k=k, c=0.0, bf_bits=1 << 36)
    hb.build(gbytes)
    assert hb.probe_mode().startswith("bitvector") or "bitvector" in hb.probe_mode()
    del gbytes
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(20261005)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    off = torch.arange(0, n + 1, device=dev, dtype=torch.int64) * k
    hits = assoc = 0
    for c in range(chunks):
        seq = acgt[torch.randint(0, 4, (n * k,), generator=gen, device=dev)]
        torch.cuda.synchronize()
        got = []
        for ctx in (h, hb):
            r = ctx.classify_device(n, seq.data_ptr(), off.data_ptr(), max_read_len=k)
            g = np.empty(n + 1, np.uint32)
            hip_memcpy_dtoh(g, r.gene_off, g.nbytes)
            ids = np.empty(int(r.n_assoc), np.uint16)
            if len(ids):
                hip_memcpy_dtoh(ids, r.gene_ids, ids.nbytes)
            got.append((g, ids))
        assert ", 8, " in h.last_kernel(), h.last_kernel()
        assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1]), "chunk %d" % c
        hits += int((np.diff(got[0][0].astype(np.int64)) > 0).sum())
        assoc += len(got[0][1])
        del seq, got
    # a random 17-mer hits a set bit with probability (set bits) / 2^36 on top of being one of the reference's own k-mers
    total = n * chunks
    p_ref = info["n_set_bits"] / (4.0 ** k / 2)
    p_fp = info["n_set_bits"] / float(1 << 36)
    assert 0.95 * total * (p_ref + p_fp) < hits < 1.05 * total * (p_ref + p_fp), (hits, total * (p_ref + p_fp))
    assert assoc >= hits
    h.close()
    hb.close()
