"""CPU tests (no GPU): the oracle against the reference's golden vectors.

 * example/*.truth.* (the reference's only golden files, README.md:60-69)
 * XXH64 known answers (python-xxhash)                     tests/golden/xxh64_kat.json
 * outputs of the REAL reference primitives (oracle/_ref)  tests/golden/ref_primitives.json
 * live cross-checks against oracle/_ref when the .so is present
"""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from tests import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TRUTH_MD5 = {
    "ENSG00000277117.truth.ssv": "b69162f990ff6c47b720ef5dd45af7b7",
    "sharked.sample_1.truth.fq": "b6965c372016c75e9a3d1022f1b412ae",
    "sharked.sample_2.truth.fq": "4e2251652761e60feaa2bd060ebe3947",
}


def _md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def test_fixture_provenance(example_dir):
    for name, md5 in TRUTH_MD5.items():
        assert _md5(os.path.join(example_dir, name)) == md5
    assert _md5(os.path.join(example_dir, "ENSG00000277117.fa")) == "fb5caa73df156f03daf32694c7984069"
    assert _md5(os.path.join(example_dir, "sample_1.fq")) == "5d9e324cf0a820161d9144912fcfac66"
    assert _md5(os.path.join(example_dir, "sample_2.fq")) == "77b41f242495b4b6067fcd25b783adf9"


def test_oracle_cli_reproduces_example_truth(oracle, example_dir, tmp_path):
    """BASELINE config 1 on the CPU: README.md:63-66 command line, byte-identical outputs"""
    ssv, o1, o2 = tmp_path / "o.ssv", tmp_path / "o1.fq", tmp_path / "o2.fq"
    oracle.run_cli(["-r", os.path.join(example_dir, "ENSG00000277117.fa"), "-1", os.path.join(example_dir, "sample_1.fq"),
                    "-2", os.path.join(example_dir, "sample_2.fq"), "-o", str(o1), "-p", str(o2)], str(ssv))
    assert _md5(ssv) == TRUTH_MD5["ENSG00000277117.truth.ssv"]
    assert _md5(o1) == TRUTH_MD5["sharked.sample_1.truth.fq"]
    assert _md5(o2) == TRUTH_MD5["sharked.sample_2.truth.fq"]


def test_oracle_batch_api_reproduces_example_truth(oracle, example_dir):
    """same through the in-memory batch API used as checker for the GPU path (1 and 4 threads)"""
    fa = synth.read_fasta(os.path.join(example_dir, "ENSG00000277117.fa"))
    r1 = synth.read_fastq(os.path.join(example_dir, "sample_1.fq"))
    r2 = synth.read_fastq(os.path.join(example_dir, "sample_2.fq"))
    o = oracle.Shark(k=17, c=0.6, bf_bits=1 << 33)
    assert o.build([s for _, s in fa]) == 1
    assert o.num_kmer() == 17483
    words = o.bf_words()
    first = np.flatnonzero(words)[:3]
    pos = [int(w) * 64 + int(np.log2(int(words[w]) & -int(words[w]))) for w in first]
    assert pos == [543356, 1120387, 1209327]                   # first three set positions
    batch = synth.batch_from_lists([s for _, s, _ in r1], [s for _, s, _ in r2])
    truth = open(os.path.join(example_dir, "ENSG00000277117.truth.ssv"), "rb").read()
    for nt in (1, 4):
        goff, gids = o.classify(batch["seq1"], batch["off1"], batch["seq2"], batch["off2"], nthreads=nt)
        lines = b"".join(r1[i][0] + b" " + fa[gids[j]][0] + b"\n" for i in range(len(r1)) for j in range(goff[i], goff[i + 1]))
        assert lines == truth


def test_xxh64_known_answers(oracle):
    L = oracle.lib()
    kat = json.load(open(os.path.join(GOLD, "xxh64_kat.json")))
    for v, h in kat["u64_le_seed0"]:
        assert L.so_get_hash(v) == h
    for hx, seed, h in kat["bytes"]:
        b = bytes.fromhex(hx)
        assert L.so_xxh64(b, len(b), seed) == h
    # SURVEY.md 8a row 6
    assert L.so_get_hash(0) == 0x34c96acdcadb1bbb
    assert L.so_get_hash(1) == 0x9f29cb17a2a49995
    assert L.so_get_hash(0x3ffffffff) == 0x31eb3411f5fef9d8
    assert L.so_get_hash(0x3fffffffffffffff) == 0xcc4e8923c52e58a0
    # first k-mer of the example gene
    assert L.so_get_hash(0x1e8a496ed) == 0xfc1d69b91093b18d
    assert 0xfc1d69b91093b18d % (1 << 33) == 4573081997


def test_oracle_matches_reference_primitive_fixtures(oracle):
    """fixtures produced by the real reference code (tests/golden/gen_ref_primitives.py)"""
    L = oracle.lib()
    fx = json.load(open(os.path.join(GOLD, "ref_primitives.json")))
    for c in range(128):
        assert L.so_to_int(bytes([c])) == fx["to_int"][c]
    for c in range(128, 256):
        assert L.so_to_int(bytes([c])) == 0
    for p in fx["kmer_prims"]:
        assert L.so_revcompl(p["kmer"], p["k"]) == p["revcompl"]
        assert L.so_lsappend(p["kmer"], p["c"], p["k"]) == p["lsappend"]
        assert L.so_rsprepend(p["kmer"], p["c"], p["k"]) == p["rsprepend"]
        assert L.so_get_hash(p["kmer"]) == p["hash"]
    for case in fx["build_kmer"]:
        s = case["seq"].encode()
        for p0, (v, p1) in enumerate(case["results"]):
            p = C.c_int(p0)
            assert L.so_build_kmer(s, len(s), C.byref(p), case["k"]) == v
            assert p.value == p1
    # FastqSplitter join + mask
    m1, m2 = fx["fastq_records"]["mate1"], fx["fastq_records"]["mate2"]
    for case in fx["fastq_splitter"]:
        assert len(case["reads"]) == len(m1)
        for i, row in enumerate(case["reads"]):
            s1, q1 = m1[i][1].encode(), m1[i][2].encode()
            s2, q2 = m2[i][1].encode(), m2[i][2].encode()
            out = C.create_string_buffer(len(s1) + len(s2) + 2)
            n = L.so_join_mask(s1, len(s1), q1, s2 if case["paired"] else None, len(s2) if case["paired"] else 0,
                               q2 if case["paired"] else None, int(case["paired"]), bytes([case["q"]]), out)
            assert out.raw[:n].hex() == row["joined_hex"], (case["variant"], case["paired"], case["q"], i)
            assert row["id1"] == m1[i][0].split()[0] and row["seq1"] == m1[i][1] and row["qual1"] == m1[i][2]


def test_oracle_cli_reader_matches_reference_records(oracle, tmp_path):
    """the oracle CLI's FASTA/FASTQ reader sees the records the reference's kseq sees"""
    fx = json.load(open(os.path.join(GOLD, "ref_primitives.json")))
    # FASTA legend order incl. an empty record and a record without valid k-mers: run the CLI and read back
    fa = tmp_path / "x.fa"
    fa.write_text(fx["fasta"]["text"])
    recs = fx["fasta"]["records"]
    assert [r[0] for r in recs] == ["g1", "g2", "empty", "g4"]
    assert recs[0][1] == "ACGTACGTACGTACGT" and recs[2][1] == "" and recs[3][1] == "NNNNACGT"
    # a read equal to g2's first 60 bases must be reported under the name the reference's numbering gives
    g2 = recs[1][1]
    fq = tmp_path / "r.fq"
    fq.write_text("@q1 c\n%s\n+\n%s\n" % (g2[:60], "I" * 60))
    ssv = tmp_path / "o.ssv"
    oracle.run_cli(["-r", str(fa), "-1", str(fq), "-k", "11", "-B", str(1 << 20), "-o", str(tmp_path / "o.fq")], str(ssv))
    assert ssv.read_text() == "q1 g2\n"


def test_live_reference_primitives(oracle):
    """when oracle/_ref is built, compare on fresh random inputs (not only the committed fixture)"""
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref/libsharkref.so not built")
    L = oracle.lib()
    rng = np.random.default_rng(99)
    for _ in range(2000):
        k = int(rng.integers(1, 32))
        v = int(rng.integers(0, 1 << 62)) & ((1 << (2 * k)) - 1)
        c = int(rng.integers(0, 4))
        assert L.so_revcompl(v, k) == R.ref_revcompl(v, k)
        assert L.so_lsappend(v, c, k) == R.ref_lsappend(v, c, k)
        assert L.so_rsprepend(v, c, k) == R.ref_rsprepend(v, c, k)
        assert L.so_get_hash(v) == R.ref_get_hash(v)
    alpha = np.frombuffer(b"ACGTacgtNn.", dtype=np.uint8)
    for _ in range(300):
        n = int(rng.integers(0, 120))
        s = alpha[rng.integers(0, len(alpha), size=n)].tobytes()
        k = int(rng.integers(1, 32))
        for p0 in range(0, n + 1, 3):
            p, q = C.c_int(p0), C.c_int(p0)
            assert L.so_build_kmer(s, n, C.byref(p), k) == R.ref_build_kmer(s, n, C.byref(q), k)
            assert p.value == q.value


def _brute_force_kmers(s, k):
    """definition used by the HIP kernels: the k-mer starting at i exists iff all k characters are valid"""
    code = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}
    out = []
    for i in range(0, len(s) - k + 1):
        w = s[i:i + k]
        if all(ch in code for ch in w):
            fw = 0
            for ch in w:
                fw = (fw << 2) | code[ch]
            rc = 0
            for ch in reversed(w):
                rc = (rc << 2) | (3 - code[ch])
            out.append(min(fw, rc))
    return out


def test_rolling_walk_equals_positional_definition(oracle):
    """KmerBuilder's rolling walk with restarts (KmerBuilder.hpp:40-72) visits exactly the all-valid windows"""
    L = oracle.lib()
    rng = np.random.default_rng(3)
    alpha = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    for _ in range(200):
        n = int(rng.integers(0, 150))
        k = int(rng.integers(1, 32))
        s = alpha[rng.integers(0, len(alpha), size=n)].tobytes()
        buf = (C.c_uint64 * max(n, 1))()
        cnt = L.so_kmer_builder(s, n, k, buf)
        want = [L.so_get_hash(x) for x in _brute_force_kmers(s, k)]
        assert list(buf[:cnt]) == want


def test_threads_do_not_change_results(oracle):
    rng = np.random.default_rng(8)
    genes = synth.make_genes(rng, 25, 200, 1500, share_every=3)
    o = oracle.Shark(k=13, c=0.5, bf_bits=1 << 20)
    o.build([bytes(g) for g in genes])
    b = synth.make_reads(rng, genes, 1200, read_len=100, paired=True, on_target=0.7, var_len=True)
    g1, i1 = o.classify(b["seq1"], b["off1"], b["seq2"], b["off2"], nthreads=1)
    g4, i4 = o.classify(b["seq1"], b["off1"], b["seq2"], b["off2"], nthreads=4)
    assert np.array_equal(g1, g4) and np.array_equal(i1, i4)
    assert (np.diff(g1.astype(np.int64)) > 1).sum() > 0, "the set must contain ties"
    # batch API == per-read API
    for i in range(0, 1200, 97):
        s1 = bytes(b["seq1"][int(b["off1"][i]):int(b["off1"][i + 1])])
        s2 = bytes(b["seq2"][int(b["off2"][i]):int(b["off2"][i + 1])])
        genes_i, mx, mk, ln = o.analyze(s1 + b"N" + s2)
        assert tuple(genes_i) == tuple(int(x) for x in i1[g1[i]:g1[i + 1]])


def test_gene_numbering_quirk_in_oracle(oracle):
    rng = np.random.default_rng(9)
    g1, g2, g3 = (synth.random_seq(rng, 400) for _ in range(3))
    recs = [bytes(g1), b"N" * 50, b"ACGT", bytes(g2), b"ACGTNACGTNACGTNACGTNACGTN", b"", bytes(g3)]
    o = oracle.Shark(k=17, c=0.6, bf_bits=1 << 22)
    assert o.build(recs) == 5                                   # main.cpp:165 skips ++nidx twice
    assert set(np.unique(o.index_kmer())) == {0, 2, 4}


# ---------------------------------------------------------------------------
# hand-worked cases: expected values derived on paper from the reference source
# (tests/golden/handworked.json), so that a misreading shared by the oracle and
# the kernels would show up here
# ---------------------------------------------------------------------------
def _handworked():
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "handworked.json")))["cases"]


def L_hash(oracle, v):
    return int(oracle.lib().so_get_hash(v))


def _handworked_batch(case):
    from tests import synth
    paired = case["reads"][0]["m2"] is not None
    m1 = [r["m1"].encode() for r in case["reads"]]
    q1 = [r["q1"].encode() for r in case["reads"]]
    m2 = [r["m2"].encode() for r in case["reads"]] if paired else None
    q2 = [r["q2"].encode() for r in case["reads"]] if paired else None
    return synth.batch_from_lists(m1, m2, q1, q2), paired


@pytest.mark.parametrize("case", _handworked(), ids=lambda c: c["name"])
def test_handworked_cases(oracle, case, tmp_path):
    o = oracle.Shark(k=case["k"], c=case["c"], bf_bits=case["bf_bits"], min_quality=case["q"], single=case["single"])
    o.build([seq.encode() for _, seq in case["fasta"]])
    assert o.num_kmer() == case.get("set_bits", case["distinct_kmers"])    # no filter collision (the derivations' one assumption), or exactly the case's
    if "xxh64_mod_64" in case:                                         # the case's filter positions: python-xxhash, not the oracle's own hash
        import struct
        import xxhash
        for kmer, pos in case["xxh64_mod_64"].items():
            v = 0
            for ch in kmer:
                v = (v << 2) | "ACGT".index(ch)
            assert xxhash.xxh64(struct.pack("<Q", v), seed=0).intdigest() % 64 == pos == L_hash(oracle, v) % 64, kmer
    L = oracle.lib()
    batch, paired = _handworked_batch(case)
    for r in case["reads"]:
        s1, s2 = r["m1"].encode(), (r["m2"] or "").encode()
        out = C.create_string_buffer(len(s1) + len(s2) + 2)
        n = L.so_join_mask(s1, len(s1), r["q1"].encode(), s2 if paired else None, len(s2), r["q2"].encode() if paired else None,
                           int(paired), bytes([case["q"]]), out)
        genes, mx, mk, ln = o.analyze(out.raw[:n])
        assert (ln, [mx, mk], genes) == (r["len"], r["best"], r["genes"]), r["id"]
    goff, gids = o.classify(batch["seq1"], batch["off1"], batch["seq2"], batch["off2"], batch["qual1"], batch["qual2"])
    assert [list(map(int, gids[goff[i]:goff[i + 1]])) for i in range(len(case["reads"]))] == [r["genes"] for r in case["reads"]]
    if case.get("no_cli"):                                             # (a filter size the CLI's -b, which counts in GB, cannot name)
        names = [n_ for n_, _ in case["fasta"]]
        assert "".join("%s %s\n" % (r["id"], names[g]) for r in case["reads"] for g in r["genes"]) == case["ssv"]
        return
    # and end to end through the oracle CLI: ssv bytes
    fa = tmp_path / "g.fa"
    fa.write_text("".join(">%s\n%s\n" % (n_, s_) for n_, s_ in case["fasta"]))
    f1 = tmp_path / "r1.fq"
    f1.write_text("".join("@%s\n%s\n+\n%s\n" % (r["id"], r["m1"], r["q1"]) for r in case["reads"]))
    args = ["-r", str(fa), "-1", str(f1), "-k", str(case["k"]), "-c", str(case["c"]), "-q", str(case["q"]), "-b", "1",
            "-o", str(tmp_path / "o1.fq")]
    if paired:
        f2 = tmp_path / "r2.fq"
        f2.write_text("".join("@%s\n%s\n+\n%s\n" % (r["id"], r["m2"], r["q2"]) for r in case["reads"]))
        args += ["-2", str(f2), "-p", str(tmp_path / "o2.fq")]
    if case["single"]:
        args.append("-s")
    oracle.run_cli(args, str(tmp_path / "out.ssv"))
    assert (tmp_path / "out.ssv").read_text() == case["ssv"]


@pytest.mark.parametrize("k,bf_bits,n_genes,gene_len,threads", [(17, 1 << 26, 300, 2500, 4), (5, 4099, 40, 300, 3), (31, 1 << 30, 120, 1500, 8),
                                                                (9, 1 << 18, 66_000, 40, 5)])
def test_multithreaded_build_equals_the_serial_one(oracle, k, bf_bits, n_genes, gene_len, threads):
    """so_shark_build_mt (the scale tests' 60 000-gene indices in seconds instead of minutes) must leave exactly the index
    so_shark_build leaves: filter words, number of set bits, every gene list in order -- with the numbering quirk (a record
    without a valid k-mer, main.cpp:165), records shorter than k, genes sharing halves, a filter dense with collisions
    (4 099 bits), and more than 65 536 genes (the uint16_t comparison of bloomfilter.h:72 then appends per occurrence)"""
    rng = np.random.default_rng(k * 7 + n_genes)
    genes = [bytes(g) for g in synth.make_genes(rng, n_genes, max(k + 3, gene_len // 3), gene_len, share_every=3)]
    genes.insert(len(genes) // 3, b"N" * (k + 20))          # at least k long, no valid k-mer: takes no gene number
    genes.insert(len(genes) // 2, b"ACGT"[:max(1, min(4, k - 1))])   # shorter than k: takes one
    g = bytearray(genes[5])
    g[len(g) // 2] = ord("N")
    genes[5] = bytes(g)
    a = oracle.Shark(k=k, c=0.6, bf_bits=bf_bits)
    na = a.build(genes)
    b = oracle.Shark(k=k, c=0.6, bf_bits=bf_bits)
    nb = b.build(genes, nthreads=threads)
    assert na == nb == len(genes) - 1
    assert a.num_kmer() == b.num_kmer() > 0
    assert np.array_equal(a.bf_words(), b.bf_words())
    assert np.array_equal(a.index_kmer(), b.index_kmer())
    reads = synth.make_reads(rng, [np.frombuffer(x, dtype=np.uint8) for x in genes[:20]], 500, read_len=max(2 * k, 40), on_target=0.7)
    ra = a.classify(reads["seq1"], reads["off1"], reads["seq2"], reads["off2"], nthreads=2)
    rb = b.classify(reads["seq1"], reads["off1"], reads["seq2"], reads["off2"], nthreads=2)
    assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
    a.close()
    b.close()
