"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the
reference's golden files.  Bit-exact: everything here is integer / byte work;
the one fp64 compare (ReadAnalyzer.hpp:104) is evaluated identically.

Run on the GPU box with `pytest -m gpu`."""
import os

import numpy as np
import pytest

try:  # torch bundles its own HIP runtime: load it BEFORE libsharkhip so one runtime serves both
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "bitvector", "no-lds-table", "force-generic", "ktable"])
def probe(request, monkeypatch):
    """run a test once with the index's automatic probe structure (position table where possible; tiny indices: the exact
    table in LDS for uniform batches), once forced onto the plain filter words (+rank directory), once without the
    LDS-resident table (so that tiny indices also exercise the LDS-summary + position-table chain on uniform batches), and
    once with SHK_FORCE_GENERIC=1: every batch through classify_fast_kernel / process_read, whose table-mode instantiations
    otherwise only see batches with reads of more than 512 bases; "ktable": probes through the minimiser-bucketed table"""
    monkeypatch.delenv("SHK_PROBE", raising=False)
    monkeypatch.delenv("SHK_NO_LDS_TABLE", raising=False)
    monkeypatch.delenv("SHK_FORCE_GENERIC", raising=False)
    for v in ("SHK_KTAB", "SHK_NO_LDS_SUMMARY", "SHK_NO_SUMMARY"):
        monkeypatch.delenv(v, raising=False)
    if request.param == "ktable":
        # the k-mer keyed, minimiser-bucketed table (k = 15 ... 17; other k: the plain position table), which is otherwise built
        # for tables beyond the caches only -- and used where the index's chain is `table`: no summaries in front of it
        monkeypatch.setenv("SHK_KTAB", "1")
        monkeypatch.setenv("SHK_NO_LDS_SUMMARY", "1")
        monkeypatch.setenv("SHK_NO_SUMMARY", "1")
    if request.param == "bitvector":
        monkeypatch.setenv("SHK_PROBE", "bitvector")
    elif request.param == "no-lds-table":
        monkeypatch.setenv("SHK_NO_LDS_TABLE", "1")
    elif request.param == "force-generic":
        monkeypatch.setenv("SHK_FORCE_GENERIC", "1")
    return request.param


def _hip(**kw):
    from shark_amd import SharkHip
    return SharkHip(**kw)


def _build_both(oracle, genes, **kw):
    o = oracle.Shark(k=kw.get("k", 17), c=kw.get("c", 0.6), bf_bits=kw.get("bf_bits", 1 << 33),
                     min_quality=kw.get("min_quality", 0), single=kw.get("single", False))
    nidx = o.build([bytes(g) for g in genes])
    h = _hip(**kw)
    info = h.build([bytes(g) for g in genes])
    assert info["nidx"] == nidx
    assert info["n_records"] == len(genes)
    return o, h, info


def _compare_index(o, h, info):
    assert info["n_set_bits"] == o.num_kmer()
    ow = o.bf_words()
    hw = h.copy_bf()
    assert np.array_equal(ow, hw), "Bloom filter words differ"
    # lists: the oracle's select-encoded lists vs the explicit CSR
    off, ids = h.copy_lists()
    assert int(off[-1]) == len(ids) == len(o.index_kmer())
    assert np.array_equal(ids, o.index_kmer())
    assert np.all(np.diff(off.astype(np.int64)) >= 1), "every set bit owns a non-empty list"


def _compare_classify(o, h, batch, nthreads=2):
    og, oi = o.classify(batch["seq1"], batch["off1"], batch["seq2"], batch["off2"], batch["qual1"], batch["qual2"],
                        nthreads=nthreads)
    hg, hi = h.classify(batch["seq1"], batch["off1"], batch["seq2"], batch["off2"], batch["qual1"], batch["qual2"])
    assert np.array_equal(og, hg), "gene_off differs at read %d" % int(np.argmax(og != hg))
    assert np.array_equal(oi, hi)
    return og, oi


def _probe_every_kmer(o, h, genes, k, stride=1):
    """every (stride-th) reference k-mer as a single-end read of its own: the associations must be the oracle's, i.e. each
    k-mer must be found with its whole gene list -- the key-by-key check of whatever structure the index is probed through
    (whole reads cannot give it: a single lost k-mer hides behind its neighbours' coverage)"""
    kmers = [g[i:i + k] for g in genes for i in range(0, len(g) - k + 1, stride)]
    og, _ = _compare_classify(o, h, synth.batch_from_lists(kmers, None))
    return og, len(kmers)


# ---------------------------------------------------------------------------
# BASELINE config 1: the bundled example, k=17 c=0.6 bf=1GB
# ---------------------------------------------------------------------------
def test_example_bit_exact(oracle, probe, example_dir, tmp_path):
    fa = synth.read_fasta(os.path.join(example_dir, "ENSG00000277117.fa"))
    r1 = synth.read_fastq(os.path.join(example_dir, "sample_1.fq"))
    r2 = synth.read_fastq(os.path.join(example_dir, "sample_2.fq"))
    h = _hip(k=17, c=0.6, bf_bits=1 << 33)
    info = h.build([s for _, s in fa])
    assert info["n_set_bits"] == 17483          # SURVEY.md 8a row 11 (k=17, B=2^33)
    assert info["n_ref_kmers"] == 18158
    batch = synth.batch_from_lists([s for _, s, _ in r1], [s for _, s, _ in r2])
    goff, gids = h.classify(batch["seq1"], batch["off1"], batch["seq2"], batch["off2"])
    # render ssv + FASTQ exactly as ReadOutput.hpp:37-50 does and compare with the truth files
    names = [n for n, _ in fa]
    ssv, fq1, fq2 = [], [], []
    for i in range(len(r1)):
        for j in range(goff[i], goff[i + 1]):
            ssv.append(r1[i][0] + b" " + names[gids[j]] + b"\n")
        if goff[i + 1] > goff[i]:
            fq1.append(b"@" + r1[i][0] + b"\n" + r1[i][1] + b"\n+\n" + r1[i][2] + b"\n")
            fq2.append(b"@" + r2[i][0] + b"\n" + r2[i][1] + b"\n+\n" + r2[i][2] + b"\n")
    assert b"".join(ssv) == open(os.path.join(example_dir, "ENSG00000277117.truth.ssv"), "rb").read()
    assert b"".join(fq1) == open(os.path.join(example_dir, "sharked.sample_1.truth.fq"), "rb").read()
    assert b"".join(fq2) == open(os.path.join(example_dir, "sharked.sample_2.truth.fq"), "rb").read()
    assert len(ssv) == 1929
    # per-gene counts (the quantity all-reduced across GPUs)
    assert int(h.gene_counts(4)[0]) == 1929
    # and the oracle agrees on the index too
    o = oracle.Shark(k=17, c=0.6, bf_bits=1 << 33)
    o.build([s for _, s in fa])
    _compare_index(o, h, info)


# ---------------------------------------------------------------------------
# synthetic multi-gene sets: ties, N, lowercase, collisions (small filter)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("k,bf_bits,paired,read_len", [
    (17, 1 << 26, True, 150),
    (31, 1 << 26, True, 150),
    (17, 1 << 18, True, 100),     # dense filter: cross-gene collisions, long lists
    (5, 1 << 12, True, 60),       # tiny k: every k-mer in many genes
    (1, 1 << 10, False, 40),
    (17, 1000003, True, 150),     # non power-of-two size: true modulo
    (21, 3 << 20, False, 120),
])
def test_synthetic_parity(oracle, probe, k, bf_bits, paired, read_len):
    rng = np.random.default_rng(1234 + k)
    genes = synth.make_genes(rng, 40, 100, 1500, share_every=4)
    o, h, info = _build_both(oracle, genes, k=k, bf_bits=bf_bits)
    if probe == "bitvector":
        assert "table" not in h.probe_mode()
    else:
        assert "table" in h.probe_mode()
        assert h.probe_mode().endswith("-mod") == bool(bf_bits & (bf_bits - 1))
    _compare_index(o, h, info)
    batch = synth.make_reads(rng, genes, 3000, read_len=read_len, paired=paired, on_target=0.6,
                             n_rate=0.01, lower_rate=0.05, var_len=True)
    goff, _ = _compare_classify(o, h, batch)
    assert goff[-1] > 0
    _probe_every_kmer(o, h, genes, k)


def test_quality_mask_and_single(oracle):
    rng = np.random.default_rng(77)
    genes = synth.make_genes(rng, 30, 300, 2000, share_every=3)
    for single in (False, True):
        for q in (20, 2, 41):
            o, h, info = _build_both(oracle, genes, k=17, bf_bits=1 << 24, min_quality=q, single=single, c=0.4)
            batch = synth.make_reads(rng, genes, 2000, read_len=150, paired=True, on_target=0.7, qual=True, var_len=True)
            _compare_classify(o, h, batch)
            batch = synth.make_reads(rng, genes, 500, read_len=90, paired=False, on_target=0.7, qual=True)
            _compare_classify(o, h, batch)


def test_confidence_edges(oracle):
    rng = np.random.default_rng(5)
    genes = synth.make_genes(rng, 10, 300, 800)
    batch = synth.make_reads(rng, genes, 1500, read_len=100, paired=True, on_target=0.8, sub_rate=0.05)
    for c in (0.0, 1.0, 0.5, 0.3333333333333333, 0.83):
        o, h, _ = _build_both(oracle, genes, k=11, bf_bits=1 << 20, c=c)
        _compare_classify(o, h, batch)


def test_gene_numbering_quirk(oracle):
    """main.cpp:165: a record >= k long without any valid k-mer does not
    advance nidx; records shorter than k do."""
    rng = np.random.default_rng(9)
    g1, g2, g3 = (synth.random_seq(rng, 400) for _ in range(3))
    recs = [bytes(g1), b"N" * 50, b"ACGT", bytes(g2), b"ACGTNACGTNACGTNACGTNACGTN", b"", bytes(g3)]
    o, h, info = _build_both(oracle, recs, k=17, bf_bits=1 << 22)
    assert info["nidx"] == 5 and info["n_records"] == 7
    _compare_index(o, h, info)
    batch = synth.make_reads(rng, [g1, g2, g3], 600, read_len=100, paired=True, on_target=0.9)
    goff, gids = _compare_classify(o, h, batch)
    assert set(int(x) for x in np.unique(gids)) == {0, 2, 4}


def test_many_ties_overflow_inline(oracle, probe):
    """more than SHK_INLINE_IDS genes tie: the general kernel writes the list"""
    rng = np.random.default_rng(11)
    core = synth.random_seq(rng, 600)
    genes = [np.concatenate([core, synth.random_seq(rng, 50 + 7 * i)]) for i in range(12)]
    genes += synth.make_genes(rng, 5, 300, 600)
    o, h, info = _build_both(oracle, genes, k=17, bf_bits=1 << 24)
    _compare_index(o, h, info)
    batch = synth.make_reads(rng, [core], 400, read_len=120, paired=True, on_target=1.0, sub_rate=0.0, n_rate=0.0)
    goff, gids = _compare_classify(o, h, batch)
    assert (np.diff(goff.astype(np.int64)) == 12).sum() > 300
    t = h.timing()
    assert t["last_n_tie"] > 300
    assert np.array_equal(h.gene_counts(20), np.bincount(gids, minlength=20)[:20].astype(np.uint64))


def test_long_and_ragged_reads(oracle, probe):
    """reads beyond the LDS specialisation (general kernel), empty reads, reads
    shorter than k, reads of only N"""
    rng = np.random.default_rng(13)
    genes = synth.make_genes(rng, 8, 3000, 9000, share_every=2)
    o, h, info = _build_both(oracle, genes, k=19, bf_bits=1 << 25)
    m1, m2 = [], []
    for i in range(300):
        g = genes[i % len(genes)]
        L1 = int(rng.integers(0, 2500))
        L2 = int(rng.integers(0, 2500))
        st = int(rng.integers(0, len(g) - 2500))
        a = g[st:st + L1].copy()
        b = synth.revcomp(g[st:st + 2500])[:L2].copy()
        if i % 7 == 0 and L1:
            a[rng.integers(0, L1, size=max(1, L1 // 20))] = ord("N")
        m1.append(a.tobytes())
        m2.append(b.tobytes())
    m1 += [b"", b"ACGT", b"N" * 100, b"A" * 18, bytes(genes[0][:19])]
    m2 += [b"", b"", b"N" * 3, b"", b""]
    batch = synth.batch_from_lists(m1, m2)
    goff, _ = _compare_classify(o, h, batch)
    assert h.timing()["last_n_long"] > 100
    assert goff[-1] > 100


def test_empty_batch_and_state_machine(oracle):
    from shark_amd import SharkHip, SharkHipError
    h = SharkHip(k=17, bf_bits=1 << 20)
    with pytest.raises(SharkHipError):
        h.classify(np.zeros(0, np.uint8), np.zeros(1, np.uint64))      # not finalized yet
    h.build([b"ACGTACGTACGTACGTACGTACGTACGTAAAC"])
    assert h.ref_add(b"ACGT") != 0                                      # no going back (bloomfilter.h:104-110)
    goff, gids = h.classify(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert list(goff) == [0] and len(gids) == 0
    # empty reference
    h2 = SharkHip(k=17, bf_bits=1 << 20)
    info = h2.build([])
    assert info["n_set_bits"] == 0 and info["tot_idx"] == 0
    b = synth.batch_from_lists([b"ACGTACGTACGTACGTACGTACGT"], [b"ACGTTGCATGCATGCATGCATGCA"])
    goff, gids = h2.classify(b["seq1"], b["off1"], b["seq2"], b["off2"])
    assert list(goff) == [0, 0]


def test_device_resident_api_and_roundtrip_properties():
    """size-independent properties at a larger scale (no oracle): reads cut
    from a gene without errors are always assigned to it; random reads never are
    (filter is sparse); results do not depend on batch splitting."""
    rng = np.random.default_rng(21)
    genes = synth.make_genes(rng, 3, 20000, 20000)
    h = _hip(k=17, c=0.6, bf_bits=1 << 33)
    h.build([bytes(g) for g in genes])
    n = 200000
    batch = synth.make_reads(rng, genes[:1], n, read_len=150, paired=True, on_target=0.5, sub_rate=0.0, n_rate=0.0)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) for k, v in batch.items() if v is not None}
    r = h.classify_device(n, t["seq1"].data_ptr(), t["off1"].data_ptr(), t["seq2"].data_ptr(), t["off2"].data_ptr(), max_read_len=150)
    assert r.n == n
    goff, gids = h.classify(batch["seq1"], batch["off1"], batch["seq2"], batch["off2"])
    assert int(r.n_assoc) == int(goff[-1])
    cnt = np.diff(goff.astype(np.int64))
    assert 0.45 * n < (cnt == 1).sum() < 0.55 * n and (cnt > 1).sum() == 0
    assert np.all(gids == 0)
    # split invariance
    half = n // 2
    o1 = batch["off1"][:half + 1]
    o2 = batch["off2"][:half + 1]
    g1, i1 = h.classify(batch["seq1"][:int(o1[-1])], o1, batch["seq2"][:int(o2[-1])], o2)
    assert np.array_equal(g1, goff[:half + 1]) and np.array_equal(i1, gids[:int(goff[half])])
    # exact work counters agree with first principles: every valid k-mer is probed once
    w = h.count_work(n, t["seq1"].data_ptr(), t["off1"].data_ptr(), t["seq2"].data_ptr(), t["off2"].data_ptr())
    assert w["n_kmers"] == n * 2 * (150 - 17 + 1)
    assert w["n_bases"] == n * 300
    assert w["n_hits"] >= (cnt == 1).sum() * 2 * 134


# ---------------------------------------------------------------------------
# the drop-in CLI: same flags, byte-identical ssv + FASTQ (README.md:63-69)
# ---------------------------------------------------------------------------
def _run_shark(args, cwd):
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shark_amd", "bin", "shark")
    return subprocess.run([exe] + args, cwd=cwd, capture_output=True)


@pytest.mark.parametrize("extra", [[], ["--batch", "777"], ["-t", "4", "--batch", "50000"]])
def test_cli_example_byte_identical(example_dir, tmp_path, extra):
    o1, o2 = tmp_path / "o1.fq", tmp_path / "o2.fq"
    r = _run_shark(["-r", os.path.join(example_dir, "ENSG00000277117.fa"), "-1", os.path.join(example_dir, "sample_1.fq"),
                    "-2", os.path.join(example_dir, "sample_2.fq"), "-o", str(o1), "-p", str(o2)] + extra, str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout == open(os.path.join(example_dir, "ENSG00000277117.truth.ssv"), "rb").read()
    assert o1.read_bytes() == open(os.path.join(example_dir, "sharked.sample_1.truth.fq"), "rb").read()
    assert o2.read_bytes() == open(os.path.join(example_dir, "sharked.sample_2.truth.fq"), "rb").read()
    assert b"[shark/Association done] Time elapsed" in r.stderr


@pytest.mark.parametrize("gz", [False, True])
def test_cli_reads_samples_from_named_pipes(example_dir, tmp_path, gz):
    """the reference opens its samples with gzopen, which reads pipes as well as files (`-1 <(zcat a.fq.gz)`): nothing in front of
    the reader may look into a pipe (two bytes taken to see whether it is gzip, or 64 KiB to guess the batch size, would be gone),
    and everything that wants to seek has to stand back.  The bundled example through two named pipes, plain and compressed:
    the truth files byte for byte."""
    import gzip
    import threading
    fifos = []
    feeders = []
    for m in (1, 2):
        data = open(os.path.join(example_dir, "sample_%d.fq" % m), "rb").read()
        if gz:
            data = gzip.compress(data, 6)
        path = str(tmp_path / ("pipe_%d" % m))
        os.mkfifo(path)
        fifos.append(path)

        def feed(path=path, data=data):
            with open(path, "wb") as f:      # (blocks until the command opens the pipe)
                f.write(data)
        t = threading.Thread(target=feed, daemon=True)
        t.start()
        feeders.append(t)
    o1, o2 = tmp_path / "o1.fq", tmp_path / "o2.fq"
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shark_amd", "bin", "shark")
    try:
        r = subprocess.run([exe, "-r", os.path.join(example_dir, "ENSG00000277117.fa"), "-1", fifos[0], "-2", fifos[1], "-o", str(o1), "-p", str(o2)],
                           cwd=str(tmp_path), capture_output=True, timeout=90)
    except subprocess.TimeoutExpired:
        for path in fifos:      # (release feeders that nobody has opened for)
            try:
                fd = os.open(path, os.O_RDONLY | os.O_NONBLOCK)
                os.close(fd)
            except OSError:
                pass
        pytest.fail("shark did not finish on named pipes (a second open of a pipe waits for a writer that never comes)")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    for t in feeders:
        t.join(timeout=30)
    assert r.stdout == open(os.path.join(example_dir, "ENSG00000277117.truth.ssv"), "rb").read()
    assert o1.read_bytes() == open(os.path.join(example_dir, "sharked.sample_1.truth.fq"), "rb").read()
    assert o2.read_bytes() == open(os.path.join(example_dir, "sharked.sample_2.truth.fq"), "rb").read()


def test_cli_matches_oracle_cli_on_options(oracle, tmp_path):
    """-q / -s / -k / single-end / gz / multi-gene with the numbering quirk: the HIP CLI and the oracle CLI
    must print the same bytes (ssv on stdout, FASTQ files)"""
    import gzip
    rng = np.random.default_rng(4242)
    genes = synth.make_genes(rng, 12, 300, 1500, share_every=3)
    fa = tmp_path / "g.fa"
    with open(fa, "w") as f:
        for i, g in enumerate(genes):
            if i == 4:
                f.write(">skipme no valid kmer\n" + "N" * 70 + "\n")
            f.write(">gene%d some description\n" % i)
            s = bytes(g).decode()
            for j in range(0, len(s), 60):
                f.write(s[j:j + 60] + "\n")
    b = synth.make_reads(rng, genes, 4000, read_len=120, paired=True, on_target=0.7, qual=True, var_len=True, lower_rate=0.02)

    def write_fq(path, seq, off, qual, tag, gz=False):
        op = gzip.open if gz else open
        with op(path, "wb") as f:
            for i in range(len(off) - 1):
                s = bytes(seq[int(off[i]):int(off[i + 1])])
                q = bytes(qual[int(off[i]):int(off[i + 1])])
                f.write(b"@read%d/%s extra\n%s\n+\n%s\n" % (i, tag, s, q))
    f1, f2 = tmp_path / "r1.fq.gz", tmp_path / "r2.fq"
    write_fq(f1, b["seq1"], b["off1"], b["qual1"], b"1", gz=True)
    write_fq(f2, b["seq2"], b["off2"], b["qual2"], b"2")
    for opts in (["-k", "17", "-c", "0.5"], ["-k", "21", "-q", "20", "-s"], ["-k", "13", "-q", "10", "-c", "0.3"]):
        for paired in (True, False):
            files = ["-r", str(fa), "-1", str(f1)] + (["-2", str(f2)] if paired else [])
            ho1, ho2, oo1, oo2 = (tmp_path / n for n in ("h1.fq", "h2.fq", "o1.fq", "o2.fq"))
            r = _run_shark(files + opts + ["-o", str(ho1), "-p", str(ho2), "--batch", "1500"], str(tmp_path))
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            ossv = tmp_path / "o.ssv"
            oracle.run_cli(files + opts + ["-o", str(oo1), "-p", str(oo2)], str(ossv))
            assert r.stdout == ossv.read_bytes(), (opts, paired)
            assert len(r.stdout) > 1000
            assert ho1.read_bytes() == oo1.read_bytes()
            if paired:
                assert ho2.read_bytes() == oo2.read_bytes()


def test_non_ascii_and_control_bytes_are_invalid(oracle):
    """bytes outside ACGTacgt -- including >= 128, NUL and '@' -- break k-mers exactly like 'N' (to_int == 0)"""
    rng = np.random.default_rng(31)
    genes = synth.make_genes(rng, 6, 400, 900)
    o, h, _ = _build_both(oracle, genes, k=15, bf_bits=1 << 22)
    b = synth.make_reads(rng, genes, 1500, read_len=110, paired=True, on_target=0.9, sub_rate=0.0, n_rate=0.0)
    for key in ("seq1", "seq2"):
        a = b[key]
        idx = rng.integers(0, len(a), size=len(a) // 40)
        a[idx] = rng.choice(np.array([0, 1, 10, 13, 32, 64, 91, 96, 123, 127, 128, 193, 225, 255], dtype=np.uint8), size=len(idx))
    goff, _ = _compare_classify(o, h, b)
    assert 200 < goff[-1] < 1500


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


@pytest.mark.parametrize("launcher,scaling,ranks", [("self", "strong", 2), ("torchrun", "strong", 2), ("self", "weak", 2), ("self", "strong", 4)])
def test_bench_ranks_dry_run(tmp_path, launcher, scaling, ranks):
    """the N>1 code path of bench.py (rank env, shards of one read set, passes per step, warm-up collective, barrier, MAX over
    ranks, count all-reduce, the per-rank split gathered over the job's channel, the other ranks' wait for rank 0) with `ranks`
    ranks sharing the one GPU of the test box over gloo; the real multi-GPU run uses RCCL inside the library.
    launcher "self": exactly the driver's form, `python3 bench.py --gpus N ...` -- bench.py starts its ranks itself
    (main.cpp:219-223 starts the reference's workers from the one command line); "torchrun": the contract's launcher form.
    Four ranks is what one card takes next to this process (the box allows six processes on it); the split for 8 is checked on
    the CPU (tests/test_dist_gloo.py)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SHARK_DIST_BACKEND="gloo")
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    reps = 2
    base = [os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--pairs", "250000", "--total-pairs", "1000000", "--reps-per-step", str(reps),
            "--scaling", scaling, "--no-configs", "--no-boundary", "--no-cpu-baseline", "--no-cli", "--no-live-counters"]
    if launcher == "self":
        cmd = ["python3"] + base + ["--gpus", str(ranks)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + base + ["--gpus", str(ranks)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out_lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(out_lines) == 1, r.stdout[-2000:]          # ONE JSON line, rank 0's
    j = json.loads(out_lines[0])
    assert j["n_gpus"] == ranks and j["ranks_seen"] == ranks and j["scaling"] == scaling and j["cpu_baseline"] is None
    assert j["config"]["gene_count_checksum"] == j["config"]["assoc_per_step"] > 0
    # every rank's split of the timed window, gathered over the job's own channel
    pr = j["per_rank"]
    assert [x["rank"] for x in pr] == list(range(ranks))
    for x in pr:
        assert x["launches"] == 2 * j["config"]["launches_per_step_per_gpu"] and x["kernel_ms"] > 0 and x["wall_ms"] >= x["kernel_ms"] * 0.5
        assert x["allreduce_ms"] >= 0 and x["barrier_wait_ms"] >= 0
        assert x["wall_ms"] + x["allreduce_ms"] + x["barrier_wait_ms"] <= j["ms_per_step"] * 2 * 1.001 + 1.0   # inside the (max over ranks) window
    # the roofline never claims more than the resource it names has (without counters: the compulsory HBM bytes)
    assert 0 < j["roofline"]["frac"] <= 1 and j["roofline"]["bound"] in ("hbm", "valu-issue", "memory-side-request-rate")
    assert j["roofline"]["hbm_compulsory"]["frac_of_hbm_peak"] <= 1
    if scaling == "strong":
        # the same read set on one GPU: same reads per step, same associations
        r1 = subprocess.run([sys.executable] + base + ["--gpus", "1"], capture_output=True, text=True, env=env, cwd=root, timeout=600)
        assert r1.returncode == 0, r1.stderr[-3000:]
        j1 = json.loads([x for x in r1.stdout.splitlines() if x.startswith("{")][-1])
        assert j1["n_gpus"] == 1 and j1["ranks_seen"] == 1 and len(j1["per_rank"]) == 1
        assert j["config"]["reads_per_step"] == j1["config"]["reads_per_step"] == 2 * 1000000 * reps
        assert j["config"]["assoc_per_step"] == j1["config"]["assoc_per_step"]
        assert j["config"]["gene_count_checksum"] == j1["config"]["gene_count_checksum"]
        assert j1["config"]["launches_per_step_per_gpu"] == ranks * j["config"]["launches_per_step_per_gpu"]
    else:
        assert j["config"]["reads_per_step"] == ranks * 2 * 250000 * reps


def test_bench_refuses_a_rank_count_that_is_not_gpus():
    """`n_gpus` on the line is what --gpus asked for or the run fails: a launcher with another number of ranks, or more
    RCCL ranks than the node has GPUs, is an error (round 2 printed a warning and measured one GPU)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr and not r.stdout.strip()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SHARK_DIST_BACKEND")}
    n_dev = torch.cuda.device_count()
    r = subprocess.run(["python3", os.path.join(root, "bench.py"), "--gpus", str(n_dev + 1), "--steps", "1", "--no-configs", "--no-boundary",
                        "--no-cpu-baseline", "--no-cli", "--no-live-counters"], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert r.returncode != 0 and "GPU(s)" in r.stderr and not r.stdout.strip()


def test_cli_block_reader_handover_on_irregular_records(oracle, tmp_path):
    """the block-parallel FASTQ reader must deliver exactly what kseq's rules deliver: strict records fast,
    then a multi-line record, CR/LF records, junk between records and a last record without newline"""
    rng = np.random.default_rng(77)
    genes = synth.make_genes(rng, 5, 500, 1500)
    fa = tmp_path / "g.fa"
    fa.write_text("".join(">g%d\n%s\n" % (i, bytes(g).decode()) for i, g in enumerate(genes)))

    def rec(i, g, L, style):
        st = int(rng.integers(0, len(g) - L))
        s = bytes(g[st:st + L]).decode()
        q = "".join(chr(int(x)) for x in rng.integers(35, 74, size=L))
        if style == "strict":
            return "@r%d desc\n%s\n+\n%s\n" % (i, s, q)
        if style == "multi":
            return "@r%d\n%s\n%s\n+r%d\n%s\n%s\n" % (i, s[:40], s[40:], i, q[:40], q[40:])
        if style == "crlf":
            return "@r%d\r\n%s\r\n+\r\n%s\r\n" % (i, s, q)
        if style == "junk":
            return "junk line\n\n@r%d\n%s\n+\n%s\n" % (i, s, q)
        return "@r%d\n%s\n+\n%s" % (i, s, q)   # no trailing newline

    styles = ["strict"] * 1300 + ["multi"] + ["strict"] * 50 + ["crlf"] * 3 + ["junk"] + ["strict"] * 20 + ["last"]
    t1 = "".join(rec(i, genes[i % 5], 100, st) for i, st in enumerate(styles))
    t2 = "".join(rec(i, genes[i % 5], 90, "strict" if st in ("multi", "junk", "last") else st) for i, st in enumerate(styles))
    f1, f2 = tmp_path / "a.fq", tmp_path / "b.fq"
    f1.write_text(t1, newline="")
    f2.write_text(t2, newline="")
    args = ["-r", str(fa), "-1", str(f1), "-2", str(f2), "-k", "15"]
    ossv = tmp_path / "o.ssv"
    oracle.run_cli(args + ["-o", str(tmp_path / "o1.fq"), "-p", str(tmp_path / "o2.fq")], str(ossv))
    assert ossv.read_bytes().count(b"\n") > 1000
    for env_serial in (False, True):
        for batch in ("333", "1000000"):
            env = dict(os.environ)
            if env_serial:
                env["SHARK_SERIAL_READER"] = "1"
            import subprocess
            exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shark_amd", "bin", "shark")
            r = subprocess.run([exe] + args + ["-o", str(tmp_path / "h1.fq"), "-p", str(tmp_path / "h2.fq"), "--batch", batch],
                               capture_output=True, env=env, cwd=str(tmp_path))
            assert r.returncode == 0, r.stderr.decode()[-1500:]
            assert r.stdout == ossv.read_bytes(), (env_serial, batch)
            assert (tmp_path / "h1.fq").read_bytes() == (tmp_path / "o1.fq").read_bytes()
            assert (tmp_path / "h2.fq").read_bytes() == (tmp_path / "o2.fq").read_bytes()


def test_cli_irregular_records_that_keep_the_four_line_alignment(oracle, tmp_path):
    """irregular records that do NOT shift the four-line rhythm -- an empty read (a trimmer's), a lone CR, a NUL, a
    sequence/quality length mismatch (which ends kseq's stream) -- in the middle of a file read by many parallel readers
    with small batches: every batch behind such a record still validates, so a reader that is ahead must not hand its batch
    to a GPU before the batches in front of it are known to be strict.  ssv, both FASTQ outputs and the per-gene counts must
    be the serial kseq-rule reader's (the oracle CLI's), run after run."""
    import subprocess
    rng = np.random.default_rng(77)
    genes = synth.make_genes(rng, 4, 600, 1200)
    fa = tmp_path / "g.fa"
    fa.write_text("".join(">g%d\n%s\n" % (i, bytes(g).decode()) for i, g in enumerate(genes)))

    def rec(i, g, L, style):
        st = int(rng.integers(0, len(g) - L))
        s = bytes(g[st:st + L]).decode()
        q = "".join(chr(int(x)) for x in rng.integers(35, 74, size=L))
        if style == "empty":
            return "@r%d\n\n+\n\n" % i
        if style == "cr":
            return "@r%d\n%s\r\n+\n%s\r\n" % (i, s, q)
        if style == "nul":
            return "@r%d\n%s\x00%s\n+\n%s\n" % (i, s[:30], s[31:], q)
        if style == "mismatch":
            return "@r%d\n%s\n+\n%s\n" % (i, s, q[:-7])
        return "@r%d\n%s\n+\n%s\n" % (i, s, q)

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shark_amd", "bin", "shark")
    for styles in (["s"] * 700 + ["empty"] + ["s"] * 1500,
                   ["s"] * 901 + ["cr"] + ["s"] * 1300,
                   ["s"] * 650 + ["nul"] + ["s"] * 1400,
                   ["s"] * 1000 + ["mismatch"] + ["s"] * 1200,
                   ["s"] * 300 + ["empty"] + ["s"] * 400 + ["nul"] + ["s"] * 500 + ["mismatch"] + ["s"] * 900):
        t1 = "".join(rec(i, genes[i % 4], 100, st) for i, st in enumerate(styles))
        t2 = "".join(rec(i, genes[i % 4], 100, "s") for i, st in enumerate(styles))
        f1, f2 = tmp_path / "a.fq", tmp_path / "b.fq"
        f1.write_bytes(t1.encode("latin-1"))
        f2.write_bytes(t2.encode("latin-1"))
        args = ["-r", str(fa), "-1", str(f1), "-2", str(f2), "-k", "15"]
        ossv = tmp_path / "o.ssv"
        oracle.run_cli(args + ["-o", str(tmp_path / "o1.fq"), "-p", str(tmp_path / "o2.fq")], str(ossv))
        want = ossv.read_bytes()
        assert want.count(b"\n") > 500
        want_counts = {}
        for line in want.splitlines():
            g = line.split()[1]
            want_counts[g] = want_counts.get(g, 0) + 1
        for rep in range(3):
            r = subprocess.run([exe] + args + ["-o", str(tmp_path / "h1.fq"), "-p", str(tmp_path / "h2.fq"), "--batch", "64", "-t", "8",
                                               "--gene-counts", str(tmp_path / "gc.txt")], capture_output=True, cwd=str(tmp_path), timeout=300)
            assert r.returncode == 0, r.stderr.decode()[-1500:]
            assert r.stdout == want, (styles.index([x for x in styles if x != "s"][0]), rep)
            assert (tmp_path / "h1.fq").read_bytes() == (tmp_path / "o1.fq").read_bytes()
            assert (tmp_path / "h2.fq").read_bytes() == (tmp_path / "o2.fq").read_bytes()
            got_counts = {ln.split()[0].encode(): int(ln.split()[1]) for ln in (tmp_path / "gc.txt").read_text().splitlines()}
            assert got_counts == want_counts


def test_cli_fixed_width_guess_with_compensating_records(oracle, tmp_path):
    """a file that passes the fixed-width shortcut's checks (size a multiple of the first record's length, sampled record
    starts in place) although one batch holds two half-length records: the surplus record must not be dropped"""
    import subprocess
    rng = np.random.default_rng(78)
    genes = synth.make_genes(rng, 3, 600, 1200)
    fa = tmp_path / "g.fa"
    fa.write_text("".join(">g%d\n%s\n" % (i, bytes(g).decode()) for i, g in enumerate(genes)))
    L = 100

    def rec(name, g, n):
        st = int(rng.integers(0, len(g) - n))
        return "@%s\n%s\n+\n%s\n" % (name, bytes(g[st:st + n]).decode(), "I" * n)

    full = len(rec("r0000", genes[0], L))                       # 11 + 2 L bytes
    recs = [rec("r%04d" % i, genes[i % 3], L) for i in range(5000)]
    # record 1234 is replaced by two records that together have its byte length: (11 + 2 h) + (12 + 2 h) = 11 + 2 L  ->  h = (L - 6) / 2
    h = (L - 6) // 2
    two = rec("x0001", genes[1], h) + rec("y00002", genes[2], h)
    assert len(two) == full
    recs[1234] = two
    f1 = tmp_path / "a.fq"
    f1.write_text("".join(recs))
    args = ["-r", str(fa), "-1", str(f1), "-k", "15"]
    ossv = tmp_path / "o.ssv"
    oracle.run_cli(args + ["-o", str(tmp_path / "o1.fq")], str(ossv))
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shark_amd", "bin", "shark")
    r = subprocess.run([exe] + args + ["-o", str(tmp_path / "h1.fq"), "--batch", "100", "-t", "4"], capture_output=True, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert r.stdout == ossv.read_bytes()
    assert (tmp_path / "h1.fq").read_bytes() == (tmp_path / "o1.fq").read_bytes()


def test_full_size_properties_config2():
    """BASELINE configs[1] at full size (10 M pairs 2x150 bp, 1 gene, k=17, 2^33-bit filter) through
    size-independent properties of the domain:
      * mate swap and read reverse-complement leave every read's gene set unchanged (canonical k-mers; the
        per-gene coverage is a union of k-mer intervals, which mirrors under reverse complement)
      * results do not depend on how the batch is split or ordered
      * the per-gene counters equal a histogram of the per-read results; two runs are identical
      * every error-free on-target pair is assigned, no uniform-random pair is (filter density 2e-6)"""
    from shark_amd import synth as dsynth
    from shark_amd.capi import hip_memcpy_dtoh
    n, L = 10_000_000, 150
    dev = torch.device("cuda:0")
    genes = dsynth.make_reference(1, 20000)
    h = _hip(k=17, c=0.6, bf_bits=1 << 33)
    h.build([g.tobytes() for g in genes])
    b = dsynth.make_pairs_device(n, genes, dev, seed=99, sub_rate=0.0, n_rate=0.0)
    torch.cuda.synchronize()

    def run(seq1, off1, seq2, off2, m=n):
        r = h.classify_device(m, seq1.data_ptr(), off1.data_ptr(), seq2.data_ptr(), off2.data_ptr(), max_read_len=L)
        goff = np.empty(m + 1, np.uint32)
        hip_memcpy_dtoh(goff, r.gene_off, goff.nbytes)
        gids = np.empty(int(r.n_assoc), np.uint16)
        if len(gids):
            hip_memcpy_dtoh(gids, r.gene_ids, gids.nbytes)
        return goff, gids

    h.gene_counts_reset()
    goff, gids = run(b["seq1"], b["off1"], b["seq2"], b["off2"])
    cnt = np.diff(goff.astype(np.int64))
    assert int(h.gene_counts(4)[0]) == int((gids == 0).sum()) == int(cnt.sum())
    assert cnt.max() == 1 and 0.49 * n < cnt.sum() < 0.51 * n      # half the pairs come from the gene, error free
    # determinism
    g2, i2 = run(b["seq1"], b["off1"], b["seq2"], b["off2"])
    assert np.array_equal(goff, g2) and np.array_equal(gids, i2)
    # mate swap
    g3, _ = run(b["seq2"], b["off2"], b["seq1"], b["off1"])
    assert np.array_equal(goff, g3)
    # reverse complement of both mates (fixed length: reshape, flip, complement)
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a_, c_ in zip(b"ACGTN", b"TGCAN"):
        comp[a_] = c_
    rc1 = comp[b["seq1"].view(n, L).flip(1).to(torch.int64)].contiguous().view(-1)
    rc2 = comp[b["seq2"].view(n, L).flip(1).to(torch.int64)].contiguous().view(-1)
    torch.cuda.synchronize()
    g4, _ = run(rc1, b["off1"], rc2, b["off2"])
    assert np.array_equal(goff, g4)
    # order independence: reverse the order of the reads
    r1 = b["seq1"].view(n, L).flip(0).contiguous().view(-1)
    r2 = b["seq2"].view(n, L).flip(0).contiguous().view(-1)
    torch.cuda.synchronize()
    g5, _ = run(r1, b["off1"], r2, b["off2"])
    assert np.array_equal(np.diff(g5.astype(np.int64)), cnt[::-1])
    # split independence: first 3 333 333 pairs alone
    m = 3_333_333
    g6, _ = run(b["seq1"], b["off1"], b["seq2"], b["off2"], m=m)
    assert np.array_equal(g6, goff[:m + 1])


def test_full_size_config2_equals_the_oracle(oracle):
    """BASELINE configs[1] at full size -- the bench's own launch: 10 M pairs 2x150 bp (50 % on-target, 1 % substitutions, 0.2 % N),
    1 gene x 20 kb, k=17, c=0.6, 2^33-bit filter, resident in HBM -- equals the oracle on EVERY pair, offsets and gene ids (the
    oracle on all host threads; bench.py makes the same comparison in its cpu_baseline leg).  The launch takes the exact LDS table
    with the sparse first round, the bound cut and the early decision: each shortcut's own test uses batches of a few hundred
    reads, this one is the 10 M-pair equality."""
    from shark_amd import synth as dsynth
    from shark_amd.capi import hip_memcpy_dtoh
    n, L = 10_000_000, 150
    dev = torch.device("cuda:0")
    genes = dsynth.make_reference(1, 20000)
    h = _hip(k=17, c=0.6, bf_bits=1 << 33)
    h.build([g.tobytes() for g in genes])
    b = dsynth.make_pairs_device(n, genes, dev, seed=dsynth.SEED + 1, read_len=L, on_target=0.5)
    torch.cuda.synchronize()
    r = h.classify_device(n, b["seq1"].data_ptr(), b["off1"].data_ptr(), b["seq2"].data_ptr(), b["off2"].data_ptr(), max_read_len=L)
    assert "classify_uni_kernel<5, " in h.last_kernel() and ", 21, " in h.last_kernel() and "+sparse-first-round" in h.last_kernel(), h.last_kernel()
    goff = np.empty(n + 1, np.uint32)
    hip_memcpy_dtoh(goff, r.gene_off, goff.nbytes)
    gids = np.empty(int(r.n_assoc), np.uint16)
    hip_memcpy_dtoh(gids, r.gene_ids, gids.nbytes)
    hb = dsynth.to_host_sample(b, n, L)
    o = oracle.Shark(k=17, c=0.6, bf_bits=1 << 33)
    o.build([g.tobytes() for g in genes])
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    og, oi = o.classify(hb["seq1"], hb["off1"], hb["seq2"], hb["off2"], nthreads=threads)
    o.close()
    assert np.array_equal(og, goff), "gene_off differs at read %d" % int(np.argmax(og != goff))
    assert np.array_equal(oi, gids)
    assert 0.45 * n < int(goff[-1]) < 0.55 * n
    h.close()


@pytest.mark.parametrize("n_genes", [40, 3000, 6000])
def test_gene_counts_equal_the_histogram_of_the_gene_lists(n_genes):
    """per-gene assigned-read counters (gene_hist_kernel): the histogram of the batch's gene ids, whatever the number of genes --
    few (every workgroup's LDS table holds them all), and thousands (runs of equal genes combined in the wave, added to the counters
    directly); reads drawn from consecutive genes so that neighbouring reads often share a gene; two batches accumulate."""
    from shark_amd.capi import hip_memcpy_dtoh
    from shark_amd import synth as dsynth
    genes = dsynth.make_gencode_like_reference(n_genes)
    h = _hip(k=17, c=0.6, bf_bits=1 << 31)
    h.build([g.tobytes() for g in genes])
    dev = torch.device("cuda:0")
    n = 300_000
    want = np.zeros(65536, np.uint64)
    for seed in (5, 6):
        b = dsynth.make_pairs_device(n, genes[:64] if seed == 6 else genes, dev, seed=seed, on_target=0.9)
        r = h.classify_device(n, b["seq1"].data_ptr(), b["off1"].data_ptr(), b["seq2"].data_ptr(), b["off2"].data_ptr(), 0, 0, max_read_len=150)
        gids = np.empty(int(r.n_assoc), np.uint16)
        hip_memcpy_dtoh(gids, r.gene_ids, gids.nbytes)
        assert len(gids) > n // 2
        want += np.bincount(gids, minlength=65536).astype(np.uint64)
        got = h.gene_counts(65536)
        bad = np.flatnonzero(want != got)
        assert len(bad) == 0, (n_genes, seed, len(bad), int(bad[0]), int(want[bad[0]]), int(got[bad[0]]))
    h.close()


def test_gene_counts_allreduce_over_rccl(oracle, tmp_path, monkeypatch):
    """the sharded run's one exchange step: RCCL all-reduce of the per-gene counters (forced on with one GPU),
    through the ABI and through `shark --gene-counts`"""
    rng = np.random.default_rng(3)
    genes = synth.make_genes(rng, 9, 400, 1200)
    o, h, _ = _build_both(oracle, genes, k=17, bf_bits=1 << 22)
    b = synth.make_reads(rng, genes, 3000, read_len=100, paired=True, on_target=0.8)
    goff, gids = _compare_classify(o, h, b)
    want = np.bincount(gids, minlength=16)[:16].astype(np.uint64)
    monkeypatch.setenv("SHK_FORCE_RCCL", "1")
    assert np.array_equal(h.gene_counts_allreduce(n=16), want)
    assert np.array_equal(h.gene_counts(16), want)          # in place on the device, one rank: unchanged
    # CLI
    fa = tmp_path / "g.fa"
    fa.write_text("".join(">g%d\n%s\n" % (i, bytes(g).decode()) for i, g in enumerate(genes)))
    f1, f2 = tmp_path / "a.fq", tmp_path / "b.fq"
    for f, seq, off, tag in ((f1, b["seq1"], b["off1"], 1), (f2, b["seq2"], b["off2"], 2)):
        with open(f, "wb") as fh:
            for i in range(len(off) - 1):
                s = bytes(seq[int(off[i]):int(off[i + 1])])
                fh.write(b"@r%d/%d\n%s\n+\n%s\n" % (i, tag, s, b"I" * len(s)))
    gc = tmp_path / "counts.txt"
    r = _run_shark(["-r", str(fa), "-1", str(f1), "-2", str(f2), "-o", str(tmp_path / "o1"), "-p", str(tmp_path / "o2"),
                    "--gene-counts", str(gc), "-v"], str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    got = dict(line.split() for line in gc.read_text().splitlines())
    assert {k: int(v) for k, v in got.items()} == {"g%d" % g: int(c) for g, c in enumerate(want) if c}
    assert b"[shark/counts] %d associations" % int(want.sum()) in r.stderr
    # stdout stays pure ssv even if RCCL prints a banner while it initialises
    lines = r.stdout.splitlines()
    assert len(lines) == int(want.sum()) and all(len(x.split()) == 2 and x.startswith(b"r") for x in lines)


@pytest.mark.parametrize("read_len,paired,q", [(60, True, 0), (100, True, 20), (125, True, 0), (170, True, 0), (200, True, 30),
                                               (250, True, 0), (300, False, 0), (520, False, 20), (64, False, 0),
                                               (300, True, 0), (300, True, 20), (310, True, 0), (600, False, 0)])
def test_every_kernel_specialisation(oracle, read_len, paired, q):
    """fixed-length batches that select each unroll U in {2,3,4,5,6,8,10} of the fast kernel (paired and single-end,
    with and without the quality mask); lengths just above a specialisation's capacity go to the general kernel"""
    rng = np.random.default_rng(1000 + read_len)
    genes = synth.make_genes(rng, 12, 800, 3000, share_every=4)
    o, h, _ = _build_both(oracle, genes, k=17, bf_bits=1 << 24, min_quality=q)
    b = synth.make_reads(rng, genes, 2500, read_len=read_len, paired=paired, on_target=0.7, qual=q > 0, n_rate=0.005)
    goff, _ = _compare_classify(o, h, b)
    assert goff[-1] > 20      # (a strict quality mask leaves few reads above the confidence threshold)
    # one read far beyond the specialisation rides along: it must take the general kernel, the rest the fast one
    long_read = bytes(genes[0][:700])
    m1 = [bytes(b["seq1"][int(b["off1"][i]):int(b["off1"][i + 1])]) for i in range(300)] + [long_read]
    q1 = None if q == 0 else [bytes(b["qual1"][int(b["off1"][i]):int(b["off1"][i + 1])]) for i in range(300)] + [b"I" * 700]
    if paired:
        m2 = [bytes(b["seq2"][int(b["off2"][i]):int(b["off2"][i + 1])]) for i in range(300)] + [long_read[::-1]]
        q2 = None if q == 0 else [bytes(b["qual2"][int(b["off2"][i]):int(b["off2"][i + 1])]) for i in range(300)] + [b"I" * 700]
        bb = synth.batch_from_lists(m1, m2, q1, q2)
    else:
        bb = synth.batch_from_lists(m1, None, q1, None)
    _compare_classify(o, h, bb)
    assert h.timing()["last_n_long"] == 1


@pytest.mark.parametrize("env,n_genes,bf_bits", [({}, 1, 1 << 33), ({"SHK_NO_LDS_TABLE": "1"}, 1, 1 << 33), ({}, 12, 1 << 26),
                                                 ({"SHK_NO_LDS_SUMMARY": "1"}, 12, 1 << 26), ({"SHK_NO_LDS_SUMMARY": "1", "SHK_NO_SUMMARY": "1"}, 12, 1 << 26),
                                                 ({"SHK_PROBE": "bitvector"}, 12, 1 << 26), ({"SHK_NO_ANCHOR": "1", "SHK_NO_LDS_SUMMARY": "1"}, 12, 5 << 24)])
def test_long_pairs_2x300(oracle, monkeypatch, env, n_genes, bf_bits):
    """2 x 300 bp (MiSeq): 588 slots and 76 staging groups per pair -- the U = 10 specialisation of classify_uni_kernel with two
    staging groups per lane (round 2 sent such pairs to the general kernel), on every kind of index: the exact table in LDS, the
    LDS summary, the table modes with the anchored extension, plain filter words; uniform and trimmed batches, host and
    device-resident entry points, -q; pairs just inside and just outside the specialisation"""
    for name, v in env.items():
        monkeypatch.setenv(name, v)
    rng = np.random.default_rng(300 + n_genes)
    genes = synth.make_genes(rng, n_genes, 20_000 if n_genes == 1 else 900, 20_000 if n_genes == 1 else 3500, share_every=3)
    for q in (0, 20):
        o, h, _ = _build_both(oracle, genes, k=17, bf_bits=bf_bits, min_quality=q)
        for L, ragged in ((300, False), (300, True), (312, False), (328, False), (329, False), (600, False)):
            paired = L != 600
            if ragged:
                b = _sequenced_pairs(rng, genes, 500, L, L, True, q > 0, 0.01, 0.002, 0.002)
            else:
                b = synth.make_reads(rng, genes, 500, read_len=L, paired=paired, on_target=0.7, qual=q > 0, n_rate=0.003)
            goff, _ = _compare_classify(o, h, b)
            assert goff[-1] > 0
            if L == 329:
                assert h.timing()["last_n_long"] == 500       # 2 x 329: 336 + 313 = 649 slots, beyond the largest specialisation (2 x 328 fills its 640 exactly)
            elif not ragged:
                assert h.timing()["last_n_long"] == 0
            # the same batch resident in HBM (uniformity decided on the device)
            dev = torch.device("cuda:0")
            t = {kk: (torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) if v is not None else None) for kk, v in b.items()}
            pt = {kk: (v.data_ptr() if v is not None else 0) for kk, v in t.items()}
            n = len(b["off1"]) - 1
            r = h.classify_device(n, pt["seq1"], pt["off1"], pt["seq2"], pt["off2"], pt["qual1"], pt["qual2"], max_read_len=L)
            from shark_amd.capi import hip_memcpy_dtoh
            dg = np.empty(n + 1, np.uint32)
            hip_memcpy_dtoh(dg, r.gene_off, dg.nbytes)
            assert np.array_equal(dg, goff)
        h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [17, 31, 4])
def test_packed_position_edges(oracle, k):
    """the fast kernel addresses k-mers by packed position (mate 2 starts at L1 rounded up to 8): every
    alignment of L1, mates shorter than k, an empty mate, and pairs that fill a specialisation's capacity
    exactly / exceed it by one (general kernel), all in ONE batch so that waves see stale staging data
    of longer reads behind shorter ones"""
    rng = np.random.default_rng(77 + k)
    genes = synth.make_genes(rng, 6, 1200, 2500, share_every=3)
    o, h, _ = _build_both(oracle, genes, k=k, bf_bits=1 << 24)
    shapes = [(L1, L2) for L1 in range(140, 153) for L2 in (150, 151)]                       # every L1 mod 8
    shapes += [(150, L2) for L2 in (0, 1, k - 1, k, k + 1)] + [(L1, 150) for L1 in (0, 1, k - 1, k, k + 1)]
    shapes += [(k - 1, k - 1), (0, 0), (k, 0), (0, k)]
    for U in (2, 3, 4, 5, 6, 8):                                                              # ns = round8(L1) + L2-k+1 around 64 U
        L1 = 8 * (4 * U - 1)                                                                  # multiple of 8, about half the capacity
        for d in (-1, 0, 1):
            shapes.append((L1, 64 * U - L1 + k - 1 + d))
    m1, m2 = [], []
    for i, (L1, L2) in enumerate(shapes * 3):
        g = genes[i % len(genes)]
        st = int(rng.integers(0, len(g) - 600))
        a = g[st:st + L1].copy()
        b = synth.revcomp(g[st:st + 600])[:L2].copy()
        if i % 5 == 0 and L2 > 3:
            b[int(rng.integers(0, L2))] = ord("N")
        m1.append(a.tobytes())
        m2.append(b.tobytes())
    order = rng.permutation(len(m1))
    batch = synth.batch_from_lists([m1[i] for i in order], [m2[i] for i in order])
    goff, _ = _compare_classify(o, h, batch)
    assert goff[-1] > len(m1) // 3


@pytest.mark.gpu
@pytest.mark.parametrize("bf_bits", [3 << 32, 5 << 32, 7 << 33])
def test_cli_filter_size_not_a_power_of_two(oracle, bf_bits):
    """`-b 3` = 3 * 2^33 bits: position = hash % size through the direct 32-bit remainder
    (h = q 2^s + r, pos = (q % m) 2^s + r); sizes from 1.5 GiB (oracle memory) to `-b 7`"""
    rng = np.random.default_rng(4242)
    genes = synth.make_genes(rng, 30, 300, 2000, share_every=3)
    o, h, info = _build_both(oracle, genes, k=17, bf_bits=bf_bits)
    assert h.probe_mode() in ("table-mod", "lds-summary+table-mod")
    _compare_index(o, h, info)
    batch = synth.make_reads(rng, genes, 4000, read_len=150, paired=True, on_target=0.5, n_rate=0.005)
    goff, _ = _compare_classify(o, h, batch)
    assert goff[-1] > 1000


# ---------------------------------------------------------------------------
# the pipelined boundary (shk_classify_submit / shk_classify_wait) and the paths without a host round trip
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("trimmed", [False, True])
def test_pipelined_submit_wait_equals_oracle(oracle, monkeypatch, trimmed):
    """PIPE_DEPTH batches in flight; every batch equals the oracle; tickets are waited in order; a fourth
    outstanding submit is refused; gene counters equal the histogram over all batches.  `trimmed`: most batches of mixed read
    lengths, and every one of them sorted by length and classified class by class (SHK_CLS_MIN_FILL=1) -- three of those in
    flight, each with its own lists, beside uniform batches"""
    from shark_amd import SharkHipError
    from shark_amd.capi import SHK_PIPE_DEPTH
    if trimmed:
        monkeypatch.setenv("SHK_CLS_MIN_FILL", "1")
    rng = np.random.default_rng(31)
    genes = synth.make_genes(rng, 20, 300, 1500, share_every=3)
    o, h, _ = _build_both(oracle, genes, k=17, bf_bits=1 << 24)
    batches = [synth.make_reads(rng, genes, 700 + 200 * i, read_len=150 if i % 2 == 0 else 100, paired=True, on_target=0.7,
                                var_len=(i == 3) or (trimmed and i != 4)) for i in range(7)]
    batches.append(synth.batch_from_lists([], []))                     # an empty batch in the middle of a stream
    batches.append(synth.make_reads(rng, genes, 300, read_len=80, paired=False, on_target=0.7))
    want = [o.classify(b["seq1"], b["off1"], b["seq2"], b["off2"]) if len(b["off1"]) > 1 else (np.zeros(1, np.uint32), np.zeros(0, np.uint16))
            for b in batches]
    h.gene_counts_reset()
    tickets, got = [], []
    for i, b in enumerate(batches):
        if len(tickets) == SHK_PIPE_DEPTH:
            with pytest.raises(SharkHipError, match="not allowed"):
                h.submit(b["seq1"], b["off1"], b["seq2"], b["off2"])
            got.append(h.wait(tickets.pop(0)))
        tickets.append(h.submit(b["seq1"], b["off1"], b["seq2"], b["off2"]))
    while tickets:
        got.append(h.wait(tickets.pop(0)))
    for (wg, wi), (gg, gi) in zip(want, got):
        assert np.array_equal(wg, gg) and np.array_equal(wi, gi)
    allids = np.concatenate([w[1] for w in want])
    assert np.array_equal(h.gene_counts(32), np.bincount(allids, minlength=32)[:32].astype(np.uint64))
    if trimmed and h.probe_mode() == "lds-table":
        assert "verdict=" in h.last_kernel(), h.last_kernel()


def test_more_associations_than_reserved(oracle):
    """every read ties over 6 genes: 6n associations exceed the 2n+4096 the slot reserves, so the batch is finished by
    the overflow path (grow, redo the tail) -- and is still counted exactly once"""
    rng = np.random.default_rng(37)
    core = synth.random_seq(rng, 1200)
    genes = [core.copy() for _ in range(6)]
    o, h, _ = _build_both(oracle, genes, k=17, bf_bits=1 << 24)
    b = synth.make_reads(rng, [core], 6000, read_len=100, paired=True, on_target=1.0, sub_rate=0.0, n_rate=0.0)
    h.gene_counts_reset()
    goff, gids = _compare_classify(o, h, b)
    assert int(goff[-1]) == 6 * 6000 > 2 * 6000 + 4096
    assert np.array_equal(h.gene_counts(8), np.array([6000] * 6 + [0, 0], dtype=np.uint64))
    goff2, gids2 = _compare_classify(o, h, b)                          # now the buffer is large enough: fast path
    assert np.array_equal(h.gene_counts(8), np.array([12000] * 6 + [0, 0], dtype=np.uint64))


def test_device_api_with_a_wrong_length_bound(oracle):
    """shk_classify_device trusts max_read_len only for the choice of kernel: reads longer than the bound are found
    after the fact and finished by the general kernel; per-gene counts are still exact"""
    from shark_amd.capi import hip_memcpy_dtoh
    rng = np.random.default_rng(41)
    genes = synth.make_genes(rng, 6, 3000, 6000)
    o, h, _ = _build_both(oracle, genes, k=17, bf_bits=1 << 24)
    b1 = synth.make_reads(rng, genes, 300, read_len=100, paired=True, on_target=0.8)
    b2 = synth.make_reads(rng, genes, 100, read_len=900, paired=True, on_target=0.8)   # 900+884 slots > 512
    m1 = [bytes(b1["seq1"][int(b1["off1"][i]):int(b1["off1"][i + 1])]) for i in range(300)] + \
         [bytes(b2["seq1"][int(b2["off1"][i]):int(b2["off1"][i + 1])]) for i in range(100)]
    m2 = [bytes(b1["seq2"][int(b1["off2"][i]):int(b1["off2"][i + 1])]) for i in range(300)] + \
         [bytes(b2["seq2"][int(b2["off2"][i]):int(b2["off2"][i + 1])]) for i in range(100)]
    order = rng.permutation(400)
    b = synth.batch_from_lists([m1[i] for i in order], [m2[i] for i in order])
    og, oi = o.classify(b["seq1"], b["off1"], b["seq2"], b["off2"])
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) for k, v in b.items() if v is not None}
    for bound in (100, 0, 900):
        h.gene_counts_reset()
        r = h.classify_device(400, t["seq1"].data_ptr(), t["off1"].data_ptr(), t["seq2"].data_ptr(), t["off2"].data_ptr(), max_read_len=bound)
        goff = np.empty(401, np.uint32)
        hip_memcpy_dtoh(goff, r.gene_off, goff.nbytes)
        gids = np.empty(int(r.n_assoc), np.uint16)
        hip_memcpy_dtoh(gids, r.gene_ids, gids.nbytes)
        assert np.array_equal(goff, og) and np.array_equal(gids, oi), bound
        assert h.timing()["last_n_long"] == 100
        assert np.array_equal(h.gene_counts(8), np.bincount(oi, minlength=8)[:8].astype(np.uint64)), bound


def test_device_resident_pipeline_submit_wait(oracle):
    """shk_classify_device_submit / shk_classify_wait: the device-resident entry point without its host synchronisation -- three
    batches in flight on the context's slots, results as DEVICE pointers, equal to the oracle's for uniform batches whose
    lengths the caller vouches for (no pass over the offsets, the kernel never reads one), for uniform batches it does not
    vouch for (the device looks), for trimmed ones, with a length bound that does not hold (repaired in wait), mixed with
    host-buffer tickets in one stream; a fourth outstanding ticket and a missing bound are refused; counters = histogram"""
    from shark_amd import SharkHipError
    from shark_amd.capi import SHK_PIPE_DEPTH, hip_memcpy_dtoh
    rng = np.random.default_rng(4141)
    genes = synth.make_genes(rng, 20, 400, 2500, share_every=3)
    o, h, _ = _build_both(oracle, genes, k=17, bf_bits=1 << 30)
    dev = torch.device("cuda:0")
    specs = [dict(read_len=150, var_len=False, vouch=True), dict(read_len=150, var_len=False, vouch=False), dict(read_len=100, var_len=True, vouch=False),
             dict(read_len=150, var_len=False, vouch=True, host=True), dict(read_len=125, var_len=False, vouch=True), dict(read_len=150, var_len=False, vouch=True, bound=60),
             dict(read_len=90, var_len=False, vouch=True, paired=False), dict(read_len=150, var_len=True, vouch=False, bound=80)]
    batches, want, keep = [], [], []
    for sp in specs:
        b = synth.make_reads(rng, genes, 1500, read_len=sp["read_len"], paired=sp.get("paired", True), on_target=0.6, var_len=sp["var_len"])
        batches.append(b)
        want.append(o.classify(b["seq1"], b["off1"], b["seq2"], b["off2"]))
    h.gene_counts_reset()
    with pytest.raises(SharkHipError):
        b = batches[0]
        h.submit_device(1500, 1, 1, 1, 1, max_read_len=0)                # no bound: refused before anything is touched
    tickets, got = [], []

    def drain():
        kind, t = tickets.pop(0)
        if kind == "host":
            got.append(h.wait(t))
            return
        r = h.wait_device(t)
        g = np.empty(int(r.n) + 1, np.uint32)
        hip_memcpy_dtoh(g, r.gene_off, g.nbytes)
        ids = np.empty(int(r.n_assoc), np.uint16)
        if len(ids):
            hip_memcpy_dtoh(ids, r.gene_ids, ids.nbytes)
        got.append((g, ids))

    for sp, b in zip(specs, batches):
        if len(tickets) == SHK_PIPE_DEPTH:
            with pytest.raises(SharkHipError, match="not allowed"):
                h.submit(b["seq1"], b["off1"], b["seq2"], b["off2"])
            drain()
        if sp.get("host"):
            tickets.append(("host", h.submit(b["seq1"], b["off1"], b["seq2"], b["off2"])))
            continue
        t = {kk: torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) for kk, v in b.items() if v is not None}
        keep.append(t)
        torch.cuda.synchronize()
        paired = b["seq2"] is not None
        L = sp["read_len"]
        tickets.append(("dev", h.submit_device(1500, t["seq1"].data_ptr(), t["off1"].data_ptr(), t["seq2"].data_ptr() if paired else 0,
                                               t["off2"].data_ptr() if paired else 0, max_read_len=sp.get("bound", L),
                                               uniform_len1=L if sp["vouch"] and "bound" not in sp else 0,
                                               uniform_len2=L if sp["vouch"] and paired and "bound" not in sp else 0)))
    while tickets:
        drain()
    for i, ((wg, wi), (gg, gi)) in enumerate(zip(want, got)):
        assert np.array_equal(wg, gg) and np.array_equal(wi, gi), specs[i]
    allids = np.concatenate([w[1] for w in want])
    assert np.array_equal(h.gene_counts(32), np.bincount(allids, minlength=32)[:32].astype(np.uint64))
    # a caller that vouches for lengths the batch does not have (a base too short: every access stays inside the buffers) gets the
    # batch refused in wait -- the device compared three offsets per mate with r * length -- and a message, not results
    t = keep[0]
    bad = h.submit_device(1500, t["seq1"].data_ptr(), t["off1"].data_ptr(), t["seq2"].data_ptr(), t["off2"].data_ptr(), max_read_len=150,
                          uniform_len1=150, uniform_len2=149)
    with pytest.raises(SharkHipError, match="do not describe the batch"):
        h.wait_device(bad)
    ok = h.submit_device(1500, t["seq1"].data_ptr(), t["off1"].data_ptr(), t["seq2"].data_ptr(), t["off2"].data_ptr(), max_read_len=150,
                         uniform_len1=150, uniform_len2=150)
    r = h.wait_device(ok)
    assert int(r.n_assoc) == len(want[0][1])
    with pytest.raises(SharkHipError, match="exceed max_read_len"):
        h.submit_device(1500, t["seq1"].data_ptr(), t["off1"].data_ptr(), t["seq2"].data_ptr(), t["off2"].data_ptr(), max_read_len=100,
                        uniform_len1=150, uniform_len2=150)


def test_device_side_uniformity_check_finds_the_one_odd_read(oracle, monkeypatch):
    """device-resident batches: uniform_check_kernel decides on the device whether every read has one length per mate (two offsets
    per thread, the third from the next lane).  One read a base shorter -- first, last, odd / even index, either side of a wave and
    of a workgroup of the check, in either mate -- makes the batch ragged; an all-equal batch with an odd read count stays uniform;
    1 %, 20 % and 50 % of the reads a few bases shorter go class by class.  The associations are the oracle's every time (a wrong
    verdict would read every later read at the wrong place)."""
    monkeypatch.setenv("SHK_NO_TRO", "1")        # (the class-by-class path; the three-pairs kernel by offsets has its own test below)
    from shark_amd.capi import hip_memcpy_dtoh
    rng = np.random.default_rng(977)
    genes = synth.make_genes(rng, 1, 5000, 5000)
    o, h, _ = _build_both(oracle, genes, k=17, bf_bits=1 << 30)
    dev = torch.device("cuda:0")
    for n in (1031, 1030):
        base = synth.make_reads(rng, genes, n, read_len=120, paired=True, on_target=0.7)
        m1 = [base["seq1"][int(base["off1"][i]):int(base["off1"][i + 1])] for i in range(n)]
        m2 = [base["seq2"][int(base["off2"][i]):int(base["off2"][i + 1])] for i in range(n)]
        for odd in (None, 0, 1, 2, 63, 64, 127, 128, 129, 511, 512, 513, n - 3, n - 2, n - 1):
            for mate in ((1,) if odd is None else (1, 2)):
                a, b = list(m1), list(m2)
                if odd is not None:
                    if mate == 1:
                        a[odd] = a[odd][:-1]
                    else:
                        b[odd] = b[odd][:-1]
                bt = synth.batch_from_lists(a, b)
                og, oi = o.classify(bt["seq1"], bt["off1"], bt["seq2"], bt["off2"])
                t = {k: torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) for k, v in bt.items() if v is not None}
                r = h.classify_device(n, t["seq1"].data_ptr(), t["off1"].data_ptr(), t["seq2"].data_ptr(), t["off2"].data_ptr(), max_read_len=120)
                goff = np.empty(n + 1, np.uint32)
                hip_memcpy_dtoh(goff, r.gene_off, goff.nbytes)
                gids = np.empty(int(r.n_assoc), np.uint16)
                hip_memcpy_dtoh(gids, r.gene_ids, gids.nbytes)
                assert np.array_equal(goff, og) and np.array_equal(gids, oi), (n, odd, mate)
                assert ("verdict=uniform" in h.last_kernel()) == (odd is None), h.last_kernel()
        # 1 %, 20 %, 50 % odd reads, one to three bases shorter in either mate: a handful of classes, sorted and classified class by class
        said = set()
        for frac in (0.01, 0.2, 0.5):
            for rep in range(2):      # (behind a uniform batch the first ragged one takes the ragged instantiation)
                a, b = list(m1), list(m2)
                for i in np.flatnonzero(rng.random(n) < frac):
                    cut = int(rng.integers(1, 4))
                    if rng.random() < 0.5:
                        a[i] = a[i][:-cut]
                    else:
                        b[i] = b[i][:-cut]
                bt = synth.batch_from_lists(a, b)
                og, oi = o.classify(bt["seq1"], bt["off1"], bt["seq2"], bt["off2"])
                t = {k: torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) for k, v in bt.items() if v is not None}
                r = h.classify_device(n, t["seq1"].data_ptr(), t["off1"].data_ptr(), t["seq2"].data_ptr(), t["off2"].data_ptr(), max_read_len=120)
                goff = np.empty(n + 1, np.uint32)
                hip_memcpy_dtoh(goff, r.gene_off, goff.nbytes)
                gids = np.empty(int(r.n_assoc), np.uint16)
                hip_memcpy_dtoh(gids, r.gene_ids, gids.nbytes)
                assert np.array_equal(goff, og) and np.array_equal(gids, oi), (n, frac, rep)
                said.add(h.last_kernel().split("verdict=")[-1])
        assert "classes" in said and "uniform" not in said, said
    h.close()


@pytest.mark.parametrize("n_genes,k,q,fill,L", [(1, 17, 0, 1, 150), (1, 21, 20, 1, 150), (1, 17, 0, 16, 150), (1, 17, 0, 1, 300), (1, 17, 20, 4, 90), (6, 17, 0, 1, 150)])
def test_trimmed_batches_go_class_by_class(oracle, monkeypatch, n_genes, k, q, fill, L):
    """trimmed samples: a share of the mates is shorter than the rest.  On an index whose uniform batches take the exact table in
    LDS the device sorts such a batch by the pairs' two lengths (class_hist / class_plan / class_scatter kernels) and the CLS
    instantiation classifies it class by class, in the uniform layout of each class -- if the classes are full enough
    (SHK_CLS_MIN_FILL pairs per unit: 1 here, so that every shape below takes that path, and the default 16, where most of these
    small batches stay with the ragged instantiation).  1 %, 20 %, 50 %, 80 % and 100 % trimmed mates (either mate, both, down to
    shorter than k), paired and single-end, host batches and batches resident in HBM, one gene (sparse first round) and several
    genes (the exact table without it), 2 x 90, 2 x 150 and 2 x 300 bp (the last: more classes than the
    pre-pass counts in LDS, the shortest mates take the global counters): the oracle's associations every time."""
    monkeypatch.setenv("SHK_CLS_MIN_FILL", str(fill))
    monkeypatch.setenv("SHK_NO_TRO", "1")        # (this test: the class-by-class path; test_trimmed_batches_through_the_three_pairs_kernel: the other)
    from shark_amd.capi import hip_memcpy_dtoh
    rng = np.random.default_rng(4242 + n_genes + k)
    genes = synth.make_genes(rng, n_genes, 2500, 2600)
    o, h, _ = _build_both(oracle, genes, k=k, bf_bits=1 << 30, min_quality=q, c=0.5)
    assert h.probe_mode() == "lds-table", h.probe_mode()
    dev = torch.device("cuda:0")
    n = 3000
    for paired in (True, False):
        base = synth.make_reads(rng, genes, n, read_len=L, paired=paired, on_target=0.7, qual=q > 0)
        for frac in (0.01, 0.2, 0.5, 0.8, 1.0):
            def trim(seq, off, qual):
                m = [bytes(seq[int(off[i]):int(off[i + 1])]) for i in range(n)]
                qq = [bytes(qual[int(off[i]):int(off[i + 1])]) for i in range(n)] if qual is not None else None
                for i in range(n):
                    if rng.random() < frac:
                        keep = int(rng.integers(k - 3 if rng.random() < 0.05 else min(60, L // 2), L))
                        m[i] = m[i][:keep]
                        if qq is not None:
                            qq[i] = qq[i][:keep]
                return m, qq
            a, qa = trim(base["seq1"], base["off1"], base["qual1"])
            b, qb = trim(base["seq2"], base["off2"], base["qual2"]) if paired else (None, None)
            bt = synth.batch_from_lists(a, b, qa, qb)
            og, oi = o.classify(bt["seq1"], bt["off1"], bt["seq2"], bt["off2"], bt["qual1"], bt["qual2"])
            hg, hi = h.classify(bt["seq1"], bt["off1"], bt["seq2"], bt["off2"], bt["qual1"], bt["qual2"])
            assert np.array_equal(hg, og) and np.array_equal(hi, oi), ("host", paired, frac)
            if fill == 1:     # what the device said about the batch (shk_last_kernel)
                assert "verdict=classes" in h.last_kernel(), h.last_kernel()
            t = {kk: (torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) if v is not None else None) for kk, v in bt.items()}
            pt = {kk: (v.data_ptr() if v is not None else 0) for kk, v in t.items()}
            torch.cuda.synchronize()
            r = h.classify_device(n, pt["seq1"], pt["off1"], pt["seq2"], pt["off2"], pt["qual1"], pt["qual2"], max_read_len=L)
            goff = np.empty(n + 1, np.uint32)
            hip_memcpy_dtoh(goff, r.gene_off, goff.nbytes)
            gids = np.empty(int(r.n_assoc), np.uint16)
            hip_memcpy_dtoh(gids, r.gene_ids, gids.nbytes)
            assert np.array_equal(goff, og) and np.array_equal(gids, oi), ("resident", paired, frac)
            if fill == 1:
                assert "verdict=classes" in h.last_kernel(), h.last_kernel()
            assert int(og[-1]) > n // 4
    h.close()


@pytest.mark.parametrize("n_genes,k,L1,L2,c", [(1, 17, 150, 150, 0.6), (1, 17, 150, 150, 0.25), (5, 17, 150, 150, 0.6), (1, 21, 125, 125, 0.5), (1, 17, 151, 101, 0.6),
                                                (1, 17, 150, 0, 0.6), (3, 12, 140, 140, 0.9), (1, 17, 90, 90, 0.6)])
def test_trimmed_batches_through_the_three_pairs_kernel(oracle, monkeypatch, n_genes, k, L1, L2, c):
    """trimmed samples on an index whose uniform batches take the exact table in LDS, without qualities, at lengths for which three
    pairs share a staging pass: the TRO instantiations -- the three-pairs kernel in the lane layout of the batch's LONGEST mates, a read
    found by its offsets, what lies behind its own end marked invalid (an N would do the same to ReadAnalyzer: `len` counts valid
    characters, a k-mer with an invalid one is skipped, the step between two hits is clamped at k).  1 % ... 100 % trimmed mates, either
    mate or both, down to shorter than k and to nothing at all, with N and chimeric pairs, one gene (sparse first round, with and
    without the tiles' round in front) and several genes, host batches (the host knows: the kernel alone) and resident ones (the
    device's verdict picks it): the oracle's associations, and what ran."""
    from shark_amd.capi import hip_memcpy_dtoh
    monkeypatch.delenv("SHK_NO_TRO", raising=False)
    monkeypatch.delenv("SHK_NO_LDS_TABLE", raising=False)
    rng = np.random.default_rng(5150 + n_genes + k + L1)
    genes = synth.make_genes(rng, n_genes, 2500, 2600, share_every=3 if n_genes > 3 else 0)
    dev = torch.device("cuda:0")
    n = 2999
    for tiles in ("1", "0"):
        monkeypatch.setenv("SHK_TILE_FIRST", tiles)
        o, h, _ = _build_both(oracle, genes, k=k, bf_bits=1 << 30, c=c)
        assert h.probe_mode() == "lds-table", h.probe_mode()
        for frac in (0.01, 0.2, 0.6, 1.0):
            base = synth.make_reads(rng, genes, n, read_len=max(L1, L2), paired=L2 > 0, on_target=0.7, n_rate=0.003)
            chim = _chimeric_batch(rng, genes, n, L1, L2, False, with_n=True, qual=False, k_hint=k)

            def mates(bt, key, off, L):
                m = [bytes(bt[key][int(bt[off][i]):int(bt[off][i]) + L]) for i in range(n)]
                for i in range(n):
                    if i % 2:
                        m[i] = bytes(chim[key][int(chim[off][i]):int(chim[off][i + 1])])[:L]
                    if rng.random() < frac:
                        u = rng.random()
                        keep = 0 if u < 0.01 else (int(rng.integers(1, k)) if u < 0.05 else int(rng.integers(min(40, L // 2), L)))
                        m[i] = m[i][:keep]
                m[int(rng.integers(0, n))] = m[0][:L].ljust(L, b"A")[:L] if len(m[0]) else b"A" * L      # (at least one mate of the full length: the layout's)
                return m
            a = mates(base, "seq1", "off1", L1)
            b = mates(base, "seq2", "off2", L2) if L2 else None
            a[7] = (a[7] + b"A" * L1)[:L1]
            if b is not None:
                b[11] = (b[11] + b"C" * L2)[:L2]
            bt = synth.batch_from_lists(a, b)
            og, oi = o.classify(bt["seq1"], bt["off1"], bt["seq2"], bt["off2"])
            hg, hi = h.classify(bt["seq1"], bt["off1"], bt["seq2"], bt["off2"])
            assert np.array_equal(hg, og) and np.array_equal(hi, oi), ("host", frac, int(np.argmax(hg != og)))
            assert "offsets" in h.last_kernel(), h.last_kernel()
            assert ("+tiles-first" in h.last_kernel()) == (tiles == "1" and n_genes == 1), h.last_kernel()
            t = {kk: (torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) if v is not None else None) for kk, v in bt.items()}
            pt = {kk: (v.data_ptr() if v is not None else 0) for kk, v in t.items()}
            torch.cuda.synchronize()
            r = h.classify_device(n, pt["seq1"], pt["off1"], pt["seq2"], pt["off2"], max_read_len=max(L1, L2))
            goff = np.empty(n + 1, np.uint32)
            hip_memcpy_dtoh(goff, r.gene_off, goff.nbytes)
            gids = np.empty(int(r.n_assoc), np.uint16)
            if len(gids):
                hip_memcpy_dtoh(gids, r.gene_ids, gids.nbytes)
            assert np.array_equal(goff, og) and np.array_equal(gids, oi), ("resident", frac)
            assert "verdict=offsets" in h.last_kernel(), h.last_kernel()
            assert int(og[-1]) > 0 or c > 0.8
        h.close()


def test_class_path_follows_the_stream(oracle, monkeypatch):
    """batches resident in HBM, whose lengths only the device sees: behind a uniform batch the launches of the class-by-class path are
    left out (a sequencer's stream pays nothing for them), so the first trimmed batch of a stream takes the ragged instantiation and
    the ones behind it go class by class; a uniform batch in between is still recognised.  Same associations either way."""
    monkeypatch.setenv("SHK_NO_TRO", "1")
    from shark_amd.capi import hip_memcpy_dtoh
    rng = np.random.default_rng(77)
    genes = synth.make_genes(rng, 1, 2500, 2600)
    o, h, _ = _build_both(oracle, genes, k=17, bf_bits=1 << 30, c=0.5)
    dev = torch.device("cuda:0")
    n, L = 4000, 150
    base = synth.make_reads(rng, genes, n, read_len=L, paired=True, on_target=0.7)
    def cut(seq, off, lens):
        return [bytes(seq[int(off[i]):int(off[i]) + int(lens[i])]) for i in range(n)]
    full = np.full(n, L)
    some = np.where(rng.random(n) < 0.3, rng.choice([100, 120, 140], size=n), L)
    said = []
    for lens1, lens2 in ((full, full), (full, full), (some, full), (some, some), (full, full), (some, some), (some, some)):
        bt = synth.batch_from_lists(cut(base["seq1"], base["off1"], lens1), cut(base["seq2"], base["off2"], lens2), None, None)
        og, oi = o.classify(bt["seq1"], bt["off1"], bt["seq2"], bt["off2"], None, None)
        t = {kk: (torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) if v is not None else None) for kk, v in bt.items()}
        torch.cuda.synchronize()
        # (the caller's bound on the read length: tight, or none -- the (l1, l2) tables are then 2^20 entries and the device checks
        #  the batch's longest mates against them)
        bound = (L, 0)[len(said) % 2]
        r = h.classify_device(n, t["seq1"].data_ptr(), t["off1"].data_ptr(), t["seq2"].data_ptr(), t["off2"].data_ptr(), 0, 0, max_read_len=bound)
        goff = np.empty(n + 1, np.uint32)
        hip_memcpy_dtoh(goff, r.gene_off, goff.nbytes)
        gids = np.empty(int(r.n_assoc), np.uint16)
        hip_memcpy_dtoh(gids, r.gene_ids, gids.nbytes)
        assert np.array_equal(goff, og) and np.array_equal(gids, oi), said
        said.append(h.last_kernel().split("verdict=")[-1])
    assert said == ["uniform", "uniform", "ragged", "classes", "uniform", "ragged", "classes"], said
    h.close()


@pytest.mark.parametrize("q", [94, 95, 100, 222, 223, 256, 300])
def test_min_quality_wraps_like_the_reference_char(oracle, q):
    """argument_parser.hpp:144 stores -q in a `char` and FastqSplitter.hpp:70 adds 33 in a `char`: values above 94 wrap
    (and 256 is `no masking`); quality bytes are compared as signed chars, so bytes >= 128 are below most thresholds"""
    rng = np.random.default_rng(43)
    genes = synth.make_genes(rng, 8, 300, 900)
    o, h, _ = _build_both(oracle, genes, k=15, bf_bits=1 << 22, min_quality=q, c=0.3)
    b = synth.make_reads(rng, genes, 800, read_len=100, paired=True, on_target=0.8, qual=True)
    for key in ("qual1", "qual2"):                                      # sprinkle every byte value, negative chars included
        qa = b[key]
        idx = rng.random(len(qa)) < 0.08
        qa[idx] = rng.integers(0, 256, size=int(idx.sum())).astype(np.uint8)
    goff, _ = _compare_classify(o, h, b)
    if q in (95, 100, 256):                                             # thresholds that mask (almost) nothing
        assert goff[-1] > 300
    if q == 94:                                                         # threshold 127: every base is masked
        assert goff[-1] == 0


# ---------------------------------------------------------------------------
# hand-worked cases (tests/golden/handworked.json): expected values derived on paper from the reference source
# ---------------------------------------------------------------------------
def _handworked():
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "handworked.json")))["cases"]


@pytest.mark.parametrize("case", _handworked(), ids=lambda c: c["name"])
def test_handworked_cases(case, probe, tmp_path):
    from tests.test_oracle import _handworked_batch
    h = _hip(k=case["k"], c=case["c"], bf_bits=case["bf_bits"], min_quality=case["q"], single=case["single"])
    info = h.build([seq.encode() for _, seq in case["fasta"]])
    assert info["n_set_bits"] == case.get("set_bits", case["distinct_kmers"])
    batch, paired = _handworked_batch(case)
    goff, gids = h.classify(batch["seq1"], batch["off1"], batch["seq2"], batch["off2"], batch["qual1"], batch["qual2"])
    assert [list(map(int, gids[goff[i]:goff[i + 1]])) for i in range(len(case["reads"]))] == [r["genes"] for r in case["reads"]]
    if case.get("no_cli"):      # (the collision case: a 64-bit filter, which the CLI's -b -- in GB -- cannot name)
        return
    # end to end through the shark CLI: ssv bytes (-b 1 = 2^33 bits; the few k-mers of a case do not collide there either)
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
    r = _run_shark(args, str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert r.stdout.decode() == case["ssv"]


def test_dist_allreduce_one_process_per_gpu_form(oracle):
    """shk_dist_unique_id / shk_dist_init / shk_dist_gene_counts_allreduce: the one-process-per-GPU form of the exchange step,
    here as a world of one rank (a real RCCL communicator and a real ncclAllReduce on this GPU).  The totals land in a
    separate buffer: the local counters are unchanged, and reducing twice gives the same totals."""
    import ctypes as C
    from shark_amd.capi import SHK_DIST_ID_BYTES
    rng = np.random.default_rng(47)
    genes = synth.make_genes(rng, 9, 400, 1200)
    o, h, _ = _build_both(oracle, genes, k=17, bf_bits=1 << 22)
    b = synth.make_reads(rng, genes, 2000, read_len=100, paired=True, on_target=0.8)
    h.gene_counts_reset()
    goff, gids = _compare_classify(o, h, b)
    want = np.bincount(gids, minlength=16)[:16].astype(np.uint64)
    ident = (C.c_uint8 * SHK_DIST_ID_BYTES)()
    assert h.L.shk_dist_unique_id(ident) == 0
    assert h.L.shk_dist_init(h.h, ident, 0, 1) == 0
    assert h.L.shk_dist_init(h.h, ident, 1, 1) != 0                      # rank outside the world
    for _ in range(2):
        assert np.array_equal(h.dist_gene_counts_allreduce(16), want)
        assert np.array_equal(h.gene_counts(16), want)
    _compare_classify(o, h, b)                                             # more reads, then reduce again: twice the counts, not four times
    assert np.array_equal(h.dist_gene_counts_allreduce(16), 2 * want)


def test_more_than_65536_genes_wrap_like_the_reference(oracle):
    """small_vector.hpp:46 stores gene ids as uint16_t and bloomfilter.h:72 compares the stored id with the int index: with
    more than 65 536 genes the ids wrap (gene 65536+x is reported under x's name) and a gene above 65535 is appended once
    per k-mer occurrence, duplicates included -- which changes (cov, nk) through ReadAnalyzer.hpp:56-62,:79-86.  Well
    defined in the reference, hence reproduced: index totals, every list as a multiset, and the associations."""
    rng = np.random.default_rng(65536)
    n_genes = 70000
    genes = [synth.random_seq(rng, int(rng.integers(40, 70))) for _ in range(n_genes)]
    rep = synth.random_seq(rng, 30)
    genes[65540] = np.concatenate([rep, synth.random_seq(rng, 5), rep, synth.random_seq(rng, 20)])   # k-mers twice inside a wrapped gene
    genes[65550] = np.concatenate([rep[:25], synth.random_seq(rng, 30)])                             # ... and shared with another one
    genes[66000] = genes[464].copy()                         # 66000 & 0xFFFF == 464: two genes behind one id
    genes[69999] = np.concatenate([genes[3][:35], genes[65539][:30]])
    k = 17
    o, h, info = _build_both(oracle, genes, k=k, bf_bits=1 << 30)
    assert info["nidx"] == n_genes
    assert info["n_set_bits"] == o.num_kmer()
    off, ids = h.copy_lists()
    oi = o.index_kmer()
    assert info["tot_idx"] == len(oi) == len(ids)
    # same lists as multisets: the oracle keeps insertion order (genes below 65536 ascending, then the wrapped ones in gene
    # order), the device sorts by id
    # (list r of the oracle is the r-th run of index_kmer; equal lengths are implied by the multiset comparison below)
    assert np.array_equal(np.sort(ids), np.sort(oi))
    starts = off.astype(np.int64)
    a = np.split(ids, starts[1:-1])
    b = np.split(oi, starts[1:-1])
    assert all(np.array_equal(np.sort(x), np.sort(y)) for x, y in zip(a[:20000], b[:20000]))
    assert all(np.all(np.diff(x.astype(np.int64)) >= 0) for x in a[:20000])
    assert any(len(x) != len(np.unique(x)) for x in a), "expected duplicate ids in some list"
    # reads from wrapped genes, from genes below, from the colliding pair, and random ones
    picks = [65540, 65550, 66000, 464, 69999, 3, 65539, 12, 65536, 65535, 70000 - 1, 1000, 68000]
    m1, m2 = [], []
    for g in picks * 40:
        s_ = genes[g]
        L = int(rng.integers(20, len(s_) + 1))
        st = int(rng.integers(0, len(s_) - L + 1))
        a_ = s_[st:st + L].copy()
        if rng.random() < 0.2:
            a_[int(rng.integers(0, L))] = ord("N")
        m1.append(a_.tobytes())
        m2.append(synth.revcomp(s_)[:int(rng.integers(17, len(s_) + 1))].tobytes())
    for _ in range(200):
        m1.append(synth.random_seq(rng, 60).tobytes())
        m2.append(synth.random_seq(rng, 60).tobytes())
    # and a read far beyond the fast kernels' capacity, so that the queue of long reads is exercised in wrap mode too
    m1.append(np.concatenate([genes[65540], genes[66000], synth.random_seq(rng, 700), genes[69999]]).tobytes())
    m2.append(synth.revcomp(np.concatenate([genes[65550], genes[464]])).tobytes())
    batch = synth.batch_from_lists(m1, m2)
    for c in (0.3, 0.0):
        o2, h2, _ = _build_both(oracle, genes, k=k, bf_bits=1 << 30, c=c)
        goff, gids = _compare_classify(o2, h2, batch)
        assert goff[-1] > 300
        assert h2.timing()["last_n_long"] >= 1
    # single-end, --single
    o3, h3, _ = _build_both(oracle, genes, k=k, bf_bits=1 << 30, c=0.2, single=True)
    _compare_classify(o3, h3, synth.batch_from_lists(m1))


@pytest.mark.parametrize("bf_bits", [1 << 30, 5 << 32])
def test_panel_sized_index_uses_the_big_lds_summary(oracle, bf_bits, monkeypatch):
    """~2.4e5 set bits: too dense for the 2^18-bit LDS summary, sparse enough for the 2^20-bit one (classify_uni_kernel with a
    1024-thread workgroup per CU).  Uniform batches take it, ragged ones the index's ordinary chain; both must equal the oracle,
    and so must the same index without it."""
    rng = np.random.default_rng(2020)
    genes = synth.make_genes(rng, 100, 2000, 3000, share_every=5)
    o, h, info = _build_both(oracle, genes, k=17, bf_bits=bf_bits)
    assert 180_000 < info["n_set_bits"] < 330_000 and "lds" not in h.probe_mode()
    uni = synth.make_reads(rng, genes, 3000, read_len=150, paired=True, on_target=0.6)
    rag = synth.make_reads(rng, genes, 2000, read_len=150, paired=True, on_target=0.6, var_len=True)
    se = synth.make_reads(rng, genes, 1500, read_len=100, paired=False, on_target=0.6)
    want = [_compare_classify(o, h, b) for b in (uni, rag, se)]
    og, n = _probe_every_kmer(o, h, genes, 17, stride=2)
    assert int(og[-1]) >= n
    monkeypatch.setenv("SHK_NO_BIG_LDS_SUMMARY", "1")
    h2 = _hip(k=17, bf_bits=bf_bits)
    h2.build([bytes(g) for g in genes])
    for b, (wg, wi) in zip((uni, rag, se), want):
        g2, i2 = h2.classify(b["seq1"], b["off1"], b["seq2"], b["off2"])
        assert np.array_equal(g2, wg) and np.array_equal(i2, wi)


def test_panel_sized_index_follows_the_assigned_fraction(oracle, monkeypatch):
    """an index with the 2^20-bit LDS summary switches its uniform batches to the position-table kernel (anchored extension) while
    the batch just finished had many pairs assigned, and back -- identical results either way, in every order of batches, and with
    the switching turned off"""
    rng = np.random.default_rng(2121)
    genes = synth.make_genes(rng, 100, 2000, 3000, share_every=5)
    on = synth.make_reads(rng, genes, 3000, read_len=150, paired=True, on_target=1.0)
    off = synth.make_reads(rng, genes, 3000, read_len=150, paired=True, on_target=0.0)
    mix = synth.make_reads(rng, genes, 3000, read_len=150, paired=True, on_target=0.5)
    for always in (False, True):
        if always:
            monkeypatch.setenv("SHK_BIG_LDS_ALWAYS", "1")
        o, h, info = _build_both(oracle, genes, k=17, bf_bits=1 << 33)
        assert "lds" not in h.probe_mode()
        fr, kernels = [], set()
        for b in (on, on, off, off, mix, on, mix, off, on):
            goff, _ = _compare_classify(o, h, b)
            fr.append(int(goff[-1]) / 3000)
            kernels.add(h.last_kernel())
        assert max(fr) > 0.9 and min(fr) < 0.05
        # what ran (shk_last_kernel): with the switching on both the 128 KiB LDS summary (LSL = 20) and the position-table kernel with
        # the anchored extension took batches; with SHK_BIG_LDS_ALWAYS=1 (read once, at shk_create) only the former
        lds = {x for x in kernels if ", 20, " in x}
        tab = {x for x in kernels if "+anchored-extension" in x}
        assert lds, kernels
        assert (not tab) if always else bool(tab), kernels
        h.close()


def test_tiles_first_follows_the_assigned_fraction(oracle, monkeypatch):
    """one-gene index in LDS, uniform batches: behind a batch with a quarter of its reads assigned or more the three-pairs kernel is
    launched with the tiles' round in front (a pair from the gene ends behind a third of a hash round), behind a batch of reads from
    elsewhere without it (they would pay that third for nothing); identical results either way, in every order of batches; pairs
    with more errors than the tiles forgive, with N, and chimeric pairs on either side of the threshold go on through the usual rounds"""
    monkeypatch.delenv("SHK_TILE_FIRST", raising=False)
    monkeypatch.delenv("SHK_NO_LDS_TABLE", raising=False)
    rng = np.random.default_rng(3131)
    genes = synth.make_genes(rng, 1, 8_000, 8_000)
    mk = lambda ot, n_rate=0.0: synth.make_reads(rng, genes, 3001, read_len=150, paired=True, on_target=ot, n_rate=n_rate)
    on, off, few, mix = mk(1.0), mk(0.0), mk(0.05), mk(0.5, 0.003)
    noisy = _sequenced_pairs(rng, genes, 3001, 150, 150, False, False, 0.03, 0.002, 0.003)
    chim = _chimeric_batch(rng, genes, 3001, 150, 150, False, with_n=True, qual=False, k_hint=17)
    o, h, info = _build_both(oracle, genes, k=17, bf_bits=1 << 30)
    assert h.probe_mode() == "lds-table"
    seen = []
    for b in (on, mix, off, few, on, noisy, chim, noisy, off, mix, on):
        goff, _ = _compare_classify(o, h, b)
        n = len(b["off1"]) - 1
        seen.append((int(goff[-1]) / n, "+tiles-first" in h.last_kernel(), "+three-pairs" in h.last_kernel()))
    assert all(t for _, _, t in seen), seen
    assert not seen[0][1]                                       # (no predecessor: without)
    for (frac_before, _, _), (_, tiles, _) in zip(seen, seen[1:]):
        assert tiles == (frac_before >= 0.25), seen
    assert any(t for _, t, _ in seen) and not all(t for _, t, _ in seen[1:])
    h.close()


def test_anchored_extension_follows_the_assigned_fraction(oracle, monkeypatch):
    """table modes: a batch behind one that left nearly all of its reads unassigned is launched without the anchored extension (its
    sample is a memory round trip that pairs from elsewhere pay for nothing), the batch behind one with many reads assigned with it;
    identical results either way, in every order of batches, and with the switching turned off (SHK_ANCHOR_ALWAYS=1)"""
    rng = np.random.default_rng(2626)
    genes = synth.make_genes(rng, 24, 900, 3500, share_every=3)
    on = synth.make_reads(rng, genes, 2000, read_len=150, paired=True, on_target=1.0)
    off = synth.make_reads(rng, genes, 2000, read_len=150, paired=True, on_target=0.0)
    few = synth.make_reads(rng, genes, 2000, read_len=150, paired=True, on_target=0.02)
    some = synth.make_reads(rng, genes, 2000, read_len=150, paired=True, on_target=0.10)
    monkeypatch.setenv("SHK_NO_LDS_SUMMARY", "1")      # (the position table behind the L2 summary, as in test_anchored_extension_reads)
    for always in (False, True):
        if always:
            monkeypatch.setenv("SHK_ANCHOR_ALWAYS", "1")
        o, h, info = _build_both(oracle, genes, k=17, bf_bits=1 << 26)
        assert h.probe_mode() in ("table", "summary+table"), h.probe_mode()
        seen = []
        for b in (on, off, few, on, on, some, on, few, off, some, some, on):
            goff, _ = _compare_classify(o, h, b)
            seen.append((int(goff[-1]) / 2000, "+anchored-extension" in h.last_kernel(), "+pre-verdict" in h.last_kernel()))
        # (the first batch has no predecessor: with the extension, and with anchor_verdict_kernel in front)
        assert seen[0][1] and seen[0][2]
        for (frac_before, _, _), (_, with_ext, with_pre) in zip(seen, seen[1:]):
            assert with_ext == (always or frac_before >= 0.05), seen
            # anchor_verdict_kernel costs every pair of the batch a pass: behind a batch with fewer than 15 reads in 100 assigned it stays out
            assert with_pre == (always or frac_before >= 0.15), seen
        h.close()


@pytest.mark.parametrize("k,bf_bits,n_bases", [
    (17, 1 << 33, 25_600),      # lds-summary+table: walks per round; 25 584 keys in 32 768 slots
    (17, 5 << 32, 25_600),      # the same with hash % size
    (17, 1 << 26, 410_000),     # summary+table / table: probe-by-probe walk in the 64-VGPR instantiations; load 0.78
    (21, 1 << 30, 200_000),     # uniform batches: 128 KiB LDS summary; load 0.76
])
def test_dense_table_long_probe_paths(oracle, k, bf_bits, n_bases, monkeypatch):
    """SHK_TAB_DENSE=1 builds the position table at up to 0.8 load: most home buckets are full, a fifth of the keys live
    behind theirs, probe paths run over several buckets (the overflow mark, both walk forms, the match in either slot of a
    later bucket, the empty slot that ends a search).  Uniform, trimmed and single-end batches -- every kernel that reads
    the table -- must still equal the oracle, and the index must equal the oracle's word for word."""
    monkeypatch.setenv("SHK_TAB_DENSE", "1")
    monkeypatch.setenv("SHK_NO_LDS_TABLE", "1")   # (uniform batches of the two small cases would not read the position table otherwise)
    rng = np.random.default_rng(n_bases + k)
    n_genes = 8
    genes = synth.make_genes(rng, n_genes, n_bases // n_genes, n_bases // n_genes + 1, share_every=3)
    o, h, info = _build_both(oracle, genes, k=k, bf_bits=bf_bits)
    assert "table" in h.probe_mode()
    slots = 1024
    while slots * 8 < 10 * info["n_set_bits"]:
        slots *= 2
    assert info["n_set_bits"] / slots > 0.55, "the case is meant to load the table heavily"
    _compare_index(o, h, info)
    for b in (synth.make_reads(rng, genes, 4000, read_len=150, paired=True, on_target=0.7),
              synth.make_reads(rng, genes, 3000, read_len=150, paired=True, on_target=0.7, var_len=True, n_rate=0.01),
              synth.make_reads(rng, genes, 2000, read_len=100, paired=False, on_target=0.7),
              synth.make_reads(rng, genes, 1500, read_len=250, paired=True, on_target=0.7)):
        goff, _ = _compare_classify(o, h, b)
        assert goff[-1] > 0
    og, n = _probe_every_kmer(o, h, genes, k, stride=3 if n_bases > 100_000 else 1)
    assert int(og[-1]) >= n


@pytest.mark.parametrize("shape", ["fixed_width", "ragged", "second_file_shorter", "first_file_shorter_no_final_newline", "bgzf", "single_end_gz",
                                   "gzip_multi_member", "gzip_-9"])
def test_cli_feeds_agree_with_the_oracle_cli(oracle, tmp_path, shape):
    """every way the CLI gets its reads -- arithmetic offsets of fixed-width records, the parallel newline count, the pair
    stream ending with the shorter mate file, a last record without newline (serial reader takes over), BGZF blocks inflated
    in parallel, gzip inflated ahead of the parser -- with several GPUs' worth of queues, thread counts and batch sizes:
    ssv and both FASTQ outputs byte for byte the oracle CLI's"""
    import gzip
    import struct
    import subprocess
    import zlib
    rng = np.random.default_rng(len(shape))
    genes = synth.make_genes(rng, 6, 500, 1500, share_every=3)
    fa = tmp_path / "g.fa"
    fa.write_text("".join(">g%d\n%s\n" % (i, bytes(g).decode()) for i, g in enumerate(genes)))
    n = 2300

    def fastq(tag, count, fixed):
        out = []
        for i in range(count):
            g = genes[i % 6]
            L = 100 if fixed else int(rng.integers(40, 120))
            st = int(rng.integers(0, len(g) - L))
            s_ = bytes(g[st:st + L]) if i % 4 else bytes(synth.random_seq(rng, L))
            q = bytes(rng.integers(35, 74, size=L).astype(np.uint8))
            rid = (b"r%06d/%d" % (i, tag)) if fixed else (b"read%d/%d some text" % (i, tag))
            out.append(b"@" + rid + b"\n" + s_ + b"\n+\n" + q + b"\n")
        return b"".join(out)

    fixed = shape == "fixed_width"
    t1, t2 = fastq(1, n, fixed), fastq(2, n - 700 if shape == "second_file_shorter" else n, fixed)
    if shape == "first_file_shorter_no_final_newline":
        t1 = fastq(1, n - 450, False)[:-1]
    f1, f2 = tmp_path / "a_1.fq", tmp_path / "a_2.fq"
    f1.write_bytes(t1)
    f2.write_bytes(t2)
    paired = shape != "single_end_gz"
    if shape == "bgzf":
        for f, data in ((f1, t1), (f2, t2)):
            with open(str(f) + ".gz", "wb") as fh:
                for o_ in list(range(0, len(data), 50000)) + [None]:
                    chunk = b"" if o_ is None else data[o_:o_ + 50000]
                    c = zlib.compressobj(6, zlib.DEFLATED, -15)
                    d = c.compress(chunk) + c.flush()
                    fh.write(struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord("B"), ord("C"), 2, len(d) + 25))
                    fh.write(d)
                    fh.write(struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
        f1, f2 = tmp_path / "a_1.fq.gz", tmp_path / "a_2.fq.gz"
    if shape == "single_end_gz":
        with gzip.open(str(f1) + ".gz", "wb") as fh:
            fh.write(t1)
        f1 = tmp_path / "a_1.fq.gz"
    env = dict(os.environ)
    if shape in ("gzip_multi_member", "gzip_-9"):
        # ordinary gzip, inflated in parallel in two passes (gzip_parallel.hpp); chunks of 16 KiB so that files of this size are cut
        # into a dozen of them (the default, 4 MiB, leaves files under 12 MiB to gzread)
        env["SHARK_GZ_CHUNK"] = "16384"
        for f, data in ((f1, t1), (f2, t2)):
            if shape == "gzip_-9":
                blob = gzip.compress(data, compresslevel=9)
            else:
                cuts = [0, len(data) // 7, len(data) // 2, len(data) // 2 + 1, len(data)]
                blob = b"".join(gzip.compress(data[a:b], compresslevel=lv) for a, b, lv in zip(cuts, cuts[1:], (1, 6, 9, 3)))
            open(str(f) + ".gz", "wb").write(blob)
        f1, f2 = tmp_path / "a_1.fq.gz", tmp_path / "a_2.fq.gz"
    args = ["-r", str(fa), "-1", str(f1), "-k", "15", "-q", "4", "-c", "0.3"] + (["-2", str(f2)] if paired else [])   # (-q 4 masks ~5 % of the bases)
    ossv = tmp_path / "o.ssv"
    oracle.run_cli(args + ["-o", str(tmp_path / "o1.fq")] + (["-p", str(tmp_path / "o2.fq")] if paired else []), str(ossv))
    assert ossv.read_bytes().count(b"\n") > 500
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shark_amd", "bin", "shark")
    for extra in (["--batch", "97", "-t", "5"], ["--batch", "1000", "-t", "3"], []):
        r = subprocess.run([exe] + args + ["-o", str(tmp_path / "h1.fq")] + (["-p", str(tmp_path / "h2.fq")] if paired else []) + extra,
                           capture_output=True, cwd=str(tmp_path), env=env)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        assert r.stdout == ossv.read_bytes(), (shape, extra)
        assert (tmp_path / "h1.fq").read_bytes() == (tmp_path / "o1.fq").read_bytes()
        if paired:
            assert (tmp_path / "h2.fq").read_bytes() == (tmp_path / "o2.fq").read_bytes()


# ---------------------------------------------------------------------------
# the bound cut of classify_uni_kernel (classify_uni.hpp: rounds [0, E) first; a read none of whose first slots is in the filter
# ends there when the remaining slots cannot reach c * len): chimeric reads whose on-target part starts or ends anywhere
# in either mate, for thresholds on both sides of what the remaining rounds cover
# ---------------------------------------------------------------------------
def _chimeric_batch(rng, genes, n, L1, L2, ragged, with_n, qual, k_hint=17):
    m1s, m2s, q1s, q2s = [], [], [], []
    for _ in range(n):
        g = genes[int(rng.integers(0, len(genes)))]
        mates = []
        for L in (L1, L2):
            if L == 0:
                continue
            if mates and rng.random() < 0.3:        # mate 2 from another gene: two genes lead by turns (the early decision)
                g = genes[int(rng.integers(0, len(genes)))]
            l = int(rng.integers(max(1, (2 * L) // 3), L + 1)) if ragged else L
            m = synth.random_seq(rng, l)
            st = int(rng.integers(0, len(g) - l + 1))
            src = g[st:st + l] if rng.random() < 0.5 else synth.revcomp(g[st:st + l])
            form = int(rng.integers(0, 7))          # 0: off-target; 1: prefix; 2: suffix; 3: middle; 4: whole mate; 5, 6: mosaic
            s = int(rng.integers(0, l + 1))
            if form == 1:
                m[:s] = src[:s]
            elif form == 2:
                m[l - s:] = src[l - s:]
            elif form == 3:
                a = int(rng.integers(0, l - s + 1))
                m[a:a + s] = src[a:a + s]
            elif form == 4:
                m[:] = src
            elif form >= 5:
                # pieces of several genes side by side: together they cover much, no single gene does (the per-gene bound)
                p = 0
                while p < l:
                    w = int(rng.integers(k_hint, 3 * k_hint + 8))
                    g2 = genes[int(rng.integers(0, len(genes)))]
                    st2 = int(rng.integers(0, len(g2) - w + 1))
                    if rng.random() < 0.8:
                        m[p:p + w] = g2[st2:st2 + w][:l - p]
                    p += w
            if with_n and rng.random() < 0.4:
                m[rng.integers(0, l, size=int(rng.integers(1, 4)))] = ord("N")
            mates.append(m)
        m1s.append(mates[0])
        if L2:
            m2s.append(mates[1])
        if qual:
            for lst, m in zip((q1s, q2s), mates):
                q = np.where(rng.random(len(m)) < 0.93, rng.integers(25, 42, size=len(m)), rng.integers(2, 20, size=len(m)))
                lst.append((q + 33).astype(np.uint8))
    return synth.batch_from_lists(m1s, m2s if L2 else None, q1s if qual else None, q2s if (qual and L2) else None)


@pytest.mark.parametrize("env", [{}, {"SHK_NO_LDS_TABLE": "1"}, {"SHK_NO_LDS_SUMMARY": "1"}, {"SHK_NO_LDS_SUMMARY": "1", "SHK_NO_SUMMARY": "1"},
                                 {"BF": str(3 << 24)}, {"SHK_NO_LDS_SUMMARY": "1", "SHK_NO_SUMMARY": "1", "BF": str(3 << 24)}])
@pytest.mark.parametrize("L1,L2,k", [(150, 150, 17), (150, 150, 31), (100, 100, 17), (150, 0, 17), (250, 250, 21), (60, 50, 9)])
def test_bound_cut_chimeric_reads(oracle, monkeypatch, env, L1, L2, k):
    bf_bits = 1 << 26
    for name, v in env.items():
        if name == "BF":
            bf_bits = int(v)
        else:
            monkeypatch.setenv(name, v)
    rng = np.random.default_rng(9000 + L1 + 7 * L2 + k)
    genes = synth.make_genes(rng, 12, 1200, 4000, share_every=3)
    for c in (0.25, 0.45, 0.6, 0.75, 1.0):
        for q in (0, 20):
            o, h, info = _build_both(oracle, genes, k=k, bf_bits=bf_bits, c=c, min_quality=q)
            assert "table" in h.probe_mode()
            for ragged in (False, True):
                batch = _chimeric_batch(rng, genes, 700, L1, L2, ragged, with_n=True, qual=q > 0, k_hint=k)
                goff, _ = _compare_classify(o, h, batch)
                assert goff[-1] > 0 or c > 0.6 or q > 0
            h.close()


# ---------------------------------------------------------------------------
# the anchored extension of the table kernels (classify_uni.hpp: sample -> anchor -> compare with the reference -> early decision)
# ---------------------------------------------------------------------------
def _sequenced_pairs(rng, genes, n, L1, L2, ragged, qual, sub_rate, indel_rate, n_rate):
    """pairs as a sequencer would give them: a fragment of the concatenated reference (so some cross into the next gene), mate 2
    from its far end reverse-complemented, substitutions, small insertions / deletions, N, some mates replaced by noise or by
    another gene, either strand"""
    cat = np.concatenate(genes)
    m1s, m2s, q1s, q2s = [], [], [], []
    for _ in range(n):
        l1 = int(rng.integers(max(1, (2 * L1) // 3), L1 + 1)) if ragged else L1
        l2 = (int(rng.integers(max(1, (2 * L2) // 3), L2 + 1)) if ragged else L2) if L2 else 0
        frag = int(rng.integers(max(l1, l2), max(l1, l2) * 3 + 1))
        st = int(rng.integers(0, len(cat) - frag - 8))
        f = cat[st:st + frag + 8].copy()

        def damage(m, want):
            m = list(m)
            i = 0
            out = []
            while i < len(m):
                u = rng.random()
                if u < indel_rate / 2:
                    i += int(rng.integers(1, 4))                     # deletion
                    continue
                if u < indel_rate:
                    out.extend(synth.random_seq(rng, int(rng.integers(1, 4))))   # insertion
                out.append(m[i])
                i += 1
            m = np.array(out[:want], dtype=np.uint8)
            if len(m) < want:
                m = np.concatenate([m, synth.random_seq(rng, want - len(m))])
            sub = rng.random(want) < sub_rate
            m[sub] = synth.ACGT[rng.integers(0, 4, size=int(sub.sum()))]
            m[rng.random(want) < n_rate] = ord("N")
            return m
        a = damage(f[:l1 + 6], l1)
        b = damage(synth.revcomp(f[:frag])[:l2 + 6], l2) if L2 else None
        if rng.random() < 0.5 and L2:                                  # the other strand of the fragment
            a, b = b[:l1] if len(b) >= l1 else np.concatenate([b, synth.random_seq(rng, l1 - len(b))]), \
                   a[:l2] if len(a) >= l2 else np.concatenate([a, synth.random_seq(rng, l2 - len(a))])
        u = rng.random()
        if u < 0.08:
            a = synth.random_seq(rng, l1)                              # mate 1 is noise
        elif u < 0.16 and L2:
            b = synth.random_seq(rng, l2)
        elif u < 0.22 and L2:
            g = genes[int(rng.integers(0, len(genes)))]
            if len(g) > l2:
                st2 = int(rng.integers(0, len(g) - l2))
                b = g[st2:st2 + l2].copy()                            # mate 2 from another gene
        m1s.append(a)
        if L2:
            m2s.append(b)
        if qual:
            for lst, m in zip((q1s, q2s), (a, b) if L2 else (a,)):
                q = np.where(rng.random(len(m)) < 0.95, rng.integers(25, 42, size=len(m)), rng.integers(2, 20, size=len(m)))
                lst.append((q + 33).astype(np.uint8))
    return synth.batch_from_lists(m1s, m2s if L2 else None, q1s if qual else None, q2s if (qual and L2) else None)


@pytest.mark.parametrize("env", [{"SHK_NO_LDS_SUMMARY": "1"}, {"SHK_NO_LDS_SUMMARY": "1", "SHK_NO_SUMMARY": "1"},
                                 {"SHK_NO_LDS_SUMMARY": "1", "SHK_NO_SUMMARY": "1", "BF": str(3 << 24)}])
@pytest.mark.parametrize("L1,L2,k", [(150, 150, 17), (150, 150, 31), (100, 100, 16), (150, 0, 20), (76, 76, 11), (250, 250, 21), (300, 300, 17)])
def test_anchored_extension_reads(oracle, monkeypatch, env, L1, L2, k):
    if L1 >= 250 and "BF" not in env:
        pytest.skip("2 x 250 / 2 x 300 bp run on the table-mod chain only (suite budget); the other chains take them in test_long_pairs_2x300")
    _anchored_extension_reads(oracle, monkeypatch, env, L1, L2, k)


def _anchored_extension_reads(oracle, monkeypatch, env, L1, L2, k):
    """table modes with the reference arrays (anchor / refpay / ref2): reads that match the reference with substitutions, indels,
    N, on either strand, across gene boundaries, from shared gene halves (multi-gene lists, ties), with one mate off-target or
    from another gene; even k (palindromic k-mers); genes with N and genes shorter than k in the reference.  Every result equals
    the oracle's with the extension and without it (SHK_NO_ANCHOR=1 at index build time)."""
    bf_bits = 1 << 26
    for name, v in env.items():
        if name == "BF":
            bf_bits = int(v)
        else:
            monkeypatch.setenv(name, v)
    rng = np.random.default_rng(4100 + L1 + 3 * L2 + k)
    genes = synth.make_genes(rng, 24, 900, 3500, share_every=3)
    genes[5][100:103] = ord("N")                                      # invalid characters inside a gene
    genes[7] = synth.random_seq(rng, max(1, k - 3))                   # a record shorter than k
    pal = np.frombuffer(b"ACGT" * 16, dtype=np.uint8)                 # its own reverse complement at every even k
    genes[9][200:200 + len(pal)] = pal
    genes[11][50:50 + len(pal)] = pal
    monkeypatch.setenv("SHK_ANCHOR_ALWAYS", "1")     # (else a batch behind one with few reads assigned runs without the extension)
    for anchor in (True, False):
        if anchor:
            monkeypatch.delenv("SHK_NO_ANCHOR", raising=False)
        else:
            monkeypatch.setenv("SHK_NO_ANCHOR", "1")
        for c, single, q in ((0.6, False, 0), (0.3, True, 0), (0.9, False, 20)):
            o, h, info = _build_both(oracle, genes, k=k, bf_bits=bf_bits, c=c, min_quality=q, single=single)
            assert h.probe_mode() in ("table", "summary+table", "table-mod"), h.probe_mode()
            for ragged in (False, True):
                for sub_rate, indel_rate in ((0.0, 0.0), (0.01, 0.0), (0.03, 0.004)):
                    batch = _sequenced_pairs(rng, genes, 400, L1, L2, ragged, q > 0, sub_rate, indel_rate, 0.002)
                    goff, _ = _compare_classify(o, h, batch)
                    assert goff[-1] > 0 or c > 0.6
                    # the A side ran WITH the extension, the B side without (the switch is read when the index is built)
                    if "classify_uni_kernel" in h.last_kernel():
                        assert ("+anchored-extension" in h.last_kernel()) == anchor, h.last_kernel()
            if q == 0:
                _probe_every_kmer(o, h, genes, k, stride=7)
            h.close()


# ---------------------------------------------------------------------------
# anchor_verdict_kernel in front of the table kernels (anchor_verdict.hip; DeviceIndex::refext / refmul)
# ---------------------------------------------------------------------------
def _pairs_with_counted_mismatches(rng, genes, n, L1, L2, ragged, qual):
    """pairs cut from a gene with a CHOSEN number of disagreeing bases (0 ... 12 per pair: on both sides of what the verdict accepts
    for any threshold), spread, clustered or at the mates' ends, as substitutions or N; fragments that start in one gene's own
    sequence and run into a stretch it shares with another gene (the extents of `refext` end inside the read), mates of two
    genes, a mate shifted by an insertion, either strand"""
    m1s, m2s, q1s, q2s = [], [], [], []
    for _ in range(n):
        gi = int(rng.integers(0, len(genes)))
        g = genes[gi]
        l1 = int(rng.integers(max(1, (2 * L1) // 3), L1 + 1)) if ragged else L1
        l2 = (int(rng.integers(max(1, (2 * L2) // 3), L2 + 1)) if ragged else L2) if L2 else 0
        if len(g) < max(l1, l2) + 2:
            g = max(genes, key=len)
        frag = int(rng.integers(max(l1, l2), min(len(g), max(l1, l2) * 3) + 1))
        st = int(rng.integers(0, len(g) - frag + 1))
        f = g[st:st + frag]
        a = f[:l1].copy()
        b = synth.revcomp(f)[:l2].copy() if L2 else None
        if L2 and rng.random() < 0.1:                                     # mate 2 from another gene
            g2 = genes[int(rng.integers(0, len(genes)))]
            if len(g2) > l2:
                s2 = int(rng.integers(0, len(g2) - l2))
                b = g2[s2:s2 + l2].copy()
        if rng.random() < 0.5 and L2 and l1 == l2:
            a, b = b, a
        e = int(rng.integers(0, 13))
        mates = [a, b] if L2 else [a]
        form = int(rng.integers(0, 4))
        for _i in range(e):
            m = mates[int(rng.integers(0, len(mates)))]
            if form == 0:
                p_ = int(rng.integers(0, len(m)))                        # anywhere
            elif form == 1:
                p_ = int(min(len(m) - 1, rng.integers(0, 24)))            # the mate's first bases
            elif form == 2:
                p_ = int(max(0, len(m) - 1 - rng.integers(0, 24)))        # its last bases
            else:
                c0 = len(m) // 2
                p_ = int(np.clip(c0 + rng.integers(-10, 11), 0, len(m) - 1))   # one cluster
            if rng.random() < 0.25:
                m[p_] = ord("N")
            else:
                m[p_] = [x for x in b"ACGT" if x != m[p_]][int(rng.integers(0, 3))]          # another base
        if rng.random() < 0.05:                                           # an insertion: everything behind it is shifted
            m = mates[0]
            p_ = int(rng.integers(1, len(m)))
            m[p_:] = np.concatenate([synth.random_seq(rng, 1), m[p_:-1]])
        m1s.append(mates[0])
        if L2:
            m2s.append(mates[1])
        if qual:
            for lst, m in zip((q1s, q2s), mates):
                q = np.where(rng.random(len(m)) < 0.97, rng.integers(25, 42, size=len(m)), rng.integers(2, 20, size=len(m)))
                lst.append((q + 33).astype(np.uint8))
    return synth.batch_from_lists(m1s, m2s if L2 else None, q1s if qual else None, q2s if (qual and L2) else None)


@pytest.mark.parametrize("env", [{"SHK_NO_LDS_SUMMARY": "1"}, {"SHK_NO_LDS_SUMMARY": "1", "SHK_NO_SUMMARY": "1"},
                                 {"SHK_NO_LDS_SUMMARY": "1", "SHK_NO_SUMMARY": "1", "BF": str(3 << 24)}])
@pytest.mark.parametrize("L1,L2,k", [(150, 150, 17), (150, 150, 31), (100, 100, 16), (150, 0, 20), (76, 76, 11), (250, 250, 21), (300, 300, 17), (150, 12, 17)])
def test_verdict_by_mismatch_count(oracle, monkeypatch, env, L1, L2, k):
    """anchor_verdict_kernel: pairs that agree with the reference around their anchors are settled from bit masks -- which slots hold
    the reference's own k-mer under a single-gene list, what those cover, what all other slots cover.  Pairs with 0 ... 12
    disagreeing bases -- spread, clustered, at the ends, as N --, thresholds on both sides of what their matched slots cover,
    references with shared halves (lists of several genes under part of a read), a repeat inside a gene, a gene whose start is
    another's tail, mates of two genes, an insertion: the oracle's result with the kernel in front and without it
    (SHK_NO_PRE_VERDICT=1; SHK_NO_REFEXT=1 at build time: the index does not carry its arrays)."""
    bf_bits = 1 << 26
    for name, v in env.items():
        if name == "BF":
            bf_bits = int(v)
        else:
            monkeypatch.setenv(name, v)
    rng = np.random.default_rng(6100 + L1 + 3 * L2 + k)
    genes = synth.make_genes(rng, 18, 900, 3500, share_every=3)
    genes[4][300:303] = ord("N")
    rep = genes[6][100:100 + 3 * k].copy()                               # a repeat inside a gene: an anchor may name the other copy
    genes[6][700:700 + len(rep)] = rep
    genes[8] = np.concatenate([genes[8], genes[10][:400]])               # one gene's start is the tail of another
    monkeypatch.setenv("SHK_ANCHOR_ALWAYS", "1")
    for how in ("pre", "off", "none"):
        monkeypatch.delenv("SHK_NO_REFEXT", raising=False)
        monkeypatch.delenv("SHK_NO_PRE_VERDICT", raising=False)
        if how == "none":
            monkeypatch.setenv("SHK_NO_REFEXT", "1")
        if how == "off":
            monkeypatch.setenv("SHK_NO_PRE_VERDICT", "1")
        for c, single, q in ((0.6, False, 0), (0.2, True, 0), (0.85, False, 0), (0.97, False, 0), (0.6, False, 20)):
            if how != "pre" and c != 0.6:
                continue
            o, h, info = _build_both(oracle, genes, k=k, bf_bits=bf_bits, c=c, min_quality=q, single=single)
            assert h.probe_mode() in ("table", "summary+table", "table-mod"), h.probe_mode()
            for ragged in (False, True):
                batch = _pairs_with_counted_mismatches(rng, genes, 600, L1, L2, ragged, q > 0)
                goff, _ = _compare_classify(o, h, batch)
                assert goff[-1] > 0 or c > 0.9
                if "classify_uni_kernel" in h.last_kernel():
                    assert ("+pre-verdict" in h.last_kernel()) == (how == "pre"), h.last_kernel()      # (trimmed batches take the kernel too)
            h.close()


def test_minimiser_table_asks_what_the_device_has_left(oracle, monkeypatch):
    """the minimiser-bucketed table is optional and large (16 GiB on the configs[2] index): it is built at the size wanted only while the device
    keeps an eighth of its memory (8 GiB at least) free behind it, at half the size -- twice the load -- when that fits instead, and not at all
    otherwise; the index is complete either way and returns the same associations (SHK_TEST_MEM_FREE: as if that much were free)"""
    for v in ("SHK_PROBE", "SHK_NO_LDS_TABLE", "SHK_FORCE_GENERIC"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("SHK_KTAB", "1")
    monkeypatch.setenv("SHK_NO_LDS_SUMMARY", "1")
    monkeypatch.setenv("SHK_NO_SUMMARY", "1")
    rng = np.random.default_rng(4711)
    genes = synth.make_genes(rng, 40, 1_500, 3_000, share_every=5)
    batch = _sequenced_pairs(rng, genes, 600, 150, 150, False, False, 0.01, 0.001, 0.002)
    seen = {}
    for free in (None, (8 << 30) + (1 << 20), 1 << 20):      # plenty; room for a small table only; no room
        if free is None:
            monkeypatch.delenv("SHK_TEST_MEM_FREE", raising=False)
        else:
            monkeypatch.setenv("SHK_TEST_MEM_FREE", str(free))
        o, h, info = _build_both(oracle, genes, k=17, bf_bits=1 << 30, c=0.6)
        seen[free] = h.probe_mode()
        goff, _ = _compare_classify(o, h, batch)
        assert goff[-1] > 0
        h.close()
    assert seen[None] == "minimiser-table", seen
    assert seen[1 << 20] == "table", seen                     # (no room: the position table, as for every other k)
    assert seen[(8 << 30) + (1 << 20)] in ("minimiser-table", "table"), seen


# ---------------------------------------------------------------------------
# one-gene indices in LDS: the sparse first round (classify_uni.hpp, spT / sparse_first)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("L1,L2,k", [(150, 150, 17), (150, 150, 31), (140, 140, 17), (100, 100, 17), (125, 125, 12), (151, 101, 17), (101, 151, 17),
                                     (250, 0, 17), (300, 0, 21), (160, 160, 5), (150, 12, 17), (12, 150, 17)])
def test_sparse_first_round_one_gene_index(oracle, monkeypatch, L1, L2, k):
    """a one-gene index held in LDS probes the first 128 slots of a read (uniform batches and trimmed reads alike) in another order (even slots + tiles, then the
    rest) and settles a read as soon as a lower bound of its coverage passes c * len: reads from the gene with 0-12 % errors,
    chimeric reads whose coverage lands on either side of every threshold, reads with N and masked qualities, off-target reads --
    every result equals the oracle's, with the sparse order and (SHK_NO_SPARSE=1 at index build time) without it"""
    monkeypatch.delenv("SHK_NO_LDS_TABLE", raising=False)
    rng = np.random.default_rng(7700 + L1 + 3 * L2 + k)
    gene = synth.make_genes(rng, 1, 6_000, 6_000)[0]
    gene[4000:4003] = ord("N")
    if k <= 6:
        gene = gene[:900]          # (4^5 k-mers: nearly every k-mer of any read is in the filter)
    genes = [gene]
    n_assigned = n_tiles = 0
    for sparse in (True, "tiles", False):
        # ("tiles": the sparse order with the tiles' round of three staged pairs in front of it, for every batch it can serve -- uniform
        #  ones without qualities at U = 3 ... 5; True: never; the default follows the stream, test_tiles_first_follows_the_assigned_fraction)
        monkeypatch.setenv("SHK_TILE_FIRST", "1" if sparse == "tiles" else "0")
        if sparse:
            monkeypatch.delenv("SHK_NO_SPARSE", raising=False)
        else:
            monkeypatch.setenv("SHK_NO_SPARSE", "1")
        # (every threshold on the geometries users run, three of them -- the default, the quality mask, everything must match -- on the others)
        full_grid = (L1, L2, k) in ((150, 150, 17), (150, 150, 31), (100, 100, 17), (250, 0, 17))
        # (the B side -- the usual order, what every other test of this file runs -- on the default and the quality mask)
        for c, q in (((0.6, 0), (0.45, 20)) if not sparse else ((0.6, 0), (0.25, 0), (0.45, 20), (0.9, 0), (1.0, 0), (0.0, 0)) if full_grid else ((0.6, 0), (0.45, 20), (1.0, 0))):
            o, h, info = _build_both(oracle, genes, k=k, bf_bits=1 << 30, c=c, min_quality=q)
            assert h.probe_mode() == "lds-table", h.probe_mode()
            for ragged in (False, True):       # (trimmed reads run the same kernel on a one-gene index and plan per read)
                for sub_rate in (0.0, 0.02, 0.1):
                    batch = _sequenced_pairs(rng, genes, 350, L1, L2, ragged, q > 0, sub_rate, 0.002 if sub_rate else 0.0, 0.003)
                    goff, _ = _compare_classify(o, h, batch)
                    n_assigned += int(goff[-1])
                    # the A side ran the sparse order, the B side the usual one (the switch is read when the index is built)
                    if ", 21, " in h.last_kernel():
                        assert ("+sparse-first-round" in h.last_kernel()) == bool(sparse), h.last_kernel()
                        assert ("+tiles-first" in h.last_kernel()) <= (sparse == "tiles"), h.last_kernel()
                        n_tiles += "+tiles-first" in h.last_kernel()
                batch = _chimeric_batch(rng, genes, 700, L1, L2, ragged, with_n=True, qual=q > 0, k_hint=k)
                goff, _ = _compare_classify(o, h, batch)
                n_assigned += int(goff[-1])
            h.close()
    assert n_assigned > 0
    # (the tiles' round exists where three pairs share a staging pass: 2 x 100 ... 2 x 160 bp here)
    if (L1, L2, k) in ((150, 150, 17), (150, 150, 31), (140, 140, 17), (125, 125, 12), (151, 101, 17), (101, 151, 17)):
        assert n_tiles > 0, (n_tiles, L1, L2, k)


@pytest.mark.parametrize("n_genes,share,L1,L2,k", [(2, 0, 150, 150, 17), (10, 0, 150, 150, 17), (10, 3, 150, 150, 17), (6, 2, 100, 100, 17), (4, 2, 150, 150, 31),
                                                   (10, 3, 250, 0, 21), (3, 0, 151, 101, 17), (8, 4, 140, 140, 12)])
def test_sparse_first_rounds_on_indices_of_several_genes(oracle, monkeypatch, n_genes, share, L1, L2, k):
    """an index of SEVERAL genes held in LDS: the first 128 slots in the sparse order (even slots + tiles, then the rest of the
    prefix) settle a read when every match is a single-gene list of one gene, what they cover passes c * len and exceeds what the
    unprobed slots could still give any other gene -- 128 probes instead of 192 and no vote.  Genes without anything in common,
    genes sharing halves (their k-mers have multi-gene lists: the escape value sends such reads to the usual path), reads from one
    gene with 0-12 % errors, chimeric reads of two genes on either side of every bound, mates from different genes, N, quality
    masks, thresholds from 0 to 1, uniform and trimmed batches: the oracle's result with the sparse rounds -- asserted to be what
    ran -- and (SHK_NO_SPARSE=1 at index build time) without"""
    monkeypatch.delenv("SHK_NO_LDS_TABLE", raising=False)
    rng = np.random.default_rng(8800 + 10 * n_genes + L1 + k)
    genes = synth.make_genes(rng, n_genes, 1_200, 2_400, share_every=share)
    genes[0][700:702] = ord("N")
    n_assigned = n_ties = 0
    for sparse in (True, False):
        if sparse:
            monkeypatch.delenv("SHK_NO_SPARSE", raising=False)
        else:
            monkeypatch.setenv("SHK_NO_SPARSE", "1")
        grid = ((0.6, 0, False), (0.45, 20, False), (0.8, 0, True), (1.0, 0, False), (0.0, 0, False))
        if not sparse or (n_genes, share, L1) not in ((10, 3, 150), (2, 0, 150), (6, 2, 100)):
            grid = grid[:3]          # (the B side, and the A side of the other shapes: the default, the quality mask, --single)
        for c, q, single in grid:
            o, h, info = _build_both(oracle, genes, k=k, bf_bits=1 << 30, c=c, min_quality=q, single=single)
            assert h.probe_mode() == "lds-table", h.probe_mode()
            for ragged in (False, True):
                for sub_rate in (0.0, 0.02, 0.1):
                    batch = _sequenced_pairs(rng, genes, 400, L1, L2, ragged, q > 0, sub_rate, 0.002 if sub_rate else 0.0, 0.003)
                    goff, _ = _compare_classify(o, h, batch)
                    n_assigned += int(goff[-1])
                    n_ties += int((np.diff(goff.astype(np.int64)) > 1).sum())
                    if not ragged and ", 21, " in h.last_kernel():
                        assert ("+sparse-first-rounds" in h.last_kernel()) == sparse, h.last_kernel()
                batch = _chimeric_batch(rng, genes, 800, L1, L2, ragged, with_n=True, qual=q > 0, k_hint=k)
                goff, _ = _compare_classify(o, h, batch)
                n_assigned += int(goff[-1])
            h.close()
    assert n_assigned > 1000
    if share:
        assert n_ties > 0          # shared halves: reads with two genes came through (the escape path)


@pytest.mark.parametrize("L1,L2,k,n_genes", [(150, 150, 17, 1), (150, 150, 17, 5), (100, 100, 17, 1), (151, 101, 17, 1), (76, 76, 21, 1), (160, 160, 17, 1),
                                              (144, 145, 17, 1), (33, 17, 17, 1), (150, 0, 17, 1), (125, 125, 12, 3), (161, 160, 17, 1), (170, 170, 17, 1)])
def test_three_pairs_per_staging_pass(oracle, monkeypatch, L1, L2, k, n_genes):
    """the TRI instantiation of the exact-table kernel (uniform batches without qualities: a lane stages 16 bases, a wave three
    consecutive pairs per pass, mate 2 packed at 16 x ceil(L1 / 16)): batches of 1 ... 8 pairs and of 3 m - 1, 3 m, 3 m + 1 pairs (the
    last triple is short; the last reads of the batch take guarded loads), host and device-resident (the device's verdict picks
    between the two launched instantiations), mates of unequal length, a mate exactly one k-mer long, single-end, N and lower
    case anywhere incl. the bases beside a chunk boundary, lengths for which three pairs do not fit a pass or the specialisation is not compiled
    for them (2 x 170, 2 x 100, 33 + 17: the ordinary instantiation -- asserted): the oracle's result, and what ran"""
    from shark_amd.capi import hip_memcpy_dtoh
    monkeypatch.delenv("SHK_NO_LDS_TABLE", raising=False)
    rng = np.random.default_rng(9900 + L1 + 7 * L2 + k)
    genes = synth.make_genes(rng, n_genes, 2_000, 3_000, share_every=0)
    # what the library does: the specialisation for the pair's slots (8-base packing), three pairs per pass at U = 3 ... 5 when a
    # pair is at most 21 chunks of 16 bases and its slots still fit with mate 2 packed at a multiple of 16
    nk1, nk2 = max(0, L1 - k + 1), max(0, L2 - k + 1)
    u = max(2, -(-((((L1 + 7) // 8) * 8 + nk2) if nk2 else nk1) // 64))
    c1, c2 = (L1 + 15) // 16, (L2 + 15) // 16
    fits = u in (3, 4, 5) and c1 + c2 <= 21 and ((c1 * 16 + nk2) if nk2 else nk1) <= 64 * u
    assert fits == ((L1, L2) not in ((33, 17), (170, 170), (100, 100)))      # (2 x 100 bp: 112 + 84 slots do not fit U = 3's 192)
    dev = torch.device("cuda:0")
    for no_tri, tiles in ((False, False), (False, True), (True, False)):
        if no_tri:
            monkeypatch.setenv("SHK_NO_TRI", "1")
        else:
            monkeypatch.delenv("SHK_NO_TRI", raising=False)
        monkeypatch.setenv("SHK_TILE_FIRST", "1" if tiles else "0")      # (one-gene indices: the tiles' round in front of the three pairs)
        o, h, _ = _build_both(oracle, genes, k=k, bf_bits=1 << 30, c=0.5)
        assert h.probe_mode() == "lds-table"
        total = 0
        for n in (1, 2, 3, 4, 5, 7, 8, 191, 192, 193, 3000):
            b = synth.make_reads(rng, genes, n, read_len=max(L1, L2), paired=L2 > 0, on_target=0.6, n_rate=0.01, lower_rate=0.02)
            # cut the mates to (L1, L2)
            m1 = [bytes(b["seq1"][int(b["off1"][i]):int(b["off1"][i]) + L1]) for i in range(n)]
            m2 = [bytes(b["seq2"][int(b["off2"][i]):int(b["off2"][i]) + L2]) for i in range(n)] if L2 else None
            if n >= 191:                              # an N right at a chunk boundary of some reads
                m1 = [x[:15] + b"N" + x[16:] if i % 5 == 0 and len(x) > 16 else x for i, x in enumerate(m1)]
            bb = synth.batch_from_lists(m1, m2)
            goff, _ = _compare_classify(o, h, bb)
            total += int(goff[-1])
            lk = h.last_kernel()
            if ", 21, " in lk:
                assert ("+three-pairs" in lk) == (fits and not no_tri), (lk, n)
                assert ("+tiles-first" in lk) == (fits and not no_tri and tiles and n_genes == 1), (lk, n)
            # the same batch resident in HBM, in buffers that end with the last read
            t = {kk: torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(dev) for kk, v in bb.items() if v is not None}
            torch.cuda.synchronize()
            r = h.classify_device(n, t["seq1"].data_ptr(), t["off1"].data_ptr(), t["seq2"].data_ptr() if L2 else 0, t["off2"].data_ptr() if L2 else 0,
                                  max_read_len=max(L1, L2))
            dg = np.empty(n + 1, np.uint32)
            hip_memcpy_dtoh(dg, r.gene_off, dg.nbytes)
            og, _ = o.classify(bb["seq1"], bb["off1"], bb["seq2"], bb["off2"])
            assert np.array_equal(og, dg), (n, no_tri)
            if ", 21, " in h.last_kernel() and not no_tri and u in (3, 4, 5):     # (the caller's bound picks the specialisation; the device the instantiation)
                assert "+three-pairs-if-they-fit" in h.last_kernel()
        assert total > 500
        h.close()


# ---------------------------------------------------------------------------
# tiny indices: the exact table held in LDS (classify_uni_kernel LSL = 21, shark_internal.hpp LTAB_*)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("k,bf_bits,n_genes,gene_len,share", [
    (17, 1 << 33, 1, 20_000, 0),        # the bench index
    (17, 1 << 33, 4, 6_400, 0),         # 25 5xx keys: load 0.78 of the 2^15 slots
    (17, 1 << 30, 12, 1_500, 3),        # shared halves: multi-gene lists escape to the position table
    (31, 1 << 28, 3, 5_000, 0),
    (11, 1 << 24, 2, 3_000, 0),         # smallest filter that gets one; groups use 9 of their 13 bits
    (17, 1 << 33, 8_400, 19, 0),        # gene ids beyond the entry's 13 bits escape as well
])
def test_lds_table_tiny_indices(oracle, monkeypatch, k, bf_bits, n_genes, gene_len, share):
    monkeypatch.delenv("SHK_NO_LDS_TABLE", raising=False)
    rng = np.random.default_rng(31 * k + n_genes)
    genes = synth.make_genes(rng, n_genes, gene_len, gene_len, share_every=share)
    reads_from = genes if gene_len >= 300 else [np.concatenate(genes[i:i + 40]) for i in range(0, n_genes, 40)]
    for q, single in ((0, False), (20, True)):
        o, h, info = _build_both(oracle, genes, k=k, bf_bits=bf_bits, min_quality=q, single=single)
        assert h.probe_mode() == "lds-table", (h.probe_mode(), info)
        _compare_index(o, h, info)
        monkeypatch.setenv("SHK_NO_LDS_TABLE", "1")
        h0 = _hip(k=k, bf_bits=bf_bits, min_quality=q, single=single)
        h0.build([bytes(g) for g in genes])
        monkeypatch.delenv("SHK_NO_LDS_TABLE")
        assert h0.probe_mode() == "lds-summary+table"
        for L, paired, var in ((150, True, False), (100, True, False), (150, False, False), (75, True, False), (150, True, True)):
            batch = synth.make_reads(rng, reads_from, 1500, read_len=L, paired=paired, on_target=0.6, n_rate=0.004, qual=q > 0, var_len=var)
            og, oi = _compare_classify(o, h, batch)
            g0, i0 = h0.classify(batch["seq1"], batch["off1"], batch["seq2"], batch["off2"], batch["qual1"], batch["qual2"])
            assert np.array_equal(g0, og) and np.array_equal(i0, oi)
            assert og[-1] > 0 or gene_len < 300
        # every reference k-mer as a read of its own (single-end, length k): each one must be assigned, so a key the table
        # in LDS did not hold would show (among whole reads a single lost k-mer hides behind its neighbours' coverage)
        kmers = [g[i:i + k] for g in genes for i in range(0, len(g) - k + 1)]
        kb = synth.batch_from_lists(kmers, None, [b"I" * k] * len(kmers) if q > 0 else None)
        og, oi = _compare_classify(o, h, kb)
        assert single or int(og[-1]) >= len(kmers)
        g0, i0 = h0.classify(kb["seq1"], kb["off1"], kb["seq2"], kb["off2"], kb["qual1"], kb["qual2"])   # (and the position table)
        assert np.array_equal(g0, og) and np.array_equal(i0, oi)
        h.close()
        h0.close()


def test_lds_table_is_not_built_for_larger_indices(oracle):
    rng = np.random.default_rng(5)
    genes = synth.make_genes(rng, 3, 9_000, 9_000)      # ~27 000 keys > LTAB_MAX_KEYS
    h = _hip(k=17, bf_bits=1 << 33)
    h.build([bytes(g) for g in genes])
    assert h.probe_mode() == "lds-summary+table"
    h2 = _hip(k=17, bf_bits=1 << 34)                    # tag would need 19 bits
    h2.build([bytes(genes[0])])
    assert h2.probe_mode() == "lds-summary+table"
