"""`shark --gpus N` with N > 1 on the one GPU a test box has: `--devices 0,0[,0,0]` puts several workers (contexts) on one
device, so everything the N-worker command does beyond the one-worker command runs for real -- per-worker queues and the
batch i -> worker i mod N dispatch, N pipelines of `shk_classify_submit / _wait` side by side, the ordered drain across
workers (ReadOutput.hpp:37-50 with the reference's `-t 1` order), the parallel `shk_ref_finalize`, and the per-gene count
reduction (contexts sharing a device are summed on it; RCCL takes a device once per communicator).  The reference starts
its N workers from one command line as well (main.cpp:219-223) and its output does not depend on N at `-t 1`... here it
must not depend on N at all: ssv, both FASTQ files and --gene-counts byte-identical to `--gpus 1`, which the other CLI
tests pin to the truth files and to the oracle CLI.

Run on the GPU box with `pytest -m gpu`."""
import gzip
import hashlib
import os
import subprocess

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "shark_amd", "bin", "shark")
WORKERS = [(2, "0,0"), (4, "0,0,0,0")]


def _shark(args, cwd, timeout=600, env=None):
    return subprocess.run([EXE] + args, cwd=str(cwd), capture_output=True, timeout=timeout, env=env)


def _outputs(tmp, tag, args, extra, with_counts=True):
    o1, o2, gc = tmp / (tag + ".1.fq"), tmp / (tag + ".2.fq"), tmp / (tag + ".gc")
    cmd = args + ["-o", str(o1), "-p", str(o2)] + (["--gene-counts", str(gc)] if with_counts else []) + extra
    r = _shark(cmd, tmp)
    assert r.returncode == 0, (cmd, r.stderr.decode()[-2000:])
    return {"ssv": r.stdout, "o1": o1.read_bytes(), "o2": o2.read_bytes() if o2.exists() else b"",
            "gc": gc.read_bytes() if with_counts else b"", "stderr": r.stderr.decode()}


def _same(a, b, what):
    for key in ("ssv", "o1", "o2", "gc"):
        assert hashlib.md5(a[key]).hexdigest() == hashlib.md5(b[key]).hexdigest(), (what, key, len(a[key]), len(b[key]))


def test_devices_option_contract(example_dir, tmp_path):
    """--devices alone sets the number of workers; a list that disagrees with --gpus, or is not a list of numbers, is refused
    before any work; a device that does not exist is reported by its number"""
    base = ["-r", os.path.join(example_dir, "ENSG00000277117.fa"), "-1", os.path.join(example_dir, "sample_1.fq"),
            "-2", os.path.join(example_dir, "sample_2.fq"), "-o", str(tmp_path / "a"), "-p", str(tmp_path / "b")]
    r = _shark(base + ["--gpus", "2", "--devices", "0"], tmp_path)
    assert r.returncode == 1 and b"--devices must name as many devices as --gpus says" in r.stderr and not r.stdout
    for bad in ("0,x", "0,,0", ",", "-1", "0;0"):
        r = _shark(base + ["--devices", bad], tmp_path)
        assert r.returncode == 1 and b"comma separated list of device numbers" in r.stderr and not r.stdout, bad
    r = _shark(base + ["--devices", "0,77"], tmp_path)
    assert r.returncode == 1 and b"cannot create a context on GPU 77" in r.stderr and not r.stdout
    r = _shark(base + ["--devices", "0,0,0", "-v"], tmp_path)
    assert r.returncode == 0 and b"gpus=3 devices=0,0,0" in r.stderr
    assert r.stdout == open(os.path.join(example_dir, "ENSG00000277117.truth.ssv"), "rb").read()


@pytest.mark.parametrize("n,devs", WORKERS)
@pytest.mark.parametrize("extra", [["--batch", "100"], ["--batch", "777", "-t", "4"], []])
def test_example_truth_files_with_several_workers(example_dir, tmp_path, n, devs, extra):
    """(a) the bundled example (README.md:63-69): the truth files byte for byte from N workers, with batches small enough that
    every worker gets many (5 000 pairs: 50 batches of 100) and at the default batch size (one batch: N - 1 workers idle)"""
    args = ["-r", os.path.join(example_dir, "ENSG00000277117.fa"), "-1", os.path.join(example_dir, "sample_1.fq"),
            "-2", os.path.join(example_dir, "sample_2.fq")]
    got = _outputs(tmp_path, "n%d" % n, args, ["--gpus", str(n), "--devices", devs] + extra)
    assert got["ssv"] == open(os.path.join(example_dir, "ENSG00000277117.truth.ssv"), "rb").read()
    assert got["o1"] == open(os.path.join(example_dir, "sharked.sample_1.truth.fq"), "rb").read()
    assert got["o2"] == open(os.path.join(example_dir, "sharked.sample_2.truth.fq"), "rb").read()
    assert got["gc"] == b"ENSG00000277117 1929\n"
    assert "workers on 1 device(s)" in got["stderr"]


def _write_pairs(tmp, rng, genes, n, L, on_target, gz=False):
    """n pairs of L bases as two FASTQ files, numpy only: fragments of the genes (mate 2 reverse-complemented, 1 % substitutions,
    0.2 % N) and noise; genes with shared halves give ties"""
    cat = np.concatenate(genes)
    starts = np.cumsum([0] + [len(g) for g in genes])[:-1]
    lens = np.array([len(g) for g in genes])
    paths = [str(tmp / ("r1.fq" + (".gz" if gz else ""))), str(tmp / ("r2.fq" + (".gz" if gz else "")))]
    files = [(gzip.open(p, "wb", compresslevel=1) if gz else open(p, "wb")) for p in paths]
    chunk = 25_000      # (small enough that the allocator reuses the chunk's temporaries)
    col = np.arange(L, dtype=np.int32)
    nd = 9
    H = 2 + nd + 3
    W = H + L + 3 + L + 1
    for b0 in range(0, n, chunk):
        m = min(chunk, n - b0)
        g = rng.integers(0, len(genes), size=m)
        frag = np.minimum(lens[g], rng.integers(L, 3 * L, size=m))
        st = (starts[g] + (rng.random(m) * (lens[g] - frag + 1)).astype(np.int64)).astype(np.int32)
        frag = frag.astype(np.int32)
        on = rng.random(m) < on_target
        idx = np.arange(b0, b0 + m, dtype=np.int64)
        for mate in (0, 1):
            if mate == 0:
                pos = st[:, None] + col[None, :]
                seqs = cat[pos]
            else:
                pos = (st + frag - 1)[:, None] - col[None, :]
                seqs = synth.COMP[cat[pos]]
            r = rng.integers(0, 65536, size=(m, L), dtype=np.uint16)      # one draw per base: noise base, substitution, N
            noise = synth.ACGT[(r & 3).astype(np.uint8)]
            seqs = np.where(on[:, None], seqs, noise)
            seqs = np.where(r < 655, synth.ACGT[((r >> 2) & 3).astype(np.uint8)], seqs)          # 1 % substitutions
            seqs[(r >= 655) & (r < 786)] = ord("N")                                              # 0.2 % N
            rec = np.empty((m, W), dtype=np.uint8)
            rec[:, 0] = ord("@")
            rec[:, 1] = ord("r")
            for d in range(nd):
                rec[:, 2 + d] = ord("0") + (idx // 10 ** (nd - 1 - d)) % 10
            rec[:, H - 3] = ord("/")
            rec[:, H - 2] = ord("1") + mate
            rec[:, H - 1] = 10
            rec[:, H:H + L] = seqs
            rec[:, H + L] = 10
            rec[:, H + L + 1] = ord("+")
            rec[:, H + L + 2] = 10
            q = rng.integers(0, 256, size=(m, L), dtype=np.uint8)
            rec[:, H + L + 3:H + 2 * L + 3] = np.where(q < 4, 35 + (q & 3), 45 + q % 29)       # Phred 12-40, 1.6 % Phred 2-5
            rec[:, H + 2 * L + 3] = 10
            files[mate].write(rec.tobytes())
    for f in files:
        f.close()
    return paths


def _write_fasta(path, genes):
    with open(path, "w") as f:
        for i, g in enumerate(genes):
            f.write(">gene%d\n%s\n" % (i, bytes(g).decode()))


def test_two_million_pairs_with_ties(tmp_path):
    """(b) 2 M synthetic pairs against 40 genes of which every third shares half of its predecessor (two-gene lists, ties):
    N = 2 and N = 4 workers print what one worker prints -- ssv, both FASTQ files, per-gene counts -- with and without -q / -s;
    the ssv of the one-worker run is non-trivial (ties present)"""
    rng = np.random.default_rng(20260501)
    genes = synth.make_genes(rng, 40, 600, 3000, share_every=3)
    fa = tmp_path / "g.fa"
    _write_fasta(fa, genes)
    f1, f2 = _write_pairs(tmp_path, rng, genes, 2_000_000, 100, 0.4)
    for opts in ([], ["-q", "10", "-s", "-k", "21"]):
        args = ["-r", str(fa), "-1", f1, "-2", f2, "-t", "8"] + opts
        one = _outputs(tmp_path, "one", args, ["--gpus", "1"])
        lines = one["ssv"].splitlines()
        assert len(lines) > (700_000 if not opts else 300_000), (opts, len(lines))
        if not opts:
            names = [ln.split()[0] for ln in lines[:200_000]]
            assert len(set(names)) < len(names), "no read with two genes: the sample has no ties"
        for n, devs in WORKERS:
            many = _outputs(tmp_path, "many", args, ["--gpus", str(n), "--devices", devs])
            _same(one, many, (opts, n))
            # (a small batch size: each worker's three-deep pipeline turns over hundreds of times)
            if n == 4 and not opts:
                many = _outputs(tmp_path, "many", args, ["--gpus", str(n), "--devices", devs, "--batch", "5000"])
                _same(one, many, (opts, n, "batch 5000"))


@pytest.mark.parametrize("n,devs", WORKERS)
def test_irregular_records_hand_over_with_several_workers(oracle, tmp_path, n, devs):
    """(c) the irregular-record hand-over (test_cli_irregular_records_that_keep_the_four_line_alignment's files): a reader that
    is ahead must not hand its batch to ANY worker before the batches in front of it are known to be strict, and what the
    serial reader re-reads goes round the workers like everything else: the oracle CLI's bytes, run after run"""
    rng = np.random.default_rng(77)
    genes = synth.make_genes(rng, 4, 600, 1200)
    fa = tmp_path / "g.fa"
    _write_fasta(fa, genes)

    def rec(i, g, L, style):
        st = int(rng.integers(0, len(g) - L))
        s = bytes(g[st:st + L]).decode()
        q = "".join(chr(int(x)) for x in rng.integers(35, 74, size=L))
        if style == "empty":
            return "@r%d\n\n+\n\n" % i
        if style == "nul":
            return "@r%d\n%s\x00%s\n+\n%s\n" % (i, s[:30], s[31:], q)
        if style == "mismatch":
            return "@r%d\n%s\n+\n%s\n" % (i, s, q[:-7])
        return "@r%d\n%s\n+\n%s\n" % (i, s, q)

    for styles in (["s"] * 700 + ["empty"] + ["s"] * 1500,
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
            got = _outputs(tmp_path, "h", args, ["--batch", "64", "-t", "8", "--gpus", str(n), "--devices", devs])
            assert got["ssv"] == want, rep
            assert got["o1"] == (tmp_path / "o1.fq").read_bytes()
            assert got["o2"] == (tmp_path / "o2.fq").read_bytes()
            assert {ln.split()[0]: int(ln.split()[1]) for ln in got["gc"].splitlines()} == want_counts


@pytest.mark.parametrize("n,devs", WORKERS)
def test_gzip_sample_with_several_workers(oracle, tmp_path, n, devs):
    """(d) a gzip sample (parallel inflate, cut and parsed from memory) feeding N workers: the oracle CLI's bytes on a
    60 000-pair prefix-sized sample, and the one-worker command's on the same files at small chunks"""
    rng = np.random.default_rng(4711)
    genes = synth.make_genes(rng, 12, 500, 2000, share_every=4)
    fa = tmp_path / "g.fa"
    _write_fasta(fa, genes)
    f1, f2 = _write_pairs(tmp_path, rng, genes, 60_000, 100, 0.5, gz=True)
    args = ["-r", str(fa), "-1", f1, "-2", f2, "-k", "17"]
    ossv = tmp_path / "o.ssv"
    oracle.run_cli(args + ["-o", str(tmp_path / "o1.fq"), "-p", str(tmp_path / "o2.fq")], str(ossv))
    want = ossv.read_bytes()
    assert want.count(b"\n") > 20_000
    env = dict(os.environ, SHARK_GZ_CHUNK="65536")
    for extra, e in ((["--batch", "2000", "-t", "8"], env), (["-t", "4"], None)):
        o1, o2 = tmp_path / "h1.fq", tmp_path / "h2.fq"
        r = _shark(args + ["-o", str(o1), "-p", str(o2), "--gpus", str(n), "--devices", devs] + extra, tmp_path, env=e)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert r.stdout == want
        assert o1.read_bytes() == (tmp_path / "o1.fq").read_bytes()
        assert o2.read_bytes() == (tmp_path / "o2.fq").read_bytes()


def test_gene_counts_allreduce_with_shared_devices(oracle):
    """the library form behind --gene-counts: four contexts on device 0, each classifies its own quarter of a batch;
    shk_gene_counts_allreduce over the four = the histogram of the whole batch's gene lists = the oracle's; every context's
    totals buffer holds the totals; repeating the call gives the same totals (the per-context counters stay local)"""
    from shark_amd import SharkHip
    rng = np.random.default_rng(99)
    genes = synth.make_genes(rng, 30, 300, 1500, share_every=3)
    ctxs = [SharkHip(k=15, c=0.5, bf_bits=1 << 26, device=0) for _ in range(4)]
    for h in ctxs:
        h.build([bytes(g) for g in genes])
    o = oracle.Shark(k=15, c=0.5, bf_bits=1 << 26)
    o.build([bytes(g) for g in genes])
    want = np.zeros(len(genes), dtype=np.uint64)
    for q, h in enumerate(ctxs):
        b = synth.make_reads(np.random.default_rng(1000 + q), genes, 3000, read_len=100, on_target=0.6)
        goff, gids = h.classify(b["seq1"], b["off1"], b["seq2"], b["off2"])
        ogoff, ogids = o.classify(b["seq1"], b["off1"], b["seq2"], b["off2"], nthreads=2)
        assert np.array_equal(goff, ogoff) and np.array_equal(gids, ogids)
        want += np.bincount(ogids, minlength=len(genes)).astype(np.uint64)
    assert want.sum() > 5000
    for rep in range(2):
        tot = ctxs[0].gene_counts_allreduce(ctxs[1:], len(genes))
        assert np.array_equal(tot, want), rep
    for h in ctxs:
        mine = h.gene_counts(len(genes))
        assert mine.sum() < want.sum()          # local counters are still local
