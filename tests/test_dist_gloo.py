"""N>1 path on CPU: world_size-2 (and 3) gloo jobs shard the batches and
all-reduce per-gene counts; the result must equal the single-process run."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, args):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py")] + args, env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0


def _run(world, out):
    _launch(world, ["batches", out])
    return json.load(open(out))


@pytest.mark.timeout(600)
def test_sharded_counts_equal_single_process(oracle, tmp_path):
    one = _run(1, str(tmp_path / "w1.json"))
    two = _run(2, str(tmp_path / "w2.json"))
    three = _run(3, str(tmp_path / "w3.json"))
    assert one["lines"] > 500
    assert two["counts"] == one["counts"] and two["lines"] == one["lines"]
    assert three["counts"] == one["counts"] and three["lines"] == one["lines"]
    assert abs(two["tmax"] - 0.2) < 1e-9 and abs(three["tmax"] - 0.3) < 1e-9   # MAX over ranks


def test_batch_ownership_is_a_partition():
    from shark_amd import dist as sdist
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            seen += sdist.shard_batches(1003, 100, r, world)
        assert sorted(seen) == [(i, min(1003, i + 100)) for i in range(0, 1003, 100)]


@pytest.mark.timeout(600)
def test_byte_range_partition_reproduces_the_single_process_ssv(oracle, tmp_path):
    """the sharded run over FILES: the CLI's own record-aligned byte ranges (fastq_partition.hpp), batch i -> rank i mod N,
    ordered merge on rank 0; world sizes 1, 2 and 3 must all give the oracle CLI's ssv byte for byte"""
    import numpy as np
    from tests import synth
    rng = np.random.default_rng(77)
    genes = synth.make_genes(rng, 12, 300, 1200, share_every=4)
    fa = tmp_path / "g.fa"
    fa.write_text("".join(">gene%d some description\n%s\n" % (i, bytes(g).decode()) for i, g in enumerate(genes)))
    b = synth.make_reads(rng, genes, 1700, read_len=100, paired=True, on_target=0.7, var_len=True)
    f1, f2 = tmp_path / "s_1.fq", tmp_path / "s_2.fq"
    for f, seq, off, tag in ((f1, b["seq1"], b["off1"], 1), (f2, b["seq2"], b["off2"], 2)):
        with open(f, "wb") as fh:
            for i in range(len(off) - 1):
                s_ = bytes(seq[int(off[i]):int(off[i + 1])])
                fh.write(b"@pair%d/%d len=%d\n%s\n+\n%s\n" % (i, tag, len(s_), s_, b"I" * len(s_)))
    want = tmp_path / "want.ssv"
    oracle.run_cli(["-r", str(fa), "-1", str(f1), "-2", str(f2), "-k", "15", "-c", "0.5", "-o", str(tmp_path / "o1"), "-p", str(tmp_path / "o2")],
                   str(want))
    assert want.stat().st_size > 10000
    for world, batch in ((1, 400), (2, 256), (3, 100)):
        out = tmp_path / ("w%d.ssv" % world)
        _launch(world, ["files", str(out), str(fa), str(f1), str(f2), str(batch)])
        assert out.read_bytes() == want.read_bytes(), world
        meta = json.load(open(str(out) + ".json"))
        assert meta["batches"] == (1700 + batch - 1) // batch
        assert sum(meta["counts"]) == want.read_bytes().count(b"\n")


def test_bench_read_set_shards_are_a_partition_at_every_rank_count():
    """bench.py's strong-scaling split (the same read set whatever N): at 1, 2, 4, 8 GPUs -- and the odd counts -- the ranks' chunk
    lists tile the chunk ids, every rank gets equally many, and at the default sizes (80 M pairs, 10 M per launch) the chunks are
    the SAME 8 chunks at every power of two, so gene_count_checksum cannot depend on N.  (The 8-rank run itself is the driver's: a
    one-GPU box takes at most six processes on its card, so the GPU suite rehearses 4 ranks over gloo.)"""
    import importlib.util
    root = os.path.dirname(HERE)
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for world in (1, 2, 3, 4, 5, 6, 7, 8):
        seen, per_pass, cp0 = [], None, None
        for r in range(world):
            cp, mine, allp = bench.shard_chunks("strong", 80_000_000, 10_000_000, world, r)
            seen += mine
            assert per_pass in (None, allp) and cp0 in (None, cp)
            per_pass, cp0 = allp, cp
            assert len(mine) == len(seen) // (r + 1)
        assert sorted(seen) == list(range(len(seen))) and cp0 * len(seen) == per_pass
        if world in (1, 2, 4, 8):
            assert cp0 == 10_000_000 and len(seen) == 8 and per_pass == 80_000_000
        cp, mine, allp = bench.shard_chunks("weak", 80_000_000, 10_000_000, world, world - 1)
        assert (cp, mine, allp) == (10_000_000, [world - 1], 10_000_000 * world)
