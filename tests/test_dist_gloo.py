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


def _run(world, out):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
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
