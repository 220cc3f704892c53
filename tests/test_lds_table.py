"""The LDS-resident exact table of a tiny index (shark_amd/csrc/lds_table.hpp), checked on the CPU through the host-only
tool shark_amd/bin/shark-ltab-check: the image is built for random key sets and queried with the lookup rule the kernel
uses -- every key must be found with its payload, and a million random / near-miss hashes must be answered exactly as a
set of the keys answers them."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "shark_amd", "bin", "shark-ltab-check")


@pytest.fixture(scope="module")
def tool():
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-C", os.path.join(ROOT, "shark_amd", "csrc"), "../bin/shark-ltab-check"], check=True, stdout=subprocess.DEVNULL)
    return TOOL


@pytest.mark.parametrize("n,lgb", [(1, 33), (100, 24), (6000, 24), (15000, 28), (19984, 33), (20000, 33), (23000, 31), (26000, 30), (26000, 33)])
def test_every_key_is_found_and_nothing_else(tool, n, lgb):
    # several key sets per shape: about one set in three of 20 000 keys has two keys of a group on one base slot for a given
    # multiplier (the case the multiplier retries exist for)
    for seed in range(1, 9):
        r = json.loads(subprocess.run([tool, str(n), str(lgb), str(1000 * lgb + seed), "300000"], check=True, capture_output=True, text=True).stdout)
        assert r["built"], r
        assert r["slots_used"] == n == r["keys"], r
        assert r["missing"] == 0 and r["wrong_payload"] == 0 and r["false_pos"] == 0 and r["false_neg"] == 0, r
