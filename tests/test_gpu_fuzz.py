"""The randomised differential test (tests/fuzz_parity.py) as part of the `-m gpu` suite: a fixed-seed schedule of random
configurations, HIP path (host and device-resident entry points) against the CPU oracle, bit-exact, every case also probing
the reference's own k-mers one by one.  The schedule is biased until every probe structure shk_probe_mode() can report has
been drawn -- a structure that is never drawn is a structure that is never tested."""
import collections

import pytest

pytestmark = pytest.mark.gpu

SCHEDULE = [("", 20261003, 72), ("uni", 20270000, 44), ("mid", 20280000, 18), ("mod", 20290000, 14), ("ktab", 20300000, 18),
            ("trim", 8880001, 24), ("pre", 9990001, 32)]      # (round 6: trimmed batches by offsets; anchor_verdict_kernel in front of the table kernels)


def test_fuzz_schedule_covers_every_probe_mode():
    from tests import fuzz_parity
    modes, kernels = collections.Counter(), collections.Counter()
    n = 0
    for bias, seed0, count in SCHEDULE:
        for it in range(count):
            ok, mode, desc = fuzz_parity.run_case(seed0 + it, bias)
            assert ok, "MISMATCH: %s  (replay: python tests/fuzz_parity.py 1 %d %s)" % (desc, seed0 + it, bias)
            modes[mode] += 1
            kernels.update(desc.rsplit("feat=", 1)[1].split("+"))
            n += 1
    print("fuzz: %d cases, probe modes %s" % (n, dict(modes)))
    print("fuzz: kernels of the device-resident calls %s" % dict(kernels))
    # (the paths that are chosen per batch on the device: drawn as well, or the schedule proves nothing about them)
    for f in ("offsets", "pre", "classes", "tiles", "anch", "tri"):
        assert kernels[f] > 0, "never drawn: %s (%s)" % (f, dict(kernels))
    missing = [m for m in fuzz_parity.ALL_MODES if modes[m] == 0]
    assert not missing, "probe modes never drawn: %s (histogram %s)" % (missing, dict(modes))
