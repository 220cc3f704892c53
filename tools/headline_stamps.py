#!/usr/bin/env python3
"""The cycle budget of the headline kernel (classify_uni_kernel, exact table in LDS, three pairs per staging pass) by phase, from the
s_memtime stamps of a diagnostic build (-DSHK_STAMPS=1 of classify_uni_u5.hip, see tools/stamps.sh): every wave sums the clock
between its phase boundaries, the host reads the sums back.  Run with SHK_LIB_PATH pointing at that build:

    SHK_LIB_PATH=$PWD/tools/variants/stamps.so python3 tools/headline_stamps.py [--ot 0,0.5,1] > profiles/r05_headline_stamps.json

The workload is bench.py's (BASELINE configs[1]: 1 gene x 20 kb, k = 17, 2^33 bits, 10 M pairs 2 x 150 bp).  Per on-target rate:
ticks per pair and phase (a wave's clock, so 4 waves per SIMD overlap: the sum over phases = the wave's whole residence, which the
script checks against launch time x waves), the share of each phase, and the kernel time of the stamped build next to the
product build's (what the stamps themselves cost)."""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shark_amd import SharkHip, synth, capi

PHASES = ["loop head: next triple's loads issued", "staging: 16 bases per lane -> code streams in LDS, wave barrier",
          "a pair's set-up (threshold, invalid-character ballot, plan)", "round A: windows, canonical form, XXH64, D[g], T[slot], ballot",
          "round A: validation, coverage, verdict (97 % of the on-target pairs end here)",
          "round B: windows, canonical form, XXH64, D[g], T[slot], ballot",
          "rest of a pair: bound cut / usual order / vote / result write", "loop tail: retire the next triple's loads"]
EXTRA = {10: "(set-up) from the last pair's end to this pair's staging area", 11: "(set-up) invalid-character ballot, threshold",
         12: "(set-up) entry of the compile-time plan: table constants, state"}

ap = argparse.ArgumentParser()
ap.add_argument("--ot", default="0,0.5,1")
ap.add_argument("--pairs", type=int, default=10_000_000)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
lib = capi.load()
if not hasattr(lib, "shk_debug_read_stamps"):
    sys.exit("this library has no stamps: build tools/variants/stamps.so (tools/stamps.sh) and set SHK_LIB_PATH")
lib.shk_debug_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
lib.shk_debug_read_stamps.restype = ctypes.c_int
lib.shk_debug_read_wave_times.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
lib.shk_debug_read_wave_times.restype = ctypes.c_int


def read_stamps(reset=True):
    buf = (ctypes.c_ulonglong * 16)()
    assert lib.shk_debug_read_stamps(buf, 1 if reset else 0) == 0
    return [int(x) for x in buf]


dev = torch.device("cuda:0")
genes = synth.make_reference(1, 20000)
out = {"what": __doc__.split("\n\n")[0], "k": 17, "bf_log2": 33, "pairs": a.pairs, "phases": PHASES, "on_target": {}}
props = torch.cuda.get_device_properties(0)
for ot in [float(x) for x in a.ot.split(",")]:
    b = synth.make_pairs_device(a.pairs, genes, dev, seed=synth.SEED + 7, read_len=150, on_target=ot, with_qual=False, sub_rate=0.01)
    torch.cuda.synchronize()
    ptr = {k: (v.data_ptr() if v is not None else 0) for k, v in b.items()}
    h = SharkHip(k=17, c=0.6, bf_bits=1 << 33)
    h.build([g.tobytes() for g in genes])
    h.classify_device(a.pairs, ptr["seq1"], ptr["off1"], ptr["seq2"], ptr["off2"], 0, 0, max_read_len=150)
    read_stamps(True)
    h.timing_enable(True)
    for _ in range(a.reps):
        r = h.classify_device(a.pairs, ptr["seq1"], ptr["off1"], ptr["seq2"], ptr["off2"], 0, 0, max_read_len=150)
    tm = h.timing()
    st = read_stamps(True)
    wt = (ctypes.c_ulonglong * 8192)()
    assert lib.shk_debug_read_wave_times(wt) == 0
    wt = np.array(wt, dtype=np.uint64).reshape(4096, 2).astype(np.int64)
    wt = wt[wt[:, 1] > 0]
    t_first = int(wt[:, 0].min())
    starts, ends = (wt[:, 0] - t_first) * 0.01, (wt[:, 1] - t_first) * 0.01      # microseconds behind the first wave's start
    wg_end = [round(float(x), 1) for x in ends.reshape(-1, 16).max(axis=1)] if len(ends) % 16 == 0 else []
    pct = lambda v: [round(float(np.percentile(v, q)), 1) for q in (0, 10, 50, 90, 100)]
    kernel_ms = tm["total_ms"] / tm["n_launches"]
    ticks = [x / a.reps for x in st[:8]]
    extra = {EXTRA[i]: st[i] / a.reps for i in EXTRA}
    ticks[2] += sum(extra.values())      # (the finer stamps split phase 2)
    triples, pairs = st[8] / a.reps, st[9] / a.reps
    total = sum(ticks)
    n_waves = props.multi_processor_count * 16
    row = {"kernel": h.last_kernel(), "kernel_ms_stamped_build": round(kernel_ms, 3), "n_assoc": int(r.n_assoc), "pairs_stamped": pairs, "triples_stamped": triples,
           "waves": n_waves, "ticks_per_wave": round(total / n_waves, 1),
           "shader_clock_GHz": round(0.1 * st[13] / st[14], 4) if st[14] else None,      # (s_memtime over s_memrealtime, 100 MHz, per wave around its loop)
           "wave_residence_of_launch": round((st[14] / a.reps / n_waves) / (kernel_ms * 1e5), 4) if st[14] else None,
           "ticks_per_pair": {PHASES[i]: round(ticks[i] / pairs, 2) for i in range(8)},
           "share": {PHASES[i]: round(ticks[i] / total, 4) for i in range(8)},
           "wave_loop_start_us_behind_the_first_p0_10_50_90_100": pct(starts), "wave_loop_end_us_p0_10_50_90_100": pct(ends),
           "wave_loop_us_p0_10_50_90_100": pct(ends - starts),
           "workgroup_end_us_by_blockIdx": wg_end,
           "set_up_split_ticks_per_pair": {k: round(v / pairs, 2) for k, v in extra.items()},
           "ticks_per_pair_total": round(total / pairs, 2)}
    out["on_target"]["%.2f" % ot] = row
    print("[stamps] on-target %.2f: %.3f ms, %s GHz, %s %s %s" % (ot, kernel_ms, row["shader_clock_GHz"], {k: row[k] for k in row if k.startswith("wave_loop")}, {k[:14]: v for k, v in row["share"].items()}, row["set_up_split_ticks_per_pair"]), file=sys.stderr, flush=True)
    h.close()
    del b
print(json.dumps(out, indent=1))
