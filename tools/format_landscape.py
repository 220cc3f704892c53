#!/usr/bin/env python3
"""The text tables of profiles/r0N_landscape.txt from the JSON lines tools/gpu_landscape.sh leaves in a directory:
python tools/format_landscape.py gpurun_out/r5land 'round 5: ...title...' > profiles/r05_landscape.txt"""
import json, os, sys
d, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "kernel time per 10 M pairs")
SECTIONS = [("sizes", "index sizes at 50 % on-target"), ("on_target", "on-target rates 0 / 100 %"),
            ("pre_ab", "anchor_verdict_kernel in front of the table kernels on / off (SHK_NO_PRE_VERDICT=1)"),
            ("ktab_ab", "the minimiser table on / off (SHK_NO_KTAB=1 at index build time: the position table), 10 000 genes at 2^33 bits and the configs[2] index"),
            ("anchor_ab", "the anchored extension on / off (SHK_NO_ANCHOR=1)"), ("sparse_ab", "the sparse first round(s) on / off (SHK_NO_SPARSE=1): one gene"),
            ("sparse_multi_ab", "... 9 genes (nothing shared) and 10 genes (one sharing half of another)"),
            ("len300", "2 x 300 bp (5 M pairs)"), ("len250", "2 x 250 bp (5 M pairs)"), ("len100", "2 x 100 bp"), ("k31q20", "the configs[4] shape: k = 31, -q 20, single-end mode, 2^37 bits"),
            ("ragged", "trimmed reads (tools/ragged_rate.py)")]
print("# " + title + "\n")
for name, head in SECTIONS:
    f = os.path.join(d, name + ".jsonl")
    if not os.path.exists(f) or not os.path.getsize(f):
        continue
    print("## " + head)
    for ln in open(f):
        ln = ln.strip()
        if not ln.startswith("{"):
            continue
        r = json.loads(ln)
        if "kernel_ms" in r and "genes" in r and "bf_log2" in r:
            on = r.get("anchored", r.get("with"))
            print("genes %-6d bf 2^%d k %2d q %2d on-target %.2f read_len %3d pairs %d mode %-22s kernel_ms %7.3f n_assoc %d%s" % (
                r["genes"], r["bf_log2"], r["k"], r["q"], r["on_target"], r["read_len"], r["pairs"], r["mode"], r["kernel_ms"], r["n_assoc"],
                "" if on is None else ("  [on]" if on else "  [off]")))
        elif "anchored_equals_plain" in r:
            print("   -> results equal: %s" % r["anchored_equals_plain"])
        else:
            print("   " + json.dumps(r))
    print()
