#!/usr/bin/env python3
"""Generates tests/golden/xxh64_kat.json: XXH64 known answers from python-xxhash
(the canonical XXH64 by Y. Collet, which xxhash.hpp -- RedSpah xxhash_cpp 0.6.5
-- implements).  Pins the only hash the hot path uses: XXH64 of the 8
little-endian bytes of a k-mer, seed 0 (kmer_utils.hpp:81-83), plus a few
other lengths/seeds so the oracle's full restatement is covered."""
import json
import os
import random

import xxhash

rng = random.Random(20200901)
u64 = []
for v in [0, 1, 2, 3, 0x3ffffffff, 0x3fffffffffffffff, 0xffffffffffffffff, 0x1e8a496ed]:
    u64.append([v, xxhash.xxh64_intdigest(v.to_bytes(8, "little"), seed=0)])
for bits in (2, 10, 34, 62, 64):
    for _ in range(40):
        v = rng.getrandbits(bits)
        u64.append([v, xxhash.xxh64_intdigest(v.to_bytes(8, "little"), seed=0)])
byt = []
for n in (0, 1, 3, 4, 7, 8, 9, 15, 16, 31, 32, 33, 63, 64, 100):
    for seed in (0, 1, 0x9E3779B185EBCA87):
        b = bytes(rng.getrandbits(8) for _ in range(n))
        byt.append([b.hex(), seed, xxhash.xxh64_intdigest(b, seed=seed)])
out = {"generator": "python-xxhash %s" % xxhash.VERSION, "u64_le_seed0": u64, "bytes": byt}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "xxh64_kat.json"), "w"))
print(len(u64), len(byt))
