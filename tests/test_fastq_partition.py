"""CPU tests of the record-aligned byte-range partition behind the `shark` CLI's parallel readers
(shark_amd/csrc/fastq_partition.hpp, through the host-only tool shark_amd/bin/shark-fastq-parts)."""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "shark_amd", "bin", "shark-fastq-parts")


@pytest.fixture(scope="module")
def tool():
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-C", os.path.join(ROOT, "shark_amd", "csrc"), "../bin/shark-fastq-parts"], check=True, stdout=subprocess.DEVNULL)
    return TOOL


def parts(tool, batch, threads, *files):
    r = subprocess.run([tool, str(batch), str(threads)] + [str(f) for f in files], capture_output=True, text=True, check=True)
    return json.loads(r.stdout)


def write_fastq(path, n, rng, tag, min_len=30, max_len=160, id_extra=True):
    recs = []
    with open(path, "wb") as f:
        for i in range(n):
            L = int(rng.integers(min_len, max_len + 1))
            seq = bytes(rng.choice(list(b"ACGTN"), size=L).astype(np.uint8))
            qual = bytes(rng.integers(33, 74, size=L).astype(np.uint8))
            rid = b"read%d/%d" % (i, tag)
            hdr = rid + (b" extra words" if id_extra and i % 3 == 0 else b"")
            f.write(b"@" + hdr + b"\n" + seq + b"\n+\n" + qual + b"\n")
            recs.append((rid, seq, qual))
    return recs


def parse_range(path, b, e):
    """independent parse of a strict four-line byte range"""
    data = open(path, "rb").read()[b:e]
    lines = data.split(b"\n")
    assert lines[-1] == b""
    lines = lines[:-1]
    assert len(lines) % 4 == 0
    return [(lines[i][1:].split()[0], lines[i + 1], lines[i + 3]) for i in range(0, len(lines), 4)]


@pytest.mark.parametrize("batch,threads", [(1, 1), (7, 3), (64, 8), (1000, 4), (5000, 2)])
def test_partition_tiles_the_pair_stream(tool, tmp_path, batch, threads):
    rng = np.random.default_rng(batch)
    r1 = write_fastq(tmp_path / "a_1.fq", 1234, rng, 1)
    r2 = write_fastq(tmp_path / "a_2.fq", 1234, rng, 2, min_len=20, max_len=90)   # mates of other lengths: other byte offsets
    t = parts(tool, batch, threads, tmp_path / "a_1.fq", tmp_path / "a_2.fq")
    assert t["ok"] and t["n_records"] == 1234
    got1, got2, pos1, pos2 = [], [], 0, 0
    for b1, e1, b2, e2, n, regular in t["batches"]:
        assert regular and b1 == pos1 and b2 == pos2
        p1, p2 = parse_range(tmp_path / "a_1.fq", b1, e1), parse_range(tmp_path / "a_2.fq", b2, e2)
        assert len(p1) == len(p2) == n
        got1 += p1
        got2 += p2
        pos1, pos2 = e1, e2
    assert got1 == r1 and got2 == r2
    assert pos1 == os.path.getsize(tmp_path / "a_1.fq") and pos2 == os.path.getsize(tmp_path / "a_2.fq")
    assert [x[4] for x in t["batches"]] == [min(batch, 1234 - i) for i in range(0, 1234, batch)]


def test_pair_stream_ends_with_the_shorter_file(tool, tmp_path):
    rng = np.random.default_rng(5)
    write_fastq(tmp_path / "a_1.fq", 500, rng, 1)
    write_fastq(tmp_path / "a_2.fq", 333, rng, 2)
    t = parts(tool, 100, 4, tmp_path / "a_1.fq", tmp_path / "a_2.fq")
    assert t["ok"] and t["n_records"] == 333 and t["records_1"] == 500 and t["records_2"] == 333
    assert [x[4] for x in t["batches"]] == [100, 100, 100, 33]
    b1, e1 = t["batches"][-1][0], t["batches"][-1][1]
    assert len(parse_range(tmp_path / "a_1.fq", b1, e1)) == 33          # exactly the 33 records the last pairs need


def test_irregular_records_are_noticed_by_the_batch_that_owns_them(tool, tmp_path):
    rng = np.random.default_rng(9)
    write_fastq(tmp_path / "a.fq", 400, rng, 1)
    lines = open(tmp_path / "a.fq", "rb").read().split(b"\n")
    # record 250: sequence and quality wrapped over two lines each (legal FASTQ, not four-line)
    i = 250 * 4
    s, q = lines[i + 1], lines[i + 3]
    lines[i:i + 4] = [lines[i], s[:10], s[10:], b"+", q[:10], q[10:]]
    open(tmp_path / "b.fq", "wb").write(b"\n".join(lines))
    t = parts(tool, 100, 4, tmp_path / "b.fq")
    flags = [x[5] for x in t["batches"]]
    assert flags[0] and flags[1] and not flags[2]                      # batches 0,1 are strict; batch 2 holds the wrapped record
    # (what lies behind the first irregular batch is never used: the serial reader takes over there)
    # CR/LF line ends, a missing final newline, gzip: not for the parallel readers
    open(tmp_path / "c.fq", "wb").write(open(tmp_path / "a.fq", "rb").read().replace(b"\n", b"\r\n"))
    assert not any(x[5] for x in parts(tool, 100, 2, tmp_path / "c.fq")["batches"])
    open(tmp_path / "d.fq", "wb").write(open(tmp_path / "a.fq", "rb").read()[:-1])
    t = parts(tool, 100, 2, tmp_path / "d.fq")
    assert t["n_records"] == 399 and all(x[5] for x in t["batches"])   # the unterminated last record is left to the serial reader
    with gzip.open(tmp_path / "e.fq.gz", "wb") as f:
        f.write(open(tmp_path / "a.fq", "rb").read())
    assert parts(tool, 100, 2, tmp_path / "e.fq.gz") == {"ok": False}
    open(tmp_path / "empty.fq", "wb").close()
    t = parts(tool, 100, 2, tmp_path / "empty.fq")
    assert t["ok"] and t["n_records"] == 0 and t["batches"] == []


def test_reader_delivers_the_same_records_from_plain_gzip_and_bgzf(tool, tmp_path):
    """the serial kseq-rule reader behind a read-ahead thread: plain file, ordinary gzip (inflated ahead of the parser) and
    BGZF (blocks inflated in parallel) must deliver identical records; a gzip member behind BGZF blocks is read too"""
    import struct
    import zlib
    rng = np.random.default_rng(3)
    write_fastq(tmp_path / "t.fq", 30000, rng, 1)
    data = open(tmp_path / "t.fq", "rb").read()
    with gzip.open(tmp_path / "t.fq.gz", "wb", compresslevel=4) as f:
        f.write(data)

    def bgzf(payload, path, blk=60000, eof=True):
        with open(path, "wb") as f:
            for o in list(range(0, len(payload), blk)) + ([None] if eof else []):
                chunk = b"" if o is None else payload[o:o + blk]
                c = zlib.compressobj(6, zlib.DEFLATED, -15)
                d = c.compress(chunk) + c.flush()
                f.write(struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord("B"), ord("C"), 2, len(d) + 25))
                f.write(d)
                f.write(struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    bgzf(data, tmp_path / "t.bgzf.gz")
    assert gzip.open(tmp_path / "t.bgzf.gz").read() == data                     # a valid gzip file for everybody else
    bgzf(data, tmp_path / "t.noeof.gz", eof=False)
    with open(tmp_path / "t.mixed.gz", "wb") as f:
        f.write(open(tmp_path / "t.noeof.gz", "rb").read())
        f.write(gzip.compress(b"@tail\nACGT\n+\nIIII\n"))

    def rec(path):
        return json.loads(subprocess.run([tool, "--records", str(path)], capture_output=True, text=True, check=True).stdout)
    want = rec(tmp_path / "t.fq")
    assert want["ok"] and want["records"] == 30000 and not want["bgzf"]
    for name, is_bgzf in (("t.fq.gz", False), ("t.bgzf.gz", True), ("t.noeof.gz", True)):
        got = rec(tmp_path / name)
        assert (got["records"], got["bases"], got["fnv"], got["bgzf"]) == (want["records"], want["bases"], want["fnv"], is_bgzf), name
    mixed = rec(tmp_path / "t.mixed.gz")
    assert mixed["records"] == 30001 and mixed["bases"] == want["bases"] + 4
    # a corrupted block ends the stream instead of delivering garbage
    raw = bytearray(open(tmp_path / "t.bgzf.gz", "rb").read())
    raw[len(raw) // 2] ^= 0xFF
    open(tmp_path / "t.bad.gz", "wb").write(bytes(raw))
    assert rec(tmp_path / "t.bad.gz")["records"] < 30000


# ---------------------------------------------------------------------------
# the lean reader (fastq_lean_reader.hpp): what the CLI's reader threads copy and what its output stage reads back
# ---------------------------------------------------------------------------
def lean(tool, batch, path, qual=False):
    r = subprocess.run([tool, "--lean", str(batch), str(path)] + (["qual"] if qual else []), capture_output=True, text=True, check=True)
    return json.loads(r.stdout)


def serial(tool, path):
    r = subprocess.run([tool, "--records", str(path)], capture_output=True, text=True, check=True)
    return json.loads(r.stdout)


def write_fixed(path, n, rng, L=100, name_digits=6, plus_name=False):
    with open(path, "wb") as f:
        for i in range(n):
            seq = bytes(rng.choice(list(b"ACGTN"), size=L).astype(np.uint8))
            qual = bytes(rng.integers(33, 74, size=L).astype(np.uint8))     # includes '@', '+', '>' inside the line
            rid = b"r%0*d/1" % (name_digits, i)
            f.write(b"@" + rid + b" x\n" + seq + b"\n+" + (rid if plus_name else b"") + b"\n" + qual + b"\n")


@pytest.mark.parametrize("batch", [1, 7, 64, 1000, 100000])
@pytest.mark.parametrize("kind", ["fixed", "fixed_plus_name", "variable", "long_records"])
def test_lean_reader_delivers_what_the_serial_reader_delivers(tool, tmp_path, batch, kind):
    """strict four-line files of one layout (the 32-bytes-at-a-time check), of many layouts (line by line), with records longer
    than the reader's 1 MiB buffer: names, sequences and qualities equal the serial kseq-rule reader's, whether the output
    stage fetches a record alone or a run of records at once"""
    rng = np.random.default_rng(11 + batch)
    p = tmp_path / "a.fq"
    if kind == "fixed":
        write_fixed(p, 3000, rng)
    elif kind == "fixed_plus_name":
        write_fixed(p, 3000, rng, L=31, plus_name=True)
    elif kind == "variable":
        write_fastq(p, 3000, rng, 1)
    else:
        write_fastq(p, 40, rng, 1, min_len=200_000, max_len=700_000, id_extra=False)
    want = serial(tool, p)
    for qual in (False, True):
        got = lean(tool, batch, p, qual)
        assert got["ok"] and got["first_irregular_batch"] == -1, got
        assert (got["records"], got["bases"], got["fnv"]) == (want["records"], want["bases"], want["fnv"])
        if kind.startswith("fixed"):
            assert got["fixed_width_file"] and got["fixed_width_batches"] == got["batches"]


@pytest.mark.parametrize("damage", ["empty_read", "length_mismatch", "lone_cr", "nul_in_seq", "nul_in_name", "seq_starts_with_at", "extra_line",
                                    "missing_plus", "two_half_records"])
def test_lean_reader_stops_at_the_first_irregular_batch(tool, tmp_path, damage):
    """every irregular record -- also those that keep the four-line rhythm or the file's fixed width -- makes its batch (and
    nothing before it) irregular, with the fast layout check exactly as with the line-by-line path"""
    rng = np.random.default_rng(5)
    L, n, bad = 60, 2000, 1234
    recs = []
    for i in range(n):
        seq = bytes(rng.choice(list(b"ACGT"), size=L).astype(np.uint8))
        qual = bytes(rng.integers(35, 74, size=L).astype(np.uint8))
        recs.append([b"@r%05d" % i, seq, b"+", qual])
    r = recs[bad]
    if damage == "empty_read":
        r[1], r[3] = b"", b""
    elif damage == "length_mismatch":
        r[3] = r[3][:-5] + b"\n" + r[3][-4:]           # same bytes, one more line: the width stays
        r[3] = r[3].replace(b"\n", b"")[:-3]
    elif damage == "lone_cr":
        r[1] = r[1][:-1] + b"\r"
    elif damage == "nul_in_seq":
        r[1] = r[1][:10] + b"\x00" + r[1][11:]
    elif damage == "nul_in_name":
        r[0] = b"@r\x00" + r[0][3:]
    elif damage == "seq_starts_with_at":
        r[1] = b"@" + r[1][1:]
    elif damage == "extra_line":
        r[1] = r[1][:30] + b"\n" + r[1][31:]            # a newline inside the sequence: same width, five lines
    elif damage == "missing_plus":
        r[2] = b"-"
    elif damage == "two_half_records":
        W = len(b"\n".join(r)) + 1                      # two records in the width of one: 2 * (8 + 2 h + 4) + pad = W
        h = (W - 24) // 4
        a = [b"@r%05d" % bad, r[1][:h], b"+", r[3][:h]]
        b = [b"@x%05d" % bad + b"y" * (W - 24 - 4 * h), r[1][h:2 * h], b"+", r[3][h:2 * h]]
        recs[bad] = a + b
        assert len(b"\n".join(recs[bad])) + 1 == W
    p = tmp_path / "a.fq"
    p.write_bytes(b"".join(b"\n".join(x) + b"\n" for x in recs))
    for batch in (1, 100, 5000):
        got = lean(tool, batch, p)
        if damage == "two_half_records" and not got["fixed_width_file"]:
            continue
        assert got["ok"] and got["first_irregular_batch"] == bad // batch, (damage, batch, got)
        assert got["records"] == (bad // batch) * batch


# ---------------------------------------------------------------------------
# ordinary gzip, inflated in parallel (gzip_parallel.hpp, through the host-only tool shark_amd/bin/shark-gunzip)
# ---------------------------------------------------------------------------
GUNZIP = os.path.join(ROOT, "shark_amd", "bin", "shark-gunzip")


@pytest.fixture(scope="module")
def gunzip_tool():
    if not os.path.exists(GUNZIP):
        subprocess.run(["make", "-C", os.path.join(ROOT, "shark_amd", "csrc"), "../bin/shark-gunzip"], check=True, stdout=subprocess.DEVNULL)
    return GUNZIP


def _gunzip(tool_path, path, threads, chunk, **env_more):
    env = dict(os.environ, SHARK_GZ_CHUNK=str(chunk), **env_more)
    r = subprocess.run([tool_path, str(path), str(threads)], capture_output=True, env=env, timeout=300)
    return r.returncode, r.stdout


def _fastq_text(rng, n, L=150, real_names=True):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    bases = acgt[rng.integers(0, 4, size=(n, L))]
    quals = (rng.integers(2, 42, size=(n, L)) + 33).astype(np.uint8)
    out = []
    for i in range(n):
        name = b"@A00123:45:HXXXXXXXX:1:%d:%d:%d 1:N:0:ACGT" % (1101 + i // 5000, int(rng.integers(1000, 30000)), int(rng.integers(1000, 30000))) if real_names else b"@r%09d/1" % i
        out.append(name + b"\n" + bases[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("shape", ["gzip_-1", "gzip_-6", "gzip_-9", "gzip_multi_member", "stored_member_in_the_middle", "synthetic_names_constant_quality",
                                   "huffman_only", "fixed_huffman_member", "stored_only", "sync_flush_points", "stored_second_chunk"])
def test_parallel_gunzip_equals_zlib(gunzip_tool, tmp_path, shape):
    """every chunking (chunks of 64 KiB ... 1 MiB of compressed bytes: hundreds of block searches, windows handed from chunk to
    chunk) and every thread count gives exactly the bytes zlib gives: the levels gzip writes, several members in one file (a new
    member has nothing in front of it), a member of stored blocks (its chunks find no block to start from and are inflated by the
    chunk in front of them), and text whose back-references never stop pointing into the unseen window (constant name prefix and
    quality line: the symbols stay markers to the end of every chunk)"""
    import zlib
    rng = np.random.default_rng(abs(hash(shape)) % 1000)
    text = _fastq_text(rng, 24000, real_names=shape != "synthetic_names_constant_quality")
    if shape == "synthetic_names_constant_quality":
        lines = text.split(b"\n")[:-1]
        text = b"\n".join(l if i % 4 != 3 else b"I" * len(l) for i, l in enumerate(lines)) + b"\n"
    path = tmp_path / "t.gz"
    if shape.startswith("gzip_-"):
        open(path, "wb").write(gzip.compress(text, compresslevel=int(shape[-1])))
    elif shape == "gzip_multi_member":
        cut = [0, len(text) // 5, len(text) // 5 + 1, len(text) // 2, len(text)]
        open(path, "wb").write(b"".join(gzip.compress(text[a:b], compresslevel=lv) for a, b, lv in zip(cut, cut[1:], (6, 1, 9, 4))))
    elif shape == "huffman_only":
        # literals only: every block is a dynamic block WITHOUT distance codes (the decoder's run-of-literals table does all the work,
        # and the end-of-block symbol arrives in a block that has no distance table)
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_HUFFMAN_ONLY)
        open(path, "wb").write(co.compress(text) + co.flush())
    elif shape == "fixed_huffman_member":
        # a member of fixed-Huffman blocks (Z_FIXED) between two ordinary ones
        a, b = len(text) // 3, 2 * len(text) // 3
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
        open(path, "wb").write(gzip.compress(text[:a], 6) + co.compress(text[a:b]) + co.flush() + gzip.compress(text[b:], 1))
    elif shape == "stored_only":
        # no block start anywhere: the first chunk inflates the whole file, every other chunk's search comes back empty
        open(path, "wb").write(gzip.compress(text, 0))
    elif shape == "sync_flush_points":
        # Z_SYNC_FLUSH every 300 000 bytes: empty stored blocks, byte-aligned block starts
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        open(path, "wb").write(b"".join(co.compress(text[i:i + 300000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(text), 300000)) + co.flush())
    elif shape == "stored_second_chunk":
        # 45 KB of an ordinary member, then 200 KB stored, then ordinary members: with chunks of 64 KiB the second and third chunk hold no block
        # start, the fourth does -- the file is taken on all the same (one chunk without a start does not decide for the file)
        a, b = 100_000, 300_000
        open(path, "wb").write(gzip.compress(text[:a], 6) + gzip.compress(text[a:b], 0) + gzip.compress(text[b:], 6))
    elif shape == "stored_member_in_the_middle":
        a, b = len(text) // 3, len(text) // 3 + 600_000
        open(path, "wb").write(gzip.compress(text[:a], 6) + gzip.compress(text[a:b], 0) + gzip.compress(text[b:], 6))
    else:
        open(path, "wb").write(gzip.compress(text, compresslevel=6))
    assert gzip.decompress(open(path, "rb").read()) == text
    for chunk, threads in ((65536, 8), (200_000, 3), (1 << 20, 2)):
        rc, out = _gunzip(gunzip_tool, path, threads, chunk)
        if os.path.getsize(path) < 3 * chunk:      # fewer than three chunks: not worth it, left to gzread
            assert rc == 3 and out == b"", (shape, chunk)
            continue
        # a stream whose SECOND chunk holds no block start is left to gzread as well (rc 3: a stream of stored or fixed-Huffman blocks
        # would be inflated by its first chunk alone); whatever is taken on is delivered exactly
        assert (rc == 3 and out == b"") or (rc == 0 and out == text), (shape, chunk, threads, len(out), len(text))
        if shape == "stored_only":
            assert rc == 3
        if shape == "stored_second_chunk" and chunk == 65536:
            assert rc == 0
        if shape.startswith("gzip_-") or shape == "synthetic_names_constant_quality":
            assert rc == 0
        # ... and taken on regardless (SHARK_GZ_FORCE_PARALLEL=1) every shape is delivered exactly: as one text per chunk, and with
        # the text handed out in pieces of 70 000 symbols (a chunk that inflates far beyond its territory; here: every chunk)
        for more in ({}, {"SHARK_GZ_PIECE": "70000"}):
            rc, out = _gunzip(gunzip_tool, path, threads, chunk, SHARK_GZ_FORCE_PARALLEL="1", **more)
            assert rc == 0 and out == text, (shape, chunk, threads, more, len(out), len(text))


def test_parallel_gunzip_holds_bounded_memory_through_a_long_stretch_without_block_starts(gunzip_tool, tmp_path):
    """an ordinary member, then a member of fixed-Huffman blocks that inflates to 140 MB (no chunk inside it finds a block start: the
    chunk in front of it inflates all of it), then an ordinary member again: the text is exact and the process stays far below the
    size of the text -- the chunk hands its text out in pieces (gzread needs constant memory for the same stream); without the
    pieces it held the whole stretch as 16-bit symbols plus its text: three times the text"""
    import hashlib
    import re
    import zlib
    rng = np.random.default_rng(12)
    text = _fastq_text(rng, 20000, real_names=False)                     # 6.5 MB
    reps = 22
    co = zlib.compressobj(1, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
    mid = b"".join(co.compress(text) for _ in range(reps)) + co.flush()
    path = tmp_path / "long.gz"
    open(path, "wb").write(gzip.compress(text, 6) + mid + gzip.compress(text, 1))
    want = hashlib.md5(text * (reps + 2)).hexdigest()
    total = len(text) * (reps + 2)
    env = dict(os.environ, SHARK_GZ_CHUNK="300000", SHARK_GZ_PIECE=str(2 << 20))
    r = subprocess.run([gunzip_tool, str(path), "4"], capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    assert len(r.stdout) == total and hashlib.md5(r.stdout).hexdigest() == want
    rss_kb = int(re.search(r"maxrss_kb (\d+)", r.stderr.decode()).group(1))
    # (152 MB of text from 113 MB of file: the pages of the mapped file are given back behind the consumer)
    assert rss_kb * 1024 < 64 << 20 < total // 2, (rss_kb, os.path.getsize(path), total)


def test_parallel_gunzip_declines_what_it_cannot_do_and_stops_where_the_stream_breaks(gunzip_tool, tmp_path):
    """small files, files that are not gzip and gzip of something that is not text are left to gzread (exit code 3 of the tool =
    usable() false); a truncated file and a file with a damaged byte deliver a PREFIX of the text and end -- nothing behind the damage,
    although the chunks behind it had found their block starts"""
    import zlib
    rng = np.random.default_rng(5)
    text = _fastq_text(rng, 12000)
    gz = gzip.compress(text, 6)
    open(tmp_path / "small.gz", "wb").write(gzip.compress(text[:3000], 6))
    open(tmp_path / "plain.fq", "wb").write(text)
    open(tmp_path / "binary.gz", "wb").write(gzip.compress(rng.integers(0, 256, size=2_000_000, dtype=np.uint8).tobytes(), 6))
    for name in ("small.gz", "plain.fq", "binary.gz"):
        rc, out = _gunzip(gunzip_tool, tmp_path / name, 4, 65536)
        assert rc == 3 and out == b"", name
    open(tmp_path / "trunc.gz", "wb").write(gz[:len(gz) * 2 // 3])
    rc, out = _gunzip(gunzip_tool, tmp_path / "trunc.gz", 4, 65536)
    want = zlib.decompressobj(31).decompress(gz[:len(gz) * 2 // 3])
    assert rc == 0 and text.startswith(out) and len(want) - 70000 <= len(out) <= len(want), (len(out), len(want))
    # a damaged byte inside a Huffman block: the codes resynchronise, zlib delivers a few wrong bytes and notices at the trailer's
    # CRC-32 -- behind everything it delivered, and the reference's reader ignores that error (kseq.h: a negative gzread is end of
    # file).  The parallel inflate gives exactly what raw inflate of the damaged stream gives; when that is invalid, a prefix.
    n_checked = 0
    for at in (len(gz) // 2, len(gz) // 3 + 17, len(gz) * 3 // 4 + 5):
        bad = bytearray(gz)
        bad[at] ^= 0x5A
        open(tmp_path / "bad.gz", "wb").write(bytes(bad))
        rc, out = _gunzip(gunzip_tool, tmp_path / "bad.gz", 4, 65536)
        assert rc == 0
        d = zlib.decompressobj(-15)
        try:
            want = d.decompress(bytes(bad[10:]))       # (the header is ten bytes: no name, no extra field)
            assert out == want, (at, len(out), len(want))
            n_checked += 1
        except zlib.error:
            good = len(os.path.commonprefix([out, text]))
            assert good > 0 and len(out) - good < 400_000, (at, good, len(out))
    assert n_checked >= 1


def test_reader_over_parallel_gunzip_delivers_the_serial_readers_records(tool, tmp_path):
    """the kseq-rule reader on top of the parallel inflate (what `shark` does with a .fq.gz sample) delivers the records it delivers
    with SHARK_GZ_SERIAL=1 (zlib's gzread on a read-ahead thread) and from the plain file"""
    rng = np.random.default_rng(8)
    text = _fastq_text(rng, 30000)
    open(tmp_path / "t.fq", "wb").write(text)
    open(tmp_path / "t.fq.gz", "wb").write(gzip.compress(text[:len(text) // 2], 1) + gzip.compress(text[len(text) // 2:], 9))

    def rec(path, **env):
        return json.loads(subprocess.run([tool, "--records", str(path)], capture_output=True, text=True, check=True, env=dict(os.environ, **env)).stdout)
    want = rec(tmp_path / "t.fq")
    par = rec(tmp_path / "t.fq.gz", SHARK_GZ_CHUNK="100000")
    ser = rec(tmp_path / "t.fq.gz", SHARK_GZ_SERIAL="1")
    assert want["records"] == 30000
    assert (par["records"], par["bases"], par["fnv"]) == (ser["records"], ser["bases"], ser["fnv"]) == (want["records"], want["bases"], want["fnv"])
