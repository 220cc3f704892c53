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
