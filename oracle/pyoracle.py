"""ctypes bindings for the CPU ORACLE (oracle/liboracle.so) and, when built,
the reference primitives (oracle/_ref/libsharkref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under shark_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
CLI_PATH = os.path.join(HERE, "oracle_cli")
REF_PATH = os.path.join(HERE, "_ref", "libsharkref.so")


def build():
    """(Re)build the oracle; also builds oracle/_ref when /root/reference exists."""
    subprocess.run(["make", "-C", HERE, "all"], check=True, stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        u64, u8, i64, p = C.c_uint64, C.c_uint8, C.c_int64, C.c_void_p
        L.so_to_int.restype = u8; L.so_to_int.argtypes = [C.c_char]
        L.so_reverse_char.restype = u8; L.so_reverse_char.argtypes = [u8]
        L.so_revcompl.restype = u64; L.so_revcompl.argtypes = [u64, u8]
        L.so_build_kmer.restype = i64; L.so_build_kmer.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), u8]
        L.so_lsappend.restype = u64; L.so_lsappend.argtypes = [u64, u64, u64]
        L.so_rsprepend.restype = u64; L.so_rsprepend.argtypes = [u64, u64, u64]
        L.so_get_hash.restype = u64; L.so_get_hash.argtypes = [u64]
        L.so_xxh64.restype = u64; L.so_xxh64.argtypes = [C.c_char_p, C.c_size_t, u64]
        L.so_kmer_builder.restype = C.c_size_t; L.so_kmer_builder.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, p]
        L.so_join_mask.restype = C.c_size_t
        L.so_join_mask.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p,
                                   C.c_int, C.c_char, C.c_char_p]
        L.so_shark_new.restype = p; L.so_shark_new.argtypes = [C.c_uint32, C.c_double, u64, C.c_int, C.c_int]
        L.so_shark_free.restype = None; L.so_shark_free.argtypes = [p]
        L.so_shark_build.restype = C.c_int; L.so_shark_build.argtypes = [p, p, p, C.c_size_t]
        L.so_shark_build_mt.restype = C.c_int; L.so_shark_build_mt.argtypes = [p, p, p, C.c_size_t, C.c_int]
        L.so_shark_bf.restype = p; L.so_shark_bf.argtypes = [p]
        L.so_analyze_read.restype = C.c_int
        L.so_analyze_read.argtypes = [p, C.c_char_p, C.c_size_t, p, C.c_int, p, p, p]
        L.so_classify_batch.restype = C.c_int
        L.so_classify_batch.argtypes = [p, u64, p, p, p, p, p, p, C.c_int, p, C.POINTER(p)]
        L.so_free.restype = None; L.so_free.argtypes = [p]
        L.so_bf_size.restype = u64; L.so_bf_size.argtypes = [p]
        L.so_bf_words.restype = p; L.so_bf_words.argtypes = [p]
        L.so_bf_num_kmer.restype = u64; L.so_bf_num_kmer.argtypes = [p]
        L.so_bf_tot_idx.restype = u64; L.so_bf_tot_idx.argtypes = [p]
        L.so_bf_index_kmer.restype = p; L.so_bf_index_kmer.argtypes = [p]
        L.so_bf_rank.restype = u64; L.so_bf_rank.argtypes = [p, u64]
        L.so_bf_get_index.restype = None
        L.so_bf_get_index.argtypes = [p, u64, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib = L
    return _lib


_ref = None


def ref():
    """The reference primitives, or None when oracle/_ref has not been built."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_PATH):
            return None
        R = C.CDLL(REF_PATH)
        u64, u8, i64, p = C.c_uint64, C.c_uint8, C.c_int64, C.c_void_p
        R.ref_to_int.restype = u8; R.ref_to_int.argtypes = [C.c_ubyte]
        R.ref_reverse_char.restype = u8; R.ref_reverse_char.argtypes = [u8]
        R.ref_revcompl.restype = u64; R.ref_revcompl.argtypes = [u64, u8]
        R.ref_build_kmer.restype = i64; R.ref_build_kmer.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), u8]
        R.ref_lsappend.restype = u64; R.ref_lsappend.argtypes = [u64, u64, u64]
        R.ref_rsprepend.restype = u64; R.ref_rsprepend.argtypes = [u64, u64, u64]
        R.ref_get_hash.restype = u64; R.ref_get_hash.argtypes = [u64]
        R.ref_xxh64.restype = u64; R.ref_xxh64.argtypes = [C.c_char_p, C.c_size_t, u64]
        R.ref_small_vector.restype = C.c_size_t; R.ref_small_vector.argtypes = [p, C.c_size_t, p, p]
        R.ref_fasta_read.restype = p; R.ref_fasta_read.argtypes = [C.c_char_p]
        R.ref_fasta_count.restype = C.c_size_t; R.ref_fasta_count.argtypes = [p]
        R.ref_fasta_name.restype = C.c_char_p; R.ref_fasta_name.argtypes = [p, C.c_size_t]
        R.ref_fasta_seq.restype = C.c_char_p; R.ref_fasta_seq.argtypes = [p, C.c_size_t]
        R.ref_fasta_seq_len.restype = C.c_size_t; R.ref_fasta_seq_len.argtypes = [p, C.c_size_t]
        R.ref_fasta_free.restype = None; R.ref_fasta_free.argtypes = [p]
        R.ref_fastq_read.restype = p; R.ref_fastq_read.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        R.ref_fastq_count.restype = C.c_size_t; R.ref_fastq_count.argtypes = [p]
        R.ref_fastq_joined.restype = C.POINTER(C.c_char); R.ref_fastq_joined.argtypes = [p, C.c_size_t, C.POINTER(C.c_size_t)]
        for name in ("ref_fastq_id", "ref_fastq_seq", "ref_fastq_qual"):
            getattr(R, name).restype = C.c_char_p
            getattr(R, name).argtypes = [p, C.c_size_t, C.c_int]
        R.ref_fastq_free.restype = None; R.ref_fastq_free.argtypes = [p]
        _ref = R
    return _ref


def _as_bytes(a):
    if a is None:
        return None
    if isinstance(a, (bytes, bytearray)):
        return np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


class Shark:
    """main.cpp orchestration over the oracle: build an index, classify reads."""

    def __init__(self, k=17, c=0.6, bf_bits=1 << 33, min_quality=0, single=False):
        self.L = lib()
        self.k, self.c, self.bf_bits, self.min_quality, self.single = k, c, bf_bits, min_quality, single
        self.h = self.L.so_shark_new(k, c, bf_bits, min_quality, int(bool(single)))
        if not self.h:
            raise MemoryError("so_shark_new failed")
        self.nidx = None

    def close(self):
        if self.h:
            self.L.so_shark_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build(self, seqs, nthreads=1):
        """seqs: list of bytes, FASTA records in file order.  nthreads > 1: so_shark_build_mt (same index; the scale tests)."""
        n = len(seqs)
        bufs = [C.create_string_buffer(bytes(s), len(s) + 1) for s in seqs]
        arr = (C.c_char_p * max(n, 1))(*[C.cast(b, C.c_char_p) for b in bufs])
        lens = (C.c_uint64 * max(n, 1))(*[len(s) for s in seqs])
        if nthreads > 1:
            self.nidx = self.L.so_shark_build_mt(self.h, arr, lens, n, int(nthreads))
        else:
            self.nidx = self.L.so_shark_build(self.h, arr, lens, n)
        return self.nidx

    # -- index introspection -------------------------------------------------
    def bf(self):
        return self.L.so_shark_bf(self.h)

    def bf_words(self):
        nw = (self.bf_bits + 63) // 64
        ptr = self.L.so_bf_words(self.bf())
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(nw,))

    def num_kmer(self):
        return self.L.so_bf_num_kmer(self.bf())

    def index_kmer(self):
        n = self.L.so_bf_tot_idx(self.bf())
        ptr = self.L.so_bf_index_kmer(self.bf())
        if n == 0:
            return np.zeros(0, np.uint16)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint16)), shape=(n,)).copy()

    def get_index(self, kmer):
        s, e = C.c_int(), C.c_int()
        self.L.so_bf_get_index(self.bf(), kmer, C.byref(s), C.byref(e))
        return s.value, e.value

    # -- classification ------------------------------------------------------
    def analyze(self, read):
        """one joined/masked read string -> (genes, max, maxk, len)"""
        cap = 1 << 16
        genes = (C.c_int * cap)()
        mx, mk, ln = C.c_uint(), C.c_uint(), C.c_uint()
        n = self.L.so_analyze_read(self.h, bytes(read), len(read), genes, cap, C.byref(mx), C.byref(mk), C.byref(ln))
        return list(genes[:n]), mx.value, mk.value, ln.value

    def classify(self, seq1, off1, seq2=None, off2=None, qual1=None, qual2=None, nthreads=1):
        """SoA batch (same layout as the C-ABI) -> (gene_off[n+1] u32, gene_ids u16)"""
        seq1 = _as_bytes(seq1); seq2 = _as_bytes(seq2); qual1 = _as_bytes(qual1); qual2 = _as_bytes(qual2)
        off1 = np.ascontiguousarray(off1, dtype=np.uint64)
        n = len(off1) - 1
        off2a = np.ascontiguousarray(off2, dtype=np.uint64) if off2 is not None else None
        gene_off = np.zeros(n + 1, dtype=np.uint32)
        out = C.c_void_p()

        def ptr(a):
            return a.ctypes.data_as(C.c_void_p) if a is not None else None

        rc = self.L.so_classify_batch(self.h, n, ptr(seq1), ptr(off1), ptr(seq2), ptr(off2a), ptr(qual1), ptr(qual2),
                                      nthreads, ptr(gene_off), C.byref(out))
        if rc != 0:
            raise RuntimeError("so_classify_batch failed: %d" % rc)
        tot = int(gene_off[n])
        ids = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint16)), shape=(max(tot, 1),))[:tot].copy()
        self.L.so_free(out)
        return gene_off, ids


def run_cli(args, stdout_path):
    """Run the oracle CLI (same flags as shark) writing ssv to stdout_path."""
    if not os.path.exists(CLI_PATH):
        build()
    with open(stdout_path, "wb") as f:
        subprocess.run([CLI_PATH] + list(args), check=True, stdout=f)
