"""ctypes binding of libsharkhip.so (include/shark_hip.h).

Plumbing for tests and bench.py; the product is the C ABI itself.  There is no
fallback of any kind: if the HIP library is missing this module raises, and if
no GPU is present every computing call returns an error that is raised here.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SHK_LIB_PATH") or os.path.join(HERE, "libsharkhip.so")   # (SHK_LIB_PATH: kernel experiments, tools/)

SHK_INLINE_IDS = 4


class ShkParams(C.Structure):
    _fields_ = [("k", C.c_uint32), ("c", C.c_double), ("bf_bits", C.c_uint64),
                ("min_quality", C.c_int32), ("single", C.c_int32), ("device", C.c_int32)]


class ShkIndexInfo(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("nidx", C.c_uint64), ("bf_bits", C.c_uint64),
                ("n_set_bits", C.c_uint64), ("tot_idx", C.c_uint64), ("n_ref_kmers", C.c_uint64)]


class ShkBatch(C.Structure):
    _fields_ = [("n", C.c_uint64), ("seq1", C.c_void_p), ("off1", C.c_void_p), ("seq2", C.c_void_p),
                ("off2", C.c_void_p), ("qual1", C.c_void_p), ("qual2", C.c_void_p)]


class ShkResult(C.Structure):
    _fields_ = [("n", C.c_uint64), ("gene_off", C.c_void_p), ("gene_ids", C.c_void_p), ("n_assoc", C.c_uint64)]


class ShkTiming(C.Structure):
    _fields_ = [("n_launches", C.c_uint64), ("total_ms", C.c_double), ("last_n_reads", C.c_uint64),
                ("last_n_long", C.c_uint64), ("last_n_tie", C.c_uint64), ("last_n_assoc", C.c_uint64), ("prepass_ms", C.c_double)]


class ShkWorkCounters(C.Structure):
    _fields_ = [("n_kmers", C.c_uint64), ("n_hits", C.c_uint64), ("n_list_ids", C.c_uint64), ("n_bases", C.c_uint64)]


# every symbol include/shark_hip.h declares
EXPORTS = [
    "shk_create", "shk_destroy", "shk_strerror", "shk_last_error", "shk_ref_add", "shk_ref_finalize",
    "shk_index_info_get", "shk_index_copy_bf", "shk_index_copy_lists", "shk_classify", "shk_classify_device",
    "shk_gene_counts", "shk_gene_counts_reset", "shk_timing_enable", "shk_timing_get", "shk_count_work",
    "shk_alloc_pinned", "shk_free_pinned", "shk_version", "shk_probe_mode", "shk_gene_counts_allreduce",
    "shk_classify_submit", "shk_classify_wait", "shk_dist_unique_id", "shk_dist_init", "shk_dist_gene_counts_allreduce",
    "shk_dist_info", "shk_measure_random_lookups", "shk_last_kernel", "shk_classify_device_submit",
    "shk_measure_valu_mix",
    "shk_measure_valu_mix_clock",
]
SHK_PIPE_DEPTH = 3
SHK_DIST_ID_BYTES = 128

_lib = None


class SharkHipError(RuntimeError):
    pass


def load():
    """Load libsharkhip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SharkHipError("libsharkhip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C shark_amd/csrc`); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    p = C.c_void_p
    L.shk_create.restype = C.c_int; L.shk_create.argtypes = [C.POINTER(ShkParams), C.POINTER(p)]
    L.shk_destroy.restype = None; L.shk_destroy.argtypes = [p]
    L.shk_strerror.restype = C.c_char_p; L.shk_strerror.argtypes = [C.c_int]
    L.shk_last_error.restype = C.c_char_p; L.shk_last_error.argtypes = [p]
    L.shk_ref_add.restype = C.c_int; L.shk_ref_add.argtypes = [p, C.c_char_p, C.c_uint64]
    L.shk_ref_finalize.restype = C.c_int; L.shk_ref_finalize.argtypes = [p]
    L.shk_index_info_get.restype = C.c_int; L.shk_index_info_get.argtypes = [p, C.POINTER(ShkIndexInfo)]
    L.shk_index_copy_bf.restype = C.c_int; L.shk_index_copy_bf.argtypes = [p, p, C.c_uint64]
    L.shk_index_copy_lists.restype = C.c_int; L.shk_index_copy_lists.argtypes = [p, p, p]
    L.shk_classify.restype = C.c_int; L.shk_classify.argtypes = [p, C.POINTER(ShkBatch), C.POINTER(ShkResult)]
    L.shk_classify_device.restype = C.c_int
    L.shk_classify_device.argtypes = [p, C.POINTER(ShkBatch), C.c_uint32, C.POINTER(ShkResult)]
    L.shk_gene_counts.restype = C.c_int; L.shk_gene_counts.argtypes = [p, p, C.c_uint32]
    L.shk_gene_counts_reset.restype = C.c_int; L.shk_gene_counts_reset.argtypes = [p]
    L.shk_timing_enable.restype = C.c_int; L.shk_timing_enable.argtypes = [p, C.c_int]
    L.shk_timing_get.restype = C.c_int; L.shk_timing_get.argtypes = [p, C.POINTER(ShkTiming)]
    L.shk_count_work.restype = C.c_int; L.shk_count_work.argtypes = [p, C.POINTER(ShkBatch), C.POINTER(ShkWorkCounters)]
    L.shk_alloc_pinned.restype = p; L.shk_alloc_pinned.argtypes = [C.c_size_t]
    L.shk_free_pinned.restype = None; L.shk_free_pinned.argtypes = [p]
    L.shk_version.restype = C.c_char_p; L.shk_version.argtypes = []
    L.shk_probe_mode.restype = C.c_char_p; L.shk_probe_mode.argtypes = [p]
    L.shk_last_kernel.restype = C.c_char_p; L.shk_last_kernel.argtypes = [p]
    L.shk_gene_counts_allreduce.restype = C.c_int; L.shk_gene_counts_allreduce.argtypes = [C.POINTER(p), C.c_int, p, C.c_uint32]
    L.shk_classify_submit.restype = C.c_int; L.shk_classify_submit.argtypes = [p, C.POINTER(ShkBatch), C.POINTER(C.c_uint64)]
    L.shk_classify_wait.restype = C.c_int; L.shk_classify_wait.argtypes = [p, C.c_uint64, C.POINTER(ShkResult)]
    # The later entry points, prototype by prototype: a library variant named by SHK_LIB_PATH (A/B timing, tools/build_variant.sh) may
    # come from an older tree and lack some of them -- such a name is set to None ("not callable") and every name that exists gets its
    # prototype, so that no call ever goes through ctypes' default int conversion.  The product library is checked symbol by symbol
    # in tests/test_cabi_cpu.py; without SHK_LIB_PATH a missing symbol is an error here as well.
    later = {
        "shk_classify_device_submit": (C.c_int, [p, C.POINTER(ShkBatch), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
        "shk_dist_unique_id": (C.c_int, [p]),
        "shk_dist_init": (C.c_int, [p, p, C.c_int, C.c_int]),
        "shk_dist_gene_counts_allreduce": (C.c_int, [p, p, C.c_uint32]),
        "shk_dist_info": (C.c_int, [p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "shk_measure_random_lookups": (C.c_int, [p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
        "shk_measure_valu_mix": (C.c_int, [p, C.c_int, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "shk_measure_valu_mix_clock": (C.c_int, [p, C.c_int, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    }
    variant = bool(os.environ.get("SHK_LIB_PATH"))
    for name, (res, args) in later.items():
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        elif variant:
            setattr(L, name, None)
        else:
            raise SharkHipError("libsharkhip.so lacks %s: rebuild it (make -C shark_amd/csrc)" % name)
    _lib = L
    return L


def _u8(a):
    if a is None:
        return None
    if isinstance(a, (bytes, bytearray)):
        return np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class SharkHip:
    """One classification context on one GPU (shk_ctx)."""

    def __init__(self, k=17, c=0.6, bf_bits=1 << 33, min_quality=0, single=False, device=0):
        self.L = load()
        self.h = C.c_void_p()
        prm = ShkParams(k, c, bf_bits, min_quality, int(bool(single)), device)
        rc = self.L.shk_create(C.byref(prm), C.byref(self.h))
        if rc != 0:
            self.h = C.c_void_p()
            raise SharkHipError("shk_create: %s" % self.L.shk_strerror(rc).decode())
        self.k, self.c, self.bf_bits = k, c, bf_bits

    def _check(self, rc, what):
        if rc != 0:
            raise SharkHipError("%s: %s (%s)" % (what, self.L.shk_strerror(rc).decode(),
                                                 self.L.shk_last_error(self.h).decode()))

    def close(self):
        if self.h:
            self.L.shk_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- index -----------------------------------------------------------------
    def ref_add(self, seq):
        return self.L.shk_ref_add(self.h, bytes(seq), len(seq))

    def ref_finalize(self):
        return self.L.shk_ref_finalize(self.h)

    def build(self, seqs):
        for s in seqs:
            self._check(self.ref_add(s), "shk_ref_add")
        self._check(self.ref_finalize(), "shk_ref_finalize")
        return self.index_info()

    def index_info(self):
        info = ShkIndexInfo()
        self._check(self.L.shk_index_info_get(self.h, C.byref(info)), "shk_index_info_get")
        return {f: getattr(info, f) for f, _ in ShkIndexInfo._fields_}

    def probe_mode(self):
        return self.L.shk_probe_mode(self.h).decode()

    def last_kernel(self):
        """the classify kernel instantiation the last batch ran (rocprofv3's name for it)"""
        return self.L.shk_last_kernel(self.h).decode()

    def copy_bf(self):
        nw = (self.bf_bits + 63) // 64
        w = np.zeros(nw, dtype=np.uint64)
        self._check(self.L.shk_index_copy_bf(self.h, _ptr(w), nw), "shk_index_copy_bf")
        return w

    def copy_lists(self):
        info = self.index_info()
        off = np.zeros(info["n_set_bits"] + 1, dtype=np.uint32)
        ids = np.zeros(max(info["tot_idx"], 1), dtype=np.uint16)
        self._check(self.L.shk_index_copy_lists(self.h, _ptr(off), _ptr(ids)), "shk_index_copy_lists")
        return off, ids[:info["tot_idx"]]

    # ---- classification --------------------------------------------------------
    def _host_batch(self, seq1, off1, seq2, off2, qual1, qual2):
        seq1, seq2, qual1, qual2 = _u8(seq1), _u8(seq2), _u8(qual1), _u8(qual2)
        off1 = np.ascontiguousarray(off1, dtype=np.uint64)
        off2 = np.ascontiguousarray(off2, dtype=np.uint64) if off2 is not None else None
        n = len(off1) - 1
        keep = (seq1, off1, seq2, off2, qual1, qual2)      # the library reads them until the ticket is waited for
        return ShkBatch(n, _ptr(seq1), _ptr(off1), _ptr(seq2), _ptr(off2), _ptr(qual1), _ptr(qual2)), keep

    @staticmethod
    def _host_result(r, copy=True):
        n, tot = int(r.n), int(r.n_assoc)
        gene_off = np.ctypeslib.as_array(C.cast(r.gene_off, C.POINTER(C.c_uint32)), shape=(n + 1,))
        ids = np.ctypeslib.as_array(C.cast(r.gene_ids, C.POINTER(C.c_uint16)), shape=(tot,)) if tot else np.zeros(0, np.uint16)
        return (gene_off.copy(), ids.copy()) if copy else (gene_off, ids)

    def classify(self, seq1, off1, seq2=None, off2=None, qual1=None, qual2=None):
        """host SoA batch -> (gene_off[n+1] u32, gene_ids u16)"""
        b, keep = self._host_batch(seq1, off1, seq2, off2, qual1, qual2)
        r = ShkResult()
        self._check(self.L.shk_classify(self.h, C.byref(b), C.byref(r)), "shk_classify")
        return self._host_result(r)

    def submit(self, seq1, off1, seq2=None, off2=None, qual1=None, qual2=None):
        """pipelined form: returns a ticket (an object that also keeps the host arrays alive)"""
        b, keep = self._host_batch(seq1, off1, seq2, off2, qual1, qual2)
        t = C.c_uint64()
        self._check(self.L.shk_classify_submit(self.h, C.byref(b), C.byref(t)), "shk_classify_submit")
        return (t.value, keep)

    def wait(self, ticket, copy=True):
        r = ShkResult()
        self._check(self.L.shk_classify_wait(self.h, ticket[0], C.byref(r)), "shk_classify_wait")
        return self._host_result(r, copy)

    def classify_device(self, n, seq1, off1, seq2=0, off2=0, qual1=0, qual2=0, max_read_len=0):
        """device pointers (ints) -> ShkResult with DEVICE pointers"""
        b = ShkBatch(n, seq1 or None, off1 or None, seq2 or None, off2 or None, qual1 or None, qual2 or None)
        r = ShkResult()
        self._check(self.L.shk_classify_device(self.h, C.byref(b), max_read_len, C.byref(r)), "shk_classify_device")
        return r

    def submit_device(self, n, seq1, off1, seq2=0, off2=0, qual1=0, qual2=0, max_read_len=0, uniform_len1=0, uniform_len2=0):
        """device pointers (ints) -> ticket; wait_device(ticket) gives the ShkResult with DEVICE pointers"""
        b = ShkBatch(n, seq1 or None, off1 or None, seq2 or None, off2 or None, qual1 or None, qual2 or None)
        t = C.c_uint64()
        self._check(self.L.shk_classify_device_submit(self.h, C.byref(b), max_read_len, uniform_len1, uniform_len2, C.byref(t)), "shk_classify_device_submit")
        return t.value

    def wait_device(self, ticket):
        r = ShkResult()
        self._check(self.L.shk_classify_wait(self.h, ticket, C.byref(r)), "shk_classify_wait")
        return r

    def count_work(self, n, seq1, off1, seq2=0, off2=0, qual1=0, qual2=0):
        b = ShkBatch(n, seq1 or None, off1 or None, seq2 or None, off2 or None, qual1 or None, qual2 or None)
        w = ShkWorkCounters()
        self._check(self.L.shk_count_work(self.h, C.byref(b), C.byref(w)), "shk_count_work")
        return {f: getattr(w, f) for f, _ in ShkWorkCounters._fields_}

    def gene_counts(self, n=65536):
        a = np.zeros(n, dtype=np.uint64)
        self._check(self.L.shk_gene_counts(self.h, _ptr(a), n), "shk_gene_counts")
        return a

    def gene_counts_allreduce(self, others=(), n=65536):
        """sum the per-gene counters of this context and `others` (one per GPU) over RCCL"""
        ctxs = [self] + list(others)
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        a = np.zeros(n, dtype=np.uint64)
        self._check(self.L.shk_gene_counts_allreduce(arr, len(ctxs), _ptr(a), n), "shk_gene_counts_allreduce")
        return a

    def gene_counts_reset(self):
        self._check(self.L.shk_gene_counts_reset(self.h), "shk_gene_counts_reset")

    # ---- one process per GPU ------------------------------------------------------
    def dist_init(self, sdist):
        """join the RCCL communicator of a torch.distributed job (shark_amd.dist): rank 0's unique id is
        broadcast over the job's own channel.  A gloo job (CPU tests, single-GPU dry runs) has no RCCL
        communicator; dist_gene_counts_allreduce then reduces the local counters over gloo."""
        import torch
        import torch.distributed as td
        self._td = None
        if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
            return
        if td.get_backend() != "nccl":
            self._td = td
            return
        rank, world = td.get_rank(), td.get_world_size()
        buf = (C.c_uint8 * SHK_DIST_ID_BYTES)()
        if rank == 0:
            rc = self.L.shk_dist_unique_id(buf)
            if rc != 0:
                raise SharkHipError("shk_dist_unique_id: %s" % self.L.shk_strerror(rc).decode())
        t = torch.tensor(list(buf), dtype=torch.uint8, device="cuda")
        td.broadcast(t, 0)
        ident = (C.c_uint8 * SHK_DIST_ID_BYTES)(*t.cpu().tolist())
        self._check(self.L.shk_dist_init(self.h, ident, rank, world), "shk_dist_init")

    def dist_info(self):
        """(rank, world) as the communicator the collective runs over reports them: RCCL's own ncclCommUserRank /
        ncclCommCount for an RCCL job, torch.distributed's for a gloo dry run, (0, 1) without a job"""
        td = getattr(self, "_td", None)
        if td is not None:
            return td.get_rank(), td.get_world_size()
        r, w = C.c_int(), C.c_int()
        self._check(self.L.shk_dist_info(self.h, C.byref(r), C.byref(w)), "shk_dist_info")
        return r.value, w.value

    def dist_gene_counts_allreduce(self, n=65536):
        a = np.zeros(n, dtype=np.uint64)
        self._check(self.L.shk_dist_gene_counts_allreduce(self.h, _ptr(a), n), "shk_dist_gene_counts_allreduce")
        td = getattr(self, "_td", None)
        if td is not None:
            import torch
            t = torch.from_numpy(a.astype(np.int64))
            td.all_reduce(t)
            a = t.numpy().astype(np.uint64)
        return a

    def measure_random_lookups(self, table_bytes, n_lookups=1 << 31, nontemporal=False, both_halves=False):
        """G independent random 16-byte lookups per second in a table of table_bytes on this context's GPU (both_halves: each lookup
        also reads the other 64-byte half of its 128-byte line; the figure is lines per second either way)"""
        g = C.c_double()
        self._check(self.L.shk_measure_random_lookups(self.h, table_bytes, n_lookups, int(bool(nontemporal)) | (2 if both_halves else 0), C.byref(g)),
                    "shk_measure_random_lookups")
        return g.value

    @staticmethod
    def random_lookups_made(n_lookups):
        """how many lookups (lines) shk_measure_random_lookups(…, n_lookups, …) really makes in its timed launch: 2 048 workgroups x 256
        lanes x 5 per iteration, whole iterations"""
        per_iter = 2048 * 256 * 5
        return per_iter * max(1, min(n_lookups // per_iter, 1 << 20))

    def measure_valu_mix(self, waves_per_simd=4, iters=20000):
        """(ms, wave_iterations) of the exact-table kernel's instruction mix on register operands, `waves_per_simd` waves per SIMD"""
        ms, wi = C.c_double(), C.c_uint64()
        self._check(self.L.shk_measure_valu_mix(self.h, waves_per_simd, iters, C.byref(ms), C.byref(wi)), "shk_measure_valu_mix")
        return ms.value, wi.value

    def measure_valu_mix_clock(self, waves_per_simd=4, iters=20000):
        """(ms, wave_iterations, shader_ghz): measure_valu_mix and the clock the SIMDs held meanwhile (s_memtime over s_memrealtime)"""
        ms, wi, ghz = C.c_double(), C.c_uint64(), C.c_double()
        self._check(self.L.shk_measure_valu_mix_clock(self.h, waves_per_simd, iters, C.byref(ms), C.byref(wi), C.byref(ghz)), "shk_measure_valu_mix_clock")
        return ms.value, wi.value, ghz.value

    def timing_enable(self, on=True):
        self._check(self.L.shk_timing_enable(self.h, int(on)), "shk_timing_enable")

    def timing(self):
        t = ShkTiming()
        self._check(self.L.shk_timing_get(self.h, C.byref(t)), "shk_timing_get")
        return {f: getattr(t, f) for f, _ in ShkTiming._fields_}


_hip = None


def hip_memcpy_dtoh(dst, src_ptr, nbytes):
    """copy nbytes from a device pointer into a numpy array (bench/test plumbing)"""
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.restype = C.c_int
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rc = _hip.hipMemcpy(dst.ctypes.data_as(C.c_void_p), C.c_void_p(src_ptr), nbytes, 2)  # hipMemcpyDeviceToHost
    if rc != 0:
        raise SharkHipError("hipMemcpy D2H failed: %d" % rc)
