"""ctypes binding of the CPU oracle (oracle/libntsm_oracle.so) -- test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "libntsm_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libntsm_oracle.so", "ntsm_oracle"], check=True)
        L = C.CDLL(path)
        u64p = C.POINTER(C.c_uint64)
        L.ntsm_oracle_hash64.restype = C.c_uint64
        L.ntsm_oracle_hash64.argtypes = [C.c_uint64, C.c_uint64]
        L.ntsm_oracle_mask.restype = C.c_uint64
        L.ntsm_oracle_mask.argtypes = [C.c_uint]
        L.ntsm_oracle_nt4.restype = C.c_int
        L.ntsm_oracle_nt4.argtypes = [C.c_ubyte]
        L.ntsm_oracle_kmers.restype = C.c_uint64
        L.ntsm_oracle_kmers.argtypes = [C.c_char_p, C.c_uint64, C.c_uint, u64p, u64p, u64p, C.c_uint64]
        L.ntsm_oracle_reader_open.restype = C.c_void_p
        L.ntsm_oracle_reader_open.argtypes = [C.c_char_p]
        L.ntsm_oracle_reader_next.restype = C.c_int64
        L.ntsm_oracle_reader_next.argtypes = [C.c_void_p]
        L.ntsm_oracle_reader_seq.restype = C.c_void_p
        L.ntsm_oracle_reader_seq.argtypes = [C.c_void_p]
        L.ntsm_oracle_reader_name.restype = C.c_char_p
        L.ntsm_oracle_reader_name.argtypes = [C.c_void_p]
        L.ntsm_oracle_reader_close.argtypes = [C.c_void_p]
        L.ntsm_oracle_fp_create.restype = C.c_void_p
        L.ntsm_oracle_fp_create.argtypes = [C.c_char_p, C.c_uint, C.c_double, C.c_int, C.c_void_p]
        L.ntsm_oracle_fp_destroy.argtypes = [C.c_void_p]
        L.ntsm_oracle_fp_insert_count.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.ntsm_oracle_fp_process_read.restype = C.c_int
        L.ntsm_oracle_fp_process_read.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        for f in ("total_kmers", "total_hits", "total_bases", "max_hits", "reads_processed", "n_distinct", "n_sites"):
            fn = getattr(L, "ntsm_oracle_fp_" + f)
            fn.restype = C.c_uint64
            fn.argtypes = [C.c_void_p]
        L.ntsm_oracle_fp_early_term.restype = C.c_int
        L.ntsm_oracle_fp_early_term.argtypes = [C.c_void_p]
        L.ntsm_oracle_fp_kmers.restype = C.c_uint64
        L.ntsm_oracle_fp_kmers.argtypes = [C.c_void_p, u64p, u64p, u64p, C.c_uint64]
        L.ntsm_oracle_fp_insert_count_mult.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint]
        L.ntsm_oracle_fp_print_counts.restype = C.c_int
        L.ntsm_oracle_fp_print_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.ntsm_oracle_fp_info_summary.restype = C.c_int
        L.ntsm_oracle_fp_info_summary.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def read_records(path):
    """All (name, sequence-bytes) the oracle's reader yields + its terminating code."""
    L = lib()
    r = L.ntsm_oracle_reader_open(os.fsencode(path))
    assert r, path
    out = []
    while True:
        n = L.ntsm_oracle_reader_next(r)
        if n < 0:
            break
        out.append((L.ntsm_oracle_reader_name(r), C.string_at(L.ntsm_oracle_reader_seq(r), n)))
    L.ntsm_oracle_reader_close(r)
    return out, int(n)


class OracleFP:
    """ntsm_oracle_fp wrapper: the reference FingerPrint on the CPU."""
    DBL_MAX = 1.7976931348623157e308

    def __init__(self, sites_path, k=19, cov=DBL_MAX, dupes=False):
        self.L = lib()
        self.h = self.L.ntsm_oracle_fp_create(os.fsencode(sites_path), k, cov, int(dupes), None)
        assert self.h, sites_path

    def process(self, seq):
        """processSingleRead; returns True once the -m threshold tripped."""
        return bool(self.L.ntsm_oracle_fp_process_read(self.h, seq, len(seq)))

    def insert_mult(self, seq, multiplier):
        """insertCount(seq, len, multiplier), src/FingerPrint.hpp:89 (the reference's own third parameter)."""
        assert 0 <= multiplier < 2 ** 32
        self.L.ntsm_oracle_fp_insert_count_mult(self.h, seq, len(seq), multiplier)

    def print_counts(self):
        """(rc, bytes) of printOptionalHeader() + printCountsMax(), src/FingerPrint.hpp:261-311."""
        import tempfile
        libc = C.CDLL(None)
        libc.fdopen.restype = C.c_void_p
        libc.fdopen.argtypes = [C.c_int, C.c_char_p]
        libc.fclose.argtypes = [C.c_void_p]
        with tempfile.TemporaryFile() as f:
            fh = libc.fdopen(os.dup(f.fileno()), b"w")
            rc = self.L.ntsm_oracle_fp_print_counts(self.h, fh)
            libc.fclose(fh)
            f.seek(0)
            return rc, f.read()

    def info_summary(self):
        buf = C.create_string_buffer(2048)
        n = self.L.ntsm_oracle_fp_info_summary(self.h, buf, 2048, None)
        return buf.raw[:n]

    def process_flat(self, bases, read_end):
        """Feed a flat stream read by read, stopping like computeCounts does."""
        buf = bases.tobytes()
        start = 0
        for e in read_end.tolist():
            if self.early_term:
                break
            self.process(buf[start:e])
            start = e + 1

    def __getattr__(self, name):
        if name in ("total_kmers", "total_hits", "total_bases", "max_hits", "reads_processed", "n_distinct", "n_sites"):
            return int(getattr(self.L, "ntsm_oracle_fp_" + name)(self.h))
        if name == "early_term":
            return bool(self.L.ntsm_oracle_fp_early_term(self.h))
        raise AttributeError(name)

    def kmers(self):
        n = self.L.ntsm_oracle_fp_kmers(self.h, None, None, None, 0)
        canon = np.zeros(n, np.uint64); hv = np.zeros(n, np.uint64); cnt = np.zeros(n, np.uint64)
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))
        self.L.ntsm_oracle_fp_kmers(self.h, p(canon), p(hv), p(cnt), n)
        return canon, hv, cnt

    def close(self):
        if self.h:
            self.L.ntsm_oracle_fp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- ntsmEval all-pairs scoring (oracle/ntsm_eval_oracle.c; parity with the reference unpinned) ---------------
_elib = None


class EvalPair(C.Structure):
    _fields_ = [("sum_joint", C.c_double), ("sum_single1", C.c_double), ("sum_single2", C.c_double), ("n_valid", C.c_uint64),
                ("hets1", C.c_uint32), ("homs1", C.c_uint32), ("hets2", C.c_uint32), ("homs2", C.c_uint32),
                ("shared_hets", C.c_uint32), ("shared_homs", C.c_uint32), ("ibs0", C.c_uint32), ("ibs2", C.c_uint32)]


def eval_lib():
    global _elib
    if _elib is None:
        path = os.path.join(ROOT, "oracle", "libntsm_eval_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libntsm_eval_oracle.so", "ntsm_eval_oracle"], check=True)
        L = C.CDLL(path)
        L.ntsm_eval_oracle_load.restype = C.c_void_p
        L.ntsm_eval_oracle_load.argtypes = [C.POINTER(C.c_char_p), C.c_uint]
        L.ntsm_eval_oracle_free.argtypes = [C.c_void_p]
        for f in ("samples", "sites"):
            fn = getattr(L, "ntsm_eval_oracle_" + f); fn.restype = C.c_uint; fn.argtypes = [C.c_void_p]
        for f in ("counts", "sums", "distinct"):
            fn = getattr(L, "ntsm_eval_oracle_" + f); fn.restype = C.POINTER(C.c_uint); fn.argtypes = [C.c_void_p]
        for f in ("total", "raw_total"):
            fn = getattr(L, "ntsm_eval_oracle_" + f); fn.restype = C.c_uint64; fn.argtypes = [C.c_void_p, C.c_uint]
        L.ntsm_eval_oracle_kmer_size.restype = C.c_uint
        L.ntsm_eval_oracle_kmer_size.argtypes = [C.c_void_p, C.c_uint]
        L.ntsm_eval_oracle_genotype.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_uint)]
        L.ntsm_eval_oracle_error_rate.restype = C.c_double
        L.ntsm_eval_oracle_error_rate.argtypes = [C.c_void_p, C.c_uint, C.c_uint64]
        L.ntsm_eval_oracle_pair.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.POINTER(EvalPair)]
        L.ntsm_eval_oracle_score.restype = C.c_double
        L.ntsm_eval_oracle_score.argtypes = [C.POINTER(EvalPair), C.c_double, C.c_double, C.c_double]
        _elib = L
    return _elib


class EvalOracle:
    """ntsm_eval_oracle wrapper: CompareCounts on the CPU (counts files in, per-pair records out)."""

    def __init__(self, files):
        self.L = eval_lib()
        arr = (C.c_char_p * len(files))(*[os.fsencode(f) for f in files])
        self.h = self.L.ntsm_eval_oracle_load(arr, len(files))
        assert self.h, files
        self.n, self.m = self.L.ntsm_eval_oracle_samples(self.h), self.L.ntsm_eval_oracle_sites(self.h)

    def counts(self):
        p = self.L.ntsm_eval_oracle_counts(self.h)
        return np.ctypeslib.as_array(p, shape=(self.n, self.m, 2)).copy()

    def pair(self, i, j, min_cov=1):
        r = EvalPair()
        self.L.ntsm_eval_oracle_pair(self.h, i, j, min_cov, C.byref(r))
        return r

    def genotype(self, i, min_cov=1):
        out = (C.c_uint * 3)()
        self.L.ntsm_eval_oracle_genotype(self.h, i, min_cov, out)
        return tuple(out)                                  # hets, homs, miss

    def close(self):
        if self.h:
            self.L.ntsm_eval_oracle_free(self.h)
            self.h = None

    def __del__(self):
        self.close()
