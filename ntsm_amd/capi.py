"""ctypes bindings of include/ntsm_hip.h, include/ntsm_host.h and include/ntsm_synth.h."""
import ctypes as C
import os

import numpy as np

# torch bundles its own libamdhip64/librccl (same SONAMEs as /opt/rocm's).  Loading torch FIRST makes
# the dynamic linker hand that one runtime to libntsm_hip.so too; the other order would put two HIP
# runtimes in one process (device pointers and streams would not be interchangeable).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))


class NtsmError(RuntimeError):
    pass


def _load(name):
    """The package's libraries lie beside this file.  An experiment build named through NTSM_HIP_LIB (`make tab / m12 / xlib /
    ablation`: never shipped) is looked for under build/lib/ first; a name with a slash is taken as a path."""
    cands = [name] if os.sep in name else [os.path.join(_HERE, name), os.path.join(os.path.dirname(_HERE), "build", "lib", name)]
    for path in cands:
        if os.path.exists(path):
            return C.CDLL(path, mode=C.RTLD_GLOBAL)
    raise ImportError("%s is missing: run `make` at the repo root (the HIP path has no CPU fallback)" % cands[0])


hip_lib = _load(os.environ.get("NTSM_HIP_LIB", "libntsm_hip.so"))   # env override: A/B builds of the same ABI
host_lib = _load("libntsm_host.so")
synth_lib = _load("libntsm_synth.so")

u8p, u64p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)


class Totals(C.Structure):
    _fields_ = [("total_kmers", C.c_uint64), ("total_hits", C.c_uint64), ("total_bases", C.c_uint64),
                ("reads_consumed", C.c_uint64), ("early_stop", C.c_int32), ("reserved", C.c_int32)]


class SynthShortParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("read_len", C.c_uint32), ("n_sites", C.c_uint32), ("embed_thr", C.c_uint32),
                ("sub_thr", C.c_uint32), ("n_thr", C.c_uint32), ("pad", C.c_uint32)]


class SynthLongParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("genome_len", C.c_uint64), ("n_sites", C.c_uint32), ("spacing", C.c_uint32),
                ("sub_thr", C.c_uint32), ("n_thr", C.c_uint32)]


def _sig(lib, name, res, args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = args
    return f


H = hip_lib
_sig(H, "ntsm_create", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, u64p, C.c_uint32, C.c_int, C.c_uint64])
_sig(H, "ntsm_destroy", None, [C.c_void_p])
_sig(H, "ntsm_submit", C.c_int, [C.c_void_p, u8p, C.c_uint64, u64p, C.c_uint32])
_sig(H, "ntsm_submit_pinned", C.c_int, [C.c_void_p, u8p, C.c_uint64, u64p, C.c_uint32])
_sig(H, "ntsm_set_submit_threads", C.c_int, [C.c_void_p, C.c_int])
_sig(H, "ntsm_host_pin", C.c_int, [C.c_void_p, C.c_uint64])
_sig(H, "ntsm_host_unpin", C.c_int, [C.c_void_p])
_sig(H, "ntsm_staging_acquire", C.c_int, [C.c_void_p, C.POINTER(u8p), u64p, C.POINTER(u64p), u64p])
_sig(H, "ntsm_submit_staged", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32])
_sig(H, "ntsm_set_batch_capacity", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64])
_sig(H, "ntsm_lane_open", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)])
_sig(H, "ntsm_lane_acquire", C.c_int, [C.c_void_p, C.POINTER(u8p), u64p, C.POINTER(u64p), u64p])
_sig(H, "ntsm_lane_submit", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32])
_sig(H, "ntsm_lane_close", C.c_int, [C.c_void_p])
_sig(H, "ntsm_lane_open_packed", C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)])
_sig(H, "ntsm_lane_acquire_packed", C.c_int, [C.c_void_p, C.POINTER(u8p), C.POINTER(u8p), u64p])
_sig(H, "ntsm_lane_submit_packed", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64])
_sig(H, "ntsm_warmup", C.c_int, [C.c_int, C.c_int])
_sig(H, "ntsm_staging_pool", C.c_int, [C.c_uint64])
_sig(H, "ntsm_count_resident", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int])
_sig(H, "ntsm_sync", C.c_int, [C.c_void_p, C.POINTER(Totals)])
_sig(H, "ntsm_counts", C.c_int, [C.c_void_p, u64p])
_sig(H, "ntsm_counts_device", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), u64p])
_sig(H, "ntsm_import_reduced", C.c_int, [C.c_void_p])
_sig(H, "ntsm_allreduce", C.c_int, [C.POINTER(C.c_void_p), C.c_int])
_sig(H, "ntsm_rccl_probe", C.c_int, [])
_sig(H, "ntsm_reset", C.c_int, [C.c_void_p])
_sig(H, "ntsm_set_max_hits", C.c_int, [C.c_void_p, C.c_uint64, C.c_int])
_sig(H, "ntsm_set_timing", C.c_int, [C.c_void_p, C.c_int])
_sig(H, "ntsm_get_timing", C.c_int, [C.c_void_p, u64p, C.POINTER(C.c_double)])
_sig(H, "ntsm_set_tuning", C.c_int, [C.c_void_p, C.c_int, C.c_int])
_sig(H, "ntsm_set_kernel", C.c_int, [C.c_void_p, C.c_int])
_sig(H, "ntsm_set_armed_chunk", C.c_int, [C.c_void_p, C.c_uint64])
_sig(H, "ntsm_stream", C.c_void_p, [C.c_void_p])
_sig(H, "ntsm_debug_stats", C.c_int, [C.c_void_p, u64p])
_sig(H, "ntsm_debug_fail_after", C.c_longlong, [C.c_int, C.c_longlong])
_sig(H, "ntsm_debug_run_filter", C.c_int, [u64p, C.c_uint32, C.c_uint32, u32p, u64p])
_sig(H, "ntsm_debug_form_choice", C.c_int, [u64p, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_int)])
_sig(H, "ntsm_hash64", C.c_uint64, [C.c_uint64, C.c_int])
_sig(H, "ntsm_hash64_inv", C.c_uint64, [C.c_uint64, C.c_int])
_sig(H, "ntsm_strerror", C.c_char_p, [C.c_int])
_sig(H, "ntsm_last_hip_error", C.c_int, [])
_sig(H, "ntsm_version", C.c_char_p, [])

HO = host_lib
_sig(HO, "ntsm_sites_load", C.c_int, [C.c_char_p, C.c_uint, C.c_int, C.POINTER(C.c_void_p)])
_sig(HO, "ntsm_sites_free", None, [C.c_void_p])
_sig(HO, "ntsm_sites_n_keys", C.c_uint64, [C.c_void_p])
_sig(HO, "ntsm_sites_keys", u64p, [C.c_void_p])
_sig(HO, "ntsm_sites_n_sites", C.c_uint64, [C.c_void_p])
_sig(HO, "ntsm_sites_n_erased", C.c_uint64, [C.c_void_p])
_sig(HO, "ntsm_host_max_hits", C.c_uint64, [C.c_uint64, C.c_double])
_sig(HO, "ntsm_host_flatten", C.c_int, [C.c_char_p, C.POINTER(u8p), u64p, C.POINTER(u64p), u64p, C.POINTER(C.c_int)])
_sig(HO, "ntsm_host_free", None, [C.c_void_p])
_sig(HO, "ntsm_host_gunzip", C.c_int, [C.c_char_p, C.c_int, C.c_uint, C.POINTER(u8p), u64p])
_sig(HO, "ntsm_host_flatten_parallel_gz", C.c_int, [C.c_char_p, C.c_uint, C.c_uint, C.c_uint64, C.POINTER(u8p), u64p, C.POINTER(u64p), u64p, u64p, u64p, C.POINTER(C.c_int)])
_sig(HO, "ntsm_host_early_ingest", C.c_int, [C.c_char_p, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.POINTER(u8p), u64p, u64p, u64p, u64p])
_sig(HO, "ntsm_host_early_ingest_hand_over", C.c_int, [C.c_char_p, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.c_uint64, C.POINTER(u8p), u64p, u64p, u64p, u64p, u64p])
_sig(HO, "ntsm_host_gunzip_parallel_chunk", None, [C.c_uint64])
_sig(HO, "ntsm_host_gunzip_parallel_stats", None, [u64p])
_sig(HO, "ntsm_host_granted_cpus", C.c_uint, [])
_sig(HO, "ntsm_host_ingest_plan", None, [C.c_uint, C.c_uint, C.POINTER(C.c_uint)])
_sig(HO, "ntsm_host_debug_early_alloc_fail", None, [C.c_long])
_sig(HO, "ntsm_host_debug_gz_max_tail", None, [C.c_uint64])
_sig(HO, "ntsm_host_flatten_parallel", C.c_int, [C.c_char_p, C.c_uint, C.c_uint64, C.POINTER(u8p), u64p, C.POINTER(u64p), u64p, u64p, u64p, u64p])
_sig(HO, "ntsm_host_pack2_append", C.c_uint64, [u8p, u8p, C.c_uint64, u8p, C.c_uint64, C.c_int])
_sig(HO, "ntsm_host_pack2_impl", C.c_char_p, [])
_sig(HO, "ntsm_host_format_counts", C.c_int, [C.c_void_p, u64p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)])
_sig(HO, "ntsm_host_format_summary", C.c_int, [C.c_void_p, u64p, C.c_uint64, C.c_uint64, C.c_uint64,
                                                C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), u64p])

SY = synth_lib
_sig(SY, "ntsm_synth_sites", C.c_int, [C.c_uint64, C.c_uint32, C.c_uint, u8p, C.c_char_p, u64p])
_sig(SY, "ntsm_synth_sites_keep", C.c_int, [C.c_uint64, C.c_uint32, C.c_uint, C.c_uint, u8p, C.c_char_p, u64p])
_sig(SY, "ntsm_synth_short_params", None, [C.POINTER(SynthShortParams), C.c_uint64, C.c_uint32, C.c_uint32,
                                           C.c_double, C.c_double, C.c_double])
_sig(SY, "ntsm_synth_long_params", None, [C.POINTER(SynthLongParams), C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.c_double])
_sig(SY, "ntsm_synth_long_qtable", None, [C.c_double, C.c_double, C.c_uint32, C.c_uint32, u32p])
_sig(SY, "ntsm_synth_long_layout", C.c_uint64, [C.c_uint64, u32p, C.c_uint64, C.c_uint64, u64p])
_sig(SY, "ntsm_synth_short_fill_host", None, [C.POINTER(SynthShortParams), u8p, C.c_uint64, C.c_uint64, u8p])
_sig(SY, "ntsm_synth_long_fill_host", None, [C.POINTER(SynthLongParams), u8p, u32p, C.c_uint64, C.c_uint64, u64p, u8p])
_sig(SY, "ntsm_synth_short_fill_device", C.c_int, [C.POINTER(SynthShortParams), C.c_void_p, C.c_uint64, C.c_uint64,
                                                   C.c_void_p, C.c_void_p])
_sig(SY, "ntsm_synth_long_fill_device", C.c_int, [C.POINTER(SynthLongParams), C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                                  C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p])
_sig(SY, "ntsm_synth_short_write_fastq", C.c_int, [C.POINTER(SynthShortParams), u8p, C.c_uint64, C.c_uint64, C.c_char_p])
_sig(SY, "ntsm_synth_short_write_fastq_mt", C.c_int, [C.POINTER(SynthShortParams), u8p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint])
_sig(SY, "ntsm_synth_short_write_fastq_mt_q", C.c_int, [C.POINTER(SynthShortParams), u8p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint, C.c_uint])
_sig(SY, "ntsm_synth_long_write_fastq", C.c_int, [C.POINTER(SynthLongParams), u8p, u32p, C.c_uint64, C.c_uint64, C.c_char_p])

KEYS_CANONICAL, KEYS_HASH64 = 0, 1


def _chk(rc, what):
    if rc != 0:
        raise NtsmError("%s: %s (code %d, hipError %d)" % (what, H.ntsm_strerror(rc).decode(), rc, H.ntsm_last_hip_error()))


def _p(arr, typ):
    return arr.ctypes.data_as(typ)


def hash64(key, k):
    return int(H.ntsm_hash64(int(key), int(k)))


def hash64_inv(hv, k):
    return int(H.ntsm_hash64_inv(int(hv), int(k)))


def max_hits_for(n_distinct, cov_thresh):
    return int(HO.ntsm_host_max_hits(int(n_distinct), float(cov_thresh)))


class Sites:
    """Host-side site set (FingerPrint::initCountsHash, src/FingerPrint.hpp:490-564)."""

    def __init__(self, path, k=19, allow_dupes=False):
        h = C.c_void_p()
        if HO.ntsm_sites_load(os.fsencode(path), k, int(allow_dupes), C.byref(h)) != 0:
            raise NtsmError("file %s cannot be opened" % path)
        self._h, self.k = h, k
        n = HO.ntsm_sites_n_keys(h)
        self.keys = np.ctypeslib.as_array(HO.ntsm_sites_keys(h), shape=(n,)).copy() if n else np.zeros(0, np.uint64)
        self.n_sites = int(HO.ntsm_sites_n_sites(h))
        self.n_erased = int(HO.ntsm_sites_n_erased(h))

    def format_counts(self, counts, total_kmers):
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        out, ln = C.c_void_p(), C.c_size_t()
        rc = HO.ntsm_host_format_counts(self._h, _p(counts, u64p), int(total_kmers), C.byref(out), C.byref(ln))
        data = C.string_at(out, ln.value)
        HO.ntsm_host_free(out)
        return rc, data

    def format_summary(self, counts, total_bases, total_kmers, total_hits):
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        out, ln, cov = C.c_void_p(), C.c_size_t(), C.c_uint64()
        HO.ntsm_host_format_summary(self._h, _p(counts, u64p), int(total_bases), int(total_kmers), int(total_hits),
                                    C.byref(out), C.byref(ln), C.byref(cov))
        data = C.string_at(out, ln.value)
        HO.ntsm_host_free(out)
        return data, int(cov.value)

    def __del__(self):
        if getattr(self, "_h", None):
            HO.ntsm_sites_free(self._h)
            self._h = None


def flatten_file(path):
    """Parse a FASTA/FASTQ(.gz) file into (bases uint8[n_bytes], read_end uint64[n_reads], last_rc)."""
    b, e = u8p(), u64p()
    nb, nr, rc = C.c_uint64(), C.c_uint64(), C.c_int()
    if HO.ntsm_host_flatten(os.fsencode(path), C.byref(b), C.byref(nb), C.byref(e), C.byref(nr), C.byref(rc)) != 0:
        raise NtsmError("file %s cannot be opened" % path)
    bases = np.ctypeslib.as_array(b, shape=(nb.value,)).copy() if nb.value else np.zeros(0, np.uint8)
    ends = np.ctypeslib.as_array(e, shape=(nr.value,)).copy() if nr.value else np.zeros(0, np.uint64)
    HO.ntsm_host_free(b)
    HO.ntsm_host_free(e)
    return bases, ends, rc.value


def gunzip(path, engine=0, chunk=1 << 16):
    """(bytes delivered, final status) of the gzip decoder thread (engine 0) or zlib's gzread (engine 1)."""
    b, n = u8p(), C.c_uint64()
    rc = HO.ntsm_host_gunzip(os.fsencode(path), engine, chunk, C.byref(b), C.byref(n))
    if rc == -2:
        raise NtsmError("cannot open %s" % path)
    data = C.string_at(b, n.value)
    HO.ntsm_host_free(b)
    return data, rc


def gunzip_parallel_chunk(n_bytes):
    """Compressed bytes per chunk of the parallel plain-gzip decoder (engine >= 2); 0 = default."""
    HO.ntsm_host_gunzip_parallel_chunk(C.c_uint64(n_bytes))


def granted_cpus():
    """CPUs this process is granted: min(affinity mask, cgroup CPU quota) -- host_shape.hpp"""
    return int(HO.ntsm_host_granted_cpus())


def ingest_plan(threads_asked, cpus=0):
    """dict(cpus, feeders, decoders, early_decoders): the thread counts `ntsmCount -t threads_asked` uses on a host that grants `cpus`
    CPUs (0: this one)."""
    out = (C.c_uint * 4)()
    HO.ntsm_host_ingest_plan(int(threads_asked), int(cpus), out)
    return dict(cpus=out[0], feeders=out[1], decoders=out[2], early_decoders=out[3])


def debug_early_alloc_fail(nth):
    """Test hook: the nth chunk allocation of an early ingest from now on fails (0 = off)."""
    HO.ntsm_host_debug_early_alloc_fail(C.c_long(nth))


def debug_gz_max_tail(n_bytes):
    """Test hook: longest unparsed rest a piece of the piece-parallel gzip parse may carry on (0 = default 256 MiB)."""
    HO.ntsm_host_debug_gz_max_tail(C.c_uint64(n_bytes))


def gunzip_parallel_stats():
    """(chunks spliced, chunks dropped) of the last gunzip(path, engine >= 2) call."""
    st = (C.c_uint64 * 2)()
    HO.ntsm_host_gunzip_parallel_stats(C.cast(st, u64p))
    return int(st[0]), int(st[1])


def flatten_file_parallel(path, n_threads=4, block_bytes=1 << 20):
    """Block-parallel plain-FASTQ parse (+ sequential tail); None when the file is not eligible (use flatten_file).
    Returns (bases, read_end, info) with info = dict(blocks, parallel_records, resume)."""
    b, e = u8p(), u64p()
    nb, nr, nblk, npar, res = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = HO.ntsm_host_flatten_parallel(os.fsencode(path), n_threads, block_bytes, C.byref(b), C.byref(nb), C.byref(e), C.byref(nr),
                                       C.byref(nblk), C.byref(npar), C.byref(res))
    if rc < 0:
        raise NtsmError("flatten_parallel(%s) failed: %d" % (path, rc))
    if rc == 1:
        return None
    bases = np.ctypeslib.as_array(b, shape=(nb.value,)).copy() if nb.value else np.zeros(0, np.uint8)
    ends = np.ctypeslib.as_array(e, shape=(nr.value,)).copy() if nr.value else np.zeros(0, np.uint64)
    HO.ntsm_host_free(b)
    HO.ntsm_host_free(e)
    return bases, ends, dict(blocks=int(nblk.value), parallel_records=int(npar.value), resume=int(res.value))


def flatten_file_parallel_gz(path, n_decoders=4, n_parsers=4, sink_bytes=1 << 20):
    """Parallel gzip ingest (decoder pool + piece-parallel parse + sequential rest); None when the file is not gzip.
    Returns (bases, read_end, info); reads are in file order piece by piece, not necessarily inside a piece."""
    b, e = u8p(), u64p()
    nb, nr, npc, npar, st = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
    rc = HO.ntsm_host_flatten_parallel_gz(os.fsencode(path), n_decoders, n_parsers, sink_bytes, C.byref(b), C.byref(nb), C.byref(e), C.byref(nr),
                                          C.byref(npc), C.byref(npar), C.byref(st))
    if rc < 0:
        raise NtsmError("flatten_parallel_gz(%s) failed: %d" % (path, rc))
    if rc == 1:
        return None
    bases = np.ctypeslib.as_array(b, shape=(nb.value,)).copy() if nb.value else np.zeros(0, np.uint8)
    ends = np.ctypeslib.as_array(e, shape=(nr.value,)).copy() if nr.value else np.zeros(0, np.uint64)
    HO.ntsm_host_free(b)
    HO.ntsm_host_free(e)
    return bases, ends, dict(pieces=int(npc.value), parallel_records=int(npar.value), status=int(st.value))


def early_ingest(path, n_parsers=4, n_decoders=4, block_bytes=1 << 20, chunk_positions=1 << 20, max_chunks=64, n_consumers=2, hand_over_after=None):
    """Early ingest (early_ingest.hpp) of one file into packed chunks; None when the file is not taken by that path.
    Returns (text bytes: A C G T for valid positions, N otherwise; n_reads; n_bases; parallel_records) -- and, with
    hand_over_after = n (the consumers ask for a gzip stream once n chunks are drained; its rest is read sequentially), a fifth
    element: the reads that did not go through the chunks."""
    t = u8p()
    nt_, nr, nb, npar, nrest = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    if hand_over_after is None:
        rc = HO.ntsm_host_early_ingest(os.fsencode(path), n_parsers, n_decoders, block_bytes, chunk_positions, max_chunks, n_consumers,
                                       C.byref(t), C.byref(nt_), C.byref(nr), C.byref(nb), C.byref(npar))
    else:
        rc = HO.ntsm_host_early_ingest_hand_over(os.fsencode(path), n_parsers, n_decoders, block_bytes, chunk_positions, max_chunks, n_consumers, hand_over_after,
                                                 C.byref(t), C.byref(nt_), C.byref(nr), C.byref(nb), C.byref(npar), C.byref(nrest))
    if rc == 1:
        return None
    if rc:
        raise NtsmError("early_ingest(%s) failed: %d" % (path, rc))
    text = C.string_at(t, nt_.value)
    HO.ntsm_host_free(t)
    if hand_over_after is None:
        return text, int(nr.value), int(nb.value), int(npar.value)
    return text, int(nr.value), int(nb.value), int(npar.value), int(nrest.value)


def flatten_reads(reads):
    """Flat-stream layout from a list of bytes objects."""
    parts, ends, off = [], [], 0
    for r in reads:
        parts.append(bytes(r))
        parts.append(b"N")
        off += len(r)
        ends.append(off)
        off += 1
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy(), np.asarray(ends, dtype=np.uint64)


class Context:
    """One GPU context (include/ntsm_hip.h)."""

    def __init__(self, keys, k=19, device=0, key_kind=KEYS_CANONICAL, max_hits=0):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        self._h = C.c_void_p()
        self.n_kmers, self.k = len(keys), k
        _chk(H.ntsm_create(C.byref(self._h), device, k, _p(keys, u64p), len(keys), key_kind, int(max_hits)), "ntsm_create")

    def submit(self, bases, read_end):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_end = np.ascontiguousarray(read_end, dtype=np.uint64)
        _chk(H.ntsm_submit(self._h, _p(bases, u8p), bases.size, _p(read_end, u64p), read_end.size), "ntsm_submit")

    def submit_pinned(self, bases, read_end):
        """ntsm_submit_pinned: `bases` must be a uint8 array inside memory pinned with host_pin (or hipHostMalloc); it must stay
        untouched until the second next submit_pinned has returned, or until sync()."""
        assert bases.dtype == np.uint8 and bases.flags.c_contiguous
        read_end = np.ascontiguousarray(read_end, dtype=np.uint64)
        _chk(H.ntsm_submit_pinned(self._h, _p(bases, u8p), bases.size, _p(read_end, u64p), read_end.size), "ntsm_submit_pinned")

    def set_submit_threads(self, n):
        _chk(H.ntsm_set_submit_threads(self._h, int(n)), "ntsm_set_submit_threads")

    def open_lane(self, cap_bytes=0, cap_reads=0, packed_only=False):
        """A producer lane (ntsm_lane_*): one host thread's private staging into this context."""
        return Lane(self, cap_bytes, cap_reads, packed_only)

    def set_max_hits(self, max_hits, armed=True):
        _chk(H.ntsm_set_max_hits(self._h, int(max_hits), int(bool(armed))), "ntsm_set_max_hits")

    def set_batch_capacity(self, cap_bytes, cap_reads):
        _chk(H.ntsm_set_batch_capacity(self._h, cap_bytes, cap_reads), "ntsm_set_batch_capacity")

    def count_resident(self, d_bases_ptr, n_bytes, d_read_end_ptr, n_reads, sign=1):
        _chk(H.ntsm_count_resident(self._h, C.c_void_p(d_bases_ptr), n_bytes,
                                   C.c_void_p(d_read_end_ptr) if d_read_end_ptr else None, n_reads, sign), "ntsm_count_resident")

    def sync(self):
        t = Totals()
        _chk(H.ntsm_sync(self._h, C.byref(t)), "ntsm_sync")
        return t

    def counts(self):
        out = np.zeros(self.n_kmers, dtype=np.uint64)
        _chk(H.ntsm_counts(self._h, _p(out, u64p)), "ntsm_counts")
        return out

    def counts_device(self):
        p, n = C.c_void_p(), C.c_uint64()
        _chk(H.ntsm_counts_device(self._h, C.byref(p), C.byref(n)), "ntsm_counts_device")
        return p.value, n.value

    def import_reduced(self):
        _chk(H.ntsm_import_reduced(self._h), "ntsm_import_reduced")

    def reset(self):
        _chk(H.ntsm_reset(self._h), "ntsm_reset")

    def set_timing(self, on):
        _chk(H.ntsm_set_timing(self._h, int(on)), "ntsm_set_timing")

    def get_timing(self):
        n, ms = C.c_uint64(), C.c_double()
        _chk(H.ntsm_get_timing(self._h, C.byref(n), C.byref(ms)), "ntsm_get_timing")
        return n.value, ms.value

    def set_tuning(self, filter_log2_bits=0, grid_blocks=0):
        _chk(H.ntsm_set_tuning(self._h, filter_log2_bits, grid_blocks), "ntsm_set_tuning")

    def set_kernel(self, variant):
        _chk(H.ntsm_set_kernel(self._h, int(variant)), "ntsm_set_kernel")

    def set_armed_chunk(self, chunk_bytes):
        _chk(H.ntsm_set_armed_chunk(self._h, int(chunk_bytes)), "ntsm_set_armed_chunk")

    def debug_stats(self):
        """dict(exotic_tiles, launches_tab, launches_k19, launches_generic) -- include/ntsm_hip.h ntsm_debug_stats"""
        out = np.zeros(8, dtype=np.uint64)
        _chk(H.ntsm_debug_stats(self._h, _p(out, u64p)), "ntsm_debug_stats")
        return dict(exotic_tiles=int(out[0]), launches_tab=int(out[1]), launches_k19=int(out[2]), launches_generic=int(out[3]),
                    queued_windows=int(out[4]), two_level=int(out[5]) == 1, run_form=int(out[5]) == 2, bloom_words=int(out[6]), site_minimizers=int(out[7]))

    @property
    def stream(self):
        return H.ntsm_stream(self._h)

    def close(self):
        if self._h:
            H.ntsm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Lane:
    def __init__(self, ctx, cap_bytes=0, cap_reads=0, packed_only=False):
        self._h = C.c_void_p()
        self._ctx = ctx                                   # keeps the context alive
        if packed_only:
            _chk(H.ntsm_lane_open_packed(ctx._h, cap_bytes, C.byref(self._h)), "ntsm_lane_open_packed")
        else:
            _chk(H.ntsm_lane_open(ctx._h, cap_bytes, cap_reads, C.byref(self._h)), "ntsm_lane_open")

    def submit(self, bases, read_end):
        """acquire + copy + submit; the batch must fit the lane's slot."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_end = np.ascontiguousarray(read_end, dtype=np.uint64)
        hb, hr = u8p(), u64p()
        cb, cr = C.c_uint64(), C.c_uint64()
        _chk(H.ntsm_lane_acquire(self._h, C.byref(hb), C.byref(cb), C.byref(hr), C.byref(cr)), "ntsm_lane_acquire")
        if bases.size > cb.value or read_end.size > cr.value:
            _chk(H.ntsm_lane_submit(self._h, 0, 0), "ntsm_lane_submit")     # give the slot back
            raise NtsmError("batch larger than the lane's slot")
        C.memmove(hb, bases.ctypes.data, bases.size)
        C.memmove(hr, read_end.ctypes.data, read_end.size * 8)
        _chk(H.ntsm_lane_submit(self._h, bases.size, read_end.size), "ntsm_lane_submit")

    def submit_packed(self, reads, force_scalar=False):
        """acquire_packed + pack every read (bytes objects) with the host packer + submit_packed.  force_scalar: False / 0 = the
        best form the CPU has (AVX-512 VBMI, AVX2), True / 1 = the portable one, 2 = at most AVX2"""
        pc, pv, cap = u8p(), u8p(), C.c_uint64()
        _chk(H.ntsm_lane_acquire_packed(self._h, C.byref(pc), C.byref(pv), C.byref(cap)), "ntsm_lane_acquire_packed")
        pos, n_bases = 0, 0
        for r in reads:
            if pos + (len(r) & ~31) + 32 > cap.value:                                   # pack2_extent
                _chk(H.ntsm_lane_submit_packed(self._h, 0, 0, 0), "ntsm_lane_submit_packed")   # give the slot back
                raise NtsmError("batch larger than the lane's slot")
            buf = (C.c_uint8 * max(1, len(r))).from_buffer_copy(bytes(r) if len(r) else b"\0")
            pos = HO.ntsm_host_pack2_append(pc, pv, pos, buf, len(r), int(force_scalar))
            n_bases += len(r)
        _chk(H.ntsm_lane_submit_packed(self._h, pos, len(reads), n_bases), "ntsm_lane_submit_packed")

    def close(self):
        if self._h:
            h, self._h = self._h, C.c_void_p()
            _chk(H.ntsm_lane_close(h), "ntsm_lane_close")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def allreduce(contexts):
    """ntsm_allreduce: one process, one context per GPU -- RCCL SUM of the count vectors + totals; afterwards every
    context's counts()/sync() report the job-wide result."""
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    _chk(H.ntsm_allreduce(arr, len(contexts)), "ntsm_allreduce")


FAULT_DEVICE_ALLOC, FAULT_H2D, FAULT_PINNED_ALLOC = 1, 2, 3


def debug_fail_after(kind, nth):
    """ntsm_debug_fail_after: the nth call of `kind` (FAULT_*) from now on fails; 0 disarms.  Returns the calls of that kind
    seen since the previous arming."""
    return int(H.ntsm_debug_fail_after(int(kind), int(nth)))


def debug_run_filter(keys, kib=0):
    """The run-anchored kernel's filter for `keys` (canonical 19-mer codes), built by the host code of libntsm_hip.so without a
    device: uint32 array [n_blocks, 4]."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    nb = C.c_uint64()
    _chk(H.ntsm_debug_run_filter(_p(keys, u64p), len(keys), kib, None, C.byref(nb)), "ntsm_debug_run_filter")
    out = np.zeros((nb.value, 4), dtype=np.uint32)
    _chk(H.ntsm_debug_run_filter(_p(keys, u64p), len(keys), kib, _p(out, u32p), C.byref(nb)), "ntsm_debug_run_filter")
    return out


def host_pin(arr):
    """ntsm_host_pin (hipHostRegister) on a numpy array's memory; undo with host_unpin before the array is released."""
    _chk(H.ntsm_host_pin(C.c_void_p(arr.ctypes.data), arr.nbytes), "ntsm_host_pin")


def host_unpin(arr):
    _chk(H.ntsm_host_unpin(C.c_void_p(arr.ctypes.data)), "ntsm_host_unpin")


FORM_NAMES = ("one_level", "two_level", "run", "generic")


def debug_form_choice(keys, k=19, key_kind=0):
    """Which kernel form ntsm_create would choose for `keys` (host code only, no device): one of FORM_NAMES."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    f = C.c_int(-1)
    _chk(H.ntsm_debug_form_choice(_p(keys, u64p), len(keys), k, key_kind, C.byref(f)), "ntsm_debug_form_choice")
    return FORM_NAMES[f.value]


def warmup(device=0, n_streams=0):
    _chk(H.ntsm_warmup(device, n_streams), "ntsm_warmup")


def staging_pool(n_bytes):
    _chk(H.ntsm_staging_pool(n_bytes), "ntsm_staging_pool")


class SynthShort:
    """Seeded short-read workload (ntsm_amd/csrc/synth.h); sites + reads share one object."""

    def __init__(self, sites_seed, n_sites, k=19, read_seed=7, read_len=150, p_embed=0.10, p_sub=0.01, p_n=5e-4,
                 sites_path=None, min_keep=0):
        """min_keep: smallest number of k-mer start positions a site keeps (0 = 3 .. 13 uniformly; 13 = every k-mer)."""
        self.n_sites, self.k, self.read_len = n_sites, k, read_len
        self.windows = np.zeros(n_sites * 2 * 32, dtype=np.uint8)
        nk = C.c_uint64()
        rc = SY.ntsm_synth_sites_keep(sites_seed, n_sites, k, min_keep, _p(self.windows, u8p),
                                      os.fsencode(sites_path) if sites_path else None, C.byref(nk))
        if rc:
            raise NtsmError("ntsm_synth_sites failed: %d" % rc)
        self.n_kmers = int(nk.value)
        self.params = SynthShortParams()
        SY.ntsm_synth_short_params(C.byref(self.params), read_seed, read_len, n_sites, p_embed, p_sub, p_n)
        self.stride = read_len + 1

    def host_bytes(self, r0, n_reads):
        out = np.zeros(n_reads * self.stride, dtype=np.uint8)
        SY.ntsm_synth_short_fill_host(C.byref(self.params), _p(self.windows, u8p), r0 * self.stride, out.size, _p(out, u8p))
        return out

    def read_end(self, n_reads):
        return (np.arange(n_reads, dtype=np.uint64) * np.uint64(self.stride)) + np.uint64(self.read_len)

    def device_fill(self, d_windows_ptr, r0, n_reads, d_out_ptr, stream=None):
        rc = SY.ntsm_synth_short_fill_device(C.byref(self.params), C.c_void_p(d_windows_ptr), r0 * self.stride,
                                             n_reads * self.stride, C.c_void_p(d_out_ptr), C.c_void_p(stream) if stream else None)
        if rc:
            raise NtsmError("device fill failed: %d" % rc)

    def write_fastq(self, path, r0, n_reads, threads=1, qual_model=0):
        """FASTQ of reads [r0, r0 + n_reads); threads > 1: the same bytes written by several threads (plain output).
        qual_model 0: constant 'I'; 1: Illumina-like qualities (synth.h: ntsm_synth_qual_char)."""
        rc = SY.ntsm_synth_short_write_fastq_mt_q(C.byref(self.params), _p(self.windows, u8p), r0, n_reads, os.fsencode(path), threads, qual_model)
        if rc:
            raise NtsmError("write_fastq failed: %d" % rc)


class SynthLong:
    """Seeded long-read workload (BASELINE.json configs[2]): log-normal lengths, implicit mini-genome with one
    site window every `spacing` bases, substitution errors (ntsm_amd/csrc/synth.h)."""

    def __init__(self, short: "SynthShort", read_seed=13, spacing=20000, mu=9.6, sigma=0.6, lo=200, hi=200000,
                 p_sub=0.05, p_n=5e-4):
        self.short, self.seed = short, read_seed
        self.params = SynthLongParams()
        SY.ntsm_synth_long_params(C.byref(self.params), read_seed, short.n_sites, spacing, p_sub, p_n)
        self.qtable = np.zeros(257, dtype=np.uint32)
        SY.ntsm_synth_long_qtable(mu, sigma, lo, hi, _p(self.qtable, u32p))

    def layout(self, r0, n_reads):
        ends = np.zeros(n_reads, dtype=np.uint64)
        total = SY.ntsm_synth_long_layout(self.seed, _p(self.qtable, u32p), r0, n_reads, _p(ends, u64p))
        return ends, int(total)

    def host_bytes(self, r0, n_reads):
        ends, total = self.layout(r0, n_reads)
        out = np.zeros(total, dtype=np.uint8)
        SY.ntsm_synth_long_fill_host(C.byref(self.params), _p(self.short.windows, u8p), _p(self.qtable, u32p), r0, n_reads,
                                     _p(ends, u64p), _p(out, u8p))
        return out, ends

    def device_fill(self, d_windows_ptr, r0, n_reads, d_read_end_ptr, n_bytes, d_out_ptr, stream=None):
        rc = SY.ntsm_synth_long_fill_device(C.byref(self.params), C.c_void_p(d_windows_ptr), None, r0, n_reads,
                                            C.c_void_p(d_read_end_ptr), n_bytes, C.c_void_p(d_out_ptr),
                                            C.c_void_p(stream) if stream else None)
        if rc:
            raise NtsmError("device fill failed: %d" % rc)

    def write_fastq(self, path, r0, n_reads):
        rc = SY.ntsm_synth_long_write_fastq(C.byref(self.params), _p(self.short.windows, u8p), _p(self.qtable, u32p), r0, n_reads,
                                            os.fsencode(path))
        if rc:
            raise NtsmError("write_fastq failed: %d" % rc)
