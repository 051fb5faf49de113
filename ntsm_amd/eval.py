"""ctypes plumbing for include/ntsm_eval_hip.h (all-pairs scoring of ntsmEval on the GPU); used by tests and tools.
Loaded on demand: `import ntsm_amd.eval`.  Fails loudly when libntsm_eval_hip.so has not been built."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_path = os.path.join(_HERE, "libntsm_eval_hip.so")
if not os.path.exists(_path):
    raise ImportError("%s is missing: run `make` (there is no CPU fallback)" % _path)
lib = C.CDLL(_path)

RECORD = np.dtype([("sum_joint", "<f8"), ("sum_single1", "<f8"), ("sum_single2", "<f8"), ("n_valid", "<u8"),
                   ("hets1", "<u4"), ("homs1", "<u4"), ("hets2", "<u4"), ("homs2", "<u4"),
                   ("shared_hets", "<u4"), ("shared_homs", "<u4"), ("ibs0", "<u4"), ("ibs2", "<u4")])
assert RECORD.itemsize == 64

lib.ntsm_eval_pairs.restype = C.c_int
lib.ntsm_eval_pairs.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_double)]


def pair_index(i, j, n):
    """Position of pair (i < j) in the output (include/ntsm_eval_hip.h: ntsm_eval_pair_index)."""
    return i * n - i * (i + 1) // 2 + (j - i - 1)


def pairs(counts, min_cov=1, device=0):
    """counts: uint32 array [n_samples][n_sites][2].  Returns (records for every i < j in row order, kernel ms)."""
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    n, m = counts.shape[0], counts.shape[1]
    out = np.zeros(n * (n - 1) // 2, dtype=RECORD)
    ms = C.c_double()
    rc = lib.ntsm_eval_pairs(device, counts.ctypes.data, n, m, min_cov, out.ctypes.data, C.byref(ms))
    if rc:
        raise RuntimeError("ntsm_eval_pairs failed: %d" % rc)
    return out, ms.value
