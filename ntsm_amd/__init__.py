"""ntsm_amd -- MI355X-native k-mer counting path of ntsmCount.

The product is native: ntsm_amd/libntsm_hip.so (HIP kernels + C ABI, include/ntsm_hip.h),
ntsm_amd/libntsm_host.so (host reader / site loader / report formatting) and build/ntsmCount (CLI).
This package is only the ctypes plumbing tests and bench.py use; importing it fails loudly when the
HIP library has not been built (there is no CPU fallback)."""
from . import capi
from .capi import (Context, Lane, Sites, SynthLong, SynthShort, flatten_file, hash64, hash64_inv, hip_lib, host_lib,
                   max_hits_for, synth_lib, warmup, staging_pool, allreduce, NtsmError)

__all__ = ["capi", "Context", "Lane", "warmup", "staging_pool", "Sites", "SynthShort", "SynthLong", "flatten_file", "hash64", "hash64_inv", "hip_lib",
           "host_lib", "synth_lib", "max_hits_for", "allreduce", "NtsmError"]
