"""Parity of the tabulated k = 19 kernel (ntsm_set_kernel 3, only in -DNTSM_WITH_TAB builds) against the oracle -- run by
tests/test_gpu_parity.py::test_tabulated_kernel_paths as a subprocess with NTSM_HIP_LIB=libntsm_hip_tab.so.
Inputs that take its special paths: (a) reads made of site sequence -- far more windows pass the filter than a wave's
queue slot holds, so they are looked up in line; (b) a clean stream with a few foreign bytes -- only the tiles that hold
them go to the exact kernel; (c) lowercase / U / N-rich input stays on the tabulated kernel; (d) several launches reuse
the per-stream buffers; (e) the filter sizes its block map treats differently, against the default kernel's counts;
(f) a site set big enough to choose the two-level tables by itself: variant 3 hands its exotic tiles to
the ONE-level k = 19 kernel, so ntsm_set_kernel(ctx, 3) must rebuild the one-level tables first."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ntsm_amd as nt
from oracle_binding import OracleFP

path = sys.argv[1]
sites = nt.Sites(path)
s = nt.SynthShort(sites_seed=20241218, n_sites=96287, read_seed=99)
rng = np.random.default_rng(7)
win = s.windows.reshape(-1, 32)[:, :31]
# (a) 40k site windows back to back, each followed by 'N'
pick = np.frombuffer(b"ACGT", dtype=np.uint8)[win[rng.integers(0, win.shape[0], 40_000)]]   # windows are stored as codes 0..3
dense = np.concatenate([pick, np.full((pick.shape[0], 1), ord("N"), np.uint8)], axis=1).reshape(-1)
ends = (np.arange(pick.shape[0], dtype=np.uint64) * np.uint64(32)) + np.uint64(31)
# (b)+(c) 6000 seeded reads, some lowercase, some T -> U, extra N, and three foreign bytes far apart
n = 6000
clean = s.host_bytes(0, n).copy()
low = rng.random(clean.size) < 0.3
clean[low & (clean != ord("N"))] |= 0x20
tmask = (clean == ord("T")) & (rng.random(clean.size) < 0.5)
clean[tmask] = ord("U")
clean[rng.integers(0, clean.size, 200)] = ord("n")
for r in range(n):                                   # keep the terminators
    clean[r * s.stride + s.read_len] = ord("N")
foreign = clean.copy()
for at, b in ((1000, ord("R")), (70_000, 0), (500_000, ord("-"))):
    foreign[at] = b
cends = s.read_end(n)
for name, buf, e, exotic in (("dense", dense, ends, 0), ("clean", clean, cends, 0), ("foreign", foreign, cends, 3)):
    fp = OracleFP(path)
    fp.process_flat(buf, e)
    ctx = nt.Context(sites.keys)
    ctx.set_kernel(3)
    for rep in range(2):                             # second launch: buffers reused, counts double
        ctx.submit(buf, e)
    t = ctx.sync()
    st = ctx.debug_stats()
    assert np.array_equal(ctx.counts(), 2 * fp.kmers()[2]), name
    assert (t.total_kmers, t.total_hits, t.total_bases) == (2 * fp.total_kmers, 2 * fp.total_hits, 2 * fp.total_bases), name
    assert st["launches_tab"] == 2 and st["exotic_tiles"] == 2 * exotic, (name, st)
    if name == "dense":
        assert fp.total_hits > 0.3 * fp.total_kmers and st["queued_windows"] < fp.total_hits   # most were looked up in line
    ctx.close()

# (e) every filter size the tabulated path maps differently, against the default kernel on the same reads
n = 300_000
bases, ends = s.host_bytes(0, n), s.read_end(n)
ref = nt.Context(sites.keys)
ref.submit(bases, ends)
tr, cr = ref.sync(), ref.counts()
ref.close()
for flog in (0, 20, 25, 124):
    ctx = nt.Context(sites.keys)
    if flog:
        ctx.set_tuning(flog, 0)
    ctx.set_kernel(3)
    ctx.submit(bases, ends)
    t = ctx.sync()
    assert np.array_equal(ctx.counts(), cr) and (t.total_kmers, t.total_hits) == (tr.total_kmers, tr.total_hits), flog
    ctx.close()

# (f) 250,000 sites = 4 M site k-mers: beyond the automatic switch to two levels (3.1 M)
import tempfile
with tempfile.TemporaryDirectory() as d:
    bp = os.path.join(d, "big.fa")
    sb = nt.SynthShort(sites_seed=777, n_sites=600_000, read_seed=3, p_embed=0.3, sites_path=bp)   # from 8 M keys on two levels are the automatic choice (below: the run-anchored kernel)
    big = nt.Sites(bp)
    assert len(big.keys) > 7_000_000
    n = 4000
    buf = sb.host_bytes(0, n).copy()
    for at, b in ((900, ord("R")), (150_000, 0), (400_000, ord("-")), (590_000, ord("*"))):
        buf[at] = b
    e = sb.read_end(n)
    fp = OracleFP(bp)
    fp.process_flat(buf, e)
    ctx = nt.Context(big.keys)
    assert ctx.debug_stats()["two_level"] is True
    ctx.set_kernel(3)
    assert ctx.debug_stats()["two_level"] is False
    ctx.submit(buf, e)
    t = ctx.sync()
    st = ctx.debug_stats()
    assert np.array_equal(ctx.counts(), fp.kmers()[2]) and (t.total_kmers, t.total_hits) == (fp.total_kmers, fp.total_hits)
    assert st["launches_tab"] == 1 and st["exotic_tiles"] >= 3, st
    ctx.set_kernel(0)                                    # back to automatic: two levels again, same counts
    assert ctx.debug_stats()["two_level"] is True
    ctx.submit(buf, e)
    assert np.array_equal(ctx.counts(), fp.kmers()[2])
    ctx.close()
print("ok")
