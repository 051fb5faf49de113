#!/usr/bin/env python3
"""CLI wall time on ONE plain FASTQ for several -t values, with phase times (block-parallel ingest)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ntsm_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
ts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16]
tmp = tempfile.mkdtemp(prefix="ntsm_thr_")
sp = os.path.join(tmp, "sites.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
fq = os.path.join(tmp, "reads.fq")
s.write_fastq(fq, 0, n)
print("file: %.2f GB" % (os.path.getsize(fq) / 1e9))
env = dict(os.environ, NTSM_PHASE_TIMES="1")
for extra in ({}, {"NTSM_BLOCK_BYTES": str(16 << 20)}):
    for t in ts:
        e = dict(env, **extra)
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(ROOT, "build", "ntsmCount"), "-s", sp, "-t", str(t), fq], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=e)
        dt = time.perf_counter() - t0
        ph = " | ".join(l[8:] for l in p.stderr.decode().split("\n") if l.startswith("[phase]"))
        print("-t %2d %s: %.2f s -> %.2f Gbases/s  [%s]" % (t, extra, dt, n * 150 / dt / 1e9, ph))
