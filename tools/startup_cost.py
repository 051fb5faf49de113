#!/usr/bin/env python3
"""tools/startup_cost.py -- the pieces of the CLI's start-up, one at a time in one process (GPU box): loading the sites file,
HIP bring-up (ntsm_warmup), the first ntsm_create (tables + upload, runtime already up) and a second one (the same without
any first-use cost), then the CLI's own phase lines for the bench's plain-FASTQ leg."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ntsm_amd
from ntsm_amd import capi

def lap(what, f):
    t0 = time.perf_counter()
    r = f()
    print("%-48s %.4f s" % (what, time.perf_counter() - t0), flush=True)
    return r

n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40_000_000
sp = "/tmp/startup_sites.fa"
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
for rep in range(2):
    sites = lap("Sites(...) load, k = 19", lambda: capi.Sites(sp, 19))
lap("ntsm_warmup(0, 5): HIP runtime + 5 streams", lambda: capi.warmup(0, 5))
lap("ntsm_staging_pool(96 MiB)", lambda: capi.staging_pool(96 << 20))
c1 = lap("ntsm_create #1", lambda: capi.Context(sites.keys, 19))
c2 = lap("ntsm_create #2", lambda: capi.Context(sites.keys, 19))
c3 = lap("ntsm_create #3", lambda: capi.Context(sites.keys, 19))
fq = "/tmp/startup.fq"
s.write_fastq(fq, 0, n_reads, threads=32)
exe = os.path.join(ROOT, "build", "ntsmCount")
for rep in range(6):
    t0 = time.perf_counter()
    p = subprocess.run([exe, "-s", sp, "-t", "16", fq], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, NTSM_PHASE_TIMES="1"))
    w = time.perf_counter() - t0
    lines = [l for l in p.stderr.decode().splitlines() if l.startswith("[phase]") or l.startswith("Time:")]
    print("wall %.3f s | " % w + " | ".join(l.replace("[phase] ", "").replace(fq, "fq") for l in lines), flush=True)
