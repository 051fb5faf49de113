#!/usr/bin/env python3
"""End-to-end rates that are NOT bench.py's `value`:
  (1) PCIe-inclusive: ntsm_submit from host memory (pinned double buffering, H2D + kernel overlapped);
  (2) CLI: build/ntsmCount on a FASTQ file (single-threaded parse + staging + GPU)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ntsm_amd

n_sub = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40_000_000
n_cli = int(float(sys.argv[2])) if len(sys.argv) > 2 else 3_000_000
tmp = tempfile.mkdtemp(prefix="ntsm_e2e_")
sp = os.path.join(tmp, "sites.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
t0 = time.perf_counter(); sites = ntsm_amd.Sites(sp); print("site load: %.2f s (%d k-mers)" % (time.perf_counter() - t0, len(sites.keys)))
dev = torch.device("cuda:0")
d_win = torch.from_numpy(s.windows).to(dev)
d = torch.empty(n_sub * s.stride, dtype=torch.uint8, device=dev)
s.device_fill(d_win.data_ptr(), 0, n_sub, d.data_ptr()); torch.cuda.synchronize()
host = d.cpu().numpy(); del d
t0 = time.perf_counter(); ctx = ntsm_amd.Context(sites.keys); print("ntsm_create (tables + upload): %.2f s" % (time.perf_counter() - t0))
per = 400_000                                   # reads per batch (60 MB)
ends = s.read_end(per)
ctx.submit(host[:per * s.stride], ends); ctx.sync(); ctx.reset()
t0 = time.perf_counter()
for b in range(n_sub // per):
    ctx.submit(host[b * per * s.stride:(b + 1) * per * s.stride], ends)
t = ctx.sync(); dt = time.perf_counter() - t0
print("submit path (host buffers, incl. memcpy to pinned + PCIe): %.1f Gbases/s (%d reads, %.2f s)" % (t.total_bases / dt / 1e9, n_sub, dt))
ctx.close()
fq = os.path.join(tmp, "reads.fq")
t0 = time.perf_counter(); s.write_fastq(fq, 0, n_cli); print("wrote %s in %.1f s" % (fq, time.perf_counter() - t0))
t0 = time.perf_counter(); b_, e_, _ = ntsm_amd.flatten_file(fq); dt = time.perf_counter() - t0
print("host parse only (SeqReader -> flat stream, 1 thread): %.2f s -> %.2f Gbases/s" % (dt, n_cli * 150 / dt / 1e9)); del b_, e_
t0 = time.perf_counter()
p = subprocess.run([os.path.join(ROOT, "build", "ntsmCount"), "-s", sp, fq], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
dt = time.perf_counter() - t0
print("CLI end to end: %.2f s for %d reads -> %.3f Gbases/s (includes site-table build)" % (dt, n_cli, n_cli * 150 / dt / 1e9))
print(p.stderr.decode()[-400:])

# -t N over N files (reference semantics: parallel over files)
nf = 8
parts = []
t0 = time.perf_counter()
for i in range(nf):
    fp_ = os.path.join(tmp, "part%d.fq" % i); s.write_fastq(fp_, i * (n_cli // nf), n_cli // nf); parts.append(fp_)
print("wrote %d files in %.1f s" % (nf, time.perf_counter() - t0))
for t in (1, 8):
    t0 = time.perf_counter()
    p = subprocess.run([os.path.join(ROOT, "build", "ntsmCount"), "-s", sp, "-t", str(t)] + parts, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    print("CLI %d files -t %d: %.2f s -> %.3f Gbases/s" % (nf, t, dt, (n_cli // nf) * nf * 150 / dt / 1e9))

# -t N on ONE plain FASTQ: block-parallel ingest (parallel_fastq.hpp)
for t in (1, 4, 8, 16, 32):
    t0 = time.perf_counter()
    p = subprocess.run([os.path.join(ROOT, "build", "ntsmCount"), "-s", sp, "-t", str(t), fq], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    print("CLI single file -t %d: %.2f s -> %.3f Gbases/s" % (t, dt, n_cli * 150 / dt / 1e9))
