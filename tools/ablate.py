#!/usr/bin/env python3
"""Ablations of the k=19 kernel (one process, same device): where does the time go?
  full       : hs_n10_like sites, reads with embedded site windows (the bench workload)
  no_embed   : same sites, reads without site windows (no true hits, fewer near-miss false positives)
  tiny_set   : 16 site k-mers only (filter = 64 blocks: every block load hits L1/L2, no positives, no drains)"""
import os, sys, tempfile, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ntsm_amd
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
tmp = tempfile.mkdtemp()
sp = os.path.join(tmp, "s.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
s0 = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, p_embed=0.0)
sites = ntsm_amd.Sites(sp)
d_win = torch.from_numpy(s.windows).to(dev)
d = torch.empty(n * s.stride, dtype=torch.uint8, device=dev)
def run(ctx, name):
    ctx.count_resident(d.data_ptr(), d.numel(), 0, n); ctx.sync(); ctx.set_timing(True)
    for _ in range(3): ctx.count_resident(d.data_ptr(), d.numel(), 0, n)
    t = ctx.sync(); k, ms = ctx.get_timing()
    print(json.dumps({"case": name, "ms": ms / k, "gbases_s": n * 150 / (ms / k) / 1e6, "hits_per_pass": t.total_hits // 4, "kmers_per_pass": t.total_kmers // 4}))
s.device_fill(d_win.data_ptr(), 0, n, d.data_ptr()); torch.cuda.synchronize()
ctx = ntsm_amd.Context(sites.keys); run(ctx, "full"); ctx.close()
ctx = ntsm_amd.Context(sites.keys[:16]); run(ctx, "tiny_set (reads with site windows)"); ctx.close()
s0.device_fill(d_win.data_ptr(), 0, n, d.data_ptr()); torch.cuda.synchronize()
ctx = ntsm_amd.Context(sites.keys); run(ctx, "no_embed"); ctx.close()
ctx = ntsm_amd.Context(sites.keys[:16]); run(ctx, "tiny_set + no_embed"); ctx.close()
