#!/usr/bin/env python3
"""tools/stress_sweep.py [variant:tuning ...] -- BASELINE.json configs[4] (1 M sites, 16 M k-mers) through several kernel
forms / filter sizes in ONE process (the site set is generated once).  variant = ntsm_set_kernel code (0 auto, 2 one level,
4 two levels), tuning = ntsm_set_tuning code (0 none; 10..30 / 100..130 block filter bits; 200..299 Bloom bits).  A second
tuning after a comma is applied after the first (e.g. 4:126,273).  One JSON line per configuration.
Environment: NTSM_STRESS_SITES (1e6), NTSM_STRESS_READS (1e8), NTSM_STRESS_SEED (424242), NTSM_STRESS_MIN_KEEP (0 = 3..13 k-mers per
allele; 13 = all of them: with 96287 sites and seed 20241218 that is bench.py's n10_full set of 2,503,462 k-mers),
NTSM_STRESS_P_EMBED (0.10)."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ntsm_amd

dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp(prefix="ntsm_sweep_")
sp = os.path.join(tmp, "stress.fa")
n_sites = int(float(os.environ.get("NTSM_STRESS_SITES", 1e6)))
n_reads = int(float(os.environ.get("NTSM_STRESS_READS", 1e8)))
s = ntsm_amd.SynthShort(int(os.environ.get("NTSM_STRESS_SEED", 424242)), n_sites, read_seed=9, sites_path=sp,
                        min_keep=int(os.environ.get("NTSM_STRESS_MIN_KEEP", 0)), p_embed=float(os.environ.get("NTSM_STRESS_P_EMBED", 0.10)))
sites = ntsm_amd.Sites(sp)
d_win = torch.from_numpy(s.windows).to(dev)
d_bases = torch.empty(n_reads * s.stride, dtype=torch.uint8, device=dev)
s.device_fill(d_win.data_ptr(), 0, n_reads, d_bases.data_ptr())
torch.cuda.synchronize()
ref = None
for spec in (sys.argv[1:] or ["0:0"]):
    variant, tun = spec.split(":")
    ctx = ntsm_amd.Context(sites.keys)
    ctx.set_kernel(int(variant))
    for t in tun.split(","):
        if int(t):
            ctx.set_tuning(int(t), 0)
    st = ctx.debug_stats()
    ctx.count_resident(d_bases.data_ptr(), d_bases.numel(), 0, n_reads); ctx.sync(); ctx.reset()
    ctx.set_timing(True)
    for _ in range(3):
        ctx.count_resident(d_bases.data_ptr(), d_bases.numel(), 0, n_reads)
    t = ctx.sync()
    n, ms = ctx.get_timing()
    cur = (t.total_kmers // 3, t.total_hits // 3)
    assert ref is None or cur == ref, (spec, cur, ref)
    ref = cur
    print(json.dumps({"lib": os.environ.get("NTSM_HIP_LIB", "libntsm_hip.so"), "spec": spec, "two_level": st["two_level"], "run_form": st.get("run_form", False), "bloom_MiB": st["bloom_words"] * 4 / 2 ** 20,
                      "site_minimizers": st["site_minimizers"], "site_kmers": len(sites.keys), "reads": n_reads, "kernel_ms": ms / n,
                      "gbases_per_s": n_reads * 150 / (ms / n / 1e3) / 1e9, "hits_per_pass": cur[1]}), flush=True)
    ctx.close()
