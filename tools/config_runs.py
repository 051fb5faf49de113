#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs[2] (ONT-like long reads + -m 10 early stop) and configs[4]
(1M-site stress set).  These are parity-test shapes (tests/test_gpu_parity.py checks them against the oracle at small
scale); this script only records their full-size rates for DESIGN.md.  Usage: config_runs.py [long] [stress]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ntsm_amd

dev = torch.device("cuda:0")
which = sys.argv[1:] or ["long", "stress"]
tmp = tempfile.mkdtemp(prefix="ntsm_cfg_")


def timed_pass(ctx, d_bases, n_bytes, d_ends, n_reads, reps=2):
    ctx.count_resident(d_bases.data_ptr(), n_bytes, d_ends.data_ptr() if d_ends is not None else 0, n_reads); ctx.sync(); ctx.reset()
    ctx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.count_resident(d_bases.data_ptr(), n_bytes, d_ends.data_ptr() if d_ends is not None else 0, n_reads)
    t = ctx.sync(); dt = (time.perf_counter() - t0) / reps
    n, ms = ctx.get_timing()
    return t, dt, ms / max(n, 1)


if "long" in which:
    sp = os.path.join(tmp, "n10.fa")
    s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
    sites = ntsm_amd.Sites(sp)
    n_reads = int(float(os.environ.get("NTSM_LONG_READS", 5e6)))
    spacing = int(os.environ.get("NTSM_LONG_SPACING", 16000))
    L = ntsm_amd.SynthLong(s, read_seed=13, spacing=spacing)
    ends, total = L.layout(0, n_reads)
    lens = np.diff(np.concatenate([[np.uint64(0)], ends + np.uint64(1)])).astype(np.int64) - 1
    srt = np.sort(lens)[::-1]; n50 = int(srt[np.searchsorted(np.cumsum(srt), srt.sum() / 2)])
    d_win = torch.from_numpy(s.windows).to(dev)
    d_ends = torch.from_numpy(ends.view(np.int64)).to(dev)
    d_bases = torch.empty(total, dtype=torch.uint8, device=dev)
    t0 = time.perf_counter(); L.device_fill(d_win.data_ptr(), 0, n_reads, d_ends.data_ptr(), total, d_bases.data_ptr()); torch.cuda.synchronize()
    gen = time.perf_counter() - t0
    ctx = ntsm_amd.Context(sites.keys)
    t, dt, kms = timed_pass(ctx, d_bases, total, None, n_reads, reps=2)
    full = {"config": "configs[2] long reads, no early stop", "reads": n_reads, "bases": int(lens.sum()), "n50": n50, "gen_s": gen,
            "kernel_ms": kms, "gbases_per_s": lens.sum() / (kms / 1e3) / 1e9, "hits": t.total_hits // 2, "hits_per_base": t.total_hits / 2 / lens.sum()}
    print(json.dumps(full)); ctx.close()
    thr = ntsm_amd.max_hits_for(len(sites.keys), 10.0)            # -m 10
    ctx = ntsm_amd.Context(sites.keys, max_hits=thr)
    t0 = time.perf_counter()
    ctx.count_resident(d_bases.data_ptr(), total, d_ends.data_ptr(), n_reads)
    t = ctx.sync(); dt = time.perf_counter() - t0
    print(json.dumps({"config": "configs[2] long reads with -m 10 (exact early stop: count, locate crossing read, take the rest out)",
                      "max_hits": thr, "early_stop": t.early_stop, "reads_consumed": t.reads_consumed, "frac_of_stream": t.reads_consumed / n_reads,
                      "total_hits": t.total_hits, "wall_s": dt, "gbases_per_s_of_whole_stream": lens.sum() / dt / 1e9}))
    ctx.close(); del d_bases

if "stress" in which:
    sp = os.path.join(tmp, "stress.fa")
    n_sites = int(float(os.environ.get("NTSM_STRESS_SITES", 1e6)))
    t0 = time.perf_counter(); s = ntsm_amd.SynthShort(424242, n_sites, read_seed=9, sites_path=sp); t_gen = time.perf_counter() - t0
    t0 = time.perf_counter(); sites = ntsm_amd.Sites(sp); t_load = time.perf_counter() - t0
    t0 = time.perf_counter(); ctx = ntsm_amd.Context(sites.keys); t_create = time.perf_counter() - t0
    n_reads = int(float(os.environ.get("NTSM_STRESS_READS", 2e8)))
    d_win = torch.from_numpy(s.windows).to(dev)
    d_bases = torch.empty(n_reads * s.stride, dtype=torch.uint8, device=dev)
    s.device_fill(d_win.data_ptr(), 0, n_reads, d_bases.data_ptr()); torch.cuda.synchronize()
    for flog in (0, 26, 27, 28):
        if flog:
            ctx.set_tuning(flog, 0)
        ctx.reset()
        t, dt, kms = timed_pass(ctx, d_bases, d_bases.numel(), None, n_reads, reps=2)
        print(json.dumps({"config": "configs[4] stress: %d sites, %d k-mers" % (n_sites, len(sites.keys)), "filter_log2": flog or "auto",
                          "reads": n_reads, "kernel_ms": kms, "gbases_per_s": n_reads * 150 / (kms / 1e3) / 1e9,
                          "hits_per_pass": t.total_hits // 2, "site_gen_s": t_gen, "site_load_s": t_load, "create_s": t_create}))
    ctx.close()

if "stress1" in which:                                   # one filter size per process: what tools/stress_profile.sh profiles
    sp = os.path.join(tmp, "stress.fa")
    n_sites = int(float(os.environ.get("NTSM_STRESS_SITES", 1e6)))
    s = ntsm_amd.SynthShort(424242, n_sites, read_seed=9, sites_path=sp)
    sites = ntsm_amd.Sites(sp)
    ctx = ntsm_amd.Context(sites.keys)
    flog = int(os.environ.get("NTSM_STRESS_FLOG", 0))
    if flog:
        ctx.set_tuning(flog, 0)
    if os.environ.get("NTSM_STRESS_KERNEL"):
        ctx.set_kernel(int(os.environ["NTSM_STRESS_KERNEL"]))
    n_reads = int(float(os.environ.get("NTSM_STRESS_READS", 1e8)))
    d_win = torch.from_numpy(s.windows).to(dev)
    d_bases = torch.empty(n_reads * s.stride, dtype=torch.uint8, device=dev)
    s.device_fill(d_win.data_ptr(), 0, n_reads, d_bases.data_ptr()); torch.cuda.synchronize()
    t, dt, kms = timed_pass(ctx, d_bases, d_bases.numel(), None, n_reads, reps=2)
    print(json.dumps({"config": "configs[4] stress: %d sites, %d k-mers" % (n_sites, len(sites.keys)), "filter_log2": flog or "auto",
                      "reads": n_reads, "kernel_ms": kms, "gbases_per_s": n_reads * 150 / (kms / 1e3) / 1e9, "hits_per_pass": t.total_hits // 2}))
    ctx.close()

if "ksweep" in which:                                    # other k through the minimizer-blocked kernel and through the generic one
    n_reads = int(float(os.environ.get("NTSM_KSWEEP_READS", 1e8)))
    ks = [int(x) for x in os.environ.get("NTSM_KSWEEP_K", "15,16,17,18,19,20,21,23,25,27,29,31").split(",")]
    for k in ks:
        sp = os.path.join(tmp, "k%d.fa" % k)
        s = ntsm_amd.SynthShort(20241218, 96287, k=k, read_seed=7, sites_path=sp)
        sites = ntsm_amd.Sites(sp, k=k)
        d_win = torch.from_numpy(s.windows).to(dev)
        d_bases = torch.empty(n_reads * s.stride, dtype=torch.uint8, device=dev)
        s.device_fill(d_win.data_ptr(), 0, n_reads, d_bases.data_ptr()); torch.cuda.synchronize()
        row = {"config": "k sweep: 96287 sites (window 31), %g reads of 150 bp" % n_reads, "k": k, "site_kmers": len(sites.keys)}
        ref = None
        for variant, name in ((0, "minimizer_kernel"), (1, "generic_kernel")):
            ctx = ntsm_amd.Context(sites.keys, k=k)
            ctx.set_kernel(variant)
            t, dt, kms = timed_pass(ctx, d_bases, d_bases.numel(), None, n_reads, reps=2)
            row[name + "_ms"] = kms
            row[name + "_gbases_per_s"] = n_reads * 150 / (kms / 1e3) / 1e9
            cur = (t.total_kmers, t.total_hits)
            assert ref is None or cur == ref, (k, cur, ref)      # both kernels count the same
            ref = cur
            ctx.close()
        row["hits_per_pass"] = ref[1] // 2
        print(json.dumps(row), flush=True)
        del d_bases
