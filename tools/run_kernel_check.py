#!/usr/bin/env python3
"""tools/run_kernel_check.py [reads] -- the run-anchored kernel (ntsm_set_kernel 5) beside the minimizer-blocked one (0) on the bench
set (1.54 M keys) and the 2.5 M-key set: identical per-k-mer counts and totals, then the rate of both on `reads` resident reads."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ntsm_amd
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "5"])]
tunings = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0"])]
tmp = tempfile.mkdtemp()
for keep, tag in ((0, "bench 1.54M"), (13, "n10_full 2.5M")):
    sp = os.path.join(tmp, "s%d.fa" % keep)
    s = ntsm_amd.SynthShort(20241218, 96287, read_seed=9, sites_path=sp, min_keep=keep)
    sites = ntsm_amd.Sites(sp)
    d_win = torch.from_numpy(s.windows).to(dev)
    d = torch.empty(n * s.stride, dtype=torch.uint8, device=dev)
    s.device_fill(d_win.data_ptr(), 0, n, d.data_ptr()); torch.cuda.synchronize()
    ref = None
    for v in variants:
        for tun in (tunings if v == 5 else [0]):
            ctx = ntsm_amd.Context(sites.keys)
            ctx.set_kernel(v)
            if tun:
                ctx.set_tuning(tun, 0)
            ctx.count_resident(d.data_ptr(), d.numel(), 0, n)
            t = ctx.sync()
            cnt = ctx.counts()
            cur = (t.total_kmers, t.total_hits)
            if ref is None:
                ref = (cur, cnt)
            same = cur == ref[0] and bool(np.array_equal(cnt, ref[1]))
            ctx.reset(); ctx.set_timing(True)
            for _ in range(3):
                ctx.count_resident(d.data_ptr(), d.numel(), 0, n)
            ctx.sync(); k, ms = ctx.get_timing()
            print(json.dumps({"set": tag, "variant": v, "tuning": tun, "equal_to_first": same, "kmers": cur[0], "hits": cur[1], "ref": ref[0],
                              "n_diff": int((cnt != ref[1]).sum()), "ms": round(ms / k, 3), "gbases_per_s": round(n * 150 / (ms / k) / 1e6, 1)}), flush=True)
            ctx.close()
    del d
    torch.cuda.empty_cache()
