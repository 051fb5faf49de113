"""tools/tab_stats.py -- first-level pass rate of the tabulated kernel on the bench workload (needs a GPU)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ntsm_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
tmp = tempfile.mkdtemp(); sp = os.path.join(tmp, "s.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
sites = ntsm_amd.Sites(sp)
dev = torch.device("cuda:0")
d_win = torch.from_numpy(s.windows).to(dev)
d = torch.empty(n * s.stride, dtype=torch.uint8, device=dev)
s.device_fill(d_win.data_ptr(), 0, n, d.data_ptr()); torch.cuda.synchronize()
ctx = ntsm_amd.Context(sites.keys)
ctx.count_resident(d.data_ptr(), d.numel(), 0, n)
t = ctx.sync(); st = ctx.debug_stats()
print("reads %d kmers %d hits %d (%.3f%%) queued %d (%.3f%% of windows) stats %s" % (n, t.total_kmers, t.total_hits, 100.0 * t.total_hits / t.total_kmers,
      st["queued_windows"], 100.0 * st["queued_windows"] / t.total_kmers, st))
