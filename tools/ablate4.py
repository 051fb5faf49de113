import os, sys, tempfile, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch, ntsm_amd
dev = torch.device("cuda:0"); n = 100_000_000
tmp = tempfile.mkdtemp(); sp = os.path.join(tmp, "s.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
sites = ntsm_amd.Sites(sp)
d_win = torch.from_numpy(s.windows).to(dev)
d = torch.empty(n * s.stride, dtype=torch.uint8, device=dev)
s.device_fill(d_win.data_ptr(), 0, n, d.data_ptr()); torch.cuda.synchronize()
def run(name, env=None):
    for k in ("NTSM_DEBUG_KERNEL", "NTSM_DEBUG_ZERO_FILTER", "NTSM_PREFILTER_OFF"): os.environ.pop(k, None)
    os.environ.update(env or {})
    ctx = ntsm_amd.Context(sites.keys)
    ctx.count_resident(d.data_ptr(), d.numel(), 0, n); ctx.sync(); ctx.set_timing(True)
    for _ in range(3): ctx.count_resident(d.data_ptr(), d.numel(), 0, n)
    t = ctx.sync(); k, ms = ctx.get_timing()
    print(json.dumps({"case": name, "ms": round(ms / k, 2), "gbases_s": round(n * 150 / (ms / k) / 1e6, 1), "hits": t.total_hits // 4})); ctx.close()
run("full")
run_tiny = True
run("push only (drain discards)", {"NTSM_DEBUG_KERNEL": "1"})
run("push + rebuild + prefilter load, no table", {"NTSM_DEBUG_KERNEL": "2"})
run("full minus atomics", {"NTSM_DEBUG_KERNEL": "4"})
run("second-level filter passes everything", {"NTSM_PREFILTER_OFF": "1"})
run("zero filter (no positives)", {"NTSM_DEBUG_ZERO_FILTER": "1"})

ctx = ntsm_amd.Context(sites.keys[:16])
ctx.count_resident(d.data_ptr(), d.numel(), 0, n); ctx.sync(); ctx.set_timing(True)
for _ in range(3): ctx.count_resident(d.data_ptr(), d.numel(), 0, n)
ctx.sync(); k, ms = ctx.get_timing()
print(json.dumps({"case": "16-key site set (compute floor)", "ms": round(ms / k, 2), "gbases_s": round(n * 150 / (ms / k) / 1e6, 1)})); ctx.close()
