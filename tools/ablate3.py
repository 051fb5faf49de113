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
ref = None
def run(name, flog=0, env=None):
    global ref
    for k in ("NTSM_PREFILTER_OFF", "NTSM_PREFILTER_LOG2"): os.environ.pop(k, None)
    os.environ.update(env or {})
    ctx = ntsm_amd.Context(sites.keys)
    if flog: ctx.set_tuning(flog, 0)
    ctx.count_resident(d.data_ptr(), d.numel(), 0, n); ctx.sync(); ctx.set_timing(True)
    for _ in range(3): ctx.count_resident(d.data_ptr(), d.numel(), 0, n)
    t = ctx.sync(); k, ms = ctx.get_timing(); c = ctx.counts()
    if ref is None: ref = c
    print(json.dumps({"case": name, "ms": round(ms / k, 2), "gbases_s": round(n * 150 / (ms / k) / 1e6, 1), "same_counts": bool(np.array_equal(c, ref))})); ctx.close()
run("prefilter off, main 3MiB", 0, {"NTSM_PREFILTER_OFF": "1"})
run("prefilter 1MiB, main 3MiB", 0)
run("prefilter 2MiB, main 3MiB", 0, {"NTSM_PREFILTER_LOG2": "24"})
run("prefilter 512KiB, main 3MiB", 0, {"NTSM_PREFILTER_LOG2": "22"})
run("prefilter 1MiB, main 2MiB", 24)
run("prefilter 2MiB, main 2MiB", 24, {"NTSM_PREFILTER_LOG2": "24"})
run("prefilter 1MiB, main 4MiB", 25)
