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
for name, keys in (("tiny_set", sites.keys[:16]), ("full", sites.keys)):
    ctx = ntsm_amd.Context(keys)
    for g in (256, 512, 768, 1024):
        ctx.set_tuning(0, g); ctx.reset()
        ctx.count_resident(d.data_ptr(), d.numel(), 0, n); ctx.sync(); ctx.set_timing(True)
        for _ in range(2): ctx.count_resident(d.data_ptr(), d.numel(), 0, n)
        ctx.sync(); k, ms = ctx.get_timing()
        print(json.dumps({"case": name, "grid": g, "ms": round(ms / k, 2), "gbases_s": round(n * 150 / (ms / k) / 1e6, 1)}))
    ctx.close()
