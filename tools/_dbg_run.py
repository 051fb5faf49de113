import json, os, sys, tempfile
sys.path.insert(0, '/root/repo')
import numpy as np, torch, ntsm_amd
dev = torch.device("cuda:0")
n = 20_000_000
tmp = tempfile.mkdtemp()
sp = os.path.join(tmp, "s.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=9, sites_path=sp, min_keep=0)
sites = ntsm_amd.Sites(sp)
d_win = torch.from_numpy(s.windows).to(dev)
d = torch.empty(n * s.stride, dtype=torch.uint8, device=dev)
s.device_fill(d_win.data_ptr(), 0, n, d.data_ptr()); torch.cuda.synchronize()
res = {}
for v in (0, 5):
    ctx = ntsm_amd.Context(sites.keys); ctx.set_kernel(v)
    ctx.count_resident(d.data_ptr(), d.numel(), 0, n); t = ctx.sync(); res[v] = (t.total_hits, ctx.counts()); ctx.close()
diff = np.nonzero(res[0][1] != res[5][1])[0]
print("hits", res[0][0], res[5][0], "differing keys", len(diff))
def rc(x, k=19):
    r = 0
    for b in range(k):
        r |= (3 - ((x >> (2 * b)) & 3)) << (2 * (k - 1 - b))
    return r
def h24(c): return (c * 0x9E3779) & 0xFFFFFF
for i in diff[:12]:
    x = int(sites.keys[i]); r = rc(x)
    hs = []
    for q in range(8):
        sub = (x >> (2 * (7 - q))) & 0xFFFFFF; rsub = (r >> (2 * q)) & 0xFFFFFF
        hs.append(h24(min(sub, rsub)))
    seq = "".join("ACGT"[(x >> (2 * (18 - b))) & 3] for b in range(19))
    print(i, seq, "counts", int(res[0][1][i]), int(res[5][1][i]), "h24", [hex(h) for h in hs], "argmin", [q for q in range(8) if hs[q] == min(hs)])
