#!/usr/bin/env python3
"""tools/ablate.py <mode> [reads] -- the kernel ablations behind DESIGN.md section 4.2's tables, one runner (rounds 1-4 had
ablate.py ... ablate4.py, ablate_occupancy.py, ablate_k19.sh and ab_libs.sh; tools/ablate.sh builds the libraries and calls this).

  workload   what the input costs: bench reads / reads without site windows, full site set / a 16-k-mer set whose filter is 64 blocks
  switches   the switches of ntsm_amd/csrc/ntsm_ablation.inc on an ablation build (NTSM_HIP_LIB=libntsm_hip_abl.so; counts are
             WRONG by construction): NTSM_DEBUG_KERNEL bits, empty filter, drain Bloom off / resized
  grid       launch geometry: workgroups per launch (ntsm_set_tuning grid) on the full and the 16-k-mer set
Every line is one JSON object; the library in use is named in it."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ntsm_amd

mode = sys.argv[1] if len(sys.argv) > 1 else "workload"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
sp = os.path.join(tmp, "s.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
sites = ntsm_amd.Sites(sp)
d_win = torch.from_numpy(s.windows).to(dev)
d = torch.empty(n * s.stride, dtype=torch.uint8, device=dev)
LIB = os.environ.get("NTSM_HIP_LIB", "libntsm_hip.so")


def fill(synth):
    synth.device_fill(d_win.data_ptr(), 0, n, d.data_ptr())
    torch.cuda.synchronize()


def run(ctx, case, passes=3, **extra):
    ctx.count_resident(d.data_ptr(), d.numel(), 0, n); ctx.sync(); ctx.reset(); ctx.set_timing(True)
    for _ in range(passes):
        ctx.count_resident(d.data_ptr(), d.numel(), 0, n)
    t = ctx.sync(); k, ms = ctx.get_timing(); ctx.set_timing(False)
    print(json.dumps(dict({"lib": LIB, "case": case, "ms": round(ms / k, 3), "gbases_s": round(n * 150 / (ms / k) / 1e6, 1),
                           "hits_per_pass": t.total_hits // passes, "kmers_per_pass": t.total_kmers // passes}, **extra)), flush=True)


if mode == "workload":
    fill(s)
    for name, keys in (("full set, bench reads", sites.keys), ("16-k-mer set, bench reads", sites.keys[:16])):
        ctx = ntsm_amd.Context(keys); run(ctx, name); ctx.close()
    fill(ntsm_amd.SynthShort(20241218, 96287, read_seed=7, p_embed=0.0))
    for name, keys in (("full set, reads without site windows", sites.keys), ("16-k-mer set, reads without site windows", sites.keys[:16])):
        ctx = ntsm_amd.Context(keys); run(ctx, name); ctx.close()
elif mode == "switches":
    fill(s)
    cases = [("full", {})] + [("NTSM_DEBUG_KERNEL=%d" % b, {"NTSM_DEBUG_KERNEL": str(b)}) for b in (1, 2, 4, 8, 9, 16, 32, 64)] + \
            [("empty filter", {"NTSM_DEBUG_ZERO_FILTER": "1"}), ("drain Bloom off", {"NTSM_PREFILTER_OFF": "1"})] + \
            [("drain Bloom 2^%d bits" % v, {"NTSM_PREFILTER_LOG2": str(v)}) for v in (21, 22, 24, 25)]
    # NTSM_DEBUG_KERNEL is read once per process by the launch code: one child process per value
    if len(sys.argv) > 3:
        ctx = ntsm_amd.Context(sites.keys); run(ctx, sys.argv[3]); ctx.close()
    else:
        import subprocess
        for name, env in cases:
            subprocess.run([sys.executable, os.path.abspath(__file__), "switches", str(n), name], env=dict(os.environ, **env))
elif mode == "grid":
    fill(s)
    for name, keys in (("16-k-mer set", sites.keys[:16]), ("full set", sites.keys)):
        ctx = ntsm_amd.Context(keys)
        for g in (256, 512, 768, 1024, 4096, 32768, 65536):
            ctx.set_tuning(0, g); run(ctx, name, passes=2, grid=g)
        ctx.close()
else:
    sys.exit(__doc__)
