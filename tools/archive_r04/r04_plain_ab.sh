#!/bin/bash
# same box, interleaved: block size / slot size / -t of the plain-FASTQ path through the CLI (4e7 reads, 12.6 GB, page cache)
cd "$(dirname "$0")/.." || exit 1
python - <<'PY'
import sys, os, subprocess, time, tempfile
sys.path.insert(0, '.')
import ntsm_amd
n = 40_000_000
tmp = tempfile.mkdtemp(prefix="ntsm_pl_")
sp = os.path.join(tmp, "sites.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
f = os.path.join(tmp, "r.fq")
s.write_fastq(f, 0, n, threads=32)
cases = [("default", "16", {}), ("block 4 MiB", "16", {"NTSM_BLOCK_BYTES": str(4 << 20)}), ("block 8 MiB", "16", {"NTSM_BLOCK_BYTES": str(8 << 20)}),
         ("slot 4 MiB", "16", {"NTSM_BATCH_BYTES": str(4 << 20)}), ("slot 2 MiB", "16", {"NTSM_BATCH_BYTES": str(2 << 20)}),
         ("-t 12", "12", {}), ("-t 14", "14", {}), ("-t 20", "20", {}), ("-t 24", "24", {})]
res = {}
for rep in range(5):
    for name, t, env in cases:
        t0 = time.perf_counter()
        p = subprocess.run(["build/ntsmCount", "-s", sp, "-t", t, f], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        res.setdefault(name, []).append(time.perf_counter() - t0)
        assert p.returncode == 0
for k, v in res.items():
    print("%-12s %s  median %.3f s" % (k, " ".join("%.3f" % x for x in v), sorted(v)[2]), flush=True)
PY
