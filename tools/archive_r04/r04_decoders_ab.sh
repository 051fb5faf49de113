#!/bin/bash
# same box, interleaved: decoder threads (NTSM_GZ_DECODERS), the CLI on one .gz of 4e7 reads
cd "$(dirname "$0")/.." || exit 1
python - <<'PY'
import sys, os, subprocess, time, tempfile
sys.path.insert(0, '.')
import ntsm_amd, bench
n = 40_000_000
tmp = tempfile.mkdtemp(prefix="ntsm_ck_")
sp = os.path.join(tmp, "sites.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
f = os.path.join(tmp, "r.fq")
s.write_fastq(f, 0, n, threads=32)
bench.pigz_like(f, f + ".gz", threads=48)
os.unlink(f)
res = {}
for rep in range(5):
    for chunk in ("8", "10", "12", "14", "16"):
        t0 = time.perf_counter()
        p = subprocess.run(["build/ntsmCount", "-s", sp, "-t", "16", f + ".gz"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, NTSM_GZ_DECODERS=chunk))
        res.setdefault(chunk, []).append(time.perf_counter() - t0)
        assert p.returncode == 0
for k in res:
    w = sorted(res[k])
    print("decoders %2s: %s  median %.3f s" % (k, " ".join("%.3f" % x for x in res[k]), w[2]), flush=True)
PY
