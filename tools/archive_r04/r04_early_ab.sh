#!/bin/bash
# same box, interleaved: early ingest of the first .gz forced (NTSM_EARLY=gz) against off (NTSM_NO_EARLY=1), one and two files of 4e7 reads in total
cd "$(dirname "$0")/.." || exit 1
python - <<'PY'
import sys, os, subprocess, time, tempfile, hashlib
sys.path.insert(0, '.')
import ntsm_amd, bench
n = 40_000_000
tmp = tempfile.mkdtemp(prefix="ntsm_ab_")
sp = os.path.join(tmp, "sites.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
sets = {}
for parts in (1, 2):
    gz = []
    for i in range(parts):
        f = os.path.join(tmp, "r%d_%d.fq" % (parts, i))
        s.write_fastq(f, i * (n // parts), n // parts, threads=32)
        bench.pigz_like(f, f + ".gz", threads=48)
        os.unlink(f)
        gz.append(f + ".gz")
    sets[parts] = gz
res = {}
for rep in range(6):
    for parts in (1, 2):
        for mode, env in (("early", {"NTSM_EARLY": "gz"}), ("off", {"NTSM_NO_EARLY": "1"})):
            t0 = time.perf_counter()
            p = subprocess.run(["build/ntsmCount", "-s", sp, "-t", "16"] + sets[parts], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            res.setdefault((parts, mode), []).append(time.perf_counter() - t0)
            assert p.returncode == 0
for k in sorted(res):
    w = sorted(res[k])
    print("%d file(s), early ingest %-5s: %s  median %.3f s" % (k[0], k[1], " ".join("%.3f" % x for x in res[k]), (w[2] + w[3]) / 2), flush=True)
PY
