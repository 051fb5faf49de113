#!/bin/bash
# the CLI on one and on two .gz files of 4e7 reads in total, 4 runs each
cd "$(dirname "$0")/.." || exit 1
python - <<'PY'
import sys, os, subprocess, time, tempfile, hashlib
sys.path.insert(0, '.')
import ntsm_amd, bench
n = 40_000_000
tmp = tempfile.mkdtemp(prefix="ntsm_12_")
sp = os.path.join(tmp, "sites.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
for parts in (1, 2):
    gz = []
    for i in range(parts):
        f = os.path.join(tmp, "r%d_%d.fq" % (parts, i))
        s.write_fastq(f, i * (n // parts), n // parts, threads=32)
        bench.pigz_like(f, f + ".gz", threads=48)
        os.unlink(f)
        gz.append(f + ".gz")
    ws = []
    for _ in range(5):
        t0 = time.perf_counter()
        p = subprocess.run(["build/ntsmCount", "-s", sp, "-t", "16"] + gz, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        ws.append(time.perf_counter() - t0)
        assert p.returncode == 0
    print("%d .gz file(s): %s s  best %.2f median %.2f Gbases/s  %s" % (parts, " ".join("%.3f" % w for w in ws), 6.0 / min(ws), 6.0 / sorted(ws)[2], hashlib.sha256(p.stdout).hexdigest()[:12]), flush=True)
    for f in gz:
        os.unlink(f)
PY
