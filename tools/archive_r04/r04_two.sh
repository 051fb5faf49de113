#!/bin/bash
# phases of the CLI on two .gz files (paired-end shape), 2e7 reads each
cd "$(dirname "$0")/.." || exit 1
python - <<'PY'
import sys, os, subprocess, time, tempfile
sys.path.insert(0, '.')
import ntsm_amd, bench
n = 40_000_000
tmp = tempfile.mkdtemp(prefix="ntsm_two_")
sp = os.path.join(tmp, "sites.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
gz = []
for i in range(2):
    f = os.path.join(tmp, "r%d.fq" % i)
    s.write_fastq(f, i * (n // 2), n // 2, threads=32)
    bench.pigz_like(f, f + ".gz", threads=48)
    os.unlink(f)
    gz.append(f + ".gz")
for env in ({}, {"NTSM_NO_EARLY": "1"}, {"NTSM_GZ_DECODERS": "16"}):
    for _ in range(2):
        t0 = time.perf_counter()
        p = subprocess.run(["build/ntsmCount", "-s", sp, "-t", "16"] + gz, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, NTSM_PHASE_TIMES="1", **env))
        w = time.perf_counter() - t0
        print("wall %.3f %s" % (w, env))
        for l in p.stderr.decode().split("\n"):
            if l.startswith("[phase]"):
                print("    " + l[8:].replace(tmp + "/", ""))
PY
