#!/usr/bin/env python3
"""tools/e2e_pack.py [reads] [t,t,...] -- build/ntsmCount on ONE plain FASTQ with the producer lanes sending packed codes
(default) or raw bytes (NTSM_NO_PACK=1), interleaved on the same file: wall time, parse+count phase (best of three),
identical stdout."""
import hashlib, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ntsm_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40_000_000
ts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 16, 32]
tmp = tempfile.mkdtemp(prefix="ntsm_pack_")
sp = os.path.join(tmp, "sites.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
fq = os.path.join(tmp, "reads.fq")
t0 = time.perf_counter(); s.write_fastq(fq, 0, n, threads=32); print("file: %.2f GB written in %.1f s" % (os.path.getsize(fq) / 1e9, time.perf_counter() - t0), flush=True)
sha = None
for t in ts:
    for mode, extra in (("packed", {}), ("bytes", {"NTSM_NO_PACK": "1"})):
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            p = subprocess.run([os.path.join(ROOT, "build", "ntsmCount"), "-s", sp, "-t", str(t), fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, NTSM_PHASE_TIMES="1", **extra))
            dt = time.perf_counter() - t0
            assert p.returncode == 0, p.stderr.decode()[-500:]
            h = hashlib.sha256(p.stdout).hexdigest()
            assert sha is None or h == sha, "stdout differs"
            sha = h
            pc = [l for l in p.stderr.decode().split("\n") if "parse+count" in l]
            ps = float(pc[0].split("parse+count")[1].split("s")[0]) if pc else float("nan")
            if best is None or dt < best[0]:
                best = (dt, ps)
        print("-t %2d %-11s wall %.3f s = %6.2f Gbases/s   parse+count %.3f s = %6.2f Gbases/s" % (t, mode, best[0], n * 150 / best[0] / 1e9, best[1], n * 150 / best[1] / 1e9), flush=True)
os.unlink(fq)
