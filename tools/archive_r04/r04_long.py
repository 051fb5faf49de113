#!/usr/bin/env python3
"""tools/r04_long.py -- the CLI on long reads (BASELINE configs[2]'s shape: log-normal lengths around 15 kb, 4-line FASTQ with
10-200 kb lines), plain and as one .gz, -t 16 against -t 1: same counts.txt, whole-process wall.  GPU box."""
import hashlib, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ntsm_amd, bench
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000
tmp = tempfile.mkdtemp(prefix="ntsm_long_")
sp = os.path.join(tmp, "sites.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
lg = ntsm_amd.SynthLong(s)
fq = os.path.join(tmp, "long.fq")
t0 = time.perf_counter()
lg.write_fastq(fq, 0, n)
ends, total = lg.layout(0, n)
bases = total - n
print("long reads: %d reads, %.2f Gbases, %.2f GB of FASTQ (written in %.1f s)" % (n, bases / 1e9, os.path.getsize(fq) / 1e9, time.perf_counter() - t0), flush=True)
bench.pigz_like(fq, fq + ".gz", threads=48)
exe = os.path.join(ROOT, "build", "ntsmCount")
shas = set()
for label, args in (("plain -t 1", ["-t", "1", fq]), ("plain -t 16", ["-t", "16", fq]), (".gz   -t 1", ["-t", "1", fq + ".gz"]), (".gz   -t 16", ["-t", "16", fq + ".gz"])):
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-s", sp] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, NTSM_PHASE_TIMES="1"))
        w = time.perf_counter() - t0
        assert p.returncode == 0, p.stderr[-500:]
        best = w if best is None else min(best, w)
    shas.add(hashlib.sha256(p.stdout).hexdigest()[:12])
    ph = [l[8:] for l in p.stderr.decode().split("\n") if l.startswith("[phase]") and ("parse" in l or "early" in l)]
    print("%-12s %.3f s  %.2f Gbases/s | %s" % (label, best, bases / best / 1e9, " | ".join(x.replace(tmp + "/", "")[:170] for x in ph)), flush=True)
assert len(shas) == 1, shas
print("counts.txt identical")
