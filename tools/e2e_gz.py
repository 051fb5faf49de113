#!/usr/bin/env python3
"""CLI wall time on gzip input: decoder thread (default) vs zlib's gzread (NTSM_ZLIB_ONLY=1), one file and four files
with -t 4."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ntsm_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
tmp = tempfile.mkdtemp(prefix="ntsm_gz_")
sp = os.path.join(tmp, "sites.fa")
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sp)
files = []
for i in range(4):
    fq = os.path.join(tmp, "r%d.fq" % i)
    s.write_fastq(fq, i * (n // 4), n // 4)
    files.append(fq)
t0 = time.perf_counter()
ps = [subprocess.Popen(["gzip", "-4", f]) for f in files]
[p.wait() for p in ps]
files = [f + ".gz" for f in files]
print("4 files, %.0f MB gz in total, gzip -4 took %.1f s" % (sum(os.path.getsize(f) for f in files) / 1e6, time.perf_counter() - t0))
exe = os.path.join(ROOT, "build", "ntsmCount")
outs = {}
for label, env in (("zlib gzread", {"NTSM_ZLIB_ONLY": "1"}), ("decoder thread", {})):
    for args, nf in ((["-t", "1"], 1), (["-t", "1"], 4), (["-t", "4"], 4)):
        e = dict(os.environ, **env)
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-s", sp] + args + files[:nf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        dt = time.perf_counter() - t0
        assert p.returncode == 0, p.stderr[-500:]
        outs.setdefault((tuple(args), nf), set()).add(p.stdout)
        print("%-15s %d file(s) %s: %.2f s -> %.3f Gbases/s" % (label, nf, " ".join(args), dt, (n // 4) * nf * 150 / dt / 1e9))
assert all(len(v) == 1 for v in outs.values()), "outputs differ between the two gzip paths"
print("counts.txt identical between the two paths")
