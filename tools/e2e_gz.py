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

# BGZF (bgzip-style) single file: block-parallel inflate with -t N
import struct, zlib
def bgzf_block(d, level=4):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(d) + co.flush()
    bsize = 12 + 6 + len(body) + 8
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + body +
            struct.pack("<II", zlib.crc32(d), len(d)))
fq = os.path.join(tmp, "b.fq")
nb = n // 2
s.write_fastq(fq, 0, nb)
t0 = time.perf_counter()
bg = os.path.join(tmp, "b.fq.gz")
with open(fq, "rb") as fi, open(bg, "wb") as fo:
    while True:
        d = fi.read(65280)
        if not d:
            break
        fo.write(bgzf_block(d))
    fo.write(bgzf_block(b""))
print("BGZF file: %d reads, %.0f MB (written in %.1f s)" % (nb, os.path.getsize(bg) / 1e6, time.perf_counter() - t0))
outs = set()
for t in (1, 2, 4, 8, 16):
    t0 = time.perf_counter()
    p = subprocess.run([exe, "-s", sp, "-t", str(t), bg], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    assert p.returncode == 0, p.stderr[-500:]
    outs.add(p.stdout)
    print("BGZF 1 file -t %2d: %.2f s -> %.3f Gbases/s" % (t, dt, nb * 150 / dt / 1e9))
p = subprocess.run([exe, "-s", sp, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
outs.add(p.stdout)
assert len(outs) == 1, "BGZF outputs differ"
print("BGZF counts identical for every -t and equal to the plain file's")
