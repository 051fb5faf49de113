#!/usr/bin/env python3
"""Randomised differential soak of the CLI against the CPU oracle (test tooling; uses oracle/ as the checker only):
random site sets and k, FASTQ / FASTA / gzip / BGZF inputs with ragged reads, Ns, lower case, CRLF and wrapped
records, random -t / -d / -m, random staging and block sizes, random thresholds / chunk sizes / decoder counts of the parallel gzip
ingest and of the early ingest.  stdout must be byte-identical and the summary lines
equal.   tools/soak.py [seconds] [seed]"""
import gzip, os, random, struct, subprocess, sys, tempfile, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE, ORA = os.path.join(ROOT, "build", "ntsmCount"), os.path.join(ROOT, "oracle", "ntsm_oracle")
REFGPU = os.path.join(ROOT, "oracle", "_ref", "ref_gpu_ntsmCount")   # the reference's own class bound to the library (oracle/ref_gpu_binding.cpp), when it was built
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
KEEP = (b"Total ", b"Distinct ", b"Sites Covered", b"Warning: site coverage", b"Reached desired")

def summary(err):
    return [l for l in err.split(b"\n") if l.startswith(KEEP)]

def bgzf(data):
    out = []
    for i in range(0, len(data), 65280):
        d = data[i:i + 65280]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = co.compress(d) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) + body +
                   struct.pack("<II", zlib.crc32(d), len(d)))
    return b"".join(out)

def dna(n, alphabet=b"ACGT"):
    return bytes(rng.choice(alphabet) for _ in range(n))

tmp = tempfile.mkdtemp(prefix="ntsm_soak_")
t_end, it, fails = time.time() + budget, 0, 0
while time.time() < t_end:
    it += 1
    k = rng.choice([19, 19, 19, 13, 14, 15, 16, 17, 18, 20, 21, 24, 25, 31, 32, 11])
    n_sites = rng.choice([5, 50, 400, 3000])
    # sites: ref/var windows around a SNP, k-mers joined by N like the reference's site files
    genome = dna(n_sites * 80)
    sp = os.path.join(tmp, "sites.fa")
    with open(sp, "wb") as f:
        for sidx in range(n_sites):
            c = sidx * 80 + 40
            for allele, name in ((genome[c:c + 1], b"ref"), (bytes([rng.choice([x for x in b"ACGT" if x != genome[c]])]), b"var")):
                seq = genome[c - k + 1:c] + allele + genome[c + 1:c + k]
                kmers = [seq[i:i + k] for i in range(len(seq) - k + 1)]
                f.write(b">rs%d %s\n" % (sidx, name) + b"N".join(kmers[::rng.choice([1, 1, 2])]) + b"\n")
    dupes = rng.random() < 0.5
    # reads: fragments of the genome with errors, Ns, lower case; sometimes unrelated
    files, n_files = [], rng.choice([1, 1, 2, 4])
    for fi in range(n_files):
        recs = []
        fasta = rng.random() < 0.15
        for r in range(rng.choice([20, 500, 6000, 30000])):
            L = rng.choice([1, 18, k, 50, 150, 150, 150, 400, 5000]) if rng.random() < 0.3 else 150
            a = rng.randrange(0, max(1, len(genome) - L))
            s = bytearray(genome[a:a + L] if rng.random() < 0.8 else dna(L))
            for _ in range(rng.choice([0, 0, 1, 3])):
                if s:
                    s[rng.randrange(len(s))] = rng.choice(b"ACGTNacgtn")
            s = bytes(s) if s else b"A"
            if fasta:
                recs.append(b">r%d\n%s\n" % (r, s))
            else:
                recs.append(b"@r%d some comment\n%s\n+\n%s\n" % (r, s, bytes(rng.choice(b"FF:,#@+") for _ in range(len(s)))))
        data = b"".join(recs)
        style = rng.random()
        if not fasta and style < 0.10:                                  # one wrapped record in the middle
            lines = data.split(b"\n")
            j = (len(lines) // 8) * 4 + 1
            if len(lines[j]) > 4:
                q = lines[j + 2]
                lines[j] = lines[j][:3] + b"\n" + lines[j][3:]
                lines[j + 2] = q[:5] + b"\n" + q[5:]
                data = b"\n".join(lines)
        elif style < 0.15:
            data = data.replace(b"\n", b"\r\n")
        enc = rng.random()
        path = os.path.join(tmp, "r%d.%s" % (fi, "fa" if fasta else "fq"))
        if enc < 0.25:
            path += ".gz"; blob = gzip.compress(data, rng.choice([1, 6]))
        elif enc < 0.40:
            path += ".gz"; blob = bgzf(data)
        else:
            blob = data
        open(path, "wb").write(blob)
        files.append(path)
    args = ["-s", sp, "-k", str(k)] + (["-d"] if dupes else [])
    if rng.random() < 0.25:
        args += ["-m", "%.3f" % rng.choice([0.05, 0.5, 2.0, 10.0])]
    t = rng.choice([1, 1, 2, 3, 8, 16])
    env = dict(os.environ, NTSM_BLOCK_BYTES=str(rng.choice([4096, 65536, 1 << 20, 16 << 20])),
               NTSM_BATCH_BYTES=str(rng.choice([8192, 1 << 20, 64 << 20])),
               # round 4: the parallel gzip ingest on files of any size, small chunks, and the early ingest of the first file
               NTSM_GZ_PARALLEL_MIN=str(rng.choice([100, 5000, 8 << 20])), NTSM_GZ_CHUNK=str(rng.choice([1024, 4096, 50000, 0])),
               NTSM_GZ_DECODERS=str(rng.choice([0, 2, 5])), NTSM_EARLY=rng.choice(["gz", "plain", "all"]))
    if rng.random() < 0.3:
        env["NTSM_NO_EARLY"] = "1"
    if rng.random() < 0.3:
        env["NTSM_FAST_EXIT"] = "1"                            # teardown handed to a CLONE_VM child instead of running inside exit(2) (round 5: opt-in)
    # round 5: the same inputs through every kernel form (hidden --debug-kernel: ntsm_set_kernel on every context) -- generic (1),
    # minimizer-blocked one level (2) / two levels (4, 15 <= k <= 31), run-anchored (5, k = 19) -- with -m, -d, early stop and undo
    forms = [None, None, 1] + ([2] if 13 <= k <= 31 else []) + ([4] if 15 <= k <= 31 else []) + ([5, 5] if k == 19 else [])
    form = rng.choice(forms)
    kargs = ["--debug-kernel", str(form)] if form is not None else []
    # round 6: several contexts on the one device (thread -> context round robin, one merge on the device: ntsm_allreduce)
    g = rng.choice([None, None, "0,0", "0,0,0"])
    if g:
        kargs += ["-g", g]
    ref = subprocess.run([ORA] + args + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    got = subprocess.run([EXE] + args + kargs + ["-t", str(t)] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    ok = (ref.returncode == got.returncode or (ref.returncode == 134 and got.returncode == -6)) and ref.stdout == got.stdout
    if ok and ref.returncode == 0:
        ok = summary(ref.stderr) == summary(got.stderr)
    # round 6: the same inputs through the reference's OWN FingerPrint class with INTEGRATION.md's binding around it (site loader,
    # kseq and printing are the reference's code, the counting is the library's); k = 32 is undefined behaviour in the reference
    if ok and os.path.isfile(REFGPU) and k <= 31 and rng.random() < 0.5:
        benv = dict(os.environ)
        if rng.random() < 0.5:
            benv["NTSM_REF_GPU_BATCH"] = str(rng.choice([8192, 100000]))
        bnd = subprocess.run([REFGPU] + args + ["-t", str(rng.choice([1, 3]))] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=benv)
        ok = (ref.returncode == bnd.returncode or (ref.returncode == 134 and bnd.returncode == -6)) and (ref.returncode != 0 or ref.stdout == bnd.stdout)
        if ok and ref.returncode == 0:
            ok = summary(ref.stderr) == summary(bnd.stderr)
        if not ok:
            got = bnd
            kargs = ["(reference binding)"] + [x for x in benv.items() if x[0].startswith("NTSM_REF")]
        n_binding = globals().get("n_binding", 0) + 1
    if not ok:
        fails += 1
        keep = tempfile.mkdtemp(prefix="ntsm_soak_fail_", dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
        print("MISMATCH iteration", it, "args", args, kargs, "-t", t, "env", {k: v for k, v in env.items() if k.startswith("NTSM_")}, "rc", ref.returncode, got.returncode, "kept in", keep)
        print(got.stderr.decode()[-600:])
        for p in [sp] + files:
            subprocess.run(["cp", p, keep])
print("soak: %d iterations (%d of them also through the reference binding), %d mismatches" % (it, globals().get("n_binding", 0), fails))
sys.exit(1 if fails else 0)
