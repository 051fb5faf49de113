#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED reference.

Runs only in the build container (needs oracle/_ref/ref_ntsmCount, i.e. /root/reference compiled in
place by oracle/Makefile, and build/ntsm_synth).  For every case it records the input files (data
authored here or produced by the seeded generator), the command-line flags, and the reference's
stdout / stderr / exit status.  The reference has no tests or known-answer vectors of its own
(SURVEY.md section 4), so these recordings are the parity pin for oracle/ and for the HIP path.

Usage: python tests/golden/make_golden.py          (rewrites tests/golden/cases.json + files)
"""
import gzip
import hashlib
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_ntsmCount")
SYNTH = os.path.join(ROOT, "build", "ntsm_synth")
INP = os.path.join(HERE, "inputs")
EXP = os.path.join(HERE, "expected")


def sh(cmd, **kw):
    return subprocess.run(cmd, check=True, **kw)


def run_ref(args, cwd):
    p = subprocess.run([REF] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    err = "\n".join(l for l in p.stderr.decode("latin-1").split("\n") if not l.startswith("Time: "))
    return p.returncode, p.stdout, err.encode("latin-1")


def revcomp(s):
    return s.translate(str.maketrans("ACGTacgt", "TGCAtgca"))[::-1]


def read_fasta(path):
    recs, name, seq = [], None, []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(seq)))
            name, seq = line[1:].split()[0], []
        else:
            seq.append(line)
    if name is not None:
        recs.append((name, "".join(seq)))
    return recs


def site_window(kmers_joined, k):
    """Rebuild the contiguous window spanned by the first and last k-mers of a site record (for reads)."""
    return kmers_joined.split("N")


def main():
    if not os.path.exists(REF) or not os.path.exists(SYNTH):
        sys.exit("need oracle/_ref/ref_ntsmCount and build/ntsm_synth (run `make` at the repo root)")
    os.makedirs(INP, exist_ok=True)
    os.makedirs(EXP, exist_ok=True)
    cases = []

    def add(name, args, files, note=""):
        rc, out, err = run_ref(args + files, INP)
        case = {"name": name, "args": args, "files": files, "rc": rc, "note": note}
        if rc == 0:
            open(os.path.join(EXP, name + ".stdout"), "wb").write(out)
            case["stdout"] = name + ".stdout"
        open(os.path.join(EXP, name + ".stderr"), "wb").write(err)
        case["stderr"] = name + ".stderr"
        m = re.search(rb"Total k-mers Recorded: (\d+)", err)
        case["hits"] = int(m.group(1)) if m else None
        cases.append(case)
        return case

    # ---------------------------------------------------------------- tiny synthetic set, k = 19
    sh([SYNTH, "sites", "--seed", "11", "--n-sites", "200", "--out", os.path.join(INP, "sites200.fa")])
    sh([SYNTH, "reads", "--seed", "5", "--sites-seed", "11", "--n-sites", "200", "--n-reads", "2000",
        "--p-embed", "0.5", "--out", os.path.join(INP, "reads2k.fq")])
    sh([SYNTH, "reads", "--seed", "6", "--sites-seed", "11", "--n-sites", "200", "--n-reads", "600",
        "--p-embed", "0.5", "--out", os.path.join(INP, "reads600.fq.gz")])
    sh([SYNTH, "reads", "--seed", "9", "--sites-seed", "11", "--n-sites", "200", "--n-reads", "3",
        "--p-embed", "1.0", "--p-sub", "0", "--p-n", "0", "--out", os.path.join(INP, "reads3.fq")])
    add("tiny_k19", ["-s", "sites200.fa"], ["reads2k.fq"])
    add("tiny_gz", ["-s", "sites200.fa"], ["reads600.fq.gz"], "gzip input through gzread")
    add("tiny_multifile", ["-s", "sites200.fa"], ["reads2k.fq", "reads600.fq.gz", "reads3.fq"])
    add("tiny_summary_t4", ["-s", "sites200.fa", "-t", "1"], ["reads3.fq"])

    # ---------------------------------------------------------------- other k (sites regenerated per k)
    for k in (11, 15, 25, 31):
        sf = "sites60_k%d.fa" % k
        sh([SYNTH, "sites", "--seed", "21", "--n-sites", "60", "--k", str(k), "--out", os.path.join(INP, sf)])
        rf = "reads300_k%d.fq" % k
        sh([SYNTH, "reads", "--seed", "3", "--sites-seed", "21", "--n-sites", "60", "--k", str(k), "--n-reads", "300",
            "--p-embed", "0.6", "--out", os.path.join(INP, rf)])
        add("k%d" % k, ["-s", sf, "-k", str(k)], [rf])
    # k mismatch: sites built for 19 read with -k 21 (only windows >= 21 valid: none) and -k 17
    add("k17_on_k19_sites", ["-s", "sites200.fa", "-k", "17"], ["reads2k.fq"])
    add("k21_on_k19_sites", ["-s", "sites200.fa", "-k", "21"], ["reads2k.fq"])

    # ---------------------------------------------------------------- hand-made edge reads
    sites = read_fasta(os.path.join(INP, "sites200.fa"))
    km = [s for _, s in sites]                       # k-mers joined by N, REF/VAR interleaved
    first = [s.split("N") for s in km]
    A = first[0][0]; B = first[1][0]; C = first[2][1]; D = first[5][0]; E = first[8][2]
    edge = []
    edge.append(("upper", A))
    edge.append(("lower", A.lower()))
    edge.append(("mixed", "".join(c.lower() if i % 3 else c for i, c in enumerate(B))))
    edge.append(("uracil", C.replace("T", "U")))
    edge.append(("uracil_lower", C.replace("T", "u")))
    edge.append(("revcomp", revcomp(D)))
    edge.append(("with_N_inside", A[:10] + "N" + A[10:]))
    edge.append(("N_then_kmer", "NNNN" + A + "N" + B + "NN"))
    edge.append(("iupac_breaks", A[:18] + "R" + E + "Y" + D))
    edge.append(("too_short", A[:18]))
    edge.append(("exact_k", E))
    edge.append(("k_plus_1", "G" + E))
    edge.append(("twice", A + A))
    edge.append(("spaces_tabs", A[:5] + " " + A[5:]))   # space is an invalid base (kseq keeps it)
    edge.append(("digits", "0123" + A))
    edge.append(("empty", ""))
    body = "".join(">%s some comment\n%s\n" % (n, s) for n, s in edge)
    open(os.path.join(INP, "edge.fa"), "w").write(body)
    add("edge_fasta", ["-s", "sites200.fa"], ["edge.fa"])

    # raw bytes 0..3 are valid bases (nt4 table maps them to themselves)
    raw = A.translate(str.maketrans("ACGT", "\x00\x01\x02\x03"))
    rawmix = "".join(ch if i % 2 else ch.translate(str.maketrans("ACGT", "\x00\x01\x02\x03")) for i, ch in enumerate(B))
    with open(os.path.join(INP, "rawbytes.fa"), "wb") as f:
        f.write(b">raw\n" + raw.encode("latin-1") + b"\n>rawmix\n" + rawmix.encode("latin-1") + b"\n")
        f.write(b">highbytes\n" + A[:9].encode() + b"\xff\x80" + A.encode() + b"\n")
    add("edge_rawbytes", ["-s", "sites200.fa"], ["rawbytes.fa"])

    # multi-line FASTA, CRLF, blank lines, record without trailing newline
    ml = ">ml1 multi-line\n" + "\n".join(A[i:i + 7] for i in range(0, len(A), 7)) + "\n"
    ml += ">crlf\r\n" + B[:9] + "\r\n" + B[9:] + "\r\n"
    ml += ">blank\n\n" + C[:4] + "\n\n\n" + C[4:] + "\n"
    ml += ">lonely_cr\n\r\n" + D + "\n"
    ml += ">cr_mid\n" + D[:6] + "\n\r\n" + D[6:] + "\n"
    ml += ">tabname\tcomment here\n" + E + "\n"
    ml += ">noeol\n" + A
    open(os.path.join(INP, "multiline.fa"), "w", newline="").write(ml)
    add("edge_multiline", ["-s", "sites200.fa"], ["multiline.fa"])

    # FASTQ oddities: '@' inside quality, multi-line FASTQ, '+' with repeated name, stray text before header
    fq = "junk before the first header\n"
    fq += "@q1\n" + A + "\n+\n" + "@" * len(A) + "\n"
    fq += "@q2 desc\n" + B[:10] + "\n" + B[10:] + "\n+q2 desc\n" + "I" * 10 + "\n" + "I" * (len(B) - 10) + "\n"
    fq += "@q3\n" + C + "\n+\n" + ">" * len(C) + "\n"
    fq += "@q_empty\n\n+\n\n"
    fq += "@q4\n" + D.lower() + "\n+\n" + "#" * len(D) + "\n"
    open(os.path.join(INP, "odd.fq"), "w").write(fq)
    add("edge_fastq", ["-s", "sites200.fa"], ["odd.fq"])

    # truncated quality: kseq_read returns -2 and the file ends there (reads after it are lost)
    tq = "@ok\n" + A + "\n+\n" + "I" * len(A) + "\n@bad\n" + B + "\n+\n" + "I" * 5 + "\n@after\n" + C + "\n+\n" + "I" * len(C) + "\n"
    open(os.path.join(INP, "truncqual.fq"), "w").write(tq)
    add("edge_truncated_quality", ["-s", "sites200.fa"], ["truncqual.fq"])
    tq2 = "@ok\n" + A + "\n+\n" + "I" * len(A) + "\n@noqual\n" + B + "\n+"
    open(os.path.join(INP, "noqual.fq"), "w").write(tq2)
    add("edge_missing_quality", ["-s", "sites200.fa"], ["noqual.fq"])
    open(os.path.join(INP, "empty.fa"), "w").write("")
    add("edge_empty_file", ["-s", "sites200.fa"], ["empty.fa", "reads3.fq"])
    open(os.path.join(INP, "headeronly.fa"), "w").write(">")
    add("edge_header_only", ["-s", "sites200.fa"], ["headeronly.fa", "reads3.fq"])

    # long read (> 64 KB, single line and 60-column wrapped): concatenate many site k-mers + filler
    import random
    rnd = random.Random(1234)
    parts = []
    for i in range(3000):
        parts.append(rnd.choice(first[rnd.randrange(len(first))]))
        parts.append("".join(rnd.choice("ACGT") for _ in range(rnd.randrange(0, 40))))
        if i % 97 == 0:
            parts.append("N")
    longseq = "".join(parts)
    with open(os.path.join(INP, "long.fa"), "w") as f:
        f.write(">long_single_line\n" + longseq + "\n>long_wrapped\n")
        f.write("\n".join(longseq[i:i + 60] for i in range(0, len(longseq), 60)) + "\n")
        f.write(">long_rc\n" + revcomp(longseq) + "\n")
    add("long_reads", ["-s", "sites200.fa"], ["long.fa"], "reads of %d bases" % len(longseq))

    # ---------------------------------------------------------------- -m early stop
    base = add("m_none", ["-s", "sites200.fa"], ["reads3.fq", "reads2k.fq"])
    n_distinct = int(re.search(rb"Distinct k-mers in initial set: (\d+)", open(os.path.join(EXP, "m_none.stderr"), "rb").read()).group(1))
    add("m_1_midfile", ["-s", "sites200.fa", "-m", "1"], ["reads2k.fq"], "threshold floor(n*1/2)")
    add("m_frac", ["-s", "sites200.fa", "-m", "0.37"], ["reads2k.fq"])
    add("m_zero_disabled", ["-s", "sites200.fa", "-m", "0"], ["reads2k.fq"])
    add("m_huge", ["-s", "sites200.fa", "-m", "1e300"], ["reads2k.fq"])
    add("m_first_read", ["-s", "sites200.fa", "-m", "0.0001"], ["reads3.fq", "reads2k.fq"], "threshold 0 after truncation => disabled")
    h3 = add("m_probe_reads3", ["-s", "sites200.fa"], ["reads3.fq"])["hits"]
    # threshold = hits(reads3) - 1: trips on the last read of the first file, second file untouched
    m_a = (2.0 * (h3 - 1) + 0.5) / n_distinct
    add("m_file_boundary_stop", ["-s", "sites200.fa", "-m", repr(m_a)], ["reads3.fq", "reads2k.fq"],
        "max_hits = hits(first file) - 1")
    # threshold = hits(reads3): strict '>' so counting continues into the second file
    m_b = (2.0 * h3 + 0.5) / n_distinct
    add("m_file_boundary_continue", ["-s", "sites200.fa", "-m", repr(m_b)], ["reads3.fq", "reads2k.fq"],
        "max_hits = hits(first file): strict >")
    add("m_long", ["-s", "sites200.fa", "-m", "3"], ["long.fa"])

    # ---------------------------------------------------------------- duplicate k-mers in the sites file
    dup = sites[:6] + [("dupA", sites[0][1]), ("dupA", sites[1][1])] + sites[6:10]
    # also a k-mer repeated inside one record and the reverse complement of an earlier k-mer
    dup += [("dupB", first[12][0] + "N" + first[12][0]), ("dupB", first[13][0] + "N" + revcomp(first[2][0]))]
    with open(os.path.join(INP, "sites_dupes.fa"), "w") as f:
        for i, (n, s) in enumerate(dup):
            f.write(">%s %s\n%s\n" % (n, "ref" if i % 2 == 0 else "var", s))
    add("dupes_allowed", ["-s", "sites_dupes.fa", "-d"], ["reads2k.fq", "edge.fa"], "-d: first-seen allele keeps the k-mer")
    add("dupes_abort", ["-s", "sites_dupes.fa"], ["reads2k.fq"], "no -d: m_counts.at() throws -> abort")
    # odd number of records: last REF without VAR -> vector::at throws at print time
    with open(os.path.join(INP, "sites_odd.fa"), "w") as f:
        for i, (n, s) in enumerate(sites[:5]):
            f.write(">%s\n%s\n" % (n, s))
    add("sites_odd_records", ["-s", "sites_odd.fa"], ["reads3.fq"])
    # sites file given as FASTQ / gz / lowercase
    with gzip.open(os.path.join(INP, "sites_lower.fa.gz"), "wt") as f:
        for i, (n, s) in enumerate(sites[:40]):
            f.write(">%s\n%s\n" % (n, s.lower().replace("n", "N" if i % 4 else "n")))
    add("sites_lower_gz", ["-s", "sites_lower.fa.gz"], ["reads2k.fq"])

    # ---------------------------------------------------------------- config 0: 96287 sites, 100k reads (hash only)
    big = os.path.join("/tmp", "ntsm_golden_big")
    os.makedirs(big, exist_ok=True)
    sh([SYNTH, "sites", "--seed", "20241218", "--n-sites", "96287", "--out", os.path.join(big, "hs_n10_like.fa")])
    sh([SYNTH, "reads", "--seed", "7", "--sites-seed", "20241218", "--n-sites", "96287", "--n-reads", "100000",
        "--out", os.path.join(big, "r100k.fq")])
    rc, out, err = run_ref(["-s", "hs_n10_like.fa", "r100k.fq"], big)
    assert rc == 0
    with gzip.GzipFile(os.path.join(EXP, "config0_counts.txt.gz"), "wb", mtime=0) as f:
        f.write(out)
    open(os.path.join(EXP, "config0.stderr"), "wb").write(err)
    config0 = {
        "sites": {"seed": 20241218, "n_sites": 96287, "k": 19,
                  "sha256": hashlib.sha256(open(os.path.join(big, "hs_n10_like.fa"), "rb").read()).hexdigest()},
        "reads": {"seed": 7, "n_reads": 100000, "len": 150,
                  "sha256": hashlib.sha256(open(os.path.join(big, "r100k.fq"), "rb").read()).hexdigest()},
        "counts_sha256": hashlib.sha256(out).hexdigest(),
        "counts_gz": "config0_counts.txt.gz", "stderr": "config0.stderr",
    }
    json.dump({"cases": cases, "config0": config0}, open(os.path.join(HERE, "cases.json"), "w"), indent=1)
    print("wrote %d cases" % len(cases))
    for c in cases:
        print("  %-28s rc=%-4d hits=%s" % (c["name"], c["rc"], c["hits"]))


if __name__ == "__main__":
    main()
