#!/usr/bin/env python3
"""tests/golden/make_wrap.py -- what the compiled, unmodified reference prints once per-k-mer counts reach 2^32.

printCountsMax() passes every 64-bit count through `unsigned` (src/FingerPrint.hpp:282, :289) before it takes the per-allele
maximum and sum, so values are truncated mod 2^32 and the sums wrap.  Four billion reads are out of reach for a fixture;
the reference's own FingerPrint::insertCount(seq, len, multiplier) (:89, a public member with an `unsigned` third parameter)
is not: oracle/ref_driver.cpp under NTSM_REF_INSERT_MULTIPLIER=M feeds every record of the input through it with that
multiplier, everything else (initCountsHash, printOptionalHeader, printCountsMax, printInfoSummary) is the reference as it is.
Inputs are the committed fixtures of make_golden.py (tests/golden/inputs).  Output: tests/golden/wrap.json + the recorded
stdout / stderr under tests/golden/expected/.  Run in the build container only (needs oracle/_ref)."""
import json, os, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_ntsmCount")
cases = []
for name, mult, files in (("wrap_max_unsigned", 4294967295, ["reads2k.fq"]),      # 1 -> 2^32-1, 2 -> 2^32-2, ...: the maximum is NOT the largest count
                          ("wrap_3e9", 3000000000, ["reads2k.fq", "reads3.fq"]),    # 2 occurrences pass 2^32
                          ("wrap_2p31", 2147483648, ["reads2k.fq"])):               # even counts print as 0, sums of odd ones wrap to 0 / 2^31
    p = subprocess.run([REF, "-s", "sites200.fa"] + files, cwd=os.path.join(HERE, "inputs"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, NTSM_REF_INSERT_MULTIPLIER=str(mult)))
    assert p.returncode == 0, p.stderr[-400:]
    err = b"\n".join(l for l in p.stderr.split(b"\n") if not l.startswith(b"Time: "))
    open(os.path.join(HERE, "expected", name + ".stdout"), "wb").write(p.stdout)
    open(os.path.join(HERE, "expected", name + ".stderr"), "wb").write(err)
    cases.append({"name": name, "multiplier": mult, "sites": "sites200.fa", "files": files, "stdout": name + ".stdout", "stderr": name + ".stderr"})
    print(name, len(p.stdout), err.decode().strip().split("\n")[:3])
json.dump({"cases": cases}, open(os.path.join(HERE, "wrap.json"), "w"), indent=1)
