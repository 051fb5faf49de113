#!/usr/bin/env python3
"""tests/golden/make_vvv.py -- records what the compiled reference (oracle/_ref/ref_ntsmCount = the unmodified
src/FingerPrint.hpp) prints under -v -v -v: the "Current Total: N reads, ..." lines of src/FingerPrint.hpp:70-78, one per
1,000,000 reads, for a seeded synthetic input (200 sites, 2,300,000 reads of 150 bp: two lines), with and without -m.
Output: tests/golden/vvv_progress.json (inputs are regenerated from the seeds by the test).  Run in the build container only
(needs /root/reference to have built oracle/_ref)."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ntsm_amd

PARAMS = dict(sites_seed=11, n_sites=200, read_seed=5, p_embed=0.5, n_reads=2_300_000)
ref = os.path.join(ROOT, "oracle", "_ref", "ref_ntsmCount")
out = {"params": PARAMS, "cases": []}
with tempfile.TemporaryDirectory() as d:
    sp, fq = os.path.join(d, "s.fa"), os.path.join(d, "r.fq")
    s = ntsm_amd.SynthShort(PARAMS["sites_seed"], PARAMS["n_sites"], read_seed=PARAMS["read_seed"], p_embed=PARAMS["p_embed"], sites_path=sp)
    s.write_fastq(fq, 0, PARAMS["n_reads"], threads=8)
    for extra in ([], ["-m", "3000"]):
        p = subprocess.run([ref, "-s", sp, "-v", "-v", "-v"] + extra + [fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr[-400:]
        lines = [l for l in p.stderr.decode().split("\n") if l.startswith(("Current Total:", "max count reached", "Reached desired"))]
        out["cases"].append({"extra": extra, "lines": lines})
        print(extra, lines)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "vvv_progress.json"), "w"), indent=1)
