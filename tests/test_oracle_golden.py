"""Pin the CPU oracle (oracle/ntsm_oracle.c) to the recordings of the unmodified reference.

tests/golden/ holds stdout/stderr/exit status the reference produced (tests/golden/make_golden.py);
the oracle must reproduce stdout byte-for-byte and the summary lines of stderr."""
import gzip
import hashlib
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
CASES = json.load(open(os.path.join(G, "cases.json")))


def _oracle(built):
    return os.path.join(built, "oracle", "ntsm_oracle")


def _summary(err_bytes):
    keep = (b"Total ", b"Distinct ", b"Sites Covered", b"Warning: site coverage", b"Reached desired", b"Warning: ")
    return [l for l in err_bytes.split(b"\n") if l.startswith(keep)]


@pytest.mark.parametrize("case", CASES["cases"], ids=[c["name"] for c in CASES["cases"]])
def test_oracle_matches_reference_recording(built, case):
    p = subprocess.run([_oracle(built)] + case["args"] + case["files"], cwd=os.path.join(G, "inputs"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if case["rc"] != 0:
        # reference aborts (uncaught std::out_of_range from m_counts.at / vector::at)
        assert p.returncode == 134
        return
    assert p.returncode == 0
    exp = open(os.path.join(G, "expected", case["stdout"]), "rb").read()
    assert p.stdout == exp
    exp_err = open(os.path.join(G, "expected", case["stderr"]), "rb").read()
    assert _summary(p.stderr) == _summary(exp_err)


def test_oracle_config0_sha256(built, tmp_path):
    """BASELINE.json configs[0]: 96287 sites, 100k reads of 150 bp -- regenerated from seeds."""
    c0 = CASES["config0"]
    synth = os.path.join(built, "build", "ntsm_synth")
    sites = str(tmp_path / "sites.fa")
    reads = str(tmp_path / "reads.fq")
    subprocess.run([synth, "sites", "--seed", str(c0["sites"]["seed"]), "--n-sites", str(c0["sites"]["n_sites"]),
                    "--out", sites], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([synth, "reads", "--seed", str(c0["reads"]["seed"]), "--sites-seed", str(c0["sites"]["seed"]),
                    "--n-sites", str(c0["sites"]["n_sites"]), "--n-reads", str(c0["reads"]["n_reads"]),
                    "--out", reads], check=True)
    assert hashlib.sha256(open(sites, "rb").read()).hexdigest() == c0["sites"]["sha256"]
    assert hashlib.sha256(open(reads, "rb").read()).hexdigest() == c0["reads"]["sha256"]
    p = subprocess.run([_oracle(built), "-s", sites, reads], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0
    assert hashlib.sha256(p.stdout).hexdigest() == c0["counts_sha256"]
    assert p.stdout == gzip.open(os.path.join(G, "expected", c0["counts_gz"])).read()


WRAP = json.load(open(os.path.join(G, "wrap.json")))


@pytest.mark.parametrize("case", WRAP["cases"], ids=[c["name"] for c in WRAP["cases"]])
def test_print_wrap_mod_2_32_matches_reference_recording(built, case):
    """Per-k-mer counts at and beyond 2^32: printCountsMax passes every count through `unsigned`
    (src/FingerPrint.hpp:282, :289), so values are truncated and sums wrap.  tests/golden/make_wrap.py recorded what the
    compiled reference prints when its own insertCount(seq, len, multiplier) has pushed the counts there; the oracle must
    print the same bytes, and so must the product's report code (ntsm_amd/csrc/host/report.cpp through libntsm_host.so)
    when it is handed the same 64-bit count vector."""
    import numpy as np
    import ntsm_amd
    from oracle_binding import OracleFP, read_records
    inp = os.path.join(G, "inputs")
    exp = open(os.path.join(G, "expected", case["stdout"]), "rb").read()
    exp_err = open(os.path.join(G, "expected", case["stderr"]), "rb").read()
    p = subprocess.run([_oracle(built), "-s", case["sites"], "--insert-multiplier", str(case["multiplier"])] + case["files"], cwd=inp,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == exp
    assert _summary(p.stderr) == _summary(exp_err)
    # the product's printer on the same vector
    fp = OracleFP(os.path.join(inp, case["sites"]))
    for f in case["files"]:
        for _, seq in read_records(os.path.join(inp, f))[0]:
            fp.insert_mult(seq, case["multiplier"])
    _, _, cnt = fp.kmers()
    assert int(cnt.max()) >= 2 ** 32                        # the case does reach the wrap
    assert fp.print_counts() == (0, exp)
    sites = ntsm_amd.Sites(os.path.join(inp, case["sites"]))
    assert sites.format_counts(cnt, fp.total_kmers) == (0, exp)
    text, _ = sites.format_summary(cnt, fp.total_bases, fp.total_kmers, fp.total_hits)
    assert _summary(text) == _summary(exp_err)[:len(_summary(text))] and text == fp.info_summary()
