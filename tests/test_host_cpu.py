"""CPU tests of the product's host logic (reader, site loader, report formatting, hash inverse,
C-ABI surface) against the oracle and the recorded reference outputs.  No GPU calls."""
import ctypes
import glob
import json
import os
import re

import numpy as np
import pytest

from oracle_binding import OracleFP, lib as oracle_lib, read_records

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
CASES = json.load(open(os.path.join(G, "cases.json")))["cases"]
INPUTS = sorted(glob.glob(os.path.join(G, "inputs", "*")))


@pytest.fixture(scope="module")
def nt(built):
    import ntsm_amd
    return ntsm_amd


@pytest.mark.parametrize("path", INPUTS, ids=[os.path.basename(p) for p in INPUTS])
def test_reader_matches_oracle_reader(nt, path):
    """SeqReader == kseq semantics: same records, same bytes, same terminating code."""
    recs, rc = read_records(path)
    bases, ends, last = nt.flatten_file(path)
    exp = b"".join(s + b"N" for _, s in recs)
    assert bases.tobytes() == exp
    assert last == rc
    assert len(ends) == len(recs)
    off = 0
    for (_, s), e in zip(recs, ends.tolist()):
        assert e == off + len(s)
        off = e + 1


def _case_k(case):
    return int(case["args"][case["args"].index("-k") + 1]) if "-k" in case["args"] else 19


def _case_sites(case):
    return os.path.join(G, "inputs", case["args"][case["args"].index("-s") + 1])


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_site_loader_and_report_match_reference(nt, case):
    """Site loading + counts.txt formatting, fed with the oracle's per-k-mer counts, reproduce the
    reference's recorded stdout (or its abort)."""
    k, dupes = _case_k(case), "-d" in case["args"]
    cov = float(case["args"][case["args"].index("-m") + 1]) if "-m" in case["args"] else OracleFP.DBL_MAX
    sites = nt.Sites(_case_sites(case), k=k, allow_dupes=dupes)
    fp = OracleFP(_case_sites(case), k=k, cov=cov, dupes=dupes)
    assert sites.n_sites == fp.n_sites
    assert len(sites.keys) == fp.n_distinct
    assert nt.max_hits_for(len(sites.keys), cov) == fp.max_hits
    for f in case["files"]:
        bases, ends, _ = nt.flatten_file(os.path.join(G, "inputs", f))
        fp.process_flat(bases, ends)
    canon, hv, cnt = fp.kmers()
    if case["rc"] != 0:
        # reference aborts at print time: erased duplicate k-mer or REF without VAR
        counts = np.zeros(len(sites.keys), np.uint64)
        rc, _ = sites.format_counts(counts, fp.total_kmers)
        assert rc == 1
        return
    # key order of the product == first-seen order of the oracle (no erasures in rc == 0 cases)
    assert np.array_equal(sites.keys, canon)
    assert all(nt.hash64(int(c), k) == int(h) for c, h in list(zip(canon, hv))[:200])
    rc, text = sites.format_counts(cnt, fp.total_kmers)
    assert rc == 0
    assert text == open(os.path.join(G, "expected", case["stdout"]), "rb").read()
    summ, covered = sites.format_summary(cnt, fp.total_bases, fp.total_kmers, fp.total_hits)
    exp_err = open(os.path.join(G, "expected", case["stderr"]), "rb").read()
    for line in summ.split(b"\n"):
        if line:
            assert line in exp_err.split(b"\n")


def test_hash64_and_inverse_match_oracle(nt):
    L = oracle_lib()
    rng = np.random.default_rng(1)
    for k in (1, 2, 7, 15, 16, 19, 24, 31, 32):
        mask = L.ntsm_oracle_mask(k)
        xs = [0, mask, mask >> 1, 1 & mask] + [int(x) & mask for x in rng.integers(0, 2**63, 300, dtype=np.uint64)]
        for x in xs:
            h = L.ntsm_oracle_hash64(x, mask)
            assert nt.hash64(x, k) == h
            assert nt.hash64_inv(h, k) == x
    # bijection, exhaustively for k = 8
    mask = L.ntsm_oracle_mask(8)
    hs = {nt.hash64(x, 8) for x in range(mask + 1)}
    assert len(hs) == mask + 1


def test_nt4_table_semantics():
    """Byte classes the kernel's LUT must implement (vendor/KseqHashIterator.hpp:114-127)."""
    L = oracle_lib()
    valid = {0: 0, 1: 1, 2: 2, 3: 3}
    for ch, c in zip("ACGTUacgtu", [0, 1, 2, 3, 3, 0, 1, 2, 3, 3]):
        valid[ord(ch)] = c
    for b in range(256):
        assert L.ntsm_oracle_nt4(b) == valid.get(b, 4)


def test_c_abi_exports_every_declared_symbol(nt):
    """Every function declared in include/*.h is exported by the library that implements it."""
    libs = {"ntsm_hip.h": nt.hip_lib, "ntsm_host.h": nt.host_lib, "ntsm_synth.h": nt.synth_lib}
    total = 0
    for hdr, lib in libs.items():
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names = set(re.findall(r"\b(ntsm_[a-z0-9_]+)\s*\(", text))
        assert names, hdr
        for n in sorted(names):
            assert hasattr(lib, n), "%s not exported (declared in %s)" % (n, hdr)
            total += 1
    assert total >= 40


def test_no_cpu_fallback(nt):
    """Without a GPU the product must fail loudly, never silently count on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(nt.NtsmError):
        nt.Context(np.arange(4, dtype=np.uint64), k=19)


def test_synth_generator_is_counter_based(nt, tmp_path):
    """Any slice of the stream equals the same bytes of a bigger slice; FASTQ == flat stream."""
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=5, p_embed=0.5)
    whole = s.host_bytes(0, 50)
    part = s.host_bytes(17, 9)
    assert np.array_equal(whole[17 * 151:26 * 151], part)
    assert whole[150] == ord("N") and set(np.unique(whole)) <= set(b"ACGTN")
    fq = str(tmp_path / "r.fq")
    s.write_fastq(fq, 0, 50)
    bases, ends, rc = nt.flatten_file(fq)
    assert np.array_equal(bases, whole) and np.array_equal(ends, s.read_end(50))
    # the committed tiny fixture was produced by the same generator
    ref, _, _ = nt.flatten_file(os.path.join(G, "inputs", "reads2k.fq"))
    assert np.array_equal(ref[:50 * 151], whole)


def test_reader_fast_path_across_buffer_refills(nt, tmp_path):
    """Files larger than the reader's 4 MiB buffer: records straddling a refill, FASTQ and single-line FASTA,
    plain and gzip -- same records as the oracle's kseq restatement."""
    import gzip
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=21, p_embed=0.2)
    fq = str(tmp_path / "big.fq")
    s.write_fastq(fq, 0, 30000)                          # ~9.3 MB
    raw = open(fq, "rb").read()
    fa = str(tmp_path / "big.fa")
    lines = raw.split(b"\n")
    with open(fa, "wb") as f:                            # 2-line FASTA built from the same reads, last record unterminated
        f.write(b"\n".join(b">" + lines[i][1:] + b"\n" + lines[i + 1] for i in range(0, len(lines) - 1, 4)))
    gz = str(tmp_path / "big.fq.gz")
    with gzip.open(gz, "wb", compresslevel=1) as f:
        f.write(raw)
    exp = s.host_bytes(0, 30000)
    for path in (fq, fa, gz):
        recs, rc = read_records(path)
        bases, ends, last = nt.flatten_file(path)
        assert last == rc == -1 and len(recs) == 30000 == len(ends)
        assert bases.tobytes() == b"".join(x + b"N" for _, x in recs)
        assert np.array_equal(bases, exp)


def test_host_code_under_asan_ubsan(built, tmp_path):
    """Reader, site loader and report formatting over every golden input under AddressSanitizer + UBSan
    (CPU build; GPU sanitizers are not available on this pool)."""
    import subprocess
    exe = str(tmp_path / "host_sanitize")
    host = os.path.join(ROOT, "ntsm_amd", "csrc", "host")
    srcs = [os.path.join(ROOT, "tools", "host_sanitize.cpp")] + [os.path.join(host, f) for f in ("seq_reader.cpp", "site_set.cpp", "report.cpp")]
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe] + srcs + ["-lz"], check=True)
    reads = [p for p in INPUTS if not os.path.basename(p).startswith("sites")]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    for sites, k, dupes in (("sites200.fa", 19, 0), ("sites_dupes.fa", 19, 1), ("sites_dupes.fa", 19, 0), ("sites_odd.fa", 19, 0),
                            ("sites60_k31.fa", 31, 0), ("sites60_k11.fa", 11, 0), ("sites_lower.fa.gz", 19, 0), ("sites200.fa", 32, 1), ("sites200.fa", 1, 1)):
        p = subprocess.run([exe, os.path.join(G, "inputs", sites), str(k), str(dupes)] + reads, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0, (sites, k, p.stderr.decode()[-2000:])
        assert b"records=" in p.stdout
