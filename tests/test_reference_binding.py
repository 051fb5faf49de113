"""INTEGRATION.md section 2, compiled and run: the reference's own FingerPrint class (unmodified, #included where it lies under
/root/reference by oracle/ref_gpu_binding.cpp) with the documented binding around it -- its site loader
(src/FingerPrint.hpp:490-564, keys handed over in m_counts' hash order with NTSM_KEYS_HASH64), kseq, printOptionalHeader /
printCountsMax / printInfoSummary (:261-349) -- and libntsm_hip.so where insertCount (:89-103) and the -m check (:476-487) were.

The binary (oracle/_ref/ref_gpu_ntsmCount) is built only where the reference tree exists (this container) and travels to the
GPU box like oracle/_ref/ref_ntsmCount.  On the box every recording of tests/golden/ -- what the reference itself printed on
the CPU -- must come out of it byte for byte: that is what "drop-in for this path" means.
"""
import gzip
import hashlib
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
GOLD = json.load(open(os.path.join(G, "cases.json")))
CASES = GOLD["cases"]
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_gpu_ntsmCount")
SRC = os.path.join(ROOT, "oracle", "ref_gpu_binding.cpp")
HAVE_REF = os.path.isdir("/root/reference/src")


def _summary(err):
    keep = (b"Total ", b"Distinct ", b"Sites Covered", b"Warning: site coverage", b"Reached desired", b"Warning: ")
    return [l for l in err.split(b"\n") if l.startswith(keep)]


def _run(case, env=None):
    return subprocess.run([EXE] + case["args"] + case["files"], cwd=os.path.join(G, "inputs"),
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)


# ---------------------------------------------------------------- CPU: the binding builds, says what INTEGRATION.md says, fails loudly

def test_binding_source_follows_integration_md():
    """Every C-ABI call of INTEGRATION.md section 2's stub appears in the compiled binding, and the binding includes the
    reference header instead of carrying a copy of it."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("## 2. Reference-side stub"):doc.index("## 3. Python")]
    stub = stub[stub.index("```cpp"):stub.index("```", stub.index("```cpp") + 6)]
    src = open(SRC).read()
    calls = set(re.findall(r"\b(ntsm_[a-z_0-9]+)\s*\(", stub))
    assert {"ntsm_create", "ntsm_staging_acquire", "ntsm_submit_staged", "ntsm_sync", "ntsm_counts", "ntsm_destroy"} <= calls
    for c in calls:
        assert re.search(r"\b%s\s*\(" % c, src), "%s is in INTEGRATION.md's stub but not in the compiled binding" % c
    for member in ("gpuInit", "gpuFlush", "processSingleRead", "gpuFinish", "m_gpuKeys", "NTSM_KEYS_HASH64"):
        assert member in stub and member in src, member
    assert '#include "src/FingerPrint.hpp"' in src
    assert "class FingerPrint" not in src and "initCountsHash()" not in src.replace("initCountsHash (", "")   # nothing of the class is restated


@pytest.mark.skipif(not HAVE_REF, reason="the reference tree is only present in the build container")
def test_binding_builds_against_the_unmodified_reference_header(built):
    """`make ref_gpu_binding` compiles the binding as C++11 (the reference's standard) against the header under /root/reference
    and links it to the product library; the binary resolves libntsm_hip.so from the package directory."""
    assert os.path.isfile(EXE)
    assert os.path.getmtime(EXE) >= os.path.getmtime(SRC)
    ldd = subprocess.run(["ldd", EXE], stdout=subprocess.PIPE).stdout.decode()
    line = next(l for l in ldd.split("\n") if "libntsm_hip.so" in l)
    assert os.path.realpath(line.split("=>")[1].split("(")[0].strip()) == os.path.join(ROOT, "ntsm_amd", "libntsm_hip.so")
    assert "not found" not in ldd
    # the build's outputs stay out of the history (oracle/_ref/ is git-ignored; it is NOT gpurun-ignored, so it travels)
    tracked = subprocess.run(["git", "ls-files", "oracle"], cwd=ROOT, stdout=subprocess.PIPE).stdout.decode().split()
    assert not any(t.startswith("oracle/_ref/") for t in tracked)


@pytest.mark.skipif(not os.path.isfile(EXE), reason="oracle/_ref/ref_gpu_ntsmCount not built (no reference tree here)")
def test_binding_fails_loudly_without_a_device():
    """No GPU: the reference's constructor runs (sites are loaded), then ntsm_create refuses -- exit 1, nothing on stdout."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = _run(next(c for c in CASES if c["name"] == "tiny_k19"))
    assert p.returncode == 1 and p.stdout == b""
    assert b"no GPU" in p.stderr and b"no CPU fallback" in p.stderr


# ---------------------------------------------------------------- GPU: the reference's recordings out of the reference's own class

@pytest.fixture(scope="module")
def gpu_binding(built):
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    assert os.path.isfile(EXE), "oracle/_ref/ref_gpu_ntsmCount must travel to the GPU box prebuilt (it needs /root/reference to build)"
    return EXE


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_class_with_binding_reproduces_reference_recordings(gpu_binding, case):
    """stdout byte-identical, summary lines identical, SIGABRT where the reference aborts (inside ITS printCountsMax, after
    the device counted)."""
    p = _run(case)
    if case["rc"] != 0:
        assert p.returncode == case["rc"], p.stderr[-500:]
        return
    assert p.returncode == 0, p.stderr[-500:]
    assert p.stdout == open(os.path.join(G, "expected", case["stdout"]), "rb").read()
    assert _summary(p.stderr) == _summary(open(os.path.join(G, "expected", case["stderr"]), "rb").read())


@pytest.mark.gpu
def test_reference_class_with_binding_small_batches(gpu_binding):
    """Staging slots of 20 kB (hundreds of batches; the -m crossing read inside one, at a boundary, in a later file): same bytes."""
    env = dict(os.environ, NTSM_REF_GPU_BATCH="20000")
    for name in ("tiny_multifile", "m_1_midfile", "m_frac", "m_file_boundary_stop", "m_file_boundary_continue", "dupes_allowed", "edge_fasta"):
        c = next(x for x in CASES if x["name"] == name)
        p = _run(c, env)
        assert p.returncode == 0, (name, p.stderr[-500:])
        assert p.stdout == open(os.path.join(G, "expected", c["stdout"]), "rb").read(), name
        assert _summary(p.stderr) == _summary(open(os.path.join(G, "expected", c["stderr"]), "rb").read()), name


@pytest.mark.gpu
def test_reference_class_with_binding_omp_lanes(gpu_binding):
    """-t 3 without -m: the `#pragma omp parallel for` form -- one producer lane per OpenMP thread, all counting into one
    context (the reference's shared m_counts + omp atomic, src/FingerPrint.hpp:47,:94-99).  Counts do not depend on the
    schedule: same bytes as the reference's single-thread recording; with -m the binding stays on one thread."""
    for name, extra in (("tiny_multifile", ["-t", "3"]), ("edge_empty_file", ["-t", "2"]), ("m_none", ["-t", "2"]),
                        ("m_file_boundary_stop", ["-t", "3"])):
        c = next(x for x in CASES if x["name"] == name)
        for env in (None, dict(os.environ, NTSM_REF_GPU_BATCH="20000")):
            p = subprocess.run([gpu_binding] + c["args"] + extra + c["files"], cwd=os.path.join(G, "inputs"),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert p.returncode == 0, (name, p.stderr[-500:])
            assert p.stdout == open(os.path.join(G, "expected", c["stdout"]), "rb").read(), name
            assert _summary(p.stderr) == _summary(open(os.path.join(G, "expected", c["stderr"]), "rb").read()), name


@pytest.mark.gpu
def test_reference_class_with_binding_config0(gpu_binding, tmp_path):
    """BASELINE.json configs[0] (96,287 sites, 100,000 reads): the reference's printCountsMax over device counts == the
    counts.txt the reference produced on the CPU."""
    import ntsm_amd as nt
    c0 = GOLD["config0"]
    s = nt.SynthShort(sites_seed=c0["sites"]["seed"], n_sites=c0["sites"]["n_sites"], read_seed=c0["reads"]["seed"],
                      sites_path=str(tmp_path / "sites.fa"))
    s.write_fastq(str(tmp_path / "reads.fq"), 0, c0["reads"]["n_reads"])
    assert hashlib.sha256(open(tmp_path / "reads.fq", "rb").read()).hexdigest() == c0["reads"]["sha256"]
    p = subprocess.run([gpu_binding, "-s", str(tmp_path / "sites.fa"), str(tmp_path / "reads.fq")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr[-500:]
    assert hashlib.sha256(p.stdout).hexdigest() == c0["counts_sha256"]
    assert p.stdout == gzip.open(os.path.join(G, "expected", c0["counts_gz"])).read()
    assert _summary(p.stderr) == _summary(open(os.path.join(G, "expected", c0["stderr"]), "rb").read())
