"""ntsmCount command-line behaviour that ends before any GPU work (CPU-only).

The golden cases under tests/golden were recorded through oracle/ref_driver.cpp's own 15-line flag loop, because the
reference's `main` (src/ntSeqMatchCount.cpp) needs the autoconf-generated config.h and cannot be compiled here.  These
tests pin, BY READING src/ntSeqMatchCount.cpp:53-173, the exit codes and messages of its getopt_long loop and of the
checks behind it -- they are not recorded from a reference binary."""
import os
import signal
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "ntsmCount")
SITES = os.path.join(ROOT, "tests", "golden", "inputs", "sites200.fa")
READS = os.path.join(ROOT, "tests", "golden", "inputs", "reads3.fq")


@pytest.fixture(scope="module")
def exe(built):
    assert os.path.exists(EXE)
    return EXE


def run(exe, *args):
    return subprocess.run([exe] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE)


@pytest.mark.parametrize("flag,value", [("-k", "abc"), ("-m", "x1"), ("-t", "four"), ("--kmer", "k"), ("--maxCov", "?"), ("--threads", "-")])
def test_invalid_numeric_prints_and_returns_0(exe, flag, value):
    """src/ntSeqMatchCount.cpp:90-128: a value the stringstream cannot convert -> "Error - Invalid parameter X: ..." and
    `return 0` (exit status 0, nothing on stdout), before any other check."""
    p = run(exe, "-s", SITES, flag, value, READS)
    letter = {"-k": "k", "--kmer": "k", "-m": "m", "--maxCov": "m", "-t": "t", "--threads": "t"}[flag]
    assert p.returncode == 0 and p.stdout == b""
    assert p.stderr.decode().strip() == "Error - Invalid parameter %s: %s" % (letter, value)


def test_dupes_long_form_takes_an_argument(exe):
    """src/ntSeqMatchCount.cpp:66 declares --dupes with required_argument while -d (:75 "s:t:vhk:m:do:") takes none.
    `--dupes` as the last word is a getopt error (die -> exit 1); `--dupes=1` and `-d` parse."""
    p = run(exe, "-s", SITES, READS, "--dupes")
    assert p.returncode == 1 and b"requires an argument" in p.stderr and b"Try '--help' for more information." in p.stderr
    for ok in (["--dupes=1"], ["--dupes", "x"], ["-d"]):
        p = run(exe, "-s", SITES, *ok)                   # no input files: fails later, but not in the option loop
        assert p.returncode == 1 and b"requires an argument" not in p.stderr and b"invalid option" not in p.stderr
        assert b"Error: Need input files" in p.stderr


def test_checks_after_the_option_loop(exe):
    """src/ntSeqMatchCount.cpp:147-173: k > 32, missing -s, no input files -> their messages in this order, then
    "Try '--help' ..." and exit status 1; an unknown option also ends there."""
    p = run(exe, "-k", "33")
    assert p.returncode == 1 and p.stdout == b""
    assert p.stderr.decode().split("\n")[:4] == ["Error: k cannot be greater than 32", "Error: Missing variants (-s) file",
                                                  "Error: Need input files", "Try '--help' for more information."]
    p = run(exe, "-s", SITES)
    assert p.returncode == 1 and p.stderr.decode().split("\n")[:2] == ["Error: Need input files", "Try '--help' for more information."]
    p = run(exe, READS)
    assert p.returncode == 1 and p.stderr.decode().split("\n")[:2] == ["Error: Missing variants (-s) file", "Try '--help' for more information."]
    p = run(exe, "-s", SITES, "-k", "32", "-Z", READS)
    assert p.returncode == 1 and b"Try '--help' for more information." in p.stderr
    p = run(exe, "-s", SITES, "-k", "0", READS)          # deviation: the reference accepts k = 0
    assert p.returncode == 1 and b"Error: k must be at least 1" in p.stderr


def test_missing_input_file_aborts(exe):
    """src/ntSeqMatchCount.cpp:160 `assert(Util::fexists(...))` is live in release builds (no -DNDEBUG): SIGABRT."""
    p = run(exe, "-s", SITES, "/nonexistent/reads.fq")
    assert p.returncode == -signal.SIGABRT and p.stdout == b""


def test_help_and_version(exe):
    """src/ntSeqMatchCount.cpp:34-52: --help / -h print the dialog to stderr and exit 0; --version likewise."""
    for flag in ("--help", "-h"):
        p = run(exe, flag)
        assert p.returncode == 0 and p.stdout == b"" and p.stderr.startswith(b"Usage: ntsmCount -s [FASTA] [OPTION]... [FILES...]")
        for opt in (b"-t, --threads = INT", b"-m, --maxCov = INT", b"-o, --output = STR", b"-d, --dupes", b"-s, --snp = STR", b"-k, --kmer = INT", b"--version"):
            assert opt in p.stderr
    p = run(exe, "--version")
    assert p.returncode == 0 and p.stdout == b"" and b"ntsmCount" in p.stderr
