"""CPU tests of the product's host logic (reader, site loader, report formatting, hash inverse,
C-ABI surface) against the oracle and the recorded reference outputs.  No GPU calls."""
import ctypes
import glob
import json
import os
import re
import subprocess
import sys

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


def test_round6_entry_points_check_their_arguments(nt):
    """The entry points added in round 6 refuse bad arguments before they touch the HIP runtime (no GPU needed): NULL contexts and
    buffers, thread counts out of range, unknown key kinds."""
    import ctypes as C
    H = nt.hip_lib
    u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
    ends = (C.c_uint64 * 1)(150)
    buf = (C.c_uint8 * 151)()
    assert H.ntsm_submit_pinned(None, C.cast(buf, u8p), 151, C.cast(ends, u64p), 1) == -1          # NTSM_ERR_ARG
    assert H.ntsm_set_submit_threads(None, 2) == -1
    assert H.ntsm_host_pin(None, 4096) == -1 and H.ntsm_host_pin(C.cast(buf, C.c_void_p), 0) == -1 and H.ntsm_host_unpin(None) == -1
    f = C.c_int(-1)
    keys = (C.c_uint64 * 2)(5, 9)
    assert H.ntsm_debug_form_choice(C.cast(keys, u64p), 2, 19, 7, C.byref(f)) == -1               # unknown key kind
    assert H.ntsm_debug_form_choice(C.cast(keys, u64p), 2, 0, 0, C.byref(f)) == -1                # k out of range
    assert H.ntsm_debug_form_choice(C.cast(keys, u64p), 2, 19, 0, None) == -1
    assert H.ntsm_debug_form_choice(C.cast(keys, u64p), 2, 19, 0, C.byref(f)) == 0 and f.value == 0   # two keys: the one-level form


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
    srcs = [os.path.join(ROOT, "tools", "host_sanitize.cpp")] + [os.path.join(host, f) for f in ("seq_reader.cpp", "site_set.cpp", "report.cpp", "inflate.cpp", "inflate_spec.cpp", "gz_stream.cpp", "gz_parallel.cpp", "crc32_fast.cpp")]
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe] + srcs + ["-lz", "-lpthread"], check=True)
    reads = [p for p in INPUTS if not os.path.basename(p).startswith("sites")]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    for sites, k, dupes in (("sites200.fa", 19, 0), ("sites_dupes.fa", 19, 1), ("sites_dupes.fa", 19, 0), ("sites_odd.fa", 19, 0),
                            ("sites60_k31.fa", 31, 0), ("sites60_k11.fa", 11, 0), ("sites_lower.fa.gz", 19, 0), ("sites200.fa", 32, 1), ("sites200.fa", 1, 1)):
        p = subprocess.run([exe, os.path.join(G, "inputs", sites), str(k), str(dupes)] + reads, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0, (sites, k, p.stderr.decode()[-2000:])
        assert b"records=" in p.stdout


def test_block_parallel_fastq_ingest(nt, tmp_path):
    """Block-parallel single-pass FASTQ ingest (ordered commit) == the sequential kseq-equivalent reader on every
    input it accepts: strict files are consumed entirely in parallel; at the first record that is not plain 4-line
    FASTQ the parallel phase stops at a record boundary and the sequential reader continues from that byte."""
    import gzip
    from ntsm_amd.capi import flatten_file_parallel
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=33, p_embed=0.2)
    fq = str(tmp_path / "p.fq")
    s.write_fastq(fq, 0, 30000)                                     # ~9.3 MB
    size = os.path.getsize(fq)
    seq_b, seq_e, _ = nt.flatten_file(fq)
    for threads, block in ((4, 1 << 20), (3, 700_001), (8, 64 << 10), (1, 1 << 20), (16, 4096)):
        r = flatten_file_parallel(fq, threads, block)
        assert r is not None
        assert np.array_equal(r[0], seq_b) and np.array_equal(r[1], seq_e)
        assert r[2]["parallel_records"] == 30000 and r[2]["resume"] == size and r[2]["blocks"] == -(-size // block)
    # adversarial but strict: qualities made of '@' and '+', names containing '+', ragged read lengths (records
    # longer and shorter than a block)
    rng = np.random.default_rng(3)
    recs = []
    for i in range(40000):
        n = int(rng.integers(1, 200)) if i % 5000 else 30000
        sq = "".join(rng.choice(list("ACGTN"), size=n))
        q = "".join(rng.choice(list("@+>I"), size=n))
        recs.append("@r%d+x @y\n%s\n+r%d\n%s\n" % (i, sq, i, q))
    adv = str(tmp_path / "adv.fq")
    open(adv, "w").write("".join(recs))
    a_b, a_e, rc = nt.flatten_file(adv)
    assert rc == -1 and len(a_e) == 40000
    for threads, block in ((4, 256 << 10), (7, 99_991), (5, 8192)):
        r = flatten_file_parallel(adv, threads, block)
        assert r is not None and np.array_equal(r[0], a_b) and np.array_equal(r[1], a_e)
        assert r[2]["parallel_records"] == 40000
    # files that stop being strict somewhere: parallel prefix + sequential rest == sequential reader
    raw = open(fq, "rb").read()
    lines = raw.split(b"\n")
    cases = {"crlf.fq": raw.replace(b"\n", b"\r\n"), "noeol.fq": raw[:-1]}
    w = list(lines)
    w[40001] = w[40001][:70] + b"\n" + w[40001][70:]                # one wrapped sequence line in the middle of the file
    cases["wrapped.fq"] = b"\n".join(w)
    t = list(lines)
    del t[60003]                                                    # a record without its quality line: kseq ends the file there
    cases["truncq.fq"] = b"\n".join(t)
    f = list(lines)
    f[80000] = b">" + f[80000][1:]                                  # a FASTA record in the middle
    del f[80002:80004]
    cases["mixed.fq"] = b"\n".join(f)
    e = list(lines)
    e[20001] = b""                                                  # empty sequence line
    cases["emptyseq.fq"] = b"\n".join(e)
    for name, data in cases.items():
        pth = str(tmp_path / name)
        open(pth, "wb").write(data)
        ref_b, ref_e, ref_rc = nt.flatten_file(pth)
        for threads, block in ((4, 1 << 20), (6, 50_000)):
            r = flatten_file_parallel(pth, threads, block)
            assert r is not None, name
            assert np.array_equal(r[0], ref_b) and np.array_equal(r[1], ref_e), (name, threads, block)
            assert r[2]["resume"] < len(data) or name == "noeol.fq" and r[2]["resume"] <= len(data), name
        if name == "crlf.fq":
            assert r[2]["parallel_records"] == 0 and r[2]["resume"] == 0
        if name == "wrapped.fq":
            assert r[2]["parallel_records"] == 10000
    # the LAST record is not strict and is the first record start of the final block(s): no newline at the end of the file,
    # wrapped, CRLF -- the parallel phase must stop there (resume < size) and the sequential reader must deliver it
    head = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"ACGTTGCA" * 12, b"I" * 96) for i in range(59))      # 59 strict records
    tails = {"tail_noeol.fq": b"@last\n" + b"ACGT" * 25 + b"\n+\n" + b"I" * 100,
             "tail_wrapped.fq": b"@last\n" + b"ACGT" * 12 + b"\n" + b"ACGT" * 13 + b"\n+\n" + b"I" * 48 + b"\n" + b"I" * 52 + b"\n",
             "tail_crlf.fq": b"@last\r\n" + b"ACGT" * 25 + b"\r\n+\r\n" + b"I" * 100 + b"\r\n",
             "tail_long.fq": b"@last\n" + b"ACGT" * 5000 + b"\n+\n" + b"I" * 20000}                     # spans several blocks, no newline at EOF
    for name, tail in tails.items():
        for pad in range(0, 4096, 509):                          # slide the last record across the block boundary
            data = head + b"".join(b"@p%d\n%s\n+\n%s\n" % (i, b"A" * 60, b"I" * 60) for i in range(pad // 127)) + tail
            pth = str(tmp_path / name)
            open(pth, "wb").write(data)
            ref_b, ref_e, ref_rc = nt.flatten_file(pth)
            for threads, block in ((4, 4096), (3, 8192)):
                r = flatten_file_parallel(pth, threads, block)
                if r is None:
                    continue
                assert np.array_equal(r[0], ref_b) and np.array_equal(r[1], ref_e), (name, pad, threads, block)
                assert r[2]["resume"] < len(data), (name, pad)
    # not eligible at all: gzip, FASTA, junk before the first header, small files
    bad = {"junk.fq": b"junk\n" + raw, "tiny.fq": raw[:50_000].rsplit(b"\n@", 1)[0] + b"\n",
           "fasta.fa": b"\n".join(b">" + lines[i][1:] + b"\n" + lines[i + 1] for i in range(0, len(lines) - 1, 4)) + b"\n"}
    for name, data in bad.items():
        pth = str(tmp_path / name)
        open(pth, "wb").write(data)
        assert flatten_file_parallel(pth, 4, 1 << 20) is None, name
    gz = str(tmp_path / "p.fq.gz")
    with gzip.open(gz, "wb", compresslevel=1) as fh:
        fh.write(raw)
    assert flatten_file_parallel(gz, 4, 1 << 18) is None


def test_block_parallel_ingest_under_tsan(nt, tmp_path):
    """The ordered-commit machinery (mutex/condvar/atomics, one thread per sink) under ThreadSanitizer, on a strict
    file and on one that falls back to the sequential reader in the middle; same stream hash for 1 and 8 threads."""
    import subprocess
    exe = str(tmp_path / "parallel_tsan")
    host = os.path.join(ROOT, "ntsm_amd", "csrc", "host")
    srcs = [os.path.join(ROOT, "tools", "parallel_tsan.cpp")] + [os.path.join(host, f) for f in
            ("host_capi.cpp", "early_ingest.cpp", "parallel_fastq.cpp", "seq_reader.cpp", "site_set.cpp", "report.cpp", "inflate.cpp", "inflate_spec.cpp", "gz_stream.cpp", "gz_parallel.cpp", "crc32_fast.cpp", "pack2.cpp")]
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-o", exe] + srcs + ["-lz", "-lpthread"], check=True)
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=5, p_embed=0.2)
    fq = str(tmp_path / "t.fq")
    s.write_fastq(fq, 0, 12000)
    lines = open(fq, "rb").read().split(b"\n")
    lines[24001] = lines[24001][:50] + b"\n" + lines[24001][50:]
    bad = str(tmp_path / "t_wrapped.fq")
    open(bad, "wb").write(b"\n".join(lines))
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    for path in (fq, bad):
        outs = set()
        for threads, block in ((1, 65536), (8, 65536), (8, 4096), (3, 200_000)):
            p = subprocess.run([exe, path, str(threads), str(block)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert p.returncode == 0, p.stderr.decode()[-3000:]
            outs.add(p.stdout.split(b" parallel=")[0] + b" " + p.stdout.split(b" ")[-1])
        assert len(outs) == 1, outs
    # the site loader's own threads (k-merising, hash-bucketed sorts, allele lists) on a file big enough for all of them,
    # with repeated records (duplicate k-mers, collision warnings): same keys as the sequential first pass
    sp = str(tmp_path / "sites.fa")
    nt.SynthShort(sites_seed=3, n_sites=8000, read_seed=1, sites_path=sp)
    raw = open(sp, "rb").read()
    open(sp, "wb").write(raw + b"\n".join(raw.split(b"\n")[:400]) + b"\n")
    outs = set()
    for extra in ({}, {"NTSM_SITES_SEQUENTIAL": "1"}):
        for dupes in ("0", "1"):
            p = subprocess.run([exe, sp, "sites", "19", dupes], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env, **extra))
            assert p.returncode == 0 and b"ThreadSanitizer" not in p.stderr, p.stderr.decode()[-3000:]
            outs.add((dupes, p.stdout))
    assert len(outs) == 2, outs


def _skewed_payload(rng, n):
    """Bytes whose frequencies fall off like Fibonacci numbers backwards (40 distinct values) with far repeats thrown in: the
    Huffman codes zlib builds for it reach the 15-bit limit -- literal / length codes behind the decoder's 11-bit first table,
    distance codes behind its 8-bit one, lengths and distances with all their extra bits."""
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    vals = list(range(33, 73))
    out = bytearray(rng.choices(vals, weights=fib[::-1], k=n))
    for _ in range(n // 4000):                                   # repeats at every distance up to the window, of every length
        a, ln = rng.randrange(0, n - 70000), rng.choice((3, 4, 9, 17, 33, 67, 131, 258, 300))
        b = min(n - ln, a + rng.choice((1, 2, 5, 40, 300, 5000, 20000, 32768 - 300)))
        out[b:b + ln] = out[a:a + ln]
    return bytes(out)


def _gz_member(data, level=6, strategy=0, wbits=15, memlevel=8):
    import zlib
    co = zlib.compressobj(level, zlib.DEFLATED, 16 + wbits, memlevel, strategy)
    return co.compress(data) + co.flush()


def test_gzip_decoder_matches_zlib(nt, tmp_path):
    """ntsm::Inflate / GzStream (the gzip ingest path) against zlib on streams of every block type (stored, fixed,
    dynamic), compression level, strategy and window size; sync/full flush blocks, concatenated members, all header
    fields, trailing garbage; and the failure behaviour gzread has: truncation -> the decodable bytes then a clean
    end, corrupt data / CRC / length -> -1."""
    import gzip
    import io
    import random
    import zlib
    from ntsm_amd.capi import gunzip
    rng = random.Random(1)
    p = str(tmp_path / "t.gz")
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGT") for _ in range(150)),
                                          bytes(rng.choice(b"FFFF:,#") for _ in range(150))) for i in range(6000))
    payloads = [b"", b"A", b"ACGT" * 50000, bytes(rng.getrandbits(8) for _ in range(100000)), fq,
                bytes(rng.choice(b"ab") for _ in range(150000)), b"\x00" * 2500000, bytes((i * 7 + (i >> 8)) & 0xFF for i in range(1 << 19)),
                _skewed_payload(rng, 600000)]
    n = 0
    for pi, data in enumerate(payloads):
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                for wbits, memlevel in ((15, 8), (9, 1), (12, 9)):
                    if (wbits, memlevel) != (15, 8) and not (pi in (4, 5) and level == 6):
                        continue
                    open(p, "wb").write(_gz_member(data, level, strategy, wbits, memlevel))
                    for chunk in ((1 << 16, 7) if len(data) < 20000 else (1 << 16,)):
                        got, rc = gunzip(p, 0, chunk)
                        assert rc == 0 and got == data, (pi, level, strategy, wbits, memlevel, chunk, rc, len(got))
                        n += 1
    assert n > 150
    data = b"".join(b"@r%d\nACGTTGCA%d\n+\nFFFFFFFF\n" % (i, i) for i in range(30000))
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = []
    for i in range(0, len(data), 70000):
        parts += [co.compress(data[i:i + 70000]), co.flush(zlib.Z_SYNC_FLUSH if (i // 70000) % 2 else zlib.Z_FULL_FLUSH)]
    open(p, "wb").write(b"".join(parts) + co.flush())
    assert gunzip(p) == (data, 0)
    multi = _gz_member(data[:100000]) + _gz_member(b"") + _gz_member(data[100000:], 1) + _gz_member(data[:10], 0)
    open(p, "wb").write(multi)
    assert gunzip(p) == (data + data[:10], 0)
    open(p, "wb").write(multi + b"garbage after the last member")
    assert gunzip(p) == gunzip(p, 1) == (data + data[:10], 0)
    bio = io.BytesIO()
    with gzip.GzipFile(filename="some name.fq", mode="wb", fileobj=bio, mtime=12345) as f:
        f.write(data)
    open(p, "wb").write(bio.getvalue())
    assert gunzip(p) == (data, 0)
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(data) + raw.flush()
    hdr = bytes([0x1f, 0x8b, 8, 2 | 4 | 8 | 16, 1, 2, 3, 4, 0, 3]) + (5).to_bytes(2, "little") + b"EXTRA" + b"name\0" + b"comment\0"
    hdr += (zlib.crc32(hdr) & 0xFFFF).to_bytes(2, "little")
    open(p, "wb").write(hdr + body + zlib.crc32(data).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little"))
    assert gunzip(p) == gunzip(p, 1) == (data, 0)
    good = _gz_member(data)
    for blob in (good[:-8] + bytes([good[-8] ^ 1]) + good[-7:], good[:-4] + bytes([good[-4] ^ 1]) + good[-3:]):   # CRC, ISIZE
        open(p, "wb").write(blob)
        got, rc = gunzip(p)
        assert rc == -1 and gunzip(p, 1)[1] == -1 and got == data
    for cut in [len(good) - k for k in (1, 4, 7, 8, 9, 20)] + [rng.randrange(11, len(good)) for _ in range(40)] + [10, 11, 12, 3, 9]:
        open(p, "wb").write(good[:cut])
        got, rc = gunzip(p)
        ref, rrc = gunzip(p, 1)
        assert rc == rrc == 0 and got == ref, ("truncated at", cut, rc, rrc, len(got), len(ref))
    for _ in range(120):                                            # single bit flips: same verdict as zlib
        b = bytearray(good)
        i = rng.randrange(10, len(good) - 8)
        b[i] ^= 1 << rng.randrange(8)
        open(p, "wb").write(bytes(b))
        got, rc = gunzip(p)
        ref, rrc = gunzip(p, 1)
        assert rc == rrc and (rc != 0 or got == ref), ("flip", i, rc, rrc)
    for hdr_bad in (b"\x1f\x8b\x07" + good[3:], good[:3] + b"\x20" + good[4:]):     # method != 8, reserved flag
        open(p, "wb").write(hdr_bad)
        assert gunzip(p)[1] == -1 and gunzip(p, 1)[1] == -1


def test_parallel_gzip_decoder_matches_zlib(nt, tmp_path):
    """The chunk-parallel decoder for ONE ordinary gzip stream (gz_parallel.cpp: speculative 16-bit decoding from block starts
    found by trial, spliced in by the in-order decoder) against zlib's gzread: every compression level / strategy / window of
    the sequential test's corpus with chunks small enough that each file is cut many times, sync / full flush blocks,
    concatenated members (also tiny and empty ones), trailing garbage, a stored-block-only file (no chunk ever starts: all in
    order), CRC / ISIZE damage, 51 truncation points and 160 bit flips with the same bytes / verdict as gzread."""
    import random
    import zlib
    from ntsm_amd.capi import gunzip, gunzip_parallel_chunk, gunzip_parallel_stats
    rng = random.Random(2)
    p = str(tmp_path / "t.gz")
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGT") for _ in range(150)),
                                          bytes(rng.choice(b"FFFF:,#") for _ in range(150))) for i in range(9000))
    payloads = [b"ACGT" * 200000, bytes(rng.getrandbits(8) for _ in range(300000)), fq, bytes(rng.choice(b"ab") for _ in range(400000)),
                b"\x00" * 2500000, bytes((i * 7 + (i >> 8)) & 0xFF for i in range(1 << 20)), _skewed_payload(rng, 900000)]
    try:
        n = spliced_total = 0
        for pi, data in enumerate(payloads):
            for level in (0, 1, 6, 9):
                for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                    for wbits, memlevel in ((15, 8), (9, 1), (12, 9)):
                        if (wbits, memlevel) != (15, 8) and not (pi in (2, 3) and level == 6):
                            continue
                        blob = _gz_member(data, level, strategy, wbits, memlevel)
                        open(p, "wb").write(blob)
                        for chunk in (4096, 50000):
                            gunzip_parallel_chunk(chunk)
                            got, rc = gunzip(p, 3, 1 << 16)
                            assert rc == 0 and got == data, (pi, level, strategy, wbits, memlevel, chunk, rc, len(got), len(data))
                            st = gunzip_parallel_stats()
                            spliced_total += st[0]
                            if level == 0 or strategy == zlib.Z_FIXED:
                                assert st[0] == 0                      # stored / fixed blocks only: nothing for a chunk to start at
                            n += 1
        assert n > 200 and spliced_total > 1000
        gunzip_parallel_chunk(8192)
        data = b"".join(b"@r%d\nACGTTGCA%d\n+\nFFFFFFFF\n" % (i, i) for i in range(60000))
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        parts = []
        for i in range(0, len(data), 70000):
            parts += [co.compress(data[i:i + 70000]), co.flush(zlib.Z_SYNC_FLUSH if (i // 70000) % 2 else zlib.Z_FULL_FLUSH)]
        open(p, "wb").write(b"".join(parts) + co.flush())
        assert gunzip(p, 4) == (data, 0) and gunzip_parallel_stats()[0] > 5
        # blocks that begin with a copy of the window's LAST bytes followed by the literals 0, 1, 2 ...: in a chunk that starts
        # there the symbols read 0xFFFA .. 0xFFFF, 0, 1, 2 ... -- consecutive modulo 65536, yet not a stretch of the window
        # (resolve()'s run test must not take them for one)
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        parts, plain = [], []
        for i in range(40):
            # more than a window of bytes above 0x7F between two units: the 0, 1, 2 ... of the previous one is out of reach, zlib codes them as literals
            body = bytes(rng.getrandbits(8) | 0x80 for _ in range(rng.randrange(34000, 40000)))
            tail = bytes(rng.getrandbits(8) | 0x80 for _ in range(rng.choice((3, 6, 12, 31))))
            parts += [co.compress(body + tail), co.flush(zlib.Z_SYNC_FLUSH)]
            again = tail + bytes(range(48))
            parts.append(co.compress(again))
            plain += [body, tail, again]
        want = b"".join(plain)
        open(p, "wb").write(b"".join(parts) + co.flush())
        for chunk in (20000, 30000, 50000):
            gunzip_parallel_chunk(chunk)
            assert gunzip(p, 4) == (want, 0), chunk
            assert gunzip_parallel_stats()[0] > 3
        gunzip_parallel_chunk(8192)
        multi = _gz_member(data[:700000]) + _gz_member(b"") + _gz_member(data[700000:1500000], 1) + _gz_member(data[:10], 0) + _gz_member(data[1500000:], 9) + _gz_member(b"x")
        open(p, "wb").write(multi)
        assert gunzip(p, 4) == (data + data[:10] + b"x" if False else data[:700000] + data[700000:1500000] + data[:10] + data[1500000:] + b"x", 0)
        assert gunzip_parallel_stats()[0] > 5
        open(p, "wb").write(multi + b"garbage after the last member")
        assert gunzip(p, 4) == gunzip(p, 1)
        good = _gz_member(data)
        for blob in (good[:-8] + bytes([good[-8] ^ 1]) + good[-7:], good[:-4] + bytes([good[-4] ^ 1]) + good[-3:]):   # CRC, ISIZE
            open(p, "wb").write(blob)
            got, rc = gunzip(p, 4)
            assert rc == -1 and gunzip(p, 1)[1] == -1 and got == data
        for cut in [len(good) - k for k in (1, 4, 7, 8, 9, 20)] + [rng.randrange(11, len(good)) for _ in range(40)] + [10, 11, 12, 3, 9]:
            open(p, "wb").write(good[:cut])
            got, rc = gunzip(p, 4)
            ref, rrc = gunzip(p, 1)
            assert rc == rrc == 0 and got == ref, ("truncated at", cut, rc, rrc, len(got), len(ref))
        for _ in range(160):                                            # single bit flips: same verdict as zlib, same bytes when it accepts
            b = bytearray(good)
            i = rng.randrange(10, len(good) - 8)
            b[i] ^= 1 << rng.randrange(8)
            open(p, "wb").write(bytes(b))
            got, rc = gunzip(p, 4)
            ref, rrc = gunzip(p, 1)
            assert rc == rrc and (rc != 0 or got == ref), ("flip", i, rc, rrc)
        # a real-sized FASTQ at the default chunk size
        gunzip_parallel_chunk(0)
        s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=9, p_embed=0.2)
        fqp = str(tmp_path / "big.fq")
        s.write_fastq(fqp, 0, 150000, threads=4)
        raw = open(fqp, "rb").read()
        open(p, "wb").write(_gz_member(raw, 6))
        assert gunzip(p, 4, 1 << 20) == (raw, 0) and gunzip_parallel_stats()[0] >= 2
        # a member that inflates a thousandfold: chunks give up at 64 times their size, the in-order decoder does it all
        zeros = b"\x00" * (64 << 20)
        open(p, "wb").write(_gz_member(zeros, 6))
        gunzip_parallel_chunk(8192)
        got, rc = gunzip(p, 4, 1 << 20)
        assert rc == 0 and got == zeros and gunzip_parallel_stats()[0] == 0 and gunzip_parallel_stats()[1] > 0
    finally:
        gunzip_parallel_chunk(0)


def test_parallel_gzip_under_tsan_and_asan(nt, tmp_path):
    """Producer / chunk workers / reader of the parallel plain-gzip path under ThreadSanitizer, and its speculative decoder on
    good, truncated and damaged streams under AddressSanitizer + UBSan."""
    import random
    import subprocess
    host = os.path.join(ROOT, "ntsm_amd", "csrc", "host")
    srcs = [os.path.join(ROOT, "tools", "gunzip_sanitize.cpp")] + [os.path.join(host, f) for f in ("gz_stream.cpp", "gz_parallel.cpp", "inflate.cpp", "inflate_spec.cpp", "crc32_fast.cpp")]
    rng = random.Random(9)
    data = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGT") for _ in range(100)), bytes(rng.choice(b"FFF:,") for _ in range(100))) for i in range(20000))
    good = _gz_member(data, 6)
    files = []
    blobs = [("good", good), ("trunc", good[:len(good) * 2 // 3]), ("multi", good + _gz_member(data[:5000], 1) + good)]
    for j in range(12):
        b = bytearray(good)
        for _ in range(1 + j % 3):
            b[rng.randrange(10, len(b))] ^= 1 << rng.randrange(8)
        blobs.append(("bad%d" % j, bytes(b)))
    for name, blob in blobs:
        path = str(tmp_path / (name + ".gz"))
        open(path, "wb").write(blob)
        files.append(path)
    for san, extra in (("thread", []), ("address,undefined", ["-fno-sanitize-recover=all"])):
        exe = str(tmp_path / ("gunzip_" + san.split(",")[0]))
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + san] + extra + ["-o", exe] + srcs + ["-lz", "-lpthread"], check=True)
        p = subprocess.run([exe] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1", NTSM_DECODER_THREADS="5", NTSM_PARALLEL_CHUNK="20000"))
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        out = p.stdout.split()
        assert out[0:2] == [b"%d" % len(data), b"0"] and out[3] == b"0" and out[4:6] == [b"%d" % (2 * len(data) + 5000), b"0"], out[:8]
        # a reader that walks away in the middle (kseq ends a file at the first malformed record): chunks are being resolved
        # into pieces at that moment -- nothing may be freed under the workers
        for stop in (1, 300000, 2000000):
            p = subprocess.run([exe] + files[:3], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1", NTSM_DECODER_THREADS="6", NTSM_PARALLEL_CHUNK="8000", NTSM_STOP_AFTER=str(stop)))
            assert p.returncode == 0, p.stderr.decode()[-3000:]


def _reads_of(bases, ends):
    bb, out, s = bases.tobytes(), [], 0
    for x in ends.tolist():
        out.append(bb[s:x])
        s = x + 1
    return out


def test_parallel_gzip_ingest_equals_sequential_reader(nt, tmp_path):
    """Decoder pool + piece-parallel parse (parallel_gz_fastq.hpp) against the sequential reader (itself pinned to kseq) on the
    same file: the same reads (as a multiset: inside a piece the order may differ, counting does not depend on it), for
    strict FASTQ, '@' at the start of quality lines, a wrapped record / CRLF / FASTA record in the middle (the sequential
    reader takes over there), a last record without newline, FASTA, empty and tiny files, several members, BGZF, truncated
    and CRC-damaged files, chunk sizes from a fraction of a record's worth to bigger than the file, sinks smaller than a
    piece."""
    import collections
    import random
    from ntsm_amd.capi import flatten_file, flatten_file_parallel_gz, gunzip_parallel_chunk
    rng = random.Random(3)
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=5, p_embed=0.2)
    fq = str(tmp_path / "t.fq")
    s.write_fastq(fq, 0, 30000)
    raw = open(fq, "rb").read()
    lines = raw.split(b"\n")
    cases = {"strict": raw}
    q = list(lines)
    for i in range(3, len(q) - 1, 4):                                # quality lines that start with '@' (and some with '+')
        if rng.random() < 0.3:
            q[i] = (b"@" if rng.random() < 0.7 else b"+") + q[i][1:]
    cases["at_qualities"] = b"\n".join(q)
    w = list(lines)
    w[60001] = w[60001][:50] + b"\n" + w[60001][50:]                 # one wrapped sequence line in the middle
    cases["wrapped_mid"] = b"\n".join(w)
    c = list(lines)
    c[40000] = c[40000] + b"\r"
    cases["crlf_mid"] = b"\n".join(c)
    cases["fasta_mid"] = b"\n".join(lines[:48000]) + b"\n>fa1 x\nACGTACGTACGTACGTACGTAGCTAGCATCGATCGAT\nACGT\n" + b"\n".join(lines[48000:])
    cases["no_final_newline"] = raw[:-1]
    cases["fasta"] = b"".join(b">s%d\n%s\n" % (i, bytes(rng.choice(b"ACGT") for _ in range(80))) for i in range(20000))
    cases["junk_first"] = b"junk line\n" + raw
    cases["empty"] = b""
    cases["tiny"] = b"@a\nACGT\n+\nFFFF\n"
    try:
        for name, data in cases.items():
            p = str(tmp_path / (name + ".gz"))
            open(p, "wb").write(_gz_member(data, 6 if name != "strict" else 1))
            ref_b, ref_e, ref_rc = flatten_file(p)
            ref = collections.Counter(_reads_of(ref_b, ref_e))
            for chunk, dec, par, sink in ((3000, 3, 3, 2000), (20000, 4, 5, 1 << 20), (1 << 20, 2, 2, 50000)):
                gunzip_parallel_chunk(chunk)
                b, e, info = flatten_file_parallel_gz(p, dec, par, sink)
                got = collections.Counter(_reads_of(b, e))
                assert got == ref and len(e) == len(ref_e), (name, chunk, dec, par, sink, len(e), len(ref_e), info)
                if name in ("strict", "at_qualities"):
                    assert info["parallel_records"] == len(ref_e)               # nothing left for the sequential reader
                if name in ("wrapped_mid", "crlf_mid", "fasta_mid"):
                    assert 0 < info["parallel_records"] < len(ref_e)            # parallel up to the odd record, sequential from there
                if name in ("fasta", "junk_first"):
                    assert info["parallel_records"] == 0
        gunzip_parallel_chunk(20000)
        # several members (cut anywhere, also inside records), an empty member, trailing garbage
        p = str(tmp_path / "multi.gz")
        open(p, "wb").write(_gz_member(raw[:1_000_003], 1) + _gz_member(b"") + _gz_member(raw[1_000_003:4_000_000], 9) + _gz_member(raw[4_000_000:], 6) + b"trailing")
        ref_b, ref_e, _ = flatten_file(p)
        b, e, info = flatten_file_parallel_gz(p, 4, 4, 1 << 20)
        assert collections.Counter(_reads_of(b, e)) == collections.Counter(_reads_of(ref_b, ref_e)) and info["parallel_records"] == len(ref_e) == 30000
        # BGZF: groups of members inflated in parallel, the same piece-parallel parse behind them
        p = str(tmp_path / "bgzf.gz")
        open(p, "wb").write(_bgzf(raw, level=4))
        b, e, info = flatten_file_parallel_gz(p, 4, 4, 1 << 20)
        assert collections.Counter(_reads_of(b, e)) == collections.Counter(_reads_of(ref_b, ref_e)) and info["parallel_records"] == 30000 and info["pieces"] >= 2
        # truncated file: every decodable byte, then a clean end; damaged CRC: every record, status -1
        good = _gz_member(raw, 6)
        for cut in (len(good) // 3, len(good) - 5, len(good) - 9):
            p = str(tmp_path / "cut.gz")
            open(p, "wb").write(good[:cut])
            ref_b, ref_e, _ = flatten_file(p)
            b, e, info = flatten_file_parallel_gz(p, 4, 3, 1 << 20)
            assert collections.Counter(_reads_of(b, e)) == collections.Counter(_reads_of(ref_b, ref_e)), cut
        p = str(tmp_path / "crc.gz")
        open(p, "wb").write(good[:-8] + bytes([good[-8] ^ 1]) + good[-7:])
        ref_b, ref_e, _ = flatten_file(p)
        b, e, info = flatten_file_parallel_gz(p, 4, 3, 1 << 20)
        assert collections.Counter(_reads_of(b, e)) == collections.Counter(_reads_of(ref_b, ref_e)) and info["status"] == -1 and len(e) == 30000
        bad = bytearray(good)
        bad[len(good) // 2] ^= 0x40                                    # damage in the middle: same reads as the sequential path up to the error
        p = str(tmp_path / "bad.gz")
        open(p, "wb").write(bytes(bad))
        ref_b, ref_e, _ = flatten_file(p)
        b, e, info = flatten_file_parallel_gz(p, 4, 3, 1 << 20)
        assert collections.Counter(_reads_of(b, e)) == collections.Counter(_reads_of(ref_b, ref_e))
    finally:
        gunzip_parallel_chunk(0)


def test_parallel_gzip_ingest_under_tsan(nt, tmp_path):
    """Producer, chunk workers and parsing threads of the parallel gzip ingest under ThreadSanitizer: strict file, a file that
    falls back to the sequential reader in the middle, a truncated one; the same order-free digest whatever the thread counts."""
    import subprocess
    exe = str(tmp_path / "parallel_tsan")
    host = os.path.join(ROOT, "ntsm_amd", "csrc", "host")
    srcs = [os.path.join(ROOT, "tools", "parallel_tsan.cpp")] + [os.path.join(host, f) for f in
            ("host_capi.cpp", "early_ingest.cpp", "parallel_fastq.cpp", "seq_reader.cpp", "site_set.cpp", "report.cpp", "inflate.cpp", "inflate_spec.cpp", "gz_stream.cpp", "gz_parallel.cpp", "crc32_fast.cpp", "pack2.cpp")]
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-o", exe] + srcs + ["-lz", "-lpthread"], check=True)
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=5, p_embed=0.2)
    fq = str(tmp_path / "t.fq")
    s.write_fastq(fq, 0, 12000)
    raw = open(fq, "rb").read()
    lines = raw.split(b"\n")
    lines[24001] = lines[24001][:50] + b"\n" + lines[24001][50:]
    good = _gz_member(raw, 6)
    files = {"strict.gz": good, "wrapped.gz": _gz_member(b"\n".join(lines), 6), "cut.gz": good[:len(good) * 2 // 3]}
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    for name, blob in files.items():
        path = str(tmp_path / name)
        open(path, "wb").write(blob)
        outs = set()
        for dec, par, sink, chunk in ((1, 1, 1 << 20, 20000), (5, 4, 30000, 20000), (3, 6, 1 << 20, 5000), (4, 2, 1 << 20, 1 << 20)):
            p = subprocess.run([exe, path, str(dec), str(par), str(sink), str(chunk)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert p.returncode == 0, p.stderr.decode()[-3000:]
            outs.add(p.stdout.split(b" parallel=")[0] + b" " + p.stdout.split(b" ")[-1])
        assert len(outs) == 1, outs
        # the early ingest on the same file (parsers -> packed chunks -> consumers, a small chunk budget so that parsers wait)
        if name != "cut.gz":
            outs = set()
            for par, dec, block, cpos, budget, cons, after in ((4, 4, 1 << 20, 100000, 10, 3, None), (2, 3, 1 << 20, 1 << 20, 64, 1, None), (4, 4, 1 << 20, 100000, 10, 3, 2)):
                p = subprocess.run([exe, path, "early", str(par), str(dec), str(block), str(cpos), str(budget), str(cons)] + ([str(after)] if after is not None else []),
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env, NTSM_PARALLEL_CHUNK="20000"))
                assert p.returncode == 0, p.stderr.decode()[-3000:]
                outs.add(p.stdout)
            assert len(outs) == 1 and b"reads=12000 " in outs.pop(), outs


def test_parallel_gzip_ingest_long_unparsable_rest_is_not_counted_twice(nt, tmp_path):
    """ADVICE round 4 (parallel_gz_fastq.hpp): a piece whose unparsed rest is too long to carry into the next link used to go
    back to the stream WHOLE -- after its link had held and a full sink had already flushed records of it, which the sequential
    reader then parsed a second time.  Now it commits what it parsed and the phase ends at the rest.  Provoked with a small
    limit (debug hook; the default is 256 MiB), sinks much smaller than a piece, and one record wrapped over 3,000 lines in the
    middle of strict records: the same multiset of reads as the sequential reader, for several piece sizes."""
    import collections
    from ntsm_amd.capi import debug_gz_max_tail, flatten_file, flatten_file_parallel_gz, gunzip_parallel_chunk
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=5, p_embed=0.2)
    fq = str(tmp_path / "t.fq")
    s.write_fastq(fq, 0, 12000)
    lines = open(fq, "rb").read().split(b"\n")
    long_seq = (b"ACGTTGCAAGCTTAGC" * 4 + b"\n") * 3000                       # one record, sequence wrapped over 3,000 lines (195 KB)
    long_rec = b"@wrapped\n" + long_seq + b"+\n" + (b"I" * 64 + b"\n") * 3000
    data = b"\n".join(lines[:20000]) + b"\n" + long_rec + b"\n".join(lines[20000:])
    p = str(tmp_path / "wrapped.gz")
    open(p, "wb").write(_gz_member(data, 6))
    ref_b, ref_e, _ = flatten_file(p)
    ref = collections.Counter(_reads_of(ref_b, ref_e))
    assert len(ref_e) == 12001
    try:
        debug_gz_max_tail(5000)
        for chunk, dec, par, sink in ((20000, 4, 4, 2000), (60000, 3, 2, 3000), (8000, 2, 3, 1500)):
            gunzip_parallel_chunk(chunk)
            b, e, info = flatten_file_parallel_gz(p, dec, par, sink)
            got = collections.Counter(_reads_of(b, e))
            assert got == ref and len(e) == len(ref_e), (chunk, dec, par, sink, len(e), len(ref_e), info)
            assert 0 < info["parallel_records"] < len(ref_e)
    finally:
        debug_gz_max_tail(0)
        gunzip_parallel_chunk(0)


def test_early_ingest_allocation_failure_is_an_error_not_a_shorter_run(nt, tmp_path):
    """ADVICE round 4 (early_ingest.cpp): a chunk that cannot be allocated used to look like an abandoned run -- the sink dropped
    its records and the CLI printed counts of fewer reads with exit status 0.  Now the ingest records the failure
    (EarlyIngest::failed), the host hook answers -3, and FingerPrint::drainEarly turns it into exit(1) with a message (the
    GPU-side test drives the CLI).  Failing the 1st, 3rd and 9th allocation, plain and gzip input."""
    from ntsm_amd.capi import NtsmError, debug_early_alloc_fail, early_ingest
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=5, p_embed=0.2)
    fq = str(tmp_path / "t.fq")
    s.write_fastq(fq, 0, 40000)
    gz = str(tmp_path / "t.fq.gz")
    open(gz, "wb").write(_gz_member(open(fq, "rb").read(), 4))
    ok = early_ingest(fq, 4, 3, 1 << 20, 60_000, 64, 2)
    assert ok is not None and ok[1] == 40000
    try:
        for path in (fq, gz):
            for nth in (1, 3, 9):
                debug_early_alloc_fail(nth)
                with pytest.raises(NtsmError, match="-3"):
                    early_ingest(path, 4, 3, 1 << 20, 60_000, 64, 2)
                debug_early_alloc_fail(0)
                again = early_ingest(path, 4, 3, 1 << 20, 60_000, 64, 2)     # and the next run is whole again
                assert again is not None and again[1] == 40000 and again[2] == ok[2]
    finally:
        debug_early_alloc_fail(0)


_AFFINITY_PROBE = r"""
import json, os, sys, threading, time
sys.path.insert(0, %r)
n, path = int(sys.argv[1]), sys.argv[2]
os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:n]))
from ntsm_amd.capi import granted_cpus, ingest_plan, early_ingest, flatten_file_parallel_gz, gunzip_parallel_chunk
plan = ingest_plan(16)                                   # what `ntsmCount -t 16` does on a host that grants n CPUs
base = len(os.listdir('/proc/self/task'))
peak, stop = [0], [False]
def watch():
    while not stop[0]:
        peak[0] = max(peak[0], len(os.listdir('/proc/self/task'))); time.sleep(0.0003)
w = threading.Thread(target=watch); w.start()
gunzip_parallel_chunk(20000)
b, e, info = flatten_file_parallel_gz(path, plan['decoders'], plan['feeders'], 1 << 20)
p_gz, peak[0] = peak[0] - base - 1, 0
text, n_reads, n_bases, n_par = early_ingest(path, plan['feeders'], plan['early_decoders'], 1 << 20, 60000, 64, 1)
p_early = peak[0] - base - 1 - 1                         # minus the watcher and the harness's one consumer thread
stop[0] = True; w.join()
import hashlib
runs = sorted(r for r in text.split(b'N') if r)
flat = sorted(r for r in bytes(b).split(b'N') if r)      # inside a piece the order of the records may differ: compare as multisets
print(json.dumps(dict(granted=granted_cpus(), plan=plan, peak_gz=p_gz, peak_early=p_early, n_reads=len(e), early_reads=n_reads,
                      flat_sha=hashlib.sha256(b'N'.join(flat)).hexdigest(), early_sha=hashlib.sha256(b'N'.join(runs)).hexdigest())))
"""


def test_ingest_thread_counts_follow_the_cpus_granted(nt, tmp_path):
    """the feeder / decoder counts used to be constants fitted to the 16-CPU pod.  They now come from a
    table keyed on the CPUs the process is GRANTED -- min(affinity mask, cgroup quota): host_shape.hpp -- and the host library's
    ingest path is driven here (no GPU) under affinity masks of 2, 4 and 8 CPUs with the plan `-t 16` gets there: the same
    reads (identical packed bytes, as multisets of runs for the early ingest whose chunk order is free) as on the unrestricted
    host, and never more ingest threads than 2 x CPUs (+ one coordinator thread that only waits)."""
    from ntsm_amd.capi import granted_cpus, ingest_plan
    have = len(os.sched_getaffinity(0))
    assert granted_cpus() <= have
    for cpus in (1, 2, 3, 4, 6, 8, 12, 16, 24, 64, 256):
        for asked in (1, 2, 4, 16, 64):
            p = ingest_plan(asked, cpus)
            assert p["cpus"] == cpus and 1 <= p["feeders"] <= min(asked, cpus, 16) and p["feeders"] == min(asked, cpus, 16)
            assert 1 <= p["decoders"] <= 2 * p["feeders"] and 1 <= p["early_decoders"] <= 2 * asked
            assert p["feeders"] + p["decoders"] + 1 <= max(2 * cpus, 3) + (1 if cpus >= 16 else 0), (asked, cpus, p)
    assert ingest_plan(16, 16) == dict(cpus=16, feeders=16, decoders=16, early_decoders=14)      # the measured row (rounds 3-4)
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=5, p_embed=0.2)
    fq = str(tmp_path / "t.fq")
    s.write_fastq(fq, 0, 60000)
    gz = str(tmp_path / "t.fq.gz")
    open(gz, "wb").write(_gz_member(open(fq, "rb").read(), 6))
    script = str(tmp_path / "probe.py")
    open(script, "w").write(_AFFINITY_PROBE % ROOT)
    seen = {}
    for n in (2, 4, 8, have):
        if n > have:
            continue
        p = subprocess.run([sys.executable, script, str(n), gz], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-800:]
        d = json.loads(p.stdout.decode().strip().split("\n")[-1])
        assert d["granted"] == n and d["plan"]["cpus"] == n and d["n_reads"] == d["early_reads"] == 60000
        assert d["peak_gz"] <= 2 * n + 1 and d["peak_early"] <= 2 * n + 1, d
        seen[n] = (d["flat_sha"], d["early_sha"])
    assert len(set(seen.values())) == 1, seen                     # identical bytes whatever the grant


def test_run_form_filter_has_no_false_negatives(nt, tmp_path):
    """The run-anchored kernel (kernels_run.hip) looks a k-mer up only if the signature of its anchored 16-mer is in
    the filter block of its minimizer.  Exactness therefore needs: for EVERY site k-mer, in either strand, under every alignment
    of the window to the kernel's position counter (the order key carries position mod 16 in its low bits and decides between
    equal 12-mers), the bits the DEVICE derives are bits the HOST set (tables.cpp sets the signature for every offset at which
    the minimum occurs).  Walked here with numpy on the filter image the library's host code builds without a device
    (ntsm_debug_run_filter): a 200-site set, plus the windows that stress the rule -- reverse-complement palindromes (a 12-mer
    and its twin 8 positions on), tandem repeats of period 1 .. 7, a 12-mer repeated at distance 12 -- and three filter sizes."""
    from ntsm_amd.capi import debug_run_filter
    rng = np.random.default_rng(7)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rcs = lambda x: "".join(comp[c] for c in reversed(x))
    wins = []
    for _ in range(200):
        wins.append("".join(rng.choice(list("ACGT"), size=31)))
    for _ in range(120):
        h = "".join(rng.choice(list("ACGT"), size=16))
        wins.append((h + rcs(h))[:31] if rng.random() < 0.5 else (h[:15] + "A" + rcs(h[:15])))
    for period in range(1, 8):
        for _ in range(10):
            wins.append(("".join(rng.choice(list("ACGT"), size=period)) * 31)[:31])
    for _ in range(60):
        a = "".join(rng.choice(list("ACGT"), size=31))
        wins.append(a[:12] + a[:12] + a[24:])
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    fw = []
    for w in wins:
        for s0 in range(13):
            x = 0
            for ch in w[s0:s0 + 19]:
                x = (x << 2) | code[ch]
            fw.append(x)
    fw = np.array(sorted(set(fw)), dtype=np.uint64)

    def rc19(x):
        r = np.zeros_like(x)
        y = x.copy()
        for _ in range(19):
            r = (r << np.uint64(2)) | (np.uint64(3) - (y & np.uint64(3)))
            y = y >> np.uint64(2)
        return r

    def rc16(w):                                             # 16-base word, first base in the top bits
        r = np.zeros_like(w)
        y = w.copy()
        for _ in range(16):
            r = (r << np.uint32(2)) | (np.uint32(3) - (y & np.uint32(3)))
            y = y >> np.uint32(2)
        return r
    rc = rc19(fw)
    canon = np.unique(np.minimum(fw, rc))
    for kib in (0, 64, 1024):
        blocks = debug_run_filter(canon, kib)
        nb = np.uint64(len(blocks))
        assert len(blocks) >= 16 and (kib == 0 or len(blocks) == kib * 64)
        for strand in (fw, rc):                              # the read may show either strand of a site k-mer
            srev = rc19(strand)
            # order hashes of the eight 12-mers (start offset q = 0 .. 7): (canonical code * odd) mod 2^24 (ntsm_device.h, NTSM_RUN_ORDER 0;
            # the build-time alternative, form 1, is ((sub * rsub) >> 8) & 0xFFFFFF)
            h24 = np.zeros((8, len(strand)), dtype=np.uint64)
            for q in range(8):
                sub = (strand >> np.uint64(2 * (7 - q))) & np.uint64(0xFFFFFF)
                rsub = (srev >> np.uint64(2 * q)) & np.uint64(0xFFFFFF)
                h24[q] = (np.minimum(sub, rsub) * np.uint64(0x9E3779)) & np.uint64(0xFFFFFF)
            for align in range(16):                          # position mod 16 of the window's last base
                # the 12-mer at start offset q ends 7 - q positions before the window's end
                keys = np.stack([(h24[q] << np.uint64(8)) | np.uint64((align - (7 - q)) % 16) for q in range(8)])
                win = np.argmin(keys, axis=0)                # the device's sliding minimum
                mz = keys[win, np.arange(len(strand))]
                o_right = (np.uint64(align) - (mz & np.uint64(15))) & np.uint64(15)
                assert np.array_equal(o_right, (7 - win).astype(np.uint64))
                blk = ((((mz >> np.uint64(8)) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) * nb) >> np.uint64(32)
                q = win.astype(np.uint64)
                # class R (four bases right of M inside the k-mer): bases q .. q+15; class L: bases q-4 .. q+11
                shift = np.where(q <= 3, np.uint64(2) * (np.uint64(3) - np.minimum(q, np.uint64(3))), np.uint64(2) * (np.uint64(7) - q))
                E = ((strand >> shift) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
                u = (E + rc16(E)).astype(np.uint32)
                um = (u.astype(np.uint64) * np.uint64(0x9E3779B1) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
                b = blocks[blk.astype(np.int64)]
                ok = ((b[:, 0] >> (np.uint32(31) - ((u >> np.uint32(24)) & np.uint32(31)))) & (b[:, 1] >> (np.uint32(31) - ((um >> np.uint32(24)) & np.uint32(31)))) &
                      (b[:, 2] >> (np.uint32(31) - ((um >> np.uint32(16)) & np.uint32(31)))) & (b[:, 3] >> (np.uint32(31) - ((um >> np.uint32(8)) & np.uint32(31)))) & np.uint32(1))
                assert ok.all(), (kib, align, int((~ok.astype(bool)).sum()))
    # and the filter is a filter: random 19-mers mostly fail at the automatic size
    blocks = debug_run_filter(canon, 0)
    assert (blocks != 0).any() and (np.unpackbits(blocks.view(np.uint8)).mean() < 0.5)


def _hash64_np(x, k):
    """vendor/KseqHashIterator.hpp:129-139 on a uint64 array (checked against ntsm_hash64 below)."""
    m = np.uint64((1 << (2 * k)) - 1)
    u = np.uint64
    x = x.astype(np.uint64)
    x = (~x + (x << u(21))) & m
    x ^= x >> u(24)
    x = (x + (x << u(3)) + (x << u(8))) & m
    x ^= x >> u(14)
    x = (x + (x << u(2)) + (x << u(4))) & m
    x ^= x >> u(28)
    x = (x + (x << u(31))) & m
    return x


def test_kernel_form_choice_is_a_property_of_the_key_set(nt, tmp_path):
    """VERDICT round 5 weak #4 / ADVICE: the automatic choice between the minimizer-blocked and the run-anchored kernel used to look at
    CONSECUTIVE keys, so the same set handed over in m_counts' iteration order (a robin_map: hash order, src/FingerPrint.hpp:466 --
    what the reference-side binding of INTEGRATION.md does) silently lost the run form.  The estimate is now taken on a
    minimizer-residue sample of the set: site-file order, shuffled, sorted, and hash order with NTSM_KEYS_HASH64 all answer "run"
    for the 2.5 M-key n10_full set; unrelated k-mers, sets outside the size window and other k do not."""
    from ntsm_amd.capi import debug_form_choice
    sp = str(tmp_path / "n10_full.fa")
    nt.SynthShort(sites_seed=20241218, n_sites=96287, read_seed=77, sites_path=sp, min_keep=13)
    keys = nt.Sites(sp).keys
    assert 2_400_000 < len(keys) < 2_600_000
    rng = np.random.default_rng(3)
    shuffled = keys.copy()
    rng.shuffle(shuffled)
    hv = _hash64_np(keys, 19)
    assert all(int(hv[i]) == nt.hash64(int(keys[i]), 19) for i in range(0, len(keys), 50_021))
    bucket_order = np.argsort(hv & np.uint64((1 << 23) - 1), kind="stable")     # robin_map: bucket = hash & (2^23 - 1) for 2.5 M keys
    assert debug_form_choice(keys) == "run"
    assert debug_form_choice(shuffled) == "run"
    assert debug_form_choice(np.sort(keys)) == "run"
    assert debug_form_choice(keys[bucket_order]) == "run"
    assert debug_form_choice(hv[bucket_order], key_kind=1) == "run"              # exactly what gpuInit of INTEGRATION.md passes
    assert debug_form_choice(keys, k=21) in ("one_level", "two_level")           # the run-anchored kernel exists for k = 19
    # a subset below the window, and the bench's own set (3 .. 13 k-mers kept, 1.54 M keys)
    assert debug_form_choice(keys[:1_700_000]) == "one_level"
    # unrelated k-mers inside the window: one (minimizer, signature) pair per key
    codes = np.unique(rng.integers(0, 1 << 38, size=2_050_000, dtype=np.int64).astype(np.uint64))[:2_000_000]
    assert debug_form_choice(codes) == "one_level"
    # half clustered, half unrelated, in the window: 0.55 * 0.5 + 1.0 * 0.5 < 0.8 -> still pays
    mixed = np.unique(np.concatenate([keys[:1_250_000], codes[:1_250_000]]))
    rng.shuffle(mixed)
    assert debug_form_choice(mixed) == "run"
    assert debug_form_choice(np.unique(rng.integers(0, 1 << 38, size=10_000_000, dtype=np.int64).astype(np.uint64))) == "two_level"
    assert debug_form_choice(keys[:1000], k=12) == "generic"


def test_early_ingest_packs_the_same_reads(nt, tmp_path):
    """early_ingest.hpp: the first input file parsed into packed chunks in ordinary memory while the sites load.  The chunks
    of a plain FASTQ and of the same reads as .gz hold the same reads as the sequential reader delivers: same number of reads
    and bases, and the same multiset of maximal runs of valid bases (every read is followed by at least one invalid position,
    so runs never join across reads) -- for strict files, a file with a wrapped record (parallel prefix + sequential rest),
    FASTA, small chunk budgets (parsers wait for the consumers) and a read longer than a chunk."""
    import collections
    import re
    from ntsm_amd.capi import early_ingest, flatten_file, gunzip_parallel_chunk
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=5, p_embed=0.2)
    fq = str(tmp_path / "t.fq")
    s.write_fastq(fq, 0, 40000)
    raw = open(fq, "rb").read()
    lines = raw.split(b"\n")
    w = list(lines)
    w[80001] = w[80001][:50] + b"\n" + w[80001][50:]
    long_read = b"@long\n" + b"ACGTTGCA" * 40000 + b"\n+\n" + b"I" * 320000 + b"\n"
    cases = {"strict.fq": raw, "wrapped.fq": b"\n".join(w),
             "longread.fq": b"\n".join(lines[:36000]) + b"\n" + long_read + b"\n".join(lines[36000:38000]) + b"\n"}
    tb = bytearray(b"N" * 256)
    for letters, code in ((b"Aa\x00", b"A"), (b"Cc\x01", b"C"), (b"Gg\x02", b"G"), (b"TtUu\x03", b"T")):
        for ch in letters:
            tb[ch] = code[0]
    table = bytes(tb)

    def runs_of_reads(bases, ends):
        out, st, bb = collections.Counter(), 0, bases.tobytes().translate(table)
        for x in ends.tolist():
            out.update(r for r in bb[st:x].split(b"N") if r)
            st = x + 1
        return out
    try:
        gunzip_parallel_chunk(30000)
        for name, data in cases.items():
            for gz in (False, True):
                p = str(tmp_path / (name + (".gz" if gz else "")))
                open(p, "wb").write(_gz_member(data, 6) if gz else data)
                ref_b, ref_e, _ = flatten_file(p)
                want = runs_of_reads(ref_b, ref_e)
                for par, dec, block, chunk, budget, cons in ((4, 4, 1 << 20, 1 << 20, 64, 2), (3, 2, 200_000, 40_000, 6, 1), (6, 5, 1 << 20, 300_000, 12, 3)):
                    r = early_ingest(p, par, dec, block, chunk, budget, cons)
                    assert r is not None, (name, gz)
                    text, n_reads, n_bases, n_par = r
                    assert n_reads == len(ref_e) and n_bases == int(ref_e[-1]) + 1 - len(ref_e), (name, gz, par, n_reads, len(ref_e))
                    assert collections.Counter(x for x in re.split(b"N+", text) if x) == want, (name, gz, par, dec, block, chunk)
                    if name == "strict.fq":
                        assert n_par == n_reads
                    if name == "wrapped.fq":
                        assert 0 < n_par < n_reads
                # the consumers ask for the stream in the middle (EarlyIngest::hand_over): what went through the chunks plus
                # what the stream still held is the same file; at 0 chunks nearly everything comes the second way
                for after in ((0, 3, 40) if gz else (3,)):
                    text2, n_reads2, n_bases2, _, n_rest = early_ingest(p, 4, 3, 1 << 20, 60_000, 8, 2, hand_over_after=after)
                    assert n_reads2 == len(ref_e) and n_bases2 == int(ref_e[-1]) + 1 - len(ref_e), (name, gz, after, n_reads2, len(ref_e))
                    assert collections.Counter(x for x in re.split(b"N+", text2) if x) == want, (name, gz, after)
                    assert n_rest == 0 if not gz else (n_rest > 0 or after > 0), (name, gz, after, n_rest)
        fa = str(tmp_path / "x.fa")
        open(fa, "wb").write(b"".join(b">s%d\nACGTACGTAGCTAGCTAGCTAGCATCGAT\n" % i for i in range(100000)))
        assert early_ingest(fa) is None                                  # FASTA: not for this path (the ordinary one reads it)
        assert early_ingest(str(tmp_path / "absent.fq")) is None
    finally:
        gunzip_parallel_chunk(0)


def test_gzip_reader_paths_agree(nt, tmp_path, monkeypatch):
    """SeqReader over the decoder thread == SeqReader over zlib (NTSM_ZLIB_ONLY) on every gzip input of the
    golden set and on a multi-member FASTQ."""
    import glob
    gz_inputs = sorted(glob.glob(os.path.join(G, "inputs", "*.gz")))
    assert gz_inputs
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=9, p_embed=0.2)
    fq = str(tmp_path / "m.fq")
    s.write_fastq(fq, 0, 20000)
    raw = open(fq, "rb").read()
    multi = str(tmp_path / "multi.fq.gz")
    open(multi, "wb").write(_gz_member(raw[:1_000_003], 1) + _gz_member(raw[1_000_003:4_000_000], 9) + _gz_member(raw[4_000_000:], 6))
    for path in gz_inputs + [multi]:
        monkeypatch.delenv("NTSM_ZLIB_ONLY", raising=False)
        a = nt.flatten_file(path)
        monkeypatch.setenv("NTSM_ZLIB_ONLY", "1")
        b = nt.flatten_file(path)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], path
    monkeypatch.delenv("NTSM_ZLIB_ONLY", raising=False)
    plain = nt.flatten_file(fq)
    got = nt.flatten_file(multi)
    assert np.array_equal(plain[0], got[0]) and np.array_equal(plain[1], got[1])


def test_bgzf_parallel_inflate_under_tsan(nt, tmp_path):
    """Dispatcher / worker / reader threads of the BGZF path under ThreadSanitizer (good file, truncated, corrupt)."""
    import random
    import subprocess
    exe = str(tmp_path / "gunzip_tsan")
    host = os.path.join(ROOT, "ntsm_amd", "csrc", "host")
    srcs = [os.path.join(ROOT, "tools", "gunzip_sanitize.cpp")] + [os.path.join(host, f) for f in ("gz_stream.cpp", "gz_parallel.cpp", "inflate.cpp", "inflate_spec.cpp", "crc32_fast.cpp")]
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-o", exe] + srcs + ["-lz", "-lpthread"], check=True)
    rng = random.Random(5)
    data = bytes(rng.choice(b"ACGTN\n@+F") for _ in range(3_000_000))
    good = _bgzf(data, level=1)
    bad = bytearray(good)
    bad[len(good) // 2] ^= 0x10
    files = []
    for name, blob in (("good", good), ("trunc", good[:len(good) * 2 // 3]), ("bad", bytes(bad))):
        path = str(tmp_path / (name + ".gz"))
        open(path, "wb").write(blob)
        files.append(path)
    p = subprocess.run([exe] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1", NTSM_DECODER_THREADS="6"))
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = p.stdout.split()
    assert out[0:2] == [b"%d" % len(data), b"0"] and out[3] == b"0" and out[5] == b"-1", out


def test_gzip_decoder_corrupt_streams_under_asan(nt, tmp_path):
    """Memory safety of the decoder on hostile input: byte-level mutations, spliced and truncated streams, run under
    AddressSanitizer + UBSan (the decoder must fail or finish, never read or write out of bounds)."""
    import random
    import subprocess
    import zlib
    exe = str(tmp_path / "gunzip_sanitize")
    host = os.path.join(ROOT, "ntsm_amd", "csrc", "host")
    srcs = [os.path.join(ROOT, "tools", "gunzip_sanitize.cpp")] + [os.path.join(host, f) for f in ("gz_stream.cpp", "gz_parallel.cpp", "inflate.cpp", "inflate_spec.cpp", "crc32_fast.cpp")]
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe] + srcs + ["-lz", "-lpthread"], check=True)
    rng = random.Random(7)
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGT") for _ in range(100)), b"F" * 100) for i in range(3000))
    seeds = [_gz_member(text, 6), _gz_member(text, 1, zlib.Z_FIXED), _gz_member(text, 9, zlib.Z_DEFAULT_STRATEGY, 9, 1),
             _gz_member(bytes(rng.getrandbits(8) for _ in range(70000)), 0), _gz_member(text[:50000], 6) + _gz_member(text[50000:], 6)]
    files = []
    for k in range(400):
        b = bytearray(rng.choice(seeds))
        kind = k % 4
        if kind == 0:                                               # a few random bytes replaced (often inside the code tables)
            for _ in range(rng.randrange(1, 6)):
                b[rng.randrange(10, len(b))] = rng.getrandbits(8)
        elif kind == 1:                                             # truncated, random tail appended
            b = b[:rng.randrange(10, len(b))] + bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 300)))
        elif kind == 2:                                             # spliced: head of one stream, middle of another
            o = rng.choice(seeds)
            b = b[:rng.randrange(10, len(b))] + o[rng.randrange(10, len(o)):]
        else:                                                       # header followed by noise
            b = b[:10] + bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 5000)))
        path = str(tmp_path / ("c%03d.gz" % k))
        open(path, "wb").write(bytes(b))
        files.append(path)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    for i in range(0, len(files), 100):
        p = subprocess.run([exe] + files[i:i + 100], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        assert len(p.stdout.split(b"\n")) == 101


def _bgzf(data, level=6, eof=True, blk=65280):
    """BGZF (bgzip) container: independent gzip members of <= 64 KiB with a BC extra field (SAM spec 4.1)."""
    import struct
    import zlib

    def block(d):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(d) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + body +
                struct.pack("<II", zlib.crc32(d), len(d)))
    return b"".join(block(data[i:i + blk]) for i in range(0, len(data), blk)) + (block(b"") if eof else b"")


def test_bgzf_block_parallel_inflate(nt, tmp_path):
    """BGZF input decoded by several decoder threads == zlib's gzread on the same file: whole files, no EOF marker,
    tiny blocks, plain gzip members before/after, trailing garbage, truncation, a corrupt block, a wrong BSIZE."""
    import random
    from ntsm_amd.capi import gunzip
    rng = random.Random(3)
    data = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGT") for _ in range(150)),
                                            bytes(rng.choice(b"FFFF:,#") for _ in range(150))) for i in range(20000))
    p = str(tmp_path / "t.bgzf.gz")
    open(p, "wb").write(_bgzf(data))
    for eng in (0, 1, 2, 5, 16):
        assert gunzip(p, eng, 1 << 20) == (data, 0), eng
    good = _bgzf(data)
    cut = len(data) // 3
    cases = {"noeof": _bgzf(data, eof=False), "small": _bgzf(data[:200000], blk=1000), "plain_after": _bgzf(data[:cut]) + _gz_member(data[cut:]),
             "plain_before": _gz_member(data[:cut]) + _bgzf(data[cut:]), "garbage": good + b"trailing garbage", "empty": _bgzf(b""),
             "trunc1": good[:len(good) // 2], "trunc2": good[:len(good) - 30], "trunc3": good[:777]}
    bad = bytearray(good)
    bad[len(good) // 3] ^= 0x40
    cases["flip"] = bytes(bad)
    bad = bytearray(good)
    bad[16] ^= 1
    cases["bsize"] = bytes(bad)
    for name, blob in cases.items():
        open(p, "wb").write(blob)
        ref = gunzip(p, 1)
        for eng in (0, 3, 8):
            got = gunzip(p, eng, 1 << 16)
            assert got[1] == ref[1] and (got[0] == ref[0] or ref[1] == -1), (name, eng, got[1], ref[1], len(got[0]), len(ref[0]))
    assert gunzip(p, 8)[1] == 0                                     # "bsize": falls back to the sequential decoder, still fine
    # through the reader: parallel BGZF == plain text
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=4, p_embed=0.2)
    fq = str(tmp_path / "b.fq")
    s.write_fastq(fq, 0, 15000)
    open(p, "wb").write(_bgzf(open(fq, "rb").read(), level=1))
    plain = nt.flatten_file(fq)
    nt.capi.HO.ntsm_host_gunzip                                    # (decoder thread count is process-wide: set through the hook)
    got = nt.flatten_file(p)
    assert np.array_equal(plain[0], got[0]) and np.array_equal(plain[1], got[1])


def test_parallel_site_loading_equals_sequential(nt, tmp_path):
    """Site files that are plain two-line FASTA are k-merised by several threads (site_set.cpp); keys, allele lists,
    IDs and the order of the collision warnings must equal the sequential kseq-equivalent load; anything else (CRLF,
    wrapped sequence, gzip, FASTQ, junk first) takes the sequential path and still agrees."""
    import gzip
    import subprocess
    sp = str(tmp_path / "s.fa")
    nt.SynthShort(sites_seed=3, n_sites=8000, read_seed=1, sites_path=sp)      # ~2.8 MB, 16000 records
    raw = open(sp, "rb").read()
    assert len(raw) > (1 << 20)
    lines = raw.split(b"\n")
    dup = b"\n".join(lines[:400]) + b"\n"                           # repeat 200 records at the end: duplicate k-mers, warnings
    variants = {"plain.fa": raw, "dups.fa": raw + dup, "crlf.fa": raw.replace(b"\n", b"\r\n"),
                "wrapped.fa": b"\n".join(l if i % 2 == 0 or len(l) < 30 else l[:25] + b"\n" + l[25:] for i, l in enumerate(lines)),
                "junk.fa": b"# comment\n" + raw, "noeol.fa": raw[:-1], "empty_seq.fa": raw + b">x\n\n>y\nACGTACGTACGTACGTACGTACGT\n"}
    # one subprocess per file (the warnings go to the process's stderr): four loads, markers in between
    code = ("import os, sys; sys.path.insert(0, %r); import numpy as np, ntsm_amd, hashlib\n"
            "for dupes in (False, True):\n"
            "    for seq in (False, True):\n"
            "        os.environ.pop('NTSM_SITES_SEQUENTIAL', None)\n"
            "        if seq: os.environ['NTSM_SITES_SEQUENTIAL'] = '1'\n"
            "        s = ntsm_amd.Sites(sys.argv[1], allow_dupes=dupes)\n"
            "        c = (np.arange(len(s.keys), dtype=np.uint64) * 7 + 3) %% 41; rc, txt = s.format_counts(c, 12345)\n"
            "        print(len(s.keys), s.n_sites, s.n_erased, rc, hashlib.sha256(s.keys.tobytes()).hexdigest(), hashlib.sha256(txt).hexdigest(), flush=True)\n"
            "        sys.stderr.write('==MARK==\\n'); sys.stderr.flush()\n") % ROOT
    for name, data in variants.items():
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        r = subprocess.run([sys.executable, "-c", code, p], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-1000:]
        res = r.stdout.split(b"\n")[:4]
        warn = [[l for l in part.split(b"\n") if l.startswith(b"Warning")] for part in r.stderr.split(b"==MARK==\n")[:4]]
        assert res[0] == res[1] and res[2] == res[3] and warn[0] == warn[1] and warn[2] == warn[3], name
        if name == "dups.fa":
            assert len(warn[0]) > 1000 and warn[0] == warn[2]       # the collision warnings are there, in the same order
    gz = str(tmp_path / "s.fa.gz")
    with gzip.open(gz, "wb", compresslevel=1) as f:
        f.write(raw)
    a, b = nt.Sites(sp), nt.Sites(gz)
    assert np.array_equal(a.keys, b.keys) and a.n_sites == b.n_sites
    # the grouping of equal k-mers and the allele lists are built on several threads at this size: against the oracle's
    # loader (one hash table, stream order) on the file with the repeated records, and what both print for the same reads
    for name, dupes in (("plain.fa", False), ("dups.fa", True)):
        p = str(tmp_path / name)
        sites, fp = nt.Sites(p, allow_dupes=dupes), OracleFP(p, k=19, dupes=dupes)
        assert sites.n_sites == fp.n_sites and len(sites.keys) == fp.n_distinct
        for i in range(0, 3000, 7):
            fp.process(raw.split(b"\n", 2 * i + 2)[1 + 2 * i].replace(b"N", b"A"))
        canon, _, cnt = fp.kmers()
        assert np.array_equal(sites.keys, canon)
        rc, text = sites.format_counts(cnt, fp.total_kmers)
        assert (rc, text) == fp.print_counts()


def test_reader_on_a_fifo(nt, tmp_path):
    """Input through a named pipe (e.g. `ntsmCount -s sites.fa <(zcat x.fq.gz)`): the file is opened exactly once --
    the gzip / block-parallel eligibility checks only stat() it -- and parses like the regular file, plain and gzip."""
    import gzip
    import threading
    s = nt.SynthShort(sites_seed=11, n_sites=200, read_seed=8, p_embed=0.2)
    fq = str(tmp_path / "f.fq")
    s.write_fastq(fq, 0, 4000)
    raw = open(fq, "rb").read()
    want = nt.flatten_file(fq)
    for blob in (raw, gzip.compress(raw, 1)):
        fifo = str(tmp_path / "pipe")
        if os.path.exists(fifo):
            os.unlink(fifo)
        os.mkfifo(fifo)
        opened = []

        def writer():
            with open(fifo, "wb") as f:                      # blocks until the reader opens; a second open would hang or break it
                opened.append(1)
                f.write(blob)
        th = threading.Thread(target=writer)
        th.start()
        got = nt.flatten_file(fifo)
        th.join(timeout=30)
        assert not th.is_alive() and opened == [1]
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_minimizer_plan_is_strand_symmetric_for_every_k(tmp_path):
    """tools/plan_check.cpp: the functions shared by the filter builder and the kernel (ntsm_device.h) for every k the
    minimizer-blocked kernel takes (13..31, 12-mer minimizers) and for its two-level form (15..31, 14-mer minimizers + Bloom
    word): candidate offsets symmetric inside the k-mer, and minimizer, filter-bit hash, block index, Bloom word and Bloom
    bits equal for a k-mer and its reverse complement (the reference counts canonical k-mers, vendor/KseqHashIterator.hpp:87-93;
    a strand-dependent filter address would drop one strand of a site k-mer)."""
    exe = str(tmp_path / "plan_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "plan_check.cpp")], check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE)
    assert p.returncode == 0 and b"plans: 36, problems: 0" in p.stdout, p.stdout.decode()


def test_pack2_packer_matches_the_byte_table(nt):
    """Host packer of the packed producer lanes (ntsm_amd/csrc/host/pack2.hpp): 2-bit code + validity bit per position must
    be exactly the class the reference's byte table gives the byte (vendor/KseqHashIterator.hpp:114-127: A a 0x00 | C c
    0x01 | G g 0x02 | T t U u 0x03 | invalid), for every byte value, read lengths 0..200, both implementations; every
    read starts at a multiple of 8 and is followed by 1..8 invalid positions; nothing before the batch start is touched."""
    import ctypes as C
    from ntsm_amd.capi import HO, u8p
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import lib
    L = lib()
    table = np.array([L.ntsm_oracle_nt4(b) for b in range(256)], dtype=np.uint8)     # the oracle's copy of the reference table
    assert sorted(np.flatnonzero(table < 4).tolist()) == sorted([0, 1, 2, 3] + list(b"ACGTUacgtu"))
    rng = np.random.default_rng(3)
    assert HO.ntsm_host_pack2_impl() in (b"avx512vbmi", b"avx2", b"scalar")
    for force in (0, 1, 2):                               # best available (AVX-512 VBMI / AVX2), portable, at most AVX2
        for trial in range(120):
            reads = []
            for _ in range(int(rng.integers(1, 7))):
                n = int(rng.integers(0, 201))
                mode = int(rng.integers(0, 3))
                if mode == 0:
                    r = rng.integers(0, 256, n, dtype=np.uint8)
                elif mode == 1:
                    r = np.frombuffer(b"ACGTacgtUuNn\x00\x01\x02\x03", np.uint8)[rng.integers(0, 16, n)]
                else:
                    r = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)]
                reads.append(r.astype(np.uint8).copy())
            if trial == 0:
                reads = [np.arange(256, dtype=np.uint8)]                                # every byte value once
            starts, pos = [], 0
            for r in reads:
                starts.append(pos)
                pos = (pos + len(r) + 8) & ~7
            cap = ((pos + 63) // 32) * 32 + 64
            cls = np.full(cap, 4, np.uint8)
            for st, r in zip(starts, reads):
                cls[st:st + len(r)] = table[r]
            want_valid = np.packbits((cls < 4).astype(np.uint8), bitorder="little")
            c4 = np.where(cls < 4, cls, 0).astype(np.uint8).reshape(-1, 4)
            want_codes = (c4[:, 0] | (c4[:, 1] << 2) | (c4[:, 2] << 4) | (c4[:, 3] << 6)).astype(np.uint8)
            codes, valid = np.full(cap // 4, 0xAA, np.uint8), np.full(cap // 8, 0x55, np.uint8)
            at = 0
            for r in reads:
                buf = r if len(r) else np.zeros(1, np.uint8)
                at = HO.ntsm_host_pack2_append(codes.ctypes.data_as(u8p), valid.ctypes.data_as(u8p), at, buf.ctypes.data_as(u8p), len(r), force)
            assert at == pos
            assert np.array_equal(codes[:pos // 4], want_codes[:pos // 4]) and np.array_equal(valid[:pos // 8], want_valid[:pos // 8]), (force, trial)


def test_pack2_packer_under_sanitizers(tmp_path):
    """tools/pack_sanitize.cpp under ASan + UBSan: the packer never writes past pack2_extent() nor reads past the read."""
    exe = str(tmp_path / "pack_sanitize")
    host = os.path.join(ROOT, "ntsm_amd", "csrc", "host")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", host, "-o", exe,
                    os.path.join(ROOT, "tools", "pack_sanitize.cpp"), os.path.join(host, "pack2.cpp")], check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"pack2 sanitize ok" in p.stdout, p.stderr.decode()[-2000:]
