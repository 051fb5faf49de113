"""ntsmEval all-pairs scoring: oracle/ntsm_eval_oracle.c, include/ntsm_eval_hip.h,
build/ntsmEval.

PARITY WITH THE REFERENCE IS UNPINNED: src/CompareCounts.hpp cannot be compiled in this image (it includes
vendor/kfunc.c, which needs autoconf's config.h) and the reference holds no fixtures for this path.  What these tests
pin is (CPU) the oracle against an independent statement of the formulas written here from the reference text and
against hand-checkable cases, and (GPU) the HIP library and the CLI against the oracle, bit for bit."""
import math
import os
import subprocess

import numpy as np
import pytest

from oracle_binding import EvalOracle, ROOT

EVAL = os.path.join(ROOT, "build", "ntsmEval")
ORACLE_CLI = os.path.join(ROOT, "oracle", "ntsm_eval_oracle")


def write_counts(path, counts, sums=None, distinct=None, tk=123456789, ks=19, loci=None, header=True):
    """A counts.txt as ntsmCount prints it (src/FingerPrint.hpp:261-311): tags, column header, one line per site."""
    m = counts.shape[0]
    sums = counts if sums is None else sums
    distinct = np.full((m, 2), 13, dtype=np.uint32) if distinct is None else distinct
    with open(path, "w") as f:
        f.write("#@TK\t%d\n#@KS\t%d\n" % (tk, ks))
        if header:
            f.write("#locusID\tcountAT\tcountCG\tsumAT\tsumCG\tdistinctAT\tdistinctCG\n")
        for s in range(m):
            f.write("%s\t%d\t%d\t%d\t%d\t%d\t%d\n" % (loci[s] if loci else "rs%d" % s, counts[s, 0], counts[s, 1], sums[s, 0], sums[s, 1],
                                                     distinct[s, 0], distinct[s, 1]))


def random_samples(rng, n, m, depth=8.0, related=True):
    """n samples over m sites: genotypes from a few founders so that some pairs are close, counts Poisson around depth."""
    founders = rng.integers(0, 3, size=(max(2, n // 3), m))                  # 0 = hom AT, 1 = het, 2 = hom CG
    out = np.zeros((n, m, 2), dtype=np.uint32)
    for i in range(n):
        g = founders[i % founders.shape[0]].copy() if related else rng.integers(0, 3, size=m)
        flip = rng.random(m) < (0.02 if i < founders.shape[0] else 0.3)
        g[flip] = rng.integers(0, 3, size=int(flip.sum()))
        d = depth * (0.3 + 1.4 * rng.random())
        lam_at = np.where(g == 0, d, np.where(g == 1, d / 2, 0.02))
        lam_cg = np.where(g == 2, d, np.where(g == 1, d / 2, 0.02))
        out[i, :, 0] = rng.poisson(lam_at)
        out[i, :, 1] = rng.poisson(lam_cg)
    return out


def py_pair(a, b, c):
    """The reference's per-pair quantities restated independently of the C oracle (src/CompareCounts.hpp:968-989,
    :1013-1033, :1057-1078, :1144-1196): Python floats are IEEE doubles, evaluated left to right without contraction."""
    joint = s1 = s2 = 0.0
    n = hets1 = homs1 = hets2 = homs2 = sh_het = sh_hom = ibs0 = 0
    for (a0, a1), (b0, b1) in zip(a.tolist(), b.tolist()):
        if (a0 <= c and a1 <= c) or (b0 <= c and b1 <= c):
            continue
        n += 1
        cat, ccg = a0 + b0, a1 + b1
        fat = cat / (cat + ccg) if cat > c else 0.0
        fcg = ccg / (cat + ccg) if ccg > c else 0.0
        joint += cat * fat + ccg * fcg
        s1 += a0 * (a0 / (a0 + a1) if a0 > c else 0.0) + a1 * (a1 / (a0 + a1) if a1 > c else 0.0)
        s2 += b0 * (b0 / (b0 + b1) if b0 > c else 0.0) + b1 * (b1 / (b0 + b1) if b1 > c else 0.0)
        t1 = "het" if a0 > c and a1 > c else ("at" if a0 > c else "cg")
        t2 = "het" if b0 > c and b1 > c else ("at" if b0 > c else "cg")
        hets1 += t1 == "het"; homs1 += t1 != "het"; hets2 += t2 == "het"; homs2 += t2 != "het"
        if t1 == "het" and t2 == "het":
            sh_het += 1
        elif t1 != "het" and t2 != "het":
            if t1 == t2:
                sh_hom += 1
            else:
                ibs0 += 1
    return dict(sum_joint=joint, sum_single1=s1, sum_single2=s2, n_valid=n, hets1=hets1, homs1=homs1, hets2=hets2, homs2=homs2,
                shared_hets=sh_het, shared_homs=sh_hom, ibs0=ibs0, ibs2=sh_het + sh_hom)


FIELDS = ("sum_joint", "sum_single1", "sum_single2", "n_valid", "hets1", "homs1", "hets2", "homs2", "shared_hets", "shared_homs", "ibs0", "ibs2")


def same_bits(x, y):
    return np.float64(x).tobytes() == np.float64(y).tobytes()


def files_for(tmp_path, samples, **kw):
    paths = []
    for i in range(samples.shape[0]):
        p = str(tmp_path / ("s%03d.txt" % i))
        write_counts(p, samples[i], **kw)
        paths.append(p)
    return paths


# ---------------------------------------------------------------------------------------------------- CPU
def test_oracle_parses_counts_files_like_the_reference_reader(tmp_path):
    """CompareCounts::CompareCounts (:30-114): the first file fixes the loci and the distinct columns; other files are
    matched by locus id in any order; a locus missing from a later file stays 0; tags come from '#@TK' / '#@KS' lines;
    the column-header line and empty lines are skipped."""
    rng = np.random.default_rng(1)
    c = rng.integers(0, 30, size=(2, 6, 2)).astype(np.uint32)
    a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    write_counts(a, c[0], sums=c[0] * 3, distinct=np.arange(12, dtype=np.uint32).reshape(6, 2), tk=1000, ks=19)
    order = [4, 0, 5, 2, 1]                                                  # shuffled, locus rs3 absent
    with open(b, "w") as f:
        f.write("#@KS\t21\n\n#@TK\t777\n#locusID\tcountAT\tcountCG\tsumAT\tsumCG\tdistinctAT\tdistinctCG\n")
        for s in order:
            f.write("rs%d\t%d\t%d\t%d\t%d\t1\t1\n" % (s, c[1, s, 0], c[1, s, 1], 2 * c[1, s, 0], 2 * c[1, s, 1]))
    o = EvalOracle([a, b])
    assert (o.n, o.m) == (2, 6)
    got = o.counts()
    want = c.copy(); want[1, 3] = 0
    assert np.array_equal(got, want)
    L = o.L
    assert (L.ntsm_eval_oracle_raw_total(o.h, 0), L.ntsm_eval_oracle_kmer_size(o.h, 0)) == (1000, 19)
    assert (L.ntsm_eval_oracle_raw_total(o.h, 1), L.ntsm_eval_oracle_kmer_size(o.h, 1)) == (777, 21)
    assert L.ntsm_eval_oracle_total(o.h, 1) == int(want[1].sum())
    assert np.array_equal(np.ctypeslib.as_array(L.ntsm_eval_oracle_distinct(o.h), shape=(6, 2)), np.arange(12).reshape(6, 2))
    assert np.array_equal(np.ctypeslib.as_array(L.ntsm_eval_oracle_sums(o.h), shape=(2, 6, 2))[0], c[0] * 3)
    # computeErrorRate (:1198-1216) against its formula
    sums, dist = int((c[0] * 3).sum()), int(np.arange(12).sum())
    want_err = 1.0 - math.pow(sums / (1000.0 * dist / 6200000000.0), 1.0 / 19.0)
    assert same_bits(L.ntsm_eval_oracle_error_rate(o.h, 0, 6200000000), want_err)
    o.close()


def test_oracle_pairs_match_an_independent_statement_of_the_formulas(tmp_path):
    """Every field of every pair, bit for bit, for several min_cov; plus cases with a known answer: a sample against
    itself at min_cov 0 scores -2 * 0 = -0 (joint frequencies = single frequencies, every term doubles exactly; with
    min_cov > 0 the doubled joint counts pass thresholds the single ones do not), disjoint coverage
    leaves no valid site (score = DBL_MAX), opposite homozygotes are ibs0."""
    rng = np.random.default_rng(2)
    samples = random_samples(rng, 7, 400)
    samples[6] = samples[0]                                                  # an exact duplicate
    samples[5, :200] = 0                                                     # low call rate
    o = EvalOracle(files_for(tmp_path, samples))
    for c in (0, 1, 3):
        for i in range(7):
            for j in range(i + 1, 7):
                r, w = o.pair(i, j, c), py_pair(samples[i], samples[j], c)
                for f in FIELDS:
                    g = getattr(r, f)
                    assert same_bits(g, w[f]) if f.startswith("sum") else g == w[f], (c, i, j, f, g, w[f])
    r = o.pair(0, 6, 0)                                                      # min_cov 0: the joint thresholds are the single ones
    assert r.sum_joint == r.sum_single1 + r.sum_single2 and r.ibs0 == 0 and r.shared_hets == r.hets1 == r.hets2
    score = o.L.ntsm_eval_oracle_score(r, 5.0, 5.0, 0.2)
    assert score == 0.0 and math.copysign(1.0, score) == -1.0                # prints as -0.000000, like -2.0 * 0.0 in the reference
    o.close()
    hand = np.zeros((3, 4, 2), dtype=np.uint32)
    hand[0] = [[9, 0], [0, 9], [5, 5], [0, 0]]
    hand[1] = [[0, 8], [8, 0], [4, 4], [7, 0]]                               # opposite homozygote at sites 0 and 1, het at 2
    hand[2] = [[0, 0], [0, 0], [0, 0], [3, 3]]                               # shares no covered site with sample 0
    o = EvalOracle(files_for(tmp_path, hand))
    r = o.pair(0, 1, 1)
    assert (r.n_valid, r.ibs0, r.ibs2, r.shared_hets, r.shared_homs, r.hets1, r.homs1, r.hets2, r.homs2) == (3, 2, 1, 1, 0, 1, 2, 1, 2)
    assert o.pair(0, 2, 1).n_valid == 0
    assert o.L.ntsm_eval_oracle_score(o.pair(0, 2, 1), 1.0, 1.0, 0.2) == 1.7976931348623157e308
    assert o.genotype(0, 1) == (1, 2, 1) and o.genotype(2, 1) == (1, 0, 3)  # hets, homs, miss (calcHomHetMiss, :742-767)
    o.close()


def test_eval_library_exports_and_cli_without_gpu(tmp_path):
    """The C ABI loads and exports what include/ntsm_eval_hip.h declares; the CLI's paths that need no GPU: flag errors,
    unsupported modes, a missing file (the reference asserts), and the single-file QC table byte for byte against the
    oracle's printer."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "ntsm_amd", "libntsm_eval_hip.so"))
    assert hasattr(lib, "ntsm_eval_pairs")
    hdr = open(os.path.join(ROOT, "include", "ntsm_eval_hip.h")).read()
    assert "int ntsm_eval_pairs(" in hdr and "PARITY" not in hdr or "UNPINNED" in hdr

    def run(*args):
        return subprocess.run([EVAL] + list(args), capture_output=True)
    p = run()
    assert p.returncode == 1 and b"Error: Need Input File" in p.stderr and b"Try '--help'" in p.stderr
    one = str(tmp_path / "one.txt")
    rng = np.random.default_rng(3)
    write_counts(one, rng.integers(0, 20, size=(50, 2)).astype(np.uint32))
    p = run("-s", "abc", one)
    assert p.returncode == 0 and b"Error - Invalid parameter s: abc" in p.stderr and p.stdout == b""   # message + return 0, like ntsmCount's main
    p = run("-p", "rot.tsv", one, one)
    assert p.returncode == 1 and b"not part of this build" in p.stderr
    # merge only (-e FILE -o: mergeCounts, :626-674; no GPU involved): bytes of the merged file against the oracle's
    parts = files_for(tmp_path, random_samples(rng, 3, 40), tk=1000)
    mine, ref = str(tmp_path / "merged.txt"), str(tmp_path / "merged_ref.txt")
    p = run("-e", mine, "-o", *parts)
    q = subprocess.run([ORACLE_CLI, "-e", ref, "-o"] + parts, capture_output=True)
    assert p.returncode == 0 and q.returncode == 0 and p.stdout == b"" and q.stdout == b""
    assert open(mine, "rb").read() == open(ref, "rb").read() and open(mine).readline() == "#@TK\t3000\n"
    p = run("-o", *parts)
    assert p.returncode == 1 and b"cannot be used without --merge" in p.stderr
    p = run(str(tmp_path / "nope.txt"))
    assert p.returncode < 0 or p.returncode == 134                           # abort()
    p = run("-c", "2", "-g", "3100000000", one)
    q = subprocess.run([ORACLE_CLI, "-c", "2", "-g", "3100000000", one], capture_output=True)
    assert p.returncode == 0 and q.returncode == 0 and p.stdout == q.stdout and p.stdout.startswith(b"sample\tcov\terrorRate\tmiss\thom\thet\n")
    assert not p.stdout.endswith(b"\n")                                      # computeScoreSingle prints the row without a newline (:579-582)


def readme_pairs(tmp_path):
    """Two counts files per row of tests/golden/eval_readme_rows.json whose genotype categories reproduce the row's tallies:
    het = (10, 10), hom AT = (20, 0), hom CG = (0, 20), missing = (0, 0) over 96287 sites."""
    import json
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "eval_readme_rows.json")))
    out = []
    for k, row in enumerate(doc["rows"]):
        r = dict(zip(doc["columns"], row))
        HH, Hh, hH, SS, OO = r["sharedHet"], r["het1"] - r["sharedHet"], r["het2"] - r["sharedHet"], r["sharedHom"], r["ibs0"]
        assert r["hom1"] == hH + SS + OO and r["hom2"] == Hh + SS + OO and r["n"] == HH + Hh + hH + SS + OO and r["ibs2"] == HH + SS
        e_hom1, e_het1 = r["allHom1"] - r["hom1"], r["allHet1"] - r["het1"]     # sample 1 covered, sample 2 missing
        e_hom2, e_het2 = r["allHom2"] - r["hom2"], r["allHet2"] - r["het2"]
        both = 96287 - r["n"] - (e_hom1 + e_het1 + e_hom2 + e_het2)
        assert both >= 0 and r["miss1"] == e_hom2 + e_het2 + both and r["miss2"] == e_hom1 + e_het1 + both
        het, at, cg, no = (10, 10), (20, 0), (0, 20), (0, 0)
        cats = [(HH, het, het), (Hh, het, at), (hH, cg, het), (SS, at, at), (OO, at, cg), (e_hom1, cg, no), (e_het1, het, no),
                (e_hom2, no, at), (e_het2, no, het), (both, no, no)]
        a = np.concatenate([np.tile(np.array(x, dtype=np.uint32), (c, 1)) for c, x, _ in cats if c])
        b = np.concatenate([np.tile(np.array(y, dtype=np.uint32), (c, 1)) for c, _, y in cats if c])
        assert a.shape == (96287, 2)
        fa, fb = str(tmp_path / ("readme%d_a.txt" % k)), str(tmp_path / ("readme%d_b.txt" % k))
        write_counts(fa, a); write_counts(fb, b)
        out.append((fa, fb, r))
    return out


def check_against_readme(stdout, r):
    lines = stdout.decode().split("\n")
    head, vals = lines[0].split("\t"), lines[1].split("\t")
    got = dict(zip(head, vals))
    for col, key in (("relate", "relate"), ("homConcord", "homConcord")):
        assert round(float(got[col]), 6) == float(r[key]), (col, got[col], r[key])
    for col, key in (("ibs0", "ibs0"), ("ibs2", "ibs2"), ("het1", "het1"), ("het2", "het2"), ("sharedHet", "sharedHet"), ("hom1", "hom1"), ("hom2", "hom2"),
                     ("sharedHom", "sharedHom"), ("n", "n"), ("miss1", "miss1"), ("miss2", "miss2"), ("allHom1", "allHom1"), ("allHom2", "allHom2"),
                     ("allHet1", "allHet1"), ("allHet2", "allHet2")):
        assert int(got[col]) == r[key], (col, got[col], r[key])


def test_oracle_reproduces_the_readme_example_rows(tmp_path):
    """The one recorded output the reference holds for ntsmEval: the six example rows of its README (README.md:143-150,
    tests/golden/eval_readme_rows.json).  Their input files are not in the checkout, but the tallies of a row determine the
    genotype categories of the two samples, so inputs with exactly those categories must give back the row's tallies, its
    'relate' and 'homConcord' (to the printed six decimals), missing / hom / het totals and n.  This pins calcRelatedness,
    calcHomHetMiss, gatherValidEntries and the two ratios to the reference's own numbers; the score column stays unpinned."""
    for fa, fb, r in readme_pairs(tmp_path):
        q = subprocess.run([ORACLE_CLI, "-a", fa, fb], capture_output=True)
        assert q.returncode == 0
        check_against_readme(q.stdout, r)


# ---------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hip_pairs_equal_oracle_bit_for_bit(tmp_path):
    """The HIP library against the oracle on the same counts: every record field of every pair, doubles compared by
    their bits.  Shapes around the tile sizes (256 lanes of j, 4 rows of i), min_cov 0 / 1 / 3, empty samples."""
    import ntsm_amd.eval as ev
    rng = np.random.default_rng(11)
    for n, m, c in ((2, 1, 1), (3, 7, 0), (5, 300, 1), (65, 1000, 1), (257, 200, 3), (300, 500, 1), (40, 96287, 1)):
        samples = random_samples(rng, n, m)
        if n > 4:
            samples[3] = 0                                                   # a sample without any coverage
            samples[4] = samples[1]
        rec, ms = ev.pairs(samples, min_cov=c)
        o = EvalOracle(files_for(tmp_path, samples)) if n <= 65 else None
        pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
        if len(pairs) > 400:
            pick = rng.choice(len(pairs), size=400, replace=False)
            pairs = [pairs[k] for k in pick] + [(0, n - 1), (n - 2, n - 1), (3, 4), (255 % n, (256 % n) or 1)]
            pairs = [(min(a, b), max(a, b)) for a, b in pairs if a != b]
        for i, j in pairs:
            g = rec[ev.pair_index(i, j, n)]
            w = py_pair(samples[i], samples[j], c) if o is None or m > 2000 else None
            if w is None:
                r = o.pair(i, j, c)
                w = {f: getattr(r, f) for f in FIELDS}
            for f in FIELDS:
                assert same_bits(g[f], w[f]) if f.startswith("sum") else int(g[f]) == int(w[f]), (n, m, c, i, j, f, g[f], w[f])
        if o:
            o.close()


@pytest.mark.gpu
def test_cli_equals_oracle_cli_bytes(tmp_path):
    """build/ntsmEval against the oracle's printer: stdout byte for byte, default threshold, -a, other -s / -w / -c / -g.
    The last case runs the whole tool chain: counts files printed by build/ntsmCount for reads of related and unrelated
    'individuals' (same sites, different read seeds and embed rates) are scored."""
    import ntsm_amd as nt
    rng = np.random.default_rng(12)
    files = files_for(tmp_path, random_samples(rng, 12, 3000))
    for args in ([], ["-a"], ["-a", "-s", "0.1", "-w", "0", "-c", "2"], ["-s", "5", "-g", "3100000000", "-w", "0.5"]):
        p = subprocess.run([EVAL] + args + files, capture_output=True)
        q = subprocess.run([ORACLE_CLI] + args + files, capture_output=True)
        assert p.returncode == 0 and q.returncode == 0, (args, p.stderr[-300:], q.stderr[-300:])
        assert p.stdout == q.stdout and p.stdout.count(b"\n") >= 1, args
    a = subprocess.run([EVAL, "-a"] + files, capture_output=True).stdout
    assert a.count(b"\n") == 1 + 12 * 11 // 2
    for fa, fb, r in readme_pairs(tmp_path)[:3]:                             # the reference's README example rows through the GPU path
        check_against_readme(subprocess.run([EVAL, "-a", fa, fb], capture_output=True).stdout, r)
    # counts files from the counting CLI itself
    sp = str(tmp_path / "sites.fa")
    s = nt.SynthShort(sites_seed=5, n_sites=2000, read_seed=1, p_embed=0.9, sites_path=sp)
    outs = []
    for i, seed in enumerate((1, 2, 3)):
        fq, out = str(tmp_path / ("r%d.fq" % i)), str(tmp_path / ("c%d.txt" % i))
        nt.SynthShort(sites_seed=5, n_sites=2000, read_seed=seed, p_embed=0.9).write_fastq(fq, 0, 60000)
        with open(out, "wb") as fh:
            subprocess.run([os.path.join(ROOT, "build", "ntsmCount"), "-s", sp, fq], stdout=fh, stderr=subprocess.DEVNULL, check=True)
        outs.append(out)
    p = subprocess.run([EVAL, "-a"] + outs, capture_output=True)
    q = subprocess.run([ORACLE_CLI, "-a"] + outs, capture_output=True)
    assert p.returncode == 0 and p.stdout == q.stdout and p.stdout.count(b"\n") == 4
