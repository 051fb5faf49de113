#!/usr/bin/env python3
"""bench.py -- bases/s through the ntsmCount hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the count path over the whole resident workload (BASELINE.json configs[1]:
1e9 synthetic 150 bp reads against the 96287-site hs_n10_like set, k = 19), inputs already in HBM.

N > 1 (configs[3]): one rank per GPU, every rank owns its own 1e9 reads (weak scaling); each step ends with one
RCCL SUM of the per-k-mer count vector + totals.  The ranks are started either by torch.distributed.run
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or by this script itself: `python bench.py --gpus N` with no WORLD_SIZE spawns N fresh child processes,
one per device, and passes rank 0's JSON line through; the parent never loads torch or HIP (it counts devices in sysfs).  A request for
more GPUs than the box has, or a WORLD_SIZE that differs from --gpus, ends with a non-zero exit status and no JSON
line -- never an N = 1 line for an N > 1 request.

Prints ONE JSON line (rank 0).  `roofline` is for the count kernel: algorithmic bytes = (L + 8) / L per base
 over the HIP-event launch time; `cpu_baseline` is the reference's CPU path timed on a bounded sample of
the same reads.  At N = 1 the line also carries `other_configs`: BASELINE.json configs[2] (long reads, with and
without -m 10), configs[4] (1 M sites), the host-fed path on its PCIe roofline (`feed`) and the file -> counts.txt path through build/ntsmCount, each with its own
correctness check; `n10_full` is the same measurement at the upper bound of the real sites file's size
(2.5 M site k-mers).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAFFIC_FILE = "r06_traffic.json"          # written by tools/make_traffic.py from the PMC passes of tools/profile.sh
STRESS_TRAFFIC_FILE = "r06_stress_traffic.json"
N10_FULL_TRAFFIC_FILE = "r06_n10_full_traffic.json"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy ceiling)
SITES_SEED, N_SITES, READ_SEED, READ_LEN, K = 20241218, 96287, 7, 150, 19


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=float, default=float(os.environ.get("NTSM_BENCH_READS", 1e9)),
                    help="reads per GPU (default 1e9 = BASELINE.json configs[1])")
    ap.add_argument("--cpu-sample-reads", type=int, default=2_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--filter-log2", type=int, default=0)
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto (minimizer-blocked kernel for k=19), 1 generic")
    ap.add_argument("--no-check", action="store_true", help="skip the correctness check of the timed result")
    ap.add_argument("--feed-reads", type=float, default=2e7, help="pre-parsed reads (host memory) of the host-fed legs: 2e7 = 3.02 GB of stream")
    ap.add_argument("--other-configs", default="long,stress,n10_full,feed,e2e",
                    help="comma list of the secondary single-GPU measurements appended at N = 1 ('' or 'none': skip)")
    ap.add_argument("--long-reads", type=float, default=5e6)
    ap.add_argument("--stress-sites", type=float, default=1e6)
    ap.add_argument("--stress-reads", type=float, default=1e9, help="configs[4] as BASELINE.json states it: 1e9 reads (151 GB resident)")
    ap.add_argument("--n10-full-sites", type=float, default=N_SITES)
    ap.add_argument("--n10-full-reads", type=float, default=1e9)
    ap.add_argument("--e2e-reads", type=float, default=4e7, help="reads of the FASTQ the CLI leg counts (4e7 = 12.6 GB)")
    ap.add_argument("--e2e-threads", type=int, default=16)
    ap.add_argument("--e2e-gz-single-reads", type=float, default=4e6,
                    help="reads of the sample that is compressed by ONE ordinary single-threaded gzip -6 stream (4e6 = 1.26 GB of text)")
    ap.add_argument("--e2e-qual-model", type=int, default=1, help="quality lines of the CLI legs' FASTQ: 1 Illumina-like 8-level binned, 0 constant 'I' (rounds 1-4)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch test: every rank prints its rank environment as one JSON line and exits before touching the GPU")
    return ap.parse_args(argv)


# -----------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: spawn the ranks ourselves.  Nothing here may initialise the GPU (a process that has must
# not be replaced or forked into ranks): the parent only counts devices and starts children.
# -----------------------------------------------------------------------------------------------------------------
def count_gpus_without_hip():
    """Number of GPUs this process may use, WITHOUT loading the HIP runtime (the launcher parent must stay a process that
    never initialised the GPU): KFD topology nodes with SIMDs (else DRM render nodes), narrowed by the *_VISIBLE_DEVICES
    lists.  None when neither source is readable -- the ranks then find out themselves (run_rank returns 3)."""
    import glob
    have = None
    if not os.path.exists("/dev/kfd"):
        have = 0                                         # no amdgpu compute driver at all: nothing a rank could open
    else:
        try:
            n = 0
            for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
                for line in open(f):
                    kv = line.split()
                    if len(kv) == 2 and kv[0] == "simd_count" and int(kv[1]) > 0:
                        n += 1
            have = n if n else None
        except (OSError, ValueError):
            have = None
        if have is None:
            rn = glob.glob("/dev/dri/renderD*")
            have = len(rn) if rn else None
    if have is None:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            have = min(have, len([x for x in v.split(",") if x.strip() != ""]))
    return have


def launch_ranks(args, argv):
    n = args.gpus
    if not args.dry_launch:
        have = count_gpus_without_hip()                  # no torch / HIP in this process: it only starts children
        if have is not None and have < n:
            sys.stderr.write("bench.py: --gpus %d requested but this node exposes %d GPU(s); refusing to run a smaller job "
                             "under that name\n" % (n, have))
            return 3
    with socket.socket() as s:                           # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   NTSM_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in live:                           # one rank failed: the others would wait in a collective forever
                    q.terminate()
    return rc


def cpu_baseline(synth, sites_path, n_reads, n_files=1):
    """Time the reference's CPU path on the GPU box's host cores: the compiled reference (oracle/_ref/ref_ntsmCount,
    kind "reference") when this checkout has it, else the plain-C restatement (oracle/ntsm_oracle, kind "port").
    FASTQ in, scan only (site-table build excluded); n_files > 1 = the reference's own parallelism, -t n_files over
    n_files files (src/FingerPrint.hpp:47)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_ntsmCount")
    port = os.path.join(ROOT, "oracle", "ntsm_oracle")
    out = {}
    with tempfile.TemporaryDirectory() as d:
        files = []
        for i in range(n_files):
            fq = os.path.join(d, "sample%d.fq" % i)
            synth.write_fastq(fq, i * n_reads, n_reads, threads=4)
            files.append(fq)
        runs = []
        if os.path.exists(ref):
            runs.append(("reference", [ref, "-s", sites_path, "-t", str(n_files)] + files, dict(os.environ, NTSM_REF_TIME_SCAN="1")))
        if os.path.exists(port) and n_files == 1:
            runs.append(("port", [port, "-s", sites_path, "--time-scan", files[0]], dict(os.environ)))
        for kind, cmd, env in runs:
            p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            secs = bases = None
            for line in p.stderr.decode().split("\n"):
                f = line.split()
                if line.startswith("SCAN_SECONDS"):
                    secs = float(f[1])
                    if len(f) > 3:
                        bases = int(f[3])
                elif line.startswith("Total Bases Considered:"):
                    bases = int(f[-1])
            if secs and bases:
                out[kind] = {"value": bases / secs, "unit": "bases/s", "cores": n_files, "kind": kind,
                             "sample": "%d x %d reads of the same synthetic stream as FASTQ (%d bases, %.1f s scan, table build excluded)"
                                       % (n_files, n_reads, bases, secs)}
    return out


def pmc_constants(name, kernel_variant_ok=True):
    """PMC-derived per-base constants (separate rocprofv3 --pmc passes, tools/profile.sh + tools/make_traffic.py).  They
    are tied to the build they were measured on: after any change to the kernel sources, the launch code or the build
    flags they are reported as null until the profile has been taken again."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from make_traffic import kernel_source_sha16
        tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        if tj.get("kernel_source_sha16") == kernel_source_sha16() and kernel_variant_ok:
            return tj, "profiles/%s" % name
        return None, "null: profiles/%s was measured on other kernel sources (or another kernel variant)" % name
    except Exception:
        return None, "null: no PMC profile of the current kernel sources under profiles/"


def timed_passes(ctx, ptr, n_bytes, n_reads, reps):
    """warm pass, reset, `reps` passes bracketed by the library's HIP events: (totals, ms per launch)"""
    ctx.count_resident(ptr, n_bytes, 0, n_reads)
    ctx.sync()
    ctx.reset()
    ctx.set_timing(True)
    for _ in range(reps):
        ctx.count_resident(ptr, n_bytes, 0, n_reads)
    t = ctx.sync()
    n, ms = ctx.get_timing()
    ctx.set_timing(False)
    return t, ms / max(n, 1)


def generic_reference(nt, keys, ptr, n_bytes, n_reads, device):
    """The same resident stream through the GENERIC kernel (ntsm_count_kernel: byte-wise rolling, 1-bit filter, no
    minimizers -- it shares only the key table and the counters with the kernel being timed): (k-mers, hits, counts)."""
    ref = nt.Context(keys, k=K, device=device)
    ref.set_kernel(1)
    ref.count_resident(ptr, n_bytes, 0, n_reads)
    t = ref.sync()
    out = (t.total_kmers, t.total_hits, ref.counts())
    ref.close()
    return out


def config_long(nt, torch, dev, local, synth, sites, args):
    """BASELINE.json configs[2]: ONT-like long reads (log-normal lengths, N50 ~ 20 kb, 5 % substitutions) resident in HBM,
    plain pass and -m 10 exact early stop."""
    import numpy as np
    n_reads = int(args.long_reads)
    L = nt.SynthLong(synth, read_seed=13, spacing=16000)
    ends, total = L.layout(0, n_reads)
    bases = int(total) - n_reads
    d_win = torch.from_numpy(synth.windows).to(dev)
    d_ends = torch.from_numpy(ends.view(np.int64)).to(dev)
    d_bases = torch.empty(total, dtype=torch.uint8, device=dev)
    L.device_fill(d_win.data_ptr(), 0, n_reads, d_ends.data_ptr(), total, d_bases.data_ptr())
    torch.cuda.synchronize()
    ctx = nt.Context(sites.keys, k=K, device=local)
    reps = 2
    t, ms = timed_passes(ctx, d_bases.data_ptr(), total, n_reads, reps)
    counts = ctx.counts()
    ctx.close()
    check = {}
    if not args.no_check:
        rk, rh, rc = generic_reference(nt, sites.keys, d_bases.data_ptr(), total, n_reads, local)
        assert (t.total_kmers, t.total_hits) == (reps * rk, reps * rh) and (counts == rc * reps).all(), \
            "long reads: minimizer-blocked kernel and generic kernel disagree"
        check["equals_generic_kernel_on_the_whole_stream"] = True
    alg_bytes = bases + 8 * n_reads
    out = {"workload": "configs[2]: %.3g synthetic ONT-like reads (%.1f Gbases, mean %.0f b), hs_n10_like sites" % (n_reads, bases / 1e9, bases / n_reads),
           "reads": n_reads, "bases": bases, "kernel_ms": ms, "gbases_per_s": bases / ms / 1e6,
           "roofline_frac": alg_bytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_base": alg_bytes / bases,
           "hits_per_pass": t.total_hits // reps}
    # -m 10: the exact early stop (src/FingerPrint.hpp:473-488) on the same resident stream
    thr = nt.max_hits_for(len(sites.keys), 10.0)
    ctx = nt.Context(sites.keys, k=K, device=local, max_hits=thr)
    t0 = time.perf_counter()
    ctx.count_resident(d_bases.data_ptr(), total, d_ends.data_ptr(), n_reads)
    tm = ctx.sync()
    wall = time.perf_counter() - t0
    ctx.close()
    m10 = {"max_hits": thr, "early_stop": bool(tm.early_stop), "stop_read": tm.reads_consumed, "frac_of_stream": tm.reads_consumed / n_reads,
           "total_hits": tm.total_hits, "wall_s": wall, "bases_consumed": tm.total_bases}
    if not args.no_check and tm.early_stop and tm.reads_consumed >= 1:
        # the stop read is exact iff the prefix WITHOUT it stays at or below the threshold and the prefix WITH it exceeds it;
        # both prefixes are recounted unarmed by the generic kernel (offset 0: aligned)
        r = tm.reads_consumed
        with_it = generic_reference(nt, sites.keys, d_bases.data_ptr(), int(ends[r - 1]) + 1, r, local)
        without = generic_reference(nt, sites.keys, d_bases.data_ptr(), int(ends[r - 2]) + 1, r - 1, local) if r >= 2 else (0, 0, None)
        assert with_it[1] == tm.total_hits and with_it[1] > thr >= without[1], "-m 10: stop read is not the first crossing read"
        m10["stop_read_is_first_crossing_by_generic_kernel_recount"] = True
    out["m10"] = m10
    out["check"] = check
    del d_bases
    torch.cuda.empty_cache()
    return out


def config_sites(nt, torch, dev, local, tmp, label, what, sites_seed, n_sites, min_keep, n_reads, traffic_file):
    """A resident 150 bp stream against another site set: kernel rate, roofline fraction, the whole stream checked against the
    generic kernel, PMC-derived request counts when a profile of the current kernel sources exists."""
    sp = os.path.join(tmp, label + ".fa")
    t0 = time.perf_counter()
    s = nt.SynthShort(sites_seed, n_sites, read_seed=9, sites_path=sp, min_keep=min_keep)
    sites = nt.Sites(sp, k=K)
    t_sites = time.perf_counter() - t0
    t0 = time.perf_counter()
    ctx = nt.Context(sites.keys, k=K, device=local)
    t_create = time.perf_counter() - t0
    st = ctx.debug_stats()
    two_level, run_form = st["two_level"], st.get("run_form", False)
    d_win = torch.from_numpy(s.windows).to(dev)
    asked = n_reads
    while True:
        try:
            d_bases = torch.empty(n_reads * s.stride, dtype=torch.uint8, device=dev)
            break
        except RuntimeError:                       # smaller GPU: halve until it fits, and say so in the leg
            n_reads //= 2
            if n_reads < 1000:
                raise
    s.device_fill(d_win.data_ptr(), 0, n_reads, d_bases.data_ptr())
    torch.cuda.synchronize()
    reps = 2
    t, ms = timed_passes(ctx, d_bases.data_ptr(), d_bases.numel(), n_reads, reps)
    counts = ctx.counts()
    ctx.close()
    check = {}
    if what["check"]:
        rk, rh, rc = generic_reference(nt, sites.keys, d_bases.data_ptr(), d_bases.numel(), n_reads, local)
        assert (t.total_kmers, t.total_hits) == (reps * rk, reps * rh) and (counts == rc * reps).all(), \
            "%s: minimizer-blocked kernel and generic kernel disagree" % label
        check["equals_generic_kernel_on_the_whole_stream"] = True
    bases = n_reads * READ_LEN
    armed = None
    if what.get("armed") and t.total_hits:
        # -m on a set whose unarmed batches run the run-anchored kernel: optimistic spans go through that kernel, the crossing chunk
        # through the minimizer-blocked PER_READ kernel with the one-level tables kept beside the run form's filter (VERDICT r5 weak #10:
        # that fall-back was never timed).  Threshold = 40 % of one pass's hits; the stop read is re-derived with the generic kernel.
        thr = int(0.4 * (t.total_hits // reps))
        d_ends = torch.arange(n_reads, device=dev, dtype=torch.int64) * s.stride + READ_LEN
        actx = nt.Context(sites.keys, k=K, device=local, max_hits=thr)
        t0 = time.perf_counter()
        actx.count_resident(d_bases.data_ptr(), d_bases.numel(), d_ends.data_ptr(), n_reads)
        tm = actx.sync()
        wall = time.perf_counter() - t0
        actx.close()
        r = int(tm.reads_consumed)
        armed = {"max_hits": thr, "early_stop": bool(tm.early_stop), "stop_read": r, "frac_of_stream": r / n_reads, "wall_s": wall,
                 "unarmed_kernel_time_for_that_prefix_s": ms / 1e3 * r / n_reads,
                 "armed_over_unarmed": wall / (ms / 1e3 * r / n_reads) if r else None}
        if what["check"] and tm.early_stop and r >= 2:
            with_it = generic_reference(nt, sites.keys, d_bases.data_ptr(), r * s.stride, r, local)
            without = generic_reference(nt, sites.keys, d_bases.data_ptr(), (r - 1) * s.stride, r - 1, local)
            assert with_it[1] == tm.total_hits and with_it[1] > thr >= without[1], "%s -m: stop read is not the first crossing read" % label
            armed["stop_read_is_first_crossing_by_generic_kernel_recount"] = True
        del d_ends
    tj, note = pmc_constants(traffic_file)
    out = {"workload": "%s: %.6g sites (%d site 19-mers), %.3g synthetic 150 bp reads" % (what["name"], n_sites, len(sites.keys), n_reads),
           "reads": n_reads, "reads_asked": asked, "site_kmers": len(sites.keys), "kernel_ms": ms, "gbases_per_s": bases / ms / 1e6,
           "roofline_frac": bases * (READ_LEN + 8) / READ_LEN / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, "kernel_form": "run-anchored" if run_form else "two-level" if two_level else "one-level",
           "hits_per_pass": t.total_hits // reps, "site_gen_and_load_s": t_sites, "create_s": t_create,
           # ONE traffic figure per leg, the same definition as the headline's roofline.traffic: bytes through the L2's memory-side
           # port (128 x RDREQ_128B + 64 x RDREQ_64B + 32 x other reads + write requests), per launch, from the PMC passes
           "traffic": tj["traffic_bytes_per_base"] * bases if tj and tj.get("traffic_bytes_per_base") else None,
           "traffic_bytes_per_base_from_pmc": tj.get("traffic_bytes_per_base") if tj else None,
           "traffic_over_algorithmic": tj.get("traffic_over_algorithmic") if tj else None,
           "fabric_read_requests_per_base_from_pmc": tj.get("fabric_read_requests_per_base") if tj else None,
           "l2_requests_per_base_from_pmc": tj.get("l2_requests_per_base") if tj else None,
           "l2_misses_per_base_from_pmc": tj.get("l2_misses_per_base") if tj else None,
           # memory side of the L2 (round 5): every fabric read is a 128-byte line; how many of them reach HBM cannot be counted on
           # this stack (no Infinity-Cache / HBM counter in rocprofv3 on gfx950: profiles/r05_counters/) -- see hbm_note
           "fabric_read_bytes_per_base_from_pmc": tj.get("fabric_read_bytes_per_base") if tj else None,
           "avg_fabric_read_latency_l2_clocks_from_pmc": tj.get("avg_fabric_read_latency_l2_clocks") if tj else None,
           "hbm_read_bytes_per_base": tj.get("hbm_read_bytes_per_base") if tj else None, "hbm_note": tj.get("hbm_note") if tj else None,
           "pmc_source": note,
           "check": check}
    if armed:
        out["armed"] = armed
    del d_bases
    torch.cuda.empty_cache()
    return out


def config_stress(nt, torch, dev, local, args, tmp):
    """BASELINE.json configs[4]: 1 M sites (16 M site k-mers, 512 MiB key table), 1e9 150 bp reads resident in HBM."""
    return config_sites(nt, torch, dev, local, tmp, "stress", {"name": "configs[4]", "check": not args.no_check}, 424242, int(args.stress_sites), 0,
                        int(args.stress_reads), STRESS_TRAFFIC_FILE)


def config_n10_full(nt, torch, dev, local, args, tmp):
    """The worst-case geometry of the real sites file (0.58 - 2.50 M distinct k-mers; data/human_sites_n10.fa
    is absent): the bench set's 96287 sites with EVERY one of the 13 k-mers of both alleles kept = 2,503,462 site 19-mers."""
    return config_sites(nt, torch, dev, local, tmp, "n10_full", {"name": "n10_full (upper bound of human_sites_n10.fa: 13 k-mers per allele)", "check": not args.no_check, "armed": True},
                        SITES_SEED, int(args.n10_full_sites), 13, int(args.n10_full_reads), N10_FULL_TRAFFIC_FILE)


def pigz_like(src, dst, level=6, block=64 << 20, threads=16):
    """One gzip member holding `src`, written the way pigz does it: blocks of `block` bytes deflated independently by a pool of
    threads (zlib releases the GIL), every block ended by a sync flush, one final empty block, CRC-32 of the whole (bench tooling:
    a 12 GB FASTQ through single-threaded gzip -6 would take five minutes)."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    size = os.path.getsize(src)

    def one(off):
        with open(src, "rb") as f:
            f.seek(off)
            data = f.read(block)
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        return co.compress(data) + co.flush(zlib.Z_SYNC_FLUSH), zlib.crc32(data), len(data)
    crc = 0
    with open(dst, "wb") as out, ThreadPoolExecutor(threads) as pool:
        out.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        for body, c, n in pool.map(one, range(0, size, block)):
            out.write(body)
            crc = _crc32_combine(crc, c, n)
        out.write(b"\x03\x00")                               # final (empty, fixed-Huffman) block
        out.write((crc & 0xFFFFFFFF).to_bytes(4, "little") + (size & 0xFFFFFFFF).to_bytes(4, "little"))
    return os.path.getsize(dst)


def _crc32_combine(crc1, crc2, len2):
    """zlib's crc32_combine (GF(2) matrix method), which Python's zlib module does not export."""
    if len2 == 0:
        return crc1

    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[n]) for n in range(32)]
    odd = [0xEDB88320] + [1 << n for n in range(31)]
    even = square(odd)
    odd = square(even)
    while True:
        even = square(odd)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def host_info():
    """What the CLI legs' thread counts meet on this host: CPUs online, CPUs this process may run on, and the cgroup's CPU
    quota (v2 cpu.max or v1 cfs quota / period) -- the pod that runs the driver's bench is granted 16 CPUs' worth of time on a
    256-CPU box, and every host-side number of the e2e legs is bounded by that."""
    info = {"cpus_online": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cgroup_cpu_max": None, "cgroup_cpus": None}
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        info["cgroup_cpu_max"] = " ".join(txt)
        if txt[0] != "max":
            info["cgroup_cpus"] = int(txt[0]) / int(txt[1])
    except (OSError, ValueError, IndexError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            info["cgroup_cpu_max"] = "%d %d" % (q, per)
            if q > 0:
                info["cgroup_cpus"] = q / per
        except (OSError, ValueError):
            pass
    return info


def _run_cli(exe, sites_path, threads, local, path, runs=2, env_extra=None):
    best = None
    env = {k: v for k, v in os.environ.items() if k not in ("NTSM_FAST_EXIT", "NTSM_SYNC_EXIT", "NTSM_CLEAN_EXIT")}
    env.update(NTSM_PHASE_TIMES="1", **(env_extra or {}))
    for _ in range(runs):                                  # second run: file certainly in the page cache
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-s", sites_path, "-t", str(threads), "-g", str(local), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            raise RuntimeError("ntsmCount failed: " + p.stderr.decode()[-400:])
        if best is None or wall < best[0]:
            best = (wall, p)
    wall, p = best
    phases = [l[8:] for l in p.stderr.decode().split("\n") if l.startswith("[phase]")]
    own = [l for l in p.stderr.decode().split("\n") if l.startswith("Time: ")]          # the CLI's own clock, main() to the last print
    return wall, p, phases, (float(own[-1].split()[1]) if own else None)


def _cli_both_exits(exe, sites_path, threads, local, path):
    """The CLI on one input, whole process as the parent sees it, in both exit modes: the DEFAULT (the kernel's teardown of the
    HIP process -- queues, pinned and device memory -- happens inside the process's exit, before the parent's wait() returns:
    what the reference's plain `return 0` means) and NTSM_FAST_EXIT=1 (that teardown handed to a clone(CLONE_VM) child and
    left out of the parent's sight: round 4's default, opt-in now).  `wall_s` is always the default mode's."""
    wall, p, phases, cli_s = _run_cli(exe, sites_path, threads, local, path)
    wall_fast, pf, _, cli_fast = _run_cli(exe, sites_path, threads, local, path, env_extra={"NTSM_FAST_EXIT": "1"})
    assert pf.stdout == p.stdout, "counts.txt depends on the exit mode"
    return wall, p, phases, cli_s, {"wall_s_fast_exit": wall_fast, "cli_reported_s_fast_exit": cli_fast,
                                   "exit_mode_of_wall_s": "default: synchronous (teardown inside the process's exit)",
                                   "exit_cost_s": wall - (cli_s or wall), "exit_cost_s_fast_exit": wall_fast - (cli_fast or wall_fast)}


def _gzip_single_stream(src, dst):
    """ONE ordinary gzip member written by ONE thread: the system's `gzip -6` when there is one, else Python's zlib (the same
    DEFLATE level-6 strategy as `gzip -6` up to the implementation).  Returns (writer, Popen or None)."""
    import shutil
    gz = shutil.which("gzip")
    if gz:
        return "gzip -6 (GNU gzip, one thread)", subprocess.Popen([gz, "-6", "-c", src], stdout=open(dst, "wb"))
    import threading
    import zlib

    def work():
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        with open(src, "rb") as f, open(dst, "wb") as o:
            while True:
                blk = f.read(8 << 20)
                if not blk:
                    break
                o.write(co.compress(blk))
            o.write(co.flush())
    th = threading.Thread(target=work)
    th.start()
    th.wait = th.join
    return "zlib.compressobj(6) in one thread (no gzip binary on this host)", th


def _phase_seconds(phases, key, exclude=None):
    for l in phases:
        if key in l and not (exclude and exclude in l):
            try:
                return float(l.split(key)[1].split("s")[0])
            except ValueError:
                pass
    return None


def e2e_start_single_stream(synth, args, tmp):
    """The ordinary single-threaded `gzip -6` of the sample takes about a minute of one CPU (1.26 GB of text at 20-25 MB/s): it
    is started before the GPU-bound legs (long / stress / n10_full leave the host idle) and collected by config_e2e."""
    n_single = max(1000, min(int(args.e2e_reads), int(args.e2e_gz_single_reads)))
    sample = os.path.join(tmp, "e2e_sample.fq")
    host = host_info()
    synth.write_fastq(sample, 0, n_single, threads=max(1, min(32, int(host["cgroup_cpus"] or 0) or (os.cpu_count() or 2) - 1)), qual_model=int(args.e2e_qual_model))
    t0 = time.perf_counter()
    writer, job = _gzip_single_stream(sample, sample + ".gz")
    return {"n": n_single, "sample": sample, "sample_gz": sample + ".gz", "writer": writer, "job": job, "t0": t0}


def config_e2e(nt, torch, dev, local, synth, sites, sites_path, args, tmp, single=None):
    """File -> counts.txt through the CLI (build/ntsmCount -t N), the whole process timed, on a generated FASTQ whose quality
    lines follow the Illumina-like 8-level model of synth.h (position-dependent decay, low scores in runs; gzip -6 ratio about
    3.5:1 -- rounds 1-4 wrote 150 x 'I', 6:1 and one long copy per record for a DEFLATE decoder):
      e2e_cli            the plain FASTQ; stdout must equal what the resident path prints for the same reads
      e2e_cli_gz         the same reads as ONE gzip member written pigz-style (independent 64 MiB blocks, sync flushes)
      e2e_cli_gz_single  a sample of the reads as ONE ordinary single-threaded `gzip -6` stream (no flush points: what a
                         sequencing facility's gzip writes); stdout must equal the CLI's on the same sample as plain text
    every leg in both exit modes (default synchronous; NTSM_FAST_EXIT=1), with the host's CPU grant stated."""
    n_reads = int(args.e2e_reads)
    qm = int(args.e2e_qual_model)
    host = host_info()
    gen_threads = max(1, min(32, int(host["cgroup_cpus"] or 0) or (os.cpu_count() or 2) - 1))
    fq = os.path.join(tmp, "e2e.fq")
    t0 = time.perf_counter()
    synth.write_fastq(fq, 0, n_reads, threads=gen_threads, qual_model=qm)
    t_gen = time.perf_counter() - t0
    size = os.path.getsize(fq)
    exe = os.path.join(ROOT, "build", "ntsmCount")
    quality = {0: "constant 'I'", 1: "Illumina-like, 8-level binned, position-dependent decay (synth.h: ntsm_synth_qual_char)",
               2: "Illumina-like, unbinned Phred 2..41"}.get(qm, str(qm))
    wall, p, phases, cli_s, exits = _cli_both_exits(exe, sites_path, args.e2e_threads, local, fq)
    parse_s = _phase_seconds(phases, "parse+count", exclude="inflate")
    # the same reads resident in HBM through the C ABI, printed by the same report code
    d_win = torch.from_numpy(synth.windows).to(dev)
    d_bases = torch.empty(n_reads * synth.stride, dtype=torch.uint8, device=dev)
    synth.device_fill(d_win.data_ptr(), 0, n_reads, d_bases.data_ptr())
    torch.cuda.synchronize()
    ctx = nt.Context(sites.keys, k=K, device=local)
    ctx.count_resident(d_bases.data_ptr(), d_bases.numel(), 0, n_reads)
    t = ctx.sync()
    rc, text = sites.format_counts(ctx.counts(), t.total_kmers)
    ctx.close()
    del d_bases
    torch.cuda.empty_cache()
    sha_cli, sha_res = hashlib.sha256(p.stdout).hexdigest(), hashlib.sha256(text).hexdigest()
    assert rc == 0 and sha_cli == sha_res, "CLI counts.txt differs from the resident path's"
    bases = n_reads * READ_LEN
    out = {"workload": "build/ntsmCount -t %d on one plain FASTQ of %.3g reads (%.1f GB, page cache), hs_n10_like sites" % (args.e2e_threads, n_reads, size / 1e9),
           "quality_lines": quality, "host": host,
           "reads": n_reads, "file_bytes": size, "wall_s": wall, "gbases_per_s": bases / wall / 1e9,
           "cli_reported_s": cli_s,       # inside the process; wall_s also holds its spawn from this (large) parent and its exit
           "parse_and_count_s": parse_s, "gbases_per_s_parse_and_count": bases / parse_s / 1e9 if parse_s else None,
           "text_GB_per_s": size / wall / 1e9,
           "phases": phases, "fastq_gen_s": t_gen, "counts_sha256": sha_cli, "check": {"counts_txt_equals_resident_path": True}}
    out.update(exits)
    gz_out = single_out = None
    gz = os.path.join(tmp, "e2e.fq.gz")
    if single is None:
        single = e2e_start_single_stream(synth, args, tmp)
    sample, sample_gz, n_single, writer, job, t0s = single["sample"], single["sample_gz"], single["n"], single["writer"], single["job"], single["t0"]
    try:
        t0 = time.perf_counter()
        # two CPUs of the grant are left to the single-threaded gzip that is still running beside this writer
        gz_size = pigz_like(fq, gz, threads=max(1, min(48, (int(host["cgroup_cpus"] or 0) or (os.cpu_count() or 2)) - 2)))
        t_gz = time.perf_counter() - t0
        os.unlink(fq)
        job.wait()                                             # (started before the GPU legs; neither writer is timed)
        t_single = time.perf_counter() - t0s
        if getattr(job, "returncode", 0):
            raise RuntimeError("gzip -6 of the sample failed: %r" % job.returncode)

        wall, pz, phases, cli_s, exits = _cli_both_exits(exe, sites_path, args.e2e_threads, local, gz)
        os.unlink(gz)
        assert hashlib.sha256(pz.stdout).hexdigest() == sha_cli, "counts.txt of the .gz run differs from the plain FASTQ's"
        early = any("early ingest" in l for l in phases)      # then the inflate started with the process and that line covers only the stream's rest
        infl = None if early else _phase_seconds(phases, "inflate+parse+count")
        gz_out = {"workload": "build/ntsmCount -t %d on the same reads as ONE gzip member (%.2f GB; level 6, written pigz-style in 64 MiB blocks)" % (args.e2e_threads, gz_size / 1e9),
                  "quality_lines": quality, "host": host, "compression_ratio": size / gz_size,
                  "reads": n_reads, "file_bytes": gz_size, "text_bytes": size, "wall_s": wall, "gbases_per_s": bases / wall / 1e9, "cli_reported_s": cli_s,
                  "text_GB_per_s": size / wall / 1e9,
                  "inflate_parse_count_s": infl, "gbases_per_s_inflate_parse_count": bases / infl / 1e9 if infl else None,
                  "text_GB_per_s_inflate_parse_count": size / infl / 1e9 if infl else None, "phases": phases, "gzip_s": t_gz,
                  "check": {"counts_txt_equals_plain_fastq_run": True}}
        gz_out.update(exits)

        # ONE ordinary gzip -6 stream of the sample; its counts.txt must equal the CLI's on the same sample as plain text
        s_size, s_gz_size = os.path.getsize(sample), os.path.getsize(sample_gz)
        _, pp, _, _ = _run_cli(exe, sites_path, args.e2e_threads, local, sample, runs=1)
        os.unlink(sample)
        wall, ps, phases, cli_s, exits = _cli_both_exits(exe, sites_path, args.e2e_threads, local, sample_gz)
        assert ps.stdout == pp.stdout, "counts.txt of the single-stream .gz run differs from the plain sample's"
        sb = n_single * READ_LEN
        early = any("early ingest" in l for l in phases)
        infl = None if early else _phase_seconds(phases, "inflate+parse+count")
        # a file of this size is usually inflated and parsed entirely beside the start-up (early ingest): that phase's own clock
        early_s = early_n = None
        for l in phases:
            if "early ingest" in l and " parsed " in l:
                try:
                    early_n = int(l.split(" parsed ")[1].split()[0])
                    early_s = float(l.split(") in ")[1].split(" s")[0])
                except (ValueError, IndexError):
                    pass
        single_out = {"workload": "build/ntsmCount -t %d on the first %.3g reads as ONE ordinary single-threaded gzip stream (%.2f GB)" % (args.e2e_threads, n_single, s_gz_size / 1e9),
                      "writer": writer, "quality_lines": quality, "host": host, "compression_ratio": s_size / s_gz_size,
                      "reads": n_single, "file_bytes": s_gz_size, "text_bytes": s_size, "wall_s": wall, "gbases_per_s": sb / wall / 1e9,
                      "cli_reported_s": cli_s, "text_GB_per_s": s_size / wall / 1e9,
                      "inflate_parse_count_s": infl, "text_GB_per_s_inflate_parse_count": s_size / infl / 1e9 if infl else None,
                      "early_ingest_records": early_n, "early_ingest_s": early_s,
                      "text_GB_per_s_early_ingest": (s_size * early_n / n_single) / early_s / 1e9 if early_s and early_n else None,
                      "note": "the pigz-style file of e2e_cli_gz is the same decoding problem at ten times the size: its only flush points are one per 64 MiB of "
                              "text, so all but ~200 of its ~3,400 one-MiB chunks start in the middle of the stream exactly like every chunk of this file",
                      "phases": phases, "gzip_s": t_single, "counts_sha256": hashlib.sha256(ps.stdout).hexdigest(),
                      "check": {"counts_txt_equals_plain_sample_run": True}}
        single_out.update(exits)
    except AssertionError:
        raise
    except Exception as e:
        err = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        gz_out = gz_out or err
        single_out = single_out or err
    finally:
        for f in (fq, gz, sample, sample_gz):
            if os.path.exists(f):
                os.unlink(f)
    return out, gz_out, single_out


def config_feed(local, args):
    """The HOST-FED path (what a caller of the C ABI that replaces the reference's read loop, src/FingerPrint.hpp:66-69, gets):
    pre-parsed reads in host memory pushed through every submit form of include/ntsm_hip.h by build/ntsm_feed_bench
    (tools/feed_bench.cpp; its own process: this one holds torch's context), each leg against a pinned hipMemcpyAsync ceiling
    measured in the same process, each leg's counts checked against the resident path."""
    exe = os.path.join(ROOT, "build", "ntsm_feed_bench")
    p = subprocess.run([exe, "--reads", "%d" % int(args.feed_reads), "--device", str(local), "--lanes", "1,4,16"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if p.returncode != 0:
        if b"COUNTS DIFFER" in p.stderr:
            raise AssertionError("feed: a host-fed leg's counts differ from the resident path's: " + p.stderr.decode()[-600:])
        raise RuntimeError("ntsm_feed_bench failed: " + p.stderr.decode()[-400:])
    out = json.loads(p.stdout.decode().strip().split("\n")[-1])
    assert out["all_counts_equal_resident_path"] is True
    out["roofline"] = {"bound": "pcie", "peak_GBps_measured": out["h2d_ceiling"]["GBps"], "peak_GBps_nominal": 64.0,
                       "achieved_GBps": {k: v["link_GBps"] for k, v in out["legs"].items()},
                       "frac_of_measured_ceiling": {k: v["frac_of_h2d_ceiling"] for k, v in out["legs"].items()},
                       "note": "bytes handed to the H2D copies per second; raw-byte legs move 151/150 byte per base, packed lanes 3/8 byte per position "
                               "(152 positions per 150 bp read) and are bound by the lane threads' packing, not by the link"}
    out["host"] = host_info()
    return out


def run_rank(args):
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.dry_launch:
        print(json.dumps({"dry_launch": True, "rank": rank, "local_rank": local, "world_size": world, "gpus": args.gpus,
                          "master_addr": os.environ.get("MASTER_ADDR"), "master_port": os.environ.get("MASTER_PORT"),
                          "pid": os.getpid(), "ppid": os.getppid()}), flush=True)
        return 0

    import numpy as np
    import torch
    import torch.distributed as dist
    import ntsm_amd
    from ntsm_amd.dist import check_merged, job_expectation, merge_counts

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the count path has no CPU fallback")
    if local >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d wants device %d but this node exposes %d GPU(s); refusing to run a smaller job under that name\n"
                         % (rank, local, torch.cuda.device_count()))
        return 3
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or bool(os.environ.get("NTSM_FORCE_DIST"))     # force: exercise the RCCL path on one rank
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    n_reads = int(args.reads)
    tmp = tempfile.mkdtemp(prefix="ntsm_bench_")
    sites_path = os.path.join(tmp, "hs_n10_like.fa")
    synth = ntsm_amd.SynthShort(SITES_SEED, N_SITES, k=K, read_seed=READ_SEED, read_len=READ_LEN,
                                sites_path=sites_path)
    sites = ntsm_amd.Sites(sites_path, k=K)
    ctx = ntsm_amd.Context(sites.keys, k=K, device=local)
    if args.filter_log2 or args.grid:
        ctx.set_tuning(args.filter_log2, args.grid)
    if args.kernel:
        ctx.set_kernel(args.kernel)

    # workload resident in HBM: this rank's reads [rank*n, (rank+1)*n) of the global synthetic stream
    d_win = torch.from_numpy(synth.windows).to(dev)
    while True:
        try:
            d_bases = torch.empty(n_reads * synth.stride, dtype=torch.uint8, device=dev)
            break
        except RuntimeError:                       # smaller GPU: halve until it fits, and say so in config
            n_reads //= 2
            if n_reads < 1000:
                raise
    synth.device_fill(d_win.data_ptr(), rank * n_reads, n_reads, d_bases.data_ptr())
    torch.cuda.synchronize()
    n_bytes = d_bases.numel()
    bases_per_step = n_reads * READ_LEN

    merge_s = [0.0, 0.0, 0.0]        # host seconds inside merge_counts: wait for own kernels + gather, all-reduce, import
    payload_words = [0]

    def run_step():
        ctx.count_resident(d_bases.data_ptr(), n_bytes, 0, n_reads)
        if use_dist:
            payload_words[0] = merge_counts(ctx, times=merge_s)

    # Correctness of the timed launch at full size (byte offsets far beyond 2^32), independent of the kernel being timed:
    # the same resident stream counted by the GENERIC kernel (no minimizers, no blocked filter, its own rolling code) in
    # read-aligned pieces of < 2 GiB on a second context, each piece re-based so that its offsets are small.  The timed
    # context must reproduce these totals and per-k-mer counts exactly (checked after the timed region).  Oracle parity of
    # both kernels is the job of tests/test_gpu_parity.py.
    # N > 1: EVERY rank does this for its own shard, and the job-wide expectation is formed over a route that shares
    # nothing with the one being timed: the ranks' generic-kernel vectors are summed on the HOST through a gloo group
    # (TCP over loopback, no RCCL, no device memory), their totals and SHA-256 travel by all_gather_object.
    expect = None
    expect_job = None
    rank_info = None
    gloo = None
    if use_dist:
        gloo = dist.new_group(backend="gloo")
    if not args.no_check:
        ref = ntsm_amd.Context(sites.keys, k=K, device=local)
        if args.kernel != 1:
            ref.set_kernel(1)
        piece = 13_000_000 // 16 * 16                     # 1.96 GB; a multiple of 16 reads keeps the piece bases 16-byte aligned
        for r0 in range(0, n_reads, piece):
            m = min(piece, n_reads - r0)
            ref.count_resident(d_bases.data_ptr() + r0 * synth.stride, m * synth.stride, 0, m)
        tr = ref.sync()
        expect = (tr.total_kmers, tr.total_hits, ref.counts())
        ref.close()
        if use_dist:
            jk, jh, jv, rank_info = job_expectation(rank, n_reads, expect[0], expect[1], expect[2], gloo)   # CPU tensors over gloo
            expect_job = (jk, jh, jv)

    for _ in range(args.warmup):
        run_step()
    ctx.sync()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    merge_s[:] = [0.0, 0.0, 0.0]
    ctx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    ctx.sync()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    n_launch, kernel_ms = ctx.get_timing()
    ctx.set_timing(False)
    reps = args.steps + args.warmup
    checked = merged_checked = False
    merged_totals = None
    if use_dist:
        # what the last timed step's all-reduce left: the job-wide view (counts + totals) on every rank
        merged_totals = ctx.sync()
        merged_counts = ctx.counts()
        if expect_job is not None:
            merged_checked = check_merged(rank, world, reps, (merged_totals.total_kmers, merged_totals.total_hits, merged_totals.total_bases, merged_totals.reads_consumed),
                                          merged_counts, expect_job, n_reads, bases_per_step)
        ctx.set_max_hits(0, armed=False)                   # drop the merged view: this context's own counts again
    totals = ctx.sync()
    if expect is not None:
        assert (totals.total_kmers, totals.total_hits) == (reps * expect[0], reps * expect[1]), \
            "timed launches disagree with the generic kernel's sum over < 2 GiB pieces: %r vs %d x %r" % ((totals.total_kmers, totals.total_hits), reps, expect[:2])
        assert (ctx.counts() == expect[2] * np.uint64(reps)).all(), "per-k-mer counts of the timed launches differ from the generic kernel's sum over pieces"
        checked = True

    per_rank = None
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        mine = {"rank": rank, "device": local, "elapsed_s": elapsed, "launches": n_launch, "avg_launch_ms": kernel_ms / max(n_launch, 1),
                "wait_and_gather_ms_per_step": 1e3 * merge_s[0] / args.steps, "allreduce_ms_per_step": 1e3 * merge_s[1] / args.steps,
                "import_ms_per_step": 1e3 * merge_s[2] / args.steps, "own_shard_equals_generic_kernel": checked,
                "merged_equals_host_sum_of_all_ranks": merged_checked}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine, group=gloo)
        elapsed = float(tmax.item())
        if rank_info is not None:
            for a, b in zip(per_rank, rank_info):
                a.update({"first_read": b["first_read"], "kmers_per_step": b["kmers"], "hits_per_step": b["hits"], "counts_sha256": b["counts_sha256"]})
    ctx.close()
    del d_bases
    torch.cuda.empty_cache()

    if rank == 0:
        value = world * bases_per_step * args.steps / elapsed
        launch_s = kernel_ms / 1e3 / max(n_launch, 1)
        if per_rank:                                       # N > 1: the roofline of the slowest GPU's kernel (every rank runs the same launch)
            launch_s = max(r["avg_launch_ms"] for r in per_rank) / 1e3
        bytes_per_base = (READ_LEN + 8) / READ_LEN
        achieved = bases_per_step * bytes_per_base / launch_s / 1e9
        tj, traffic_note = pmc_constants(TRAFFIC_FILE, args.kernel in (0, 2))
        traffic = valu_busy = l2_frac = traffic_over_alg = None
        if tj:
            if tj.get("traffic_bytes_per_base"):
                traffic = tj["traffic_bytes_per_base"] * bases_per_step
                traffic_over_alg = tj.get("traffic_over_algorithmic")
            valu_busy, l2_frac = tj.get("valu_busy_frac"), tj.get("l2_request_rate_frac_of_cap")
            traffic_note = ("bytes/launch through the L2's memory-side port: 128 x TCC_EA0_RDREQ_128B + 64 x RDREQ_64B + 32 x other reads + write requests "
                            "(%s; separate --pmc passes): the stream once + the filter / key-table lines that miss the L2 and are served by the "
                            "Infinity Cache; rocprofv3 on gfx950 has no HBM-side counter (profiles/r05_counters/)" % traffic_note)
        out = {
            "metric": "bases/s (and reads/s) through ntsmCount, 150 bp reads vs human_sites_n10.fa",   # BASELINE.json's metric
            "value": value, "unit": "bases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64",
            "data": "synthetic (reads: counter-based generator; sites: hs_n10_like, a 96287-site stand-in with the geometry of "
                    "data/human_sites_n10.fa, which the reference checkout does not contain)",
            "reads_per_s": world * n_reads * args.steps / elapsed,
            "config": {"workload": "configs[1]: %.3g synthetic 150 bp reads per GPU resident in HBM, hs_n10_like sites "
                                   "(96287 sites, %d distinct 19-mers), k=19" % (n_reads, len(sites.keys)),
                       "reads_per_gpu": n_reads, "read_len": READ_LEN, "k": K, "n_sites": N_SITES,
                       "parallelism": ("reads sharded over %d GPUs; one RCCL SUM of per-k-mer counts per step" % world) if world > 1
                                      else "one GPU, no collective",
                       "launched_by": "bench.py --gpus N (self-spawned ranks)" if os.environ.get("NTSM_BENCH_SELF_LAUNCHED") else
                                      ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "single process")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_over_algorithmic": traffic_over_alg,
                         "traffic_note": traffic_note,
                         "achieved_stream_only_GBs": bases_per_step * (READ_LEN + 1) / READ_LEN / launch_s / 1e9,
                         "kernel": {0: "ntsm_count_mz_kernel", 2: "ntsm_count_mz_kernel"}.get(args.kernel, "ntsm_count_kernel"),
                         "launches": n_launch, "avg_launch_ms": 1e3 * launch_s,
                         "algorithmic_bytes_per_base": bytes_per_base,
                         "valu_busy_frac_from_pmc": valu_busy,     # share of SIMD issue cycles on VALU
                         "l2_request_rate_frac_of_cap_from_pmc": l2_frac,   # the resource that binds: L2 requests/s over the measured 266 G/s cap
                         "kmer_probe_rate_per_s": totals.total_kmers / max(reps, 1) / launch_s,
                         "per_gpu": world > 1},          # N > 1: one GPU's kernel (slowest rank's average launch), not the job
            "check": {"total_kmers_per_step": totals.total_kmers // reps,        # this rank's own shard
                      "total_hits_per_step": totals.total_hits // reps,
                      "equals_generic_kernel_sum_of_pieces_below_2GiB": checked,
                      "note": "independent-kernel consistency at full size; oracle parity of both kernels: tests/test_gpu_parity.py"},
        }
        if use_dist:
            # evidence that N ranks really merged: what the collective saw, what it cost, and the merge checked against a
            # host-side sum that never touched RCCL (every rank asserted both before this line could be printed)
            out["check"].update({
                "every_rank_equals_generic_kernel_on_its_own_shard": bool(per_rank) and all(r["own_shard_equals_generic_kernel"] for r in per_rank),
                "merged_equals_host_side_gloo_sum_of_all_ranks_generic_counts": bool(per_rank) and all(r["merged_equals_host_sum_of_all_ranks"] for r in per_rank),
                "merged_total_kmers_per_step": merged_totals.total_kmers // reps, "merged_total_hits_per_step": merged_totals.total_hits // reps,
                "merged_reads_per_step": merged_totals.reads_consumed // reps, "ranks_checked": len(per_rank or [])})
            out["collective"] = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
                                 "library": "RCCL %s" % ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
                                 "op": "all_reduce SUM int64[n_kmers + 4], once per step", "payload_bytes": 8 * payload_words[0],
                                 "allreduce_ms_per_step": max(r["allreduce_ms_per_step"] for r in per_rank),
                                 "wait_and_gather_ms_per_step": max(r["wait_and_gather_ms_per_step"] for r in per_rank),
                                 "import_ms_per_step": max(r["import_ms_per_step"] for r in per_rank),
                                 "check_route": "gloo (host TCP) all_reduce of the generic kernel's per-rank vectors + all_gather_object of totals / SHA-256"}
            out["per_rank"] = per_rank
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(synth, sites_path, args.cpu_sample_reads)
            main_kind = "reference" if "reference" in cb else ("port" if "port" in cb else None)
            if main_kind:
                out["cpu_baseline"] = cb[main_kind]
                out["gpu_over_cpu"] = value / cb[main_kind]["value"]
                if main_kind == "reference":
                    # oracle/_ref is the reference compiled in place: it travels to the GPU box as a prebuilt binary only.
                    # The plain-C port is timed beside it so that the figure survives without that binary.
                    if "port" in cb:
                        out["cpu_baseline_port"] = cb["port"]
                    n_thr = max(2, min(32, (os.cpu_count() or 2) // 2))   # the reference's best case: -t N over N files
                    mt = cpu_baseline(synth, sites_path, max(1, args.cpu_sample_reads // n_thr), n_thr).get("reference")
                    if mt:
                        out["cpu_baseline_threads"] = mt
        which = [w for w in args.other_configs.split(",") if w and w != "none"] if world == 1 and not use_dist else []
        if which:
            other = {}
            single = None
            if "e2e" in which:
                try:
                    single = e2e_start_single_stream(synth, args, tmp)    # one CPU's worth of gzip -6 beside the GPU-bound legs
                except Exception:
                    single = None
            for name in which:
                t0 = time.perf_counter()
                try:
                    if name == "long":
                        other["long"] = config_long(ntsm_amd, torch, dev, local, synth, sites, args)
                    elif name == "stress":
                        other["stress"] = config_stress(ntsm_amd, torch, dev, local, args, tmp)
                    elif name == "n10_full":
                        other["n10_full"] = config_n10_full(ntsm_amd, torch, dev, local, args, tmp)
                    elif name == "feed":
                        other["feed"] = config_feed(local, args)
                    elif name == "e2e":
                        other["e2e_cli"], other["e2e_cli_gz"], other["e2e_cli_gz_single"] = config_e2e(ntsm_amd, torch, dev, local, synth, sites, sites_path, args, tmp, single)
                    else:
                        continue
                except AssertionError:
                    raise                                  # a wrong result is never reported as a number
                except Exception as e:                     # out of memory / disk on a smaller box: say so, keep the headline
                    other[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                key = "e2e_cli" if name == "e2e" else name
                if key in other:
                    other[key]["leg_wall_s"] = time.perf_counter() - t0
            out["other_configs"] = other
            # the headline's site set is a stand-in: the real human_sites_n10.fa (absent) holds 0.58 - 2.50 M distinct k-mers
            # (SURVEY.md 8a); configs[1] above sits at 1.54 M, n10_full is the upper bound -- the span a user of the real file meets
            nf = other.get("n10_full", {})
            if nf.get("roofline_frac"):
                out["roofline"]["frac_range"] = sorted([nf["roofline_frac"], out["roofline"]["frac"]])
                out["roofline"]["frac_range_note"] = ("[n10_full (%d site k-mers: upper bound of the absent human_sites_n10.fa), configs[1] stand-in "
                                                      "(%d site k-mers)], at %.3g and %.3g resident reads" % (nf["site_kmers"], len(sites.keys), nf["reads"], n_reads))
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    return 0


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if env_world is None:
        if args.gpus > 1:
            return launch_ranks(args, argv)                # spawn before anything here touches the GPU
    elif int(env_world) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks; refusing to report one as the other\n"
                         % (args.gpus, env_world))
        return 2
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
