#!/usr/bin/env python3
"""bench.py -- bases/s through the ntsmCount hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the count path over the whole resident workload (BASELINE.json configs[1]:
1e9 synthetic 150 bp reads against the 96287-site hs_n10_like set, k = 19), inputs already in HBM.
N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns its own 1e9 reads
(weak scaling, configs[3]); each step ends with one RCCL SUM of the per-k-mer count vector + totals.

Prints ONE JSON line (rank 0).  `roofline` is for the count kernel: algorithmic bytes =
(L + 8) / L per base over the HIP-event launch time; `cpu_baseline` is the CPU
restatement of the reference (oracle/ntsm_oracle) timed on a bounded sample of the same reads.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAFFIC_FILE = "r02_traffic.json"   # written by tools/make_traffic.py from the PMC passes of tools/profile.sh
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy ceiling)
SITES_SEED, N_SITES, READ_SEED, READ_LEN, K = 20241218, 96287, 7, 150, 19


def cpu_baseline(synth, sites_path, n_reads, n_files=1):
    """Time the reference's CPU path on the GPU box's host cores: the compiled reference (oracle/_ref/ref_ntsmCount,
    kind "reference") when this checkout has it, else the plain-C restatement (oracle/ntsm_oracle, kind "port").
    FASTQ in, scan only (site-table build excluded); n_files > 1 = the reference's own parallelism, -t n_files over
    n_files files (src/FingerPrint.hpp:47)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_ntsmCount")
    port = os.path.join(ROOT, "oracle", "ntsm_oracle")
    with tempfile.TemporaryDirectory() as d:
        files = []
        for i in range(n_files):
            fq = os.path.join(d, "sample%d.fq" % i)
            synth.write_fastq(fq, i * n_reads, n_reads)
            files.append(fq)
        if os.path.exists(ref):
            kind = "reference"
            p = subprocess.run([ref, "-s", sites_path, "-t", str(n_files)] + files, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                               env=dict(os.environ, NTSM_REF_TIME_SCAN="1"))
        elif os.path.exists(port) and n_files == 1:
            kind = "port"
            p = subprocess.run([port, "-s", sites_path, "--time-scan", files[0]], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        else:
            return None
    secs = bases = None
    for line in p.stderr.decode().split("\n"):
        f = line.split()
        if line.startswith("SCAN_SECONDS"):
            secs = float(f[1])
            if len(f) > 3:
                bases = int(f[3])
        elif line.startswith("Total Bases Considered:"):
            bases = int(f[-1])
    if not secs or not bases:
        return None
    return {"value": bases / secs, "unit": "bases/s", "cores": n_files, "kind": kind,
            "sample": "%d x %d reads of the same synthetic stream as FASTQ (%d bases, %.1f s scan, table build excluded)"
                      % (n_files, n_reads, bases, secs)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=float, default=float(os.environ.get("NTSM_BENCH_READS", 1e9)),
                    help="reads per GPU (default 1e9 = BASELINE.json configs[1])")
    ap.add_argument("--cpu-sample-reads", type=int, default=3_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--filter-log2", type=int, default=0)
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto (minimizer-blocked kernel for k=19), 1 generic, 3 tabulated")
    ap.add_argument("--no-check", action="store_true", help="skip the sum-of-pieces correctness check of the timed result")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ntsm_amd
    from ntsm_amd.dist import merge_counts, shard_range

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the count path has no CPU fallback")
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

    def run_step():
        ctx.count_resident(d_bases.data_ptr(), n_bytes, 0, n_reads)
        if use_dist:
            merge_counts(ctx)

    # Correctness of the timed launch at full size (byte offsets far beyond 2^32): the same resident stream counted in
    # read-aligned pieces of < 2 GiB on a second context, each piece re-based so that its offsets are small.  The timed
    # context must reproduce these totals and per-k-mer counts exactly (checked after the timed region).
    expect = None
    if not args.no_check and not use_dist:
        ref = ntsm_amd.Context(sites.keys, k=K, device=local)
        piece = 13_000_000 // 16 * 16                     # 1.96 GB; a multiple of 16 reads keeps the piece bases 16-byte aligned
        for r0 in range(0, n_reads, piece):
            m = min(piece, n_reads - r0)
            ref.count_resident(d_bases.data_ptr() + r0 * synth.stride, m * synth.stride, 0, m)
        tr = ref.sync()
        expect = (tr.total_kmers, tr.total_hits, ref.counts())
        ref.close()

    for _ in range(args.warmup):
        run_step()
    ctx.sync()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
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
    totals = ctx.sync()
    checked = False
    if expect is not None:
        reps = args.steps + args.warmup
        assert (totals.total_kmers, totals.total_hits) == (reps * expect[0], reps * expect[1]), \
            "timed launches disagree with the sum over < 2 GiB pieces: %r vs %d x %r" % ((totals.total_kmers, totals.total_hits), reps, expect[:2])
        assert (ctx.counts() == expect[2] * reps).all(), "per-k-mer counts of the timed launches differ from the sum over pieces"
        checked = True

    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        value = world * bases_per_step * args.steps / elapsed
        launch_s = kernel_ms / 1e3 / max(n_launch, 1)
        bytes_per_base = (READ_LEN + 8) / READ_LEN
        achieved = bases_per_step * bytes_per_base / launch_s / 1e9
        # PMC-derived constants (separate rocprofv3 --pmc passes, tools/profile.sh + tools/make_traffic.py): bytes per launch,
        # VALU share, L2 request rate.  They are tied to the kernel sources they were measured on: after any change to those
        # files they are reported as null until the profile has been taken again.
        traffic = valu_busy = l2_frac = None
        traffic_note = "null: no PMC profile of the current kernel sources under profiles/"
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from make_traffic import kernel_source_sha16
            tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))
            if tj.get("kernel_source_sha16") == kernel_source_sha16() and args.kernel in (0, 2):
                traffic = tj["traffic_bytes_per_base"] * bases_per_step
                valu_busy, l2_frac = tj.get("valu_busy_frac"), tj.get("l2_request_rate_frac_of_cap")
                traffic_note = ("fabric-side bytes/launch from FETCH_SIZE+WRITE_SIZE (profiles/%s): L2 misses of filter/table "
                                "served by the Infinity Cache + the stream; not HBM re-reads" % TRAFFIC_FILE)
            else:
                traffic_note = "null: profiles/%s was measured on other kernel sources (or another kernel variant)" % TRAFFIC_FILE
        except Exception:
            pass
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
                                      else "one GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": traffic_note,
                         "achieved_stream_only_GBs": bases_per_step * (READ_LEN + 1) / READ_LEN / launch_s / 1e9,
                         "kernel": {0: "ntsm_count_mz_kernel", 2: "ntsm_count_mz_kernel", 3: "ntsm_count_tab19_kernel"}.get(args.kernel, "ntsm_count_kernel"),
                         "launches": n_launch, "avg_launch_ms": 1e3 * launch_s,
                         "algorithmic_bytes_per_base": bytes_per_base,
                         "valu_busy_frac_from_pmc": valu_busy,     # share of SIMD issue cycles on VALU
                         "l2_request_rate_frac_of_cap_from_pmc": l2_frac,   # the resource that binds: L2 requests/s over the measured 266 G/s cap
                         "kmer_probe_rate_per_s": totals.total_kmers / max(args.steps + args.warmup, 1) / launch_s},
            "check": {"total_kmers_per_step": totals.total_kmers // (args.steps + args.warmup) if world == 1 else None,
                      "total_hits_per_step": totals.total_hits // (args.steps + args.warmup) if world == 1 else None,
                      "equals_sum_of_pieces_below_2GiB": checked},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(synth, sites_path, args.cpu_sample_reads)
            if cb:
                out["cpu_baseline"] = cb
                out["gpu_over_cpu"] = value / cb["value"]
                if cb["kind"] == "reference":       # the reference's best case: -t N over N files
                    n_thr = max(2, min(32, (os.cpu_count() or 2) // 2))
                    mt = cpu_baseline(synth, sites_path, max(1, args.cpu_sample_reads // n_thr), n_thr)   # same total work
                    if mt:
                        out["cpu_baseline_threads"] = mt
        print(json.dumps(out))
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
