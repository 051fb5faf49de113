"""N > 1 path on CPU: two gloo ranks shard the reads, count their shard (with the oracle, the only
CPU counter there is), SUM-all-reduce the per-k-mer vector through ntsm_amd.dist, and must equal a
single run over all reads.  Also shows why the merge is SUM and not MAX."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "inputs")


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_binding import OracleFP
    import ntsm_amd
    from ntsm_amd.dist import allreduce_sum_, shard_range
    bases, ends, _ = ntsm_amd.flatten_file(os.path.join(G, "reads2k.fq"))
    lo, hi = shard_range(len(ends), rank, world)
    start = 0 if lo == 0 else int(ends[lo - 1]) + 1
    fp = OracleFP(os.path.join(G, "sites200.fa"))
    fp.process_flat(bases[start:int(ends[hi - 1]) + 1], ends[lo:hi] - np.uint64(start))
    cnt = fp.kmers()[2]
    vec = torch.from_numpy(np.concatenate([cnt, np.array([fp.total_kmers, fp.total_hits, fp.total_bases, hi - lo],
                                                         dtype=np.uint64)]).view(np.int64).copy())
    mx = vec.clone()
    allreduce_sum_(vec)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((vec.numpy().view(np.uint64).copy(), mx.numpy().view(np.uint64).copy()))
    dist.destroy_process_group()


def test_two_rank_sum_merge_equals_single_run(built):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import OracleFP
    import ntsm_amd
    from ntsm_amd.dist import shard_range
    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert shard_range(5, 7, 8) == (5, 5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, maxed = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bases, ends, _ = ntsm_amd.flatten_file(os.path.join(G, "reads2k.fq"))
    fp = OracleFP(os.path.join(G, "sites200.fa"))
    fp.process_flat(bases, ends)
    cnt = fp.kmers()[2]
    assert np.array_equal(merged[:-4], cnt)
    assert list(merged[-4:]) == [fp.total_kmers, fp.total_hits, fp.total_bases, len(ends)]
    assert not np.array_equal(maxed[:-4], cnt)          # a MAX merge is NOT a single run
    # the merged vector formats to the reference's recorded counts.txt
    sites = ntsm_amd.Sites(os.path.join(G, "sites200.fa"))
    rc, text = sites.format_counts(merged[:-4], int(merged[-4]))
    assert rc == 0 and text == open(os.path.join(ROOT, "tests", "golden", "expected", "tiny_k19.stdout"), "rb").read()


# ---- ordered -m stop across ranks (ntsm_amd.dist.OrderedEarlyStop) on CPU: the protocol with the oracle as engine ----
class _OracleEngine:
    """count / undo / recount_armed of ntsm_amd.dist.ContextEngine, done with the CPU oracle (tests only)."""

    def __init__(self, sites_path):
        from oracle_binding import OracleFP
        self.OracleFP, self.sites_path = OracleFP, sites_path
        self.vec, self.tot, self.last = None, np.zeros(3, dtype=np.int64), None

    def _run(self, shard, budget=None):
        bases, ends = shard
        fp = self.OracleFP(self.sites_path)
        buf, start, n = bases.tobytes(), 0, 0
        for e in ends.tolist():
            fp.process(buf[start:e])
            start, n = e + 1, n + 1
            if budget is not None and fp.total_hits > budget:
                break
        return fp.kmers()[2].astype(np.int64), np.array([fp.total_kmers, fp.total_hits, fp.total_bases], dtype=np.int64), n

    def _add(self, v, t, sign):
        self.vec = sign * v if self.vec is None else self.vec + sign * v
        self.tot = self.tot + sign * t

    def count(self, shard):
        v, t, _ = self._run(shard)
        self.last = (v, t)
        self._add(v, t, 1)
        return int(t[1])

    def undo(self, shard):
        self._add(self.last[0], self.last[1], -1)

    def recount_armed(self, shard, budget):
        v, t, n = self._run(shard, budget)
        self._add(v, t, 1)
        return n


def _slice(bases, ends, lo, hi):
    if hi <= lo:
        return (bases[:0], ends[:0])
    start = 0 if lo == 0 else int(ends[lo - 1]) + 1
    return (bases[start:int(ends[hi - 1]) + 1], ends[lo:hi] - np.uint64(start))


def _ordered_worker(rank, world, port, max_hits, per_super, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ntsm_amd
    from ntsm_amd.dist import OrderedEarlyStop, allreduce_sum_, shard_range
    bases, ends, _ = ntsm_amd.flatten_file(os.path.join(G, "reads2k.fq"))
    eng = _OracleEngine(os.path.join(G, "sites200.fa"))
    stop = OrderedEarlyStop(eng, max_hits, rank=rank)
    for s0 in range(0, len(ends), per_super):
        n_sb = min(per_super, len(ends) - s0)
        lo, hi = shard_range(n_sb, rank, world)
        if stop.step(_slice(bases, ends, s0 + lo, s0 + hi), hi - lo):
            break
    vec = eng.vec if eng.vec is not None else np.zeros(1, dtype=np.int64)
    out = torch.from_numpy(np.concatenate([vec, eng.tot, np.array([stop.reads_consumed], dtype=np.int64)]).copy())
    allreduce_sum_(out)
    if rank == 0:
        q.put((out.numpy().copy(), stop.stopped, stop.stop_rank))
    dist.destroy_process_group()


def test_ordered_early_stop_across_ranks_equals_single_run(built):
    """Three gloo ranks, reads in super-batches of 250 split over the ranks: the global -m stop lands on the same read,
    with the same counts and totals, as one reference-equivalent run over the reads in the same order."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import OracleFP
    import ntsm_amd
    from ntsm_amd.dist import shard_range
    bases, ends, _ = ntsm_amd.flatten_file(os.path.join(G, "reads2k.fq"))
    world, per_super = 3, 250
    # the order the protocol defines: super-batch by super-batch, rank shards in rank order == file order here
    full = OracleFP(os.path.join(G, "sites200.fa"))
    full.process_flat(bases, ends)
    for frac in (0.37, 0.5, 0.93):
        target = int(full.total_hits * frac)
        ref = OracleFP(os.path.join(G, "sites200.fa"), cov=2.0 * (target + 0.5) / full.n_distinct)
        assert ref.max_hits == target
        ref.process_flat(bases, ends)
        assert ref.early_term and 0 < ref.reads_processed < len(ends)
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 31500 + (os.getpid() + int(frac * 100)) % 2000
        procs = [ctx.Process(target=_ordered_worker, args=(r, world, port, target, per_super, q)) for r in range(world)]
        for p in procs:
            p.start()
        merged, stopped, stop_rank = q.get(timeout=180)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert stopped and stop_rank is not None
        assert np.array_equal(merged[:-4].astype(np.uint64), ref.kmers()[2])
        assert list(merged[-4:]) == [ref.total_kmers, ref.total_hits, ref.total_bases, ref.reads_processed]


def test_bench_self_launch_spawns_one_child_per_rank():
    """`python bench.py --gpus N` with no launcher starts N fresh child processes itself (before anything in the parent
    touches the GPU) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; --dry-launch makes every child print its rank
    environment and leave before the first GPU call, so the real spawn path runs here on the CPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "4", "--warmup", "2", "--dry-launch"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-800:]
    rows = sorted((json.loads(l) for l in p.stdout.decode().split("\n") if l.startswith("{")), key=lambda r: r["rank"])
    assert [r["rank"] for r in rows] == [0, 1, 2] and [r["local_rank"] for r in rows] == [0, 1, 2]
    assert all(r["world_size"] == 3 and r["gpus"] == 3 and r["master_addr"] == "127.0.0.1" for r in rows)
    assert len({r["master_port"] for r in rows}) == 1 and len({r["pid"] for r in rows}) == 3 and len({r["ppid"] for r in rows}) == 1
    # a launcher (torch.distributed.run) whose world size differs from --gpus is refused: no rank runs, no JSON line
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert q.returncode != 0 and q.stdout == b"" and b"refusing" in q.stderr
    # more GPUs than the node has (none here): refused before any child is started
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode != 0 and r.stdout == b"" and b"refusing" in r.stderr
    # under torch.distributed.run the ranks come from the launcher and bench.py does not spawn again
    port = 29700 + os.getpid() % 200
    t = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert t.returncode == 0, t.stderr.decode()[-800:]
    rows = [json.loads(l) for l in t.stdout.decode().split("\n") if l.startswith("{")]
    assert sorted(r["rank"] for r in rows) == [0, 1] and all(r["world_size"] == 2 for r in rows)


def test_rccl_binding_resolves_every_symbol_ntsm_allreduce_calls(built):
    """ntsm_allreduce binds RCCL with dlopen on first use (ntsm_hip.hip: struct Rccl).  No box with two GPUs has run it
    yet, so the binding itself is checked here: ntsm_rccl_probe() performs exactly that dlopen + five dlsym calls and
    needs no GPU; and the five names exist in the RCCL this image ships."""
    import ctypes
    import ntsm_amd
    assert ntsm_amd.capi.H.ntsm_rccl_probe() == 0
    lib = None
    for name in ("librccl.so.1", "librccl.so"):
        try:
            lib = ctypes.CDLL(name)
            break
        except OSError:
            pass
    assert lib is not None
    for sym in ("ncclCommInitAll", "ncclGroupStart", "ncclGroupEnd", "ncclAllReduce", "ncclCommDestroy"):
        assert hasattr(lib, sym), sym


def test_bench_helpers_pigz_like_and_gpu_count(tmp_path, monkeypatch):
    """bench.py's tooling that runs without a GPU: the pigz-style writer of the e2e_cli_gz leg produces ONE gzip member that
    gzip / zlib read back byte for byte (blocks joined by sync flushes, CRC-32 combined from the blocks'), and the launcher's
    device count comes from sysfs and the *_VISIBLE_DEVICES lists, never from HIP."""
    import gzip
    import zlib
    sys.path.insert(0, ROOT)
    import bench
    data = os.urandom(70_000) + b"@r1\nACGTACGTAGCTAGCTAGCATCGATCGATCGATCAGCTAGCTAGCTAC\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" * 9000
    src, dst = str(tmp_path / "x.fq"), str(tmp_path / "x.fq.gz")
    open(src, "wb").write(data)
    for block, threads in ((50_000, 4), (1 << 20, 2), (len(data), 1)):
        n = bench.pigz_like(src, dst, block=block, threads=threads)
        blob = open(dst, "rb").read()
        assert n == len(blob) and gzip.decompress(blob) == data
        d = zlib.decompressobj(31)
        assert d.decompress(blob) == data and d.eof and d.unused_data == b""     # one member, nothing behind it
    for a, b in ((b"", b"xyz"), (b"abc", b""), (b"abc", b"defgh"), (os.urandom(1000), os.urandom(77777))):
        assert bench._crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    assert bench.count_gpus_without_hip() == 0                                   # this container has no /dev/kfd
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(os.path, "exists", lambda p: True if p == "/dev/kfd" else os.path.lexists(p))
    import glob as _glob
    real = _glob.glob
    monkeypatch.setattr(_glob, "glob", lambda pat: ["/dev/dri/renderD128", "/dev/dri/renderD129", "/dev/dri/renderD130"] if pat.startswith("/dev/dri") else ([] if "kfd" in pat else real(pat)))
    assert bench.count_gpus_without_hip() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.count_gpus_without_hip() == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "")                               # an empty list hides every device
    assert bench.count_gpus_without_hip() == 0


def _proof_worker(rank, world, port, q, corrupt_rank):
    """One rank of bench.py's merge proof with the oracle as the counting engine: `reps` passes over the rank's shard, merged
    by allreduce_sum_ over the default group (the route under test: RCCL on the GPU box, gloo here), checked against
    job_expectation() over a SEPARATE gloo group + check_merged().  corrupt_rank >= 0: that rank's contribution to the merge is
    off by one in one counter -- every rank must notice."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    side = dist.new_group(backend="gloo")
    from oracle_binding import OracleFP
    import ntsm_amd
    from ntsm_amd.dist import allreduce_sum_, check_merged, job_expectation
    bases, ends, _ = ntsm_amd.flatten_file(os.path.join(G, "reads2k.fq"))
    per = 700                                              # equal shards (weak scaling: every rank owns `per` reads)
    lo, hi = rank * per, (rank + 1) * per
    start = 0 if lo == 0 else int(ends[lo - 1]) + 1
    shard, shard_ends = bases[start:int(ends[hi - 1]) + 1], ends[lo:hi] - np.uint64(start)
    fp = OracleFP(os.path.join(G, "sites200.fa"))
    fp.process_flat(shard, shard_ends)
    one = (fp.total_kmers, fp.total_hits, fp.kmers()[2].copy())
    expect = job_expectation(rank, per, one[0], one[1], one[2], side)
    reps, shard_bases = 3, int(shard.size) - per
    local = np.concatenate([one[2] * np.uint64(reps), np.array([one[0] * reps, one[1] * reps, shard_bases * reps, per * reps], dtype=np.uint64)])
    if rank == corrupt_rank:
        local[5] += np.uint64(1)
    vec = torch.from_numpy(local.view(np.int64).copy())
    allreduce_sum_(vec)
    m = vec.numpy().view(np.uint64)
    try:
        ok = check_merged(rank, world, reps, (int(m[-4]), int(m[-3]), int(m[-2]), int(m[-1])), m[:-4], expect[:3], per, shard_bases)
        q.put((rank, "ok" if ok else "?", [r["counts_sha256"] for r in expect[3]], int(expect[0])))
    except AssertionError as e:
        q.put((rank, "caught: " + str(e)[:60], None, None))
    dist.destroy_process_group()


def test_bench_merge_proof_under_gloo(built):
    """bench.py's N > 1 self-check (ntsm_amd.dist.job_expectation + check_merged) with two CPU ranks: a correct SUM merge of
    three passes passes on both ranks with the ranks' digests gathered; a merge in which one rank's vector is off by one in
    one counter is caught on BOTH ranks."""
    ctx = mp.get_context("spawn")
    for corrupt in (-1, 1):
        q = ctx.Queue()
        port = 31500 + (os.getpid() + 7 * (corrupt + 2)) % 2000
        procs = [ctx.Process(target=_proof_worker, args=(r, 2, port, q, corrupt)) for r in range(2)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        if corrupt < 0:
            assert [g[1] for g in got] == ["ok", "ok"] and got[0][2] == got[1][2] and len(set(got[0][2])) == 2 and got[0][3] > 0, got
        else:
            assert all(g[1].startswith("caught: rank %d: merged per-k-mer counts differ" % g[0]) for g in got), got
