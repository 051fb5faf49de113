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
