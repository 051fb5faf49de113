"""Multi-GPU plumbing: reads shard across ranks with no data-path collective; the only exchange is
one SUM all-reduce of the dense per-k-mer count vector + 4 totals at the end (RCCL over xGMI when the
backend is "nccl").  SUM, not MAX: per-site maxima are taken afterwards from the summed per-k-mer
counts, which is what one reference run over all reads computes (src/FingerPrint.hpp:281-294);
ntsmEval's merge sums maxima instead (src/CompareCounts.hpp:646-657) and is not equal to a single run."""
import time

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_items for this rank."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum_(vec, group=None):
    """In-place SUM over ranks of an int64 vector (uint64 counts reinterpreted: wrap-around is identical).  With a process
    group initialised the collective is issued whatever the world size: one rank summing with itself is still a real
    RCCL call (bench.py's NTSM_FORCE_DIST leg relies on that)."""
    assert vec.dtype == torch.int64
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    return vec


class _DeviceVector:
    """Zero-copy view of the library-owned device vector for torch (CUDA array interface v2)."""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i8", "data": (ptr, False), "version": 2}


def merge_counts(ctx, group=None, times=None):
    """Job-wide counts/totals on every rank: gather dense counts on the device, all-reduce, import.
    `times` (optional list of three floats) accumulates host seconds of the three parts: [0] waiting for this rank's own
    count kernels + the dense gather, [1] the all-reduce until it has completed on the device, [2] the import."""
    t0 = time.perf_counter()
    ptr, n_words = ctx.counts_device()
    t1 = time.perf_counter()
    vec = torch.as_tensor(_DeviceVector(ptr, n_words), device="cuda")
    allreduce_sum_(vec, group)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ctx.import_reduced()
    if times is not None:
        times[0] += t1 - t0
        times[1] += t2 - t1
        times[2] += time.perf_counter() - t2
    return n_words


# ---------------------------------------------------------------------------------------------------------
# Proof of a merge (bench.py's N > 1 line): the job-wide expectation is formed over a route that shares nothing with the
# collective being checked -- CPU tensors through a gloo group, totals and digests by all_gather_object -- and the merged
# view every rank holds must equal reps x that.
# ---------------------------------------------------------------------------------------------------------
def job_expectation(rank, n_reads, kmers, hits, counts, gloo_group):
    """(sum of kmers, sum of hits, host-side SUM of the ranks' count vectors as uint64 numpy, [per-rank info dicts]).
    `counts`: this rank's expected per-k-mer counts for ONE pass over its shard (uint64 numpy), from a kernel other than the
    one whose merge is being checked."""
    import hashlib
    import numpy as np
    world = dist.get_world_size(gloo_group)
    vec = torch.from_numpy(np.ascontiguousarray(counts, dtype=np.uint64).view(np.int64).copy())
    dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=gloo_group)
    infos = [None] * world
    dist.all_gather_object(infos, {"rank": rank, "reads": int(n_reads), "first_read": rank * int(n_reads), "kmers": int(kmers), "hits": int(hits),
                                   "counts_sha256": hashlib.sha256(np.ascontiguousarray(counts, dtype=np.uint64).tobytes()).hexdigest()}, group=gloo_group)
    return sum(r["kmers"] for r in infos), sum(r["hits"] for r in infos), vec.numpy().view(np.uint64), infos


def check_merged(rank, world, reps, merged, merged_counts, expect_job, reads_per_rank, bases_per_rank):
    """Raises AssertionError unless the merged view (total_kmers, total_hits, total_bases, reads_consumed) + per-k-mer counts a
    rank holds after `reps` merged passes equals reps x the job-wide expectation of job_expectation()."""
    import numpy as np
    kmers, hits, bases, reads = merged
    assert (kmers, hits) == (reps * expect_job[0], reps * expect_job[1]), \
        "rank %d: merged totals %r differ from %d x the host-side (gloo) sum of the ranks' expectations %r" % (rank, (kmers, hits), reps, tuple(expect_job[:2]))
    assert reads == reps * world * reads_per_rank and bases == reps * world * bases_per_rank, \
        "rank %d: merged read / base totals are not those of %d ranks" % (rank, world)
    assert np.array_equal(np.asarray(merged_counts, dtype=np.uint64), expect_job[2] * np.uint64(reps)), \
        "rank %d: merged per-k-mer counts differ from the host-side (gloo) sum of the ranks' expected counts" % rank
    return True


# ---------------------------------------------------------------------------------------------------------
# Ordered -m early stop across ranks
#
# The reference stops after the first read at which the cumulative number of site-k-mer hits exceeds
# m_maxCounts (src/FingerPrint.hpp:473-488: checked after every whole read, strict '>'); which read that is
# depends on the order of the reads, so a multi-GPU run has to fix one.  Here the global order is: super-batch
# 0, 1, 2, ...; inside a super-batch rank 0's shard, then rank 1's, ...  Per super-batch:
#   1. every rank counts its shard WITHOUT the check and notes the hits it added            (no waiting)
#   2. one all-gather of those W integers
#   3. no crossing (hits so far + sum <= max): done.  Otherwise r* = first rank whose inclusive prefix exceeds
#      max: ranks > r* take their shard out again (sign = -1, exact), rank r* takes its shard out and counts it
#      again ARMED with the budget that was left when its shard began -- the library finds the exact read --,
#      ranks < r* keep theirs.  Everybody stops.
# The result (counts, totals, number of reads consumed) equals one context consuming the same reads in the same
# order with the same threshold; the cost without a crossing is one tiny all-gather per super-batch.
# ---------------------------------------------------------------------------------------------------------
def all_gather_int(value, group=None):
    """[value of rank 0, rank 1, ...] (non-negative integers below 2^63)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [int(value)]
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, mine, group=group)
    return [int(t.item()) for t in out]


class ContextEngine:
    """The three operations OrderedEarlyStop needs, on a GPU context.  A shard is (d_bases_ptr, n_bytes,
    d_read_end_ptr, n_reads): a flat stream resident in device memory (include/ntsm_hip.h)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def count(self, shard):
        ptr, n_bytes, ends, n_reads = shard
        self.ctx.set_max_hits(0, armed=False)            # also drops a merged (job-wide) view: totals are this context's own again
        before = self.ctx.sync().total_hits
        self.ctx.count_resident(ptr, n_bytes, ends, n_reads, 1)
        return self.ctx.sync().total_hits - before

    def undo(self, shard):
        ptr, n_bytes, ends, n_reads = shard
        self.ctx.count_resident(ptr, n_bytes, ends, n_reads, -1)

    def recount_armed(self, shard, budget):
        """Count the shard again, stopping after the first read at which the shard's own hits exceed budget;
        returns the number of reads consumed."""
        ptr, n_bytes, ends, n_reads = shard
        t0 = self.ctx.sync()
        self.ctx.set_max_hits(t0.total_hits + budget, armed=True)
        self.ctx.count_resident(ptr, n_bytes, ends, n_reads, 1)
        return self.ctx.sync().reads_consumed - t0.reads_consumed


class OrderedEarlyStop:
    """Deterministic global -m stop for reads sharded over ranks; see the protocol above.  `all_gather` maps this
    rank's integer to the list over ranks (default: torch.distributed); `engine` is a ContextEngine or anything
    with the same three methods (tests use a CPU engine)."""

    def __init__(self, engine, max_hits, rank=0, all_gather=None, group=None):
        self.engine, self.max_hits, self.rank = engine, int(max_hits), rank
        self.all_gather = all_gather if all_gather is not None else (lambda v: all_gather_int(v, group))
        self.hits_before = 0              # hits counted globally in the super-batches consumed so far
        self.stopped = False
        self.stop_rank = None
        self.reads_consumed = 0           # by this rank

    def step(self, shard, n_reads):
        """Consume one super-batch (this rank's shard of it).  Returns True once the threshold has tripped;
        later calls are ignored like reads after the stop in the reference (src/FingerPrint.hpp:66)."""
        if self.stopped:
            return True
        mine = self.engine.count(shard) if n_reads else 0
        per_rank = self.all_gather(mine)
        total = self.hits_before + sum(per_rank)
        if total <= self.max_hits:                       # strict '>' like the reference
            self.hits_before = total
            self.reads_consumed += n_reads
            return False
        acc = self.hits_before
        for r, h in enumerate(per_rank):
            if acc + h > self.max_hits:
                self.stop_rank = r
                break
            acc += h
        if self.rank < self.stop_rank:
            self.reads_consumed += n_reads
        elif self.rank > self.stop_rank:
            if n_reads:
                self.engine.undo(shard)
        else:
            self.engine.undo(shard)
            self.reads_consumed += self.engine.recount_armed(shard, self.max_hits - acc)
        self.stopped = True
        return True
