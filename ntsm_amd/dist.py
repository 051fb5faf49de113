"""Multi-GPU plumbing: reads shard across ranks with no data-path collective; the only exchange is
one SUM all-reduce of the dense per-k-mer count vector + 4 totals at the end (RCCL over xGMI when the
backend is "nccl").  SUM, not MAX: per-site maxima are taken afterwards from the summed per-k-mer
counts, which is what one reference run over all reads computes (src/FingerPrint.hpp:281-294);
ntsmEval's merge sums maxima instead (src/CompareCounts.hpp:646-657) and is not equal to a single run."""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_items for this rank."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum_(vec, group=None):
    """In-place SUM over ranks of an int64 vector (uint64 counts reinterpreted: wrap-around is identical)."""
    assert vec.dtype == torch.int64
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    return vec


class _DeviceVector:
    """Zero-copy view of the library-owned device vector for torch (CUDA array interface v2)."""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i8", "data": (ptr, False), "version": 2}


def merge_counts(ctx, group=None):
    """Job-wide counts/totals on every rank: gather dense counts on the device, all-reduce, import."""
    ptr, n_words = ctx.counts_device()
    vec = torch.as_tensor(_DeviceVector(ptr, n_words), device="cuda")
    allreduce_sum_(vec, group)
    torch.cuda.synchronize()
    ctx.import_reduced()
