"""Sample sharding across GPUs / ranks: contiguous ranges, as the reference shards samples across forked
workers (``sampleRanges <- getSampleRange(N, nCores)``, QUILT/R/quilt.R:691; STITCH's getSampleRange is not
vendored -- its exact rounding is unpinned, the contract kept here is "contiguous, near-equal, in order")."""
from typing import List, Tuple


def get_sample_range(N: int, n_workers: int) -> List[Tuple[int, int]]:
    """0-based half-open ranges [start, end) per worker; workers beyond N get empty ranges."""
    n = min(N, n_workers)
    bounds = [round(i * N / n) for i in range(n + 1)] if n > 0 else [0]
    out = [(bounds[i], bounds[i + 1]) for i in range(n)]
    out += [(N, N)] * (n_workers - n)
    return out


def reduce_counts(counts, group=None):
    """Sum a rank's :class:`quilt_amd.io.SummaryCounts` over all ranks, in place (writers.R:38-47: the four per-SNP count
    arrays are the only cross-shard reduction of a run).  One all-reduce of one flat fp64 vector -- on the host (gloo) or,
    under the nccl backend (= RCCL), through a device tensor; a no-op without an initialised process group."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return counts
    v = torch.from_numpy(np.ascontiguousarray(counts.as_vector(), dtype=np.float64))
    if dist.get_backend(group) == "nccl":
        v = v.cuda()
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return counts.from_vector(v.cpu().numpy())
