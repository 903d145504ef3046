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
