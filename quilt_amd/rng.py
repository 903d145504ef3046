"""Counter-based uniform stream shared by the HIP kernels and the host (gibbs.hip ``stream_uniform``):
element ``i`` of stream ``seed`` is the splitmix64 finaliser of ``seed + (i + 1) * 0x9E3779B97F4A7C15``,
top 53 bits scaled to [0, 1)."""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def stream_u64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """Elements offset .. offset + n - 1 of stream ``seed`` as raw 64-bit keys."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def stream_uniform(seed: int, n: int, offset: int = 0) -> np.ndarray:
    return (stream_u64(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


# Random subsets as the device draws them (csrc/select.hip): "the m smallest keys of a stream, in key order" (ties by
# index).  Streams of one selection seed: previously selected haplotypes at offset 0, the subsample of the last rank at
# 2^20, the draw from the rest of the panel at 2^21.
SELECT_OFFSET_PREV, SELECT_OFFSET_RANK, SELECT_OFFSET_POOL = 0, 1 << 20, 1 << 21


def keyed_subset(seed: int, n: int, m: int, offset: int) -> np.ndarray:
    """Indices (into 0 .. n-1) of the m smallest keys, in key order."""
    return np.argsort(stream_u64(seed, n, offset), kind="stable")[:m]
