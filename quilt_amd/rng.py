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


# ---------------------------------------------------------------------------------------------------------------------------
# The per-chain host stream: every draw get_and_impute_one_sample makes on the R side (functions.R:579-586, :746, the
# uniforms handed to the native calls) for one (seed, sample, Gibbs sample).  R's Mersenne-Twister cannot be reproduced
# without R, so the draws are DEFINED on the counter stream above -- rules a C++ host (csrc/impute.cpp, ``ChainStream``
# there) and numpy evaluate identically, bit for bit:
#   key        splitmix64 finalisers over (seed, sample, chain), see chain_key
#   element i  stream_u64(key, 1, i): the stream is consumed front to back, ``ctr`` counts the elements taken
#   uniform    top 53 bits / 2^53
#   integers(lo, hi)            lo + floor(uniform * (hi - lo))   (the product is exact in double for spans <= 2^63)
#   choice(n, m, replace=False) the indices of the m smallest of the next n keys (ties by index), i.e. keyed_subset
#   choice(n, size, p)          per draw the number of cumulative masses (normalised) <= uniform
# ---------------------------------------------------------------------------------------------------------------------------

def _mix64(z: int) -> int:
    z &= 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def chain_key(seed: int, i_sample: int, i_chain: int) -> int:
    """Key of the stream of (seed, global sample index, Gibbs sample 1 .. nGibbsSamples + 1)."""
    k = _mix64(int(seed) + 0x9E3779B97F4A7C15)
    k = _mix64(k ^ ((int(i_sample) + 1) * 0xD1342543DE82EF95))
    return _mix64(k ^ ((int(i_chain) + 1) * 0x2545F4914F6CDD1D))


class ChainStream:
    """The subset of numpy's Generator interface the drivers use, on the counter stream of one chain."""

    def __init__(self, seed: int, i_sample: int, i_chain: int):
        self.key = chain_key(seed, i_sample, i_chain)
        self.ctr = 0

    def _u64(self, n: int) -> np.ndarray:
        out = stream_u64(self.key, n, self.ctr)
        self.ctr += n
        return out

    def random(self, size=None):
        n = 1 if size is None else int(size)
        u = (self._u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return float(u[0]) if size is None else u

    def integers(self, low: int, high: int, size=None):
        span = float(int(high) - int(low))
        u = self.random(1 if size is None else size)
        v = np.floor(u * span).astype(np.int64) + int(low)
        return int(v[0]) if size is None else v

    def choice(self, n: int, size: int, replace: bool = True, p=None) -> np.ndarray:
        if not replace:
            if p is not None:
                raise ValueError("choice without replacement is unweighted here")
            idx = np.argsort(self._u64(int(n)), kind="stable")[: int(size)]
            return idx.astype(np.int64)
        cdf = np.cumsum(np.asarray(p if p is not None else np.full(int(n), 1.0 / int(n)), dtype=np.float64))
        cdf = cdf / cdf[-1]
        return np.minimum(np.searchsorted(cdf, self.random(int(size)), side="right"), int(n) - 1).astype(np.int64)
