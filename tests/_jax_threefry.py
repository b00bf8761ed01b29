"""``jax.random.uniform`` of the default (threefry2x32, non-partitionable) generator, re-implemented in NumPy.

The reference's MaF / LSMOP tests compare against golden values from PlatEMO on inputs drawn with
``jax.random.uniform(PRNGKey(seed), shape)`` (``tests/test_maf.py:10-11``, ``tests/test_lsmop.py:10-14``).  JAX is not
installable here; the generator is a published counter-based cipher (Threefry-2x32, 20 rounds), so the same inputs can be
regenerated and the goldens used as they stand."""
import numpy as np

_R = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, d):
    return ((x << np.uint32(d)) | (x >> np.uint32(32 - d))).astype(np.uint32)


def threefry2x32(k0, k1, x0, x1):
    k0, k1 = np.uint32(k0), np.uint32(k1)
    ks = (k0, k1, np.uint32(k0 ^ k1 ^ np.uint32(0x1BD11BDA)))
    x0 = (x0 + ks[0]).astype(np.uint32)
    x1 = (x1 + ks[1]).astype(np.uint32)
    for i in range(5):
        for r in _R[i % 2]:
            x0 = (x0 + x1).astype(np.uint32)
            x1 = _rotl(x1, r) ^ x0
        x0 = (x0 + ks[(i + 1) % 3]).astype(np.uint32)
        x1 = (x1 + ks[(i + 2) % 3] + np.uint32(i + 1)).astype(np.uint32)
    return x0, x1


def random_bits(seed: int, size: int) -> np.ndarray:
    """``size`` 32-bit words of the stream of ``PRNGKey(seed)``: the counters 0..size-1 (padded to an even count) are split
    into halves that form the two cipher words."""
    k0, k1 = np.uint32((seed >> 32) & 0xFFFFFFFF), np.uint32(seed & 0xFFFFFFFF)
    count = np.arange(size + (size % 2), dtype=np.uint32)
    if size % 2:
        count[-1] = 0
    half = count.size // 2
    with np.errstate(over="ignore"):
        y0, y1 = threefry2x32(k0, k1, count[:half], count[half:])
    return np.concatenate([y0, y1])[:size]


def uniform(seed: int, shape, minval=0.0, maxval=1.0) -> np.ndarray:
    size = int(np.prod(shape)) if len(shape) else 1
    bits = random_bits(seed, size)
    floats = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    floats = floats.reshape(shape)
    lo = np.asarray(minval, dtype=np.float32)
    hi = np.asarray(maxval, dtype=np.float32)
    return np.maximum(lo, floats * (hi - lo) + lo).astype(np.float32)


def normal(seed: int, shape) -> np.ndarray:
    """``jax.random.normal``: ``sqrt(2) * erfinv(u)`` with ``u`` uniform on ``(-1, 1)``."""
    from scipy.special import erfinv

    lo = np.nextafter(np.float32(-1.0), np.float32(1.0))
    u = uniform(seed, shape, minval=lo, maxval=1.0)
    return (np.sqrt(2.0) * erfinv(u.astype(np.float64))).astype(np.float32)
