"""Counter-based random numbers (Philox4x32-10) with explicit keys.

A key is an ``int32`` tensor of shape ``(2,)`` (or ``(..., 2)`` for a batch of keys) that lives
on the device it is used on; drawing never touches the host, so every function here can be
recorded into a HIP graph.  Element ``i`` of a draw is ``philox(key, counter = i // 4)[i % 4]``
— a pure function of ``(key, i)``:

* the same key gives the same numbers on the CPU and on the GPU, bit for bit, including the
  Gaussian (the Box–Muller transform is evaluated with the same fused-multiply-add sequence on
  both sides);
* ``offset=`` draws a window of a larger logical tensor, so a rank that owns rows
  ``[r0, r1)`` of an ``(N, D)`` population draws exactly those rows
  (``offset = r0 * D``) and the union over ranks equals the single-process draw.

This module replaces ``jax.random`` (split / uniform / normal / choice / randint /
permutation / cauchy / geometric) used throughout the reference.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Union

import torch

from .. import _native

Shape = Union[int, Sequence[int]]


def _shape(shape: Shape):
    if isinstance(shape, int):
        return (shape,)
    return tuple(int(s) for s in shape)


def _stream(t: torch.Tensor) -> int:
    from ..ops._util import stream_of

    return stream_of(t)


def PRNGKey(seed: int, device=None) -> torch.Tensor:
    """Key from an integer seed (64 bits are used)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    hi, lo = (seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF

    def s32(v):
        return v - (1 << 32) if v >= (1 << 31) else v

    # decorrelate nearby seeds with one keyed Philox evaluation on the host
    k = torch.tensor([s32(hi), s32(lo)], dtype=torch.int32)
    out = torch.empty(2, dtype=torch.int32)
    _native.cpu().evox_cpu_fold_in(k.data_ptr(), out.data_ptr(), 1, 0)
    return out.to(device) if device is not None else out


key = PRNGKey


def _check_key(k: torch.Tensor):
    if not isinstance(k, torch.Tensor) or k.dtype != torch.int32 or k.shape[-1:] != (2,):
        raise TypeError(f"expected an int32 key tensor of shape (..., 2), got {type(k)} {getattr(k, 'shape', None)}")


def split(key: torch.Tensor, num: int = 2) -> torch.Tensor:
    """``num`` statistically independent child keys: shape ``key.shape[:-1] + (num, 2)``."""
    _check_key(key)
    k = key.contiguous()
    nkeys = k.numel() // 2
    out = torch.empty(k.shape[:-1] + (num, 2), dtype=torch.int32, device=k.device)
    if k.is_cuda:
        _native.check(_native.hip().evox_hip_split(k.data_ptr(), out.data_ptr(), nkeys, num, _stream(k)), "split")
    else:
        _native.cpu().evox_cpu_split(k.data_ptr(), out.data_ptr(), nkeys, num)
    return out


def split_nested(key: torch.Tensor, num: int, which: int, sub: int) -> torch.Tensor:
    """``split(key, num)`` followed by ``split(children[which], sub)``, as one ``(num + sub, 2)`` tensor — on the GPU one
    launch instead of two (a generation of a small algorithm is a dozen launches, key handling several of them)."""
    _check_key(key)
    if key.is_cuda and key.numel() == 2:
        k = key.contiguous()
        out = torch.empty((num + sub, 2), dtype=torch.int32, device=k.device)
        _native.check(_native.hip().evox_hip_split_nested(k.data_ptr(), out.data_ptr(), int(num), int(which), int(sub), _stream(k)),
                      "split_nested")
        return out
    first = split(key, num)
    return torch.cat([first, split(first[which], sub)], dim=0)


def fold_in(key: torch.Tensor, data: int) -> torch.Tensor:
    """Derive a key from ``key`` and an integer (e.g. a generation or rank number)."""
    _check_key(key)
    k = key.contiguous()
    out = torch.empty_like(k)
    if k.is_cuda:
        _native.check(_native.hip().evox_hip_fold_in(k.data_ptr(), out.data_ptr(), k.numel() // 2, int(data), _stream(k)), "fold_in")
    else:
        _native.cpu().evox_cpu_fold_in(k.data_ptr(), out.data_ptr(), k.numel() // 2, int(data))
    return out


def _single(key: torch.Tensor) -> torch.Tensor:
    _check_key(key)
    if key.dim() != 1:
        raise ValueError("this draw takes a single key of shape (2,); split a batch of keys and loop, or fold_in")
    return key.contiguous()


def bits(key: torch.Tensor, shape: Shape = (), offset: int = 0) -> torch.Tensor:
    """Uniform random 32-bit words, returned as ``int64`` in ``[0, 2**32)``."""
    k = _single(key)
    shape = _shape(shape)
    n = math.prod(shape)
    raw = torch.empty(n, dtype=torch.int32, device=k.device)
    if k.is_cuda:
        _native.check(_native.hip().evox_hip_random_bits(k.data_ptr(), raw.data_ptr(), n, offset, _stream(k)), "bits")
    else:
        _native.cpu().evox_cpu_random_bits(k.data_ptr(), raw.data_ptr(), n, offset)
    return (raw.to(torch.int64) & 0xFFFFFFFF).reshape(shape)


def uniform(key: torch.Tensor, shape: Shape = (), dtype=torch.float32, minval: float = 0.0, maxval: float = 1.0,
            offset: int = 0) -> torch.Tensor:
    """Uniform on ``[minval, maxval)`` (24-bit resolution)."""
    k = _single(key)
    shape = _shape(shape)
    n = math.prod(shape)
    out = torch.empty(n, dtype=torch.float32, device=k.device)
    lo, hi = float(minval), float(maxval)
    if k.is_cuda:
        _native.check(_native.hip().evox_hip_uniform(k.data_ptr(), out.data_ptr(), n, offset, lo, hi, _stream(k)), "uniform")
    else:
        _native.cpu().evox_cpu_uniform(k.data_ptr(), out.data_ptr(), n, offset, lo, hi)
    out = out.reshape(shape)
    return out if dtype == torch.float32 else out.to(dtype)


def normal(key: torch.Tensor, shape: Shape = (), dtype=torch.float32, mean: float = 0.0, std: float = 1.0,
           offset: int = 0) -> torch.Tensor:
    """Gaussian draws (Box–Muller on Philox bits)."""
    k = _single(key)
    shape = _shape(shape)
    n = math.prod(shape)
    if k.is_cuda and dtype == torch.bfloat16:
        # drawn in fp32 and rounded in the same kernel: identical to ``normal(...).to(bfloat16)``
        out = torch.empty(n, dtype=torch.bfloat16, device=k.device)
        _native.check(_native.hip().evox_hip_normal_bf16(k.data_ptr(), out.data_ptr(), n, offset, float(mean), float(std),
                                                         _stream(k)), "normal_bf16")
        return out.reshape(shape)
    out = torch.empty(n, dtype=torch.float32, device=k.device)
    if k.is_cuda:
        _native.check(_native.hip().evox_hip_normal(k.data_ptr(), out.data_ptr(), n, offset, float(mean), float(std), _stream(k)), "normal")
    else:
        _native.cpu().evox_cpu_normal(k.data_ptr(), out.data_ptr(), n, offset, float(mean), float(std))
    out = out.reshape(shape)
    return out if dtype == torch.float32 else out.to(dtype)


def randint(key: torch.Tensor, shape: Shape, minval: int, maxval: int, dtype=torch.int64, offset: int = 0) -> torch.Tensor:
    """Integers in ``[minval, maxval)``."""
    k = _single(key)
    shape = _shape(shape)
    n = math.prod(shape)
    lo, hi = int(minval), int(maxval)
    if hi <= lo:
        raise ValueError("randint needs maxval > minval")
    if hi - lo > 0xFFFFFFFF:
        raise ValueError("randint range must fit in 32 bits")
    out = torch.empty(n, dtype=torch.int64, device=k.device)
    if k.is_cuda:
        _native.check(_native.hip().evox_hip_randint(k.data_ptr(), out.data_ptr(), n, offset, lo, hi, _stream(k)), "randint")
    else:
        _native.cpu().evox_cpu_randint(k.data_ptr(), out.data_ptr(), n, offset, lo, hi)
    out = out.reshape(shape)
    return out if dtype == torch.int64 else out.to(dtype)


def bernoulli(key: torch.Tensor, p=0.5, shape: Shape = ()) -> torch.Tensor:
    if isinstance(p, torch.Tensor) and shape == ():
        shape = p.shape
    return uniform(key, shape) < p


def permutation(key: torch.Tensor, x: Union[int, torch.Tensor], axis: int = 0) -> torch.Tensor:
    """Random permutation of ``range(x)`` or of tensor ``x`` along ``axis`` (argsort of random
    64-bit words: ties have probability ~ n² / 2⁶⁴)."""
    k = _single(key)
    n = x if isinstance(x, int) else x.shape[axis]
    w = bits(k, (2 * n,))
    order = torch.argsort((w[:n] << 31) ^ w[n:], stable=True)
    if isinstance(x, int):
        return order
    return x.index_select(axis, order)


def batched_permutation(key: torch.Tensor, batch: int, n: int) -> torch.Tensor:
    """``batch`` independent permutations of ``range(n)`` as a ``(batch, n)`` tensor."""
    k = _single(key)
    w = bits(k, (2, batch, n))
    return torch.argsort((w[0] << 31) ^ w[1], dim=1, stable=True)


def choice(key: torch.Tensor, a: Union[int, torch.Tensor], shape: Shape = (), replace: bool = True,
           p: Optional[torch.Tensor] = None, axis: int = 0) -> torch.Tensor:
    """Random sample from ``range(a)`` or from tensor ``a`` along ``axis``.

    Weighted sampling uses inverse-CDF search with replacement and Gumbel top-k (exponential
    race) without replacement."""
    k = _single(key)
    shape = _shape(shape)
    m = math.prod(shape)
    n = a if isinstance(a, int) else a.shape[axis]
    dev = k.device
    if p is None:
        if replace:
            idx = randint(k, (m,), 0, n)
        else:
            if m > n:
                raise ValueError("cannot take more samples than the population without replacement")
            idx = permutation(k, n)[:m]
    else:
        p = p.to(dev, torch.float32)
        if replace:
            cdf = torch.cumsum(p, 0)
            u = uniform(k, (m,)) * cdf[-1]
            idx = torch.searchsorted(cdf, u, right=True).clamp_(max=n - 1)
        else:
            if m > n:
                raise ValueError("cannot take more samples than the population without replacement")
            e = -torch.log1p(-uniform(k, (n,)))            # Exp(1)
            idx = torch.topk(-(e / p.clamp_min(1e-38)), m).indices
    idx = idx.reshape(shape)
    if isinstance(a, int):
        return idx
    return a.index_select(axis, idx.reshape(-1)).reshape(a.shape[:axis] + shape + a.shape[axis + 1:])


def cauchy(key: torch.Tensor, shape: Shape = (), loc: float = 0.0, scale: float = 1.0) -> torch.Tensor:
    u = uniform(key, shape)
    return loc + scale * torch.tan(math.pi * (u - 0.5))


def exponential(key: torch.Tensor, shape: Shape = ()) -> torch.Tensor:
    return -torch.log1p(-uniform(key, shape))


def geometric(key: torch.Tensor, p, shape: Shape = ()) -> torch.Tensor:
    """Number of Bernoulli(p) trials up to and including the first success (support 1, 2, …)."""
    if isinstance(p, torch.Tensor) and shape == ():
        shape = p.shape
    u = uniform(key, shape)
    lp = torch.log1p(-torch.as_tensor(p, dtype=torch.float32, device=u.device))
    return torch.floor(torch.log1p(-u) / lp).to(torch.int64) + 1


def rademacher(key: torch.Tensor, shape: Shape = (), dtype=torch.float32) -> torch.Tensor:
    return ((bits(key, shape) & 1) * 2 - 1).to(dtype)


def categorical(key: torch.Tensor, logits: torch.Tensor, shape: Shape = ()) -> torch.Tensor:
    """Samples from ``softmax(logits)`` over the last axis via the Gumbel-max trick."""
    shape = _shape(shape) if shape != () else ()
    full = shape + tuple(logits.shape)
    g = -torch.log(exponential(key, full).clamp_min(1e-38))
    return torch.argmax(logits + g, dim=-1)


def choice_distinct(key: torch.Tensor, n: int, batch: int, k: int, exclude: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``(batch, k)`` indices in ``[0, n)``, distinct within each row and (optionally) different
    from ``exclude[row]`` — exact uniform sampling without replacement in O(batch · k²), for the
    small ``k`` of DE-style parent selection (a permutation per row would cost O(batch · n))."""
    k0 = _single(key)
    taken = 0 if exclude is None else 1
    if k + taken > n:
        raise ValueError("not enough distinct individuals")
    # j-th pick: uniform over the n - taken - j remaining values, then shifted past the
    # already taken values in ascending order
    u = bits(k0, (batch, k))
    dev = k0.device
    picks = []
    base = [] if exclude is None else [exclude.to(dev).reshape(batch).to(torch.int64)]
    for j in range(k):
        m = n - taken - j
        r = (u[:, j] * m) >> 32                      # multiply-shift: uniform in [0, m)
        prev = base + picks
        if prev:
            srt = torch.sort(torch.stack(prev, dim=1), dim=1).values
            for c in range(srt.shape[1]):
                r = r + (r >= srt[:, c]).to(torch.int64)
        picks.append(r)
    return torch.stack(picks, dim=1)
