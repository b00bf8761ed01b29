"""Noise handling of the evolution-strategy family.

ES gradient estimates are reductions over the population of (fitness-derived weight × noise
row).  The noise is a pure function of (key, row, column), so it is *regenerated* where it is
consumed instead of being kept in memory: a population of 4096 perturbations of a 10⁶-parameter
network would otherwise pin 16 GB per copy.

``perturb``             x_i = center + scale_i-th noise row (mirrored: rows come in ± pairs)
``weighted_noise_sum``  Σ_i w_i z_i  and optionally Σ_i v_i (z_i² − 1)   — the K18 reduction
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import _native
from ..core import random as rnd
from ._util import check, hip, on_gpu, stream_of

_native.register_hip_api({
    "evox_hip_noise_wsum": (_native.I, [_native.P, _native.P, _native.P, _native.P, _native.P, _native.P, _native.I64,
                                        _native.I64, _native.I64, _native.I64, _native.I, _native.P]),
})

_native.register_hip_api({
    "evox_hip_es_perturb": (_native.I, [_native.P, _native.P, _native.P, _native.I, _native.P, _native.I, _native.I64,
                                        _native.I64, _native.I64, _native.I64, _native.I64, _native.P]),
})

_native.register_hip_api({
    "evox_hip_es_perturb_pairs": (_native.I, [_native.P, _native.P, _native.P, _native.I, _native.P, _native.I, _native.I64,
                                              _native.I64, _native.I64, _native.I64, _native.P]),
})

_CHUNK_ELEMS = 1 << 24


def noise_rows(key, n_rows: int, dim: int, row_offset: int = 0) -> torch.Tensor:
    """Rows ``[row_offset, row_offset + n_rows)`` of the logical (·, dim) Gaussian tensor."""
    return rnd.normal(key, (n_rows, dim), offset=row_offset * dim)


def _perturb_hip(key, center, scale, n_rows, dim, row_offset, half, out_dtype):
    """One fused pass: Philox normals → scale → add → (optional bf16) store.  Rows are padded to a
    multiple of four elements so that consumers can use 16-byte loads; the returned tensor is the
    ``[:, :dim]`` view of that buffer."""
    dev = center.device
    dt = torch.float32 if out_dtype is None else out_dtype
    if dt not in (torch.float32, torch.bfloat16):
        raise ValueError("perturb writes float32 or bfloat16")
    pitch = (dim + 3) // 4 * 4
    buf = torch.empty((n_rows, pitch), dtype=dt, device=dev)
    if isinstance(scale, torch.Tensor):
        sc = scale.to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
        if sc.numel() not in (1, dim):
            raise ValueError("scale must be a scalar or have one entry per dimension")
    else:
        sc = torch.full((1,), float(scale), dtype=torch.float32, device=dev)
    check(hip().evox_hip_es_perturb(key.contiguous().data_ptr(), center.contiguous().data_ptr(), sc.data_ptr(),
                                    int(sc.numel() == dim and dim > 1), buf.data_ptr(), 1 if dt == torch.bfloat16 else 0,
                                    pitch, n_rows, dim, int(row_offset), int(half), stream_of(center)), "es_perturb")
    return buf if pitch == dim else buf[:, :dim]


def perturb(key, center, scale, n_rows: int, mirrored: bool = False, row_offset: int = 0, total_rows=None,
            out_dtype=None):
    """Population ``center + scale * z``; ``scale`` scalar or (dim,).  With ``mirrored`` the
    logical population of ``total_rows`` (default ``n_rows``) is ``[+z; −z]`` stacked and
    ``row_offset`` / ``n_rows`` address a window of it (for row-sharded sampling)."""
    dim = center.shape[0]
    total = n_rows if total_rows is None else total_rows
    if on_gpu(center) and center.dtype == torch.float32:
        out = _perturb_hip(key, center, scale, n_rows, dim, row_offset, total // 2 if mirrored else 0, out_dtype)
        if mirrored and row_offset == 0 and total == n_rows and n_rows % 2 == 0:
            # the whole mirrored population in one tensor: row half+j = 2·center − row j.  Consumers
            # that stream every row (the per-individual MLP kernel) can read the first half only.
            out._evox_mirror = (center, n_rows // 2)
        return out
    if out_dtype not in (None, torch.float32):
        return perturb(key, center, scale, n_rows, mirrored, row_offset, total_rows).to(out_dtype)
    if not mirrored:
        return center + scale * noise_rows(key, n_rows, dim, row_offset)
    half = total // 2
    lo, hi = row_offset, row_offset + n_rows
    parts = []
    if lo < half:
        e = min(hi, half)
        parts.append(center + scale * noise_rows(key, e - lo, dim, lo))
    if hi > half:
        b = max(lo, half)
        parts.append(center - scale * noise_rows(key, hi - b, dim, b - half))
    return torch.cat(parts, dim=0) if len(parts) > 1 else parts[0]


def perturb_pairs(key, center, scale, noise_start: int, count: int, out_dtype=None):
    """``[center + scale·z; center − scale·z]`` for the noise rows ``noise_start … noise_start+count−1``:
    a share of a mirrored population that keeps both members of every pair together (the returned
    tensor carries the mirrored-population mark, so pair-aware consumers read its first half only)."""
    dim = center.shape[0]
    if on_gpu(center) and center.dtype == torch.float32:
        dev = center.device
        dt = torch.float32 if out_dtype is None else out_dtype
        pitch = (dim + 3) // 4 * 4
        buf = torch.empty((2 * count, pitch), dtype=dt, device=dev)
        if isinstance(scale, torch.Tensor):
            sc = scale.to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
        else:
            sc = torch.full((1,), float(scale), dtype=torch.float32, device=dev)
        check(hip().evox_hip_es_perturb_pairs(key.contiguous().data_ptr(), center.contiguous().data_ptr(), sc.data_ptr(),
                                              int(sc.numel() == dim and dim > 1), buf.data_ptr(),
                                              1 if dt == torch.bfloat16 else 0, pitch, int(noise_start), int(count), dim,
                                              stream_of(center)), "es_perturb_pairs")
        out = buf if pitch == dim else buf[:, :dim]
        out._evox_mirror = (center, count)
        return out
    z = noise_rows(key, count, dim, noise_start)
    out = torch.cat([center + scale * z, center - scale * z], dim=0)
    return out if out_dtype in (None, torch.float32) else out.to(out_dtype)


def weighted_noise_sum_reference(key, w, dim: int, v=None, row_offset: int = 0):
    """``(Σ_i w_i z_i, Σ_i v_i (z_i² − 1) or None)`` over noise rows ``row_offset + i``."""
    n = w.shape[0]
    g = torch.zeros(dim, dtype=torch.float32, device=w.device)
    h = torch.zeros(dim, dtype=torch.float32, device=w.device) if v is not None else None
    step = max(1, _CHUNK_ELEMS // max(dim, 1))
    for s in range(0, n, step):
        e = min(n, s + step)
        z = noise_rows(key, e - s, dim, row_offset + s)
        g += w[s:e] @ z
        if v is not None:
            h += v[s:e] @ (z * z - 1)
    return g, h


def weighted_noise_sum(key, w, dim: int, v=None, row_offset: int = 0):
    if on_gpu(w) and _has("evox_hip_noise_wsum"):
        n = w.shape[0]
        dev = w.device
        w = w.contiguous().float()
        vv = v.contiguous().float() if v is not None else None
        # rows are split over blockIdx.y so that narrow problems still fill the chip; partial
        # sums land in a [splits][dim] slab and are added in a fixed order
        col_threads = (dim + 3) // 4
        # (a thread regenerates ~200 instructions of noise per row: with few columns the rows must be spread thin — 8 rows
        # per thread took 75 -> 15 us for 2048 rows of a 1.5 k-parameter policy)
        splits = max(1, min(512, (256 * 256 * 4) // max(col_threads, 1), n // 8 if n >= 8 else 1))
        slab = torch.empty((2 if v is not None else 1, splits, dim), dtype=torch.float32, device=dev)
        check(hip().evox_hip_noise_wsum(key.contiguous().data_ptr(), w.data_ptr(), vv.data_ptr() if vv is not None else None,
                                        slab[0].data_ptr(), slab[1].data_ptr() if v is not None else None, None,
                                        n, dim, int(row_offset), splits, 0, stream_of(w)), "noise_wsum")
        red = slab.sum(1)
        return red[0], (red[1] if v is not None else None)
    return weighted_noise_sum_reference(key, w, dim, v, row_offset)


def _has(symbol: str) -> bool:
    try:
        getattr(hip(), symbol)
        return True
    except AttributeError:
        return False


def cast_population(pop: torch.Tensor, dtype) -> torch.Tensor:
    """``pop.to(dtype)`` that keeps the mirrored-population mark of ``perturb`` (a storage-precision
    cast leaves row half+j = 2·center − row j true up to the rounding of the stored rows, which is
    what a consumer of reduced-precision parameters accepts anyway)."""
    if pop.dim() == 2 and pop.is_cuda and pop.shape[1] % 4:
        # keep rows padded to a multiple of four elements (vector loads in the consumers)
        pitch = (pop.shape[1] + 3) // 4 * 4
        buf = torch.empty((pop.shape[0], pitch), dtype=dtype, device=pop.device)
        out = buf[:, :pop.shape[1]]
        out.copy_(pop)
    else:
        out = pop.to(dtype)
    tag = getattr(pop, "_evox_mirror", None)
    if tag is not None:
        out._evox_mirror = tag
    return out
