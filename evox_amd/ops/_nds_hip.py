"""Driver of the matrix-free non-dominated sort kernels (``csrc/nds.hip``)."""
from __future__ import annotations

from typing import Optional

import torch

from .. import _native
from ._util import check, hip, stream_of

P, I = _native.P, _native.I
_native.register_hip_api({
    "evox_hip_nds_count": (I, [P, _native.I64, I, I, P, I, P]),
    "evox_hip_nds_round": (I, [P, _native.I64, I, I, P, P, P, P, P, P, I, P]),
    "evox_hip_nds_rounds": (I, [P, _native.I64, I, I, P, P, P, P, P, P, I, I, P]),
    "evox_hip_nds_small": (I, [P, _native.I64, I, I, I, P, P, P]),
    "evox_hip_nds_peel_all": (I, [P, _native.I64, I, I, P, P, I, I, I, P, P, P]),
    "evox_hip_crowding": (I, [P, P, P, I, I, P, P]),
    "evox_hip_nds_sweep_build": (I, [P, I, I, P, P, P, P]),
    "evox_hip_nds_sweep_rounds": (I, [P, P, P, P, P, I, I, P, P, P, P, P, I, I, I, P]),
    "evox_hip_nds2_scan_max": (_native.I64, []),
    "evox_hip_nds2_scan": (I, [P, P, P, I, I, P, P, P, P]),
    "evox_hip_nds2_patience_max_fronts": (I, []),
    "evox_hip_nds2_patience": (I, [P, P, I, P, P, P]),
})

MAX_OBJECTIVES = 8
_ROUNDS_PER_CHECK = 4           # rounds issued before the first host check; doubles up to _MAX_ROUNDS_PER_CHECK
_MAX_ROUNDS_PER_CHECK = 32
_GROW_BATCH_MAX_N = 16384        # beyond this a peeling round costs more than a read-back: no speculative rounds
_SMALL_MAX_N = 1024              # up to here the whole sort is one launch of one workgroup, with no read-back
_PEEL_ALL_MAX_N = 4096           # up to here one workgroup finishes a sort that the first batch of rounds did not
_PEEL_ALL_DIRECT_N = 2048        # ... and up to here it does the whole peeling (its work is a few microseconds)


def supported(x: torch.Tensor) -> bool:
    return x.is_cuda and x.dim() == 2 and 2 <= x.shape[1] <= MAX_OBJECTIVES and x.shape[0] > 0


_SWEEP_MIN_N = 49152             # from here on three objectives take the O(N log² N)-per-front path (measured crossover)
_SWEEP_ROUNDS_PER_CHECK = 8


def _sweep_min_n() -> int:
    import os

    return int(os.environ.get("EVOX_AMD_NDS_SWEEP_MIN", str(_SWEEP_MIN_N)))


def _sweep_m2() -> bool:
    import os

    return os.environ.get("EVOX_AMD_NDS_SWEEP_M2", "0") == "1"


def _finish(rk64, host_sizes, target, cur, order, with_info):
    """Common tail of the peeling drivers: the front at which the target was reached closes the ranking, everything
    later collapses into one index."""
    cut = next((k for k, c in enumerate(host_sizes) if c >= target), cur - 1)
    while cut > 0 and host_sizes[cut] == host_sizes[cut - 1]:
        cut -= 1                                            # trailing empty rounds
    last = cut + 1
    rk64 = torch.where((rk64 < 0) | (rk64 > cut), torch.full_like(rk64, last), rk64)
    out = torch.empty_like(rk64)
    out[order] = rk64
    if with_info:
        n_before = int(host_sizes[cut - 1]) if cut > 0 else 0
        return out, (cut, n_before, int(host_sizes[cut]) - n_before)
    return out


def rank_sweep(x: torch.Tensor, until: Optional[int] = None, with_info: bool = False):
    """``rank`` for two or three objectives without the O(N²) pair tests (``csrc/nds_sweep.hip``): lexicographic
    sort, one representative per group of identical points, a static merge-sort structure over the second objective,
    and per front one segmented prefix-minimum pass plus a logarithmic look-up per point.  Returns None when a row
    holds a NaN (the caller then takes the pair-test path, whose NaN behaviour is the documented one)."""
    n, m = x.shape
    dev = x.device
    xf = x.float()
    if m == 2:
        xf = torch.cat([xf, torch.zeros(n, 1, device=dev)], dim=1)
    from . import sort as _sort

    order = _sort.lexsort_rows(xf)                   # three stable column sorts, last column first (csrc/radix.hip)
    xs = xf[order]
    is_rep = torch.ones(n, dtype=torch.bool, device=dev)
    is_rep[1:] = (xs[1:] != xs[:-1]).any(dim=1)
    idx = torch.arange(n, device=dev)
    # position of every point's representative: the first point of its group of identical points (group number by a
    # prefix sum, first member by a scatter-min; the tensor library's running maximum is a slow single-pass scan)
    gid = torch.cumsum(is_rep, 0) - 1
    rep_idx = torch.full((n,), n, dtype=torch.int64, device=dev).scatter_reduce_(0, gid, idx, "amin")[gid]
    has_nan = torch.isnan(xs).any().to(torch.int32).reshape(1)
    npad = max(2048, 1 << (n - 1).bit_length())
    levels = npad.bit_length() - 1
    f23 = torch.full((2, npad), float("inf"), dtype=torch.float32, device=dev)
    # NaN keys would break the total order the merge passes rely on (slots of a level left unwritten); rows with a NaN
    # make this function give up below, but the structure is still built and must stay well-formed
    f23[:, :n] = torch.nan_to_num(xs[:, 1:], nan=float("inf"), posinf=float("inf"), neginf=float("-inf")).T
    alive = torch.zeros(npad, dtype=torch.int32, device=dev)
    alive[:n] = is_rep
    gsize = torch.zeros(npad, dtype=torch.int32, device=dev)
    gsize.index_add_(0, rep_idx, torch.ones(n, dtype=torch.int32, device=dev))
    ordl = torch.empty((levels + 1, npad), dtype=torch.int32, device=dev)
    keyl = torch.empty((levels + 1, npad), dtype=torch.float32, device=dev)
    pos = torch.empty((levels, npad), dtype=torch.int32, device=dev)
    lib, s = hip(), stream_of(f23)
    check(lib.evox_hip_nds_sweep_build(f23[0].data_ptr(), npad, levels, ordl.data_ptr(), keyl.data_ptr(), pos.data_ptr(), s),
          "nds_sweep_build")
    chunks = npad // 2048
    pm = torch.empty((levels, npad), dtype=torch.float32, device=dev)
    agg = torch.empty(levels * chunks, dtype=torch.float32, device=dev)
    cpm = torch.empty(levels * chunks, dtype=torch.float32, device=dev)
    rk = torch.full((npad,), -1, dtype=torch.int32, device=dev)
    state = torch.zeros(2 + n + _SWEEP_ROUNDS_PER_CHECK, dtype=torch.int32, device=dev)
    target = n if until is None else max(1, min(int(until), n))
    cur = 0
    prev_ranked = -1
    while True:
        batch = min(_SWEEP_ROUNDS_PER_CHECK, n + _SWEEP_ROUNDS_PER_CHECK - cur)
        check(lib.evox_hip_nds_sweep_rounds(ordl.data_ptr(), pos.data_ptr(), f23[1].data_ptr(), alive.data_ptr(),
                                            gsize.data_ptr(), npad, levels, pm.data_ptr(), agg.data_ptr(), cpm.data_ptr(),
                                            rk.data_ptr(), state.data_ptr(), cur, batch, target, s), "nds_sweep_rounds")
        cur += batch
        host = torch.cat([state[:2 + cur], has_nan]).cpu()         # one small read-back per batch of rounds
        if int(host[-1]):
            return None
        ranked = int(host[0])
        if ranked >= target or ranked >= n or ranked == prev_ranked or cur >= n:
            break
        prev_ranked = ranked
    host_sizes = host[2:2 + cur].cumsum(0).tolist()
    rk64 = rk[:n][rep_idx].to(torch.int64)
    return _finish(rk64, host_sizes, target, cur, order, with_info)


_M2_SCAN_MIN_N = 4097             # up to 4096 points one workgroup finishes the pair-test sort just as fast


def _m2_scan() -> bool:
    import os

    return os.environ.get("EVOX_AMD_NDS_M2_SCAN", "1") == "1"


def rank_m2_scan(x: torch.Tensor, until: Optional[int] = None, with_info: bool = False):
    """``rank`` for two objectives in one launch (``csrc/nds2.hip``): the points are put in lexicographic order (one sort on a
    64-bit key) and one workgroup peels front after front — a front is the set of points whose second objective lies
    below the prefix-minimum of the remaining points in front of them (with the tie rules of dominance for equal first
    objectives), i.e. one block-wide scan per front.  Returns None when a row holds a NaN (the caller then takes the
    pair-test path)."""
    n = x.shape[0]
    dev = x.device
    xf = x.float() + 0.0                                         # -0.0 and 0.0 are one value
    from . import sort as _sort

    order = _sort.lexsort_rows(xf[:, :2])            # stable sort by the second objective, then by the first
    xs = xf[order]
    f1, f2 = xs[:, 0].contiguous(), xs[:, 1].contiguous()
    target = n if until is None else max(1, min(int(until), n))
    rk = torch.full((n,), -1, dtype=torch.int32, device=dev)
    gstart = torch.empty(n, dtype=torch.int32, device=dev)
    buf = torch.zeros(3 + n, dtype=torch.int32, device=dev)                       # fronts, ranked, NaN flag | front sizes
    buf[2] = torch.isnan(xs).any().to(torch.int32)
    check(hip().evox_hip_nds2_scan(f1.data_ptr(), f2.data_ptr(), gstart.data_ptr(), n, target, rk.data_ptr(),
                                   buf[3:].data_ptr(), buf.data_ptr(), stream_of(f2)), "nds2_scan")
    host = buf.cpu()                                                              # the one read-back of the sort
    if int(host[2]):
        return None
    cur = int(host[0])
    host_sizes = host[3:3 + cur].cumsum(0).tolist()
    return _finish(rk.to(torch.int64), host_sizes, target, cur, order, with_info)


def _m2_patience() -> bool:
    import os

    return os.environ.get("EVOX_AMD_NDS_M2_PATIENCE", "1") == "1"


def rank_m2_patience(x: torch.Tensor, until: Optional[int] = None, with_info: bool = False):
    """``rank`` for two objectives at any size (``csrc/nds2.hip::nds2_patience_kernel``): the points are put in
    lexicographic order, and a point's front is the number of fronts whose smallest (second, first) key lies below its
    own — a binary search on the fronts' tails, as for the piles of a patience sort — raised by the dominators handled in
    the same chunk of 64.  O(n log fronts), independent of the number of fronts.  All fronts are found; ``until``
    collapses the ones behind the cut afterwards.  Returns None when a row holds a NaN or there are more fronts than the
    kernel's table holds (the caller then takes another path)."""
    n = x.shape[0]
    dev = x.device
    xf = x.float() + 0.0                                         # -0.0 and 0.0 are one value
    from . import sort as _sort

    order = _sort.lexsort_rows(xf[:, :2])            # stable sort by the second objective, then by the first
    xs = xf[order]
    f1, f2 = xs[:, 0].contiguous(), xs[:, 1].contiguous()
    target = n if until is None else max(1, min(int(until), n))
    cap = int(hip().evox_hip_nds2_patience_max_fronts())
    rk = torch.zeros(n, dtype=torch.int32, device=dev)
    info = torch.zeros(3, dtype=torch.int32, device=dev)                          # fronts, overflow, NaN flag
    info[2] = torch.isnan(xs).any().to(torch.int32)
    check(hip().evox_hip_nds2_patience(f1.data_ptr(), f2.data_ptr(), n, rk.data_ptr(), info.data_ptr(), stream_of(f2)),
          "nds2_patience")
    rk64 = rk.to(torch.int64)
    sizes = torch.bincount(rk64.clamp_(0, cap - 1), minlength=cap).to(torch.int32)
    host = torch.cat([info, sizes]).cpu()                                         # the one read-back of the sort
    if int(host[2]) or int(host[1]):
        return None
    cur = int(host[0])
    host_sizes = host[3:3 + cur].cumsum(0).tolist()
    return _finish(rk64, host_sizes, target, cur, order, with_info)


def rank(x: torch.Tensor, until: Optional[int] = None, with_info: bool = False):
    """Front index of every row (int64).  With ``until`` peeling stops after the front that brings
    the number of ranked points to at least ``until``; all later points share the next index.
    ``with_info``: also return ``(cut, n_before, cut_size)`` — the index of the last front that was
    kept apart, how many points the fronts before it hold, and its own size (host integers that the
    peeling loop has read anyway)."""
    n, m = x.shape
    dev = x.device
    # two objectives have many more fronts (hundreds at this size), and a front costs the sweep a full pass: they stay
    # with the pair tests unless asked for explicitly (EVOX_AMD_NDS_SWEEP_MIN with EVOX_AMD_NDS_SWEEP_M2=1)
    if (m == 3 or (m == 2 and _sweep_m2())) and n >= _sweep_min_n():
        got = rank_sweep(x, until, with_info)
        if got is not None:
            return got
    # two objectives, every point to be ranked: the patience form, O(n log fronts) whatever the number of fronts (2e5
    # uniform points, 883 fronts: 11 ms against 23).  A partial ranking (NSGA-II keeps half of the merged population)
    # stays with the peeling paths, which stop at the target: on a converging population that is a handful of fronts
    # (NSGA-II on ZDT1 at population 1e5: 9.6 ms per generation, 11.6 with the patience form).
    whole = until is None or int(until) >= n
    if (m == 2 and n >= _M2_SCAN_MIN_N and whole and _m2_patience()
            and not torch.cuda.is_current_stream_capturing()):
        got = rank_m2_patience(x, until, with_info)
        if got is not None:
            return got
    if (m == 2 and _M2_SCAN_MIN_N <= n <= int(hip().evox_hip_nds2_scan_max()) and _m2_scan()
            and not torch.cuda.is_current_stream_capturing()):
        got = rank_m2_scan(x, until, with_info)
        if got is not None:
            return got
    if n <= _SMALL_MAX_N:
        xs = x.float()
        if xs.stride(1) != 1:
            xs = xs.contiguous()
        out = torch.empty(n, dtype=torch.int64, device=dev)
        info = torch.empty(3, dtype=torch.int32, device=dev)
        check(hip().evox_hip_nds_small(xs.data_ptr(), xs.stride(0), n, m, n if until is None else max(1, min(int(until), n)),
                                       out.data_ptr(), info.data_ptr(), stream_of(xs)), "nds_small")
        if with_info:
            cut, n_before, cut_size = info.tolist()
            return out, (cut, n_before, cut_size)
        return out
    # ascending first objective: a dominator never sits behind the point it dominates
    from . import sort as _sort

    order = _sort.argsort(x[:, 0].float())
    ld = (n + 15) // 16 * 16
    obj = torch.full((m, ld), float("inf"), dtype=torch.float32, device=dev)          # objective-major, padded columns
    obj[:, :n] = x.float()[order].T
    count = torch.zeros(n, dtype=torch.int32, device=dev)
    blocks = (n + 255) // 256
    ysplit = max(1, min(32, 1024 // blocks))
    s = stream_of(obj)
    lib = hip()
    check(lib.evox_hip_nds_count(obj.data_ptr(), ld, n, m, count.data_ptr(), ysplit, s), "nds_count")

    rk = torch.full((n,), -1, dtype=torch.int32, device=dev)
    front = torch.empty((m, ld), dtype=torch.float32, device=dev)                  # objectives of the current front
    buf = torch.zeros(2 + n + 2 * _MAX_ROUNDS_PER_CHECK, dtype=torch.int32, device=dev)
    info, sizes = buf[:2], buf[2:]                                                 # (fronts, ranked) | size of front k
    blockcnt = torch.empty(2 * blocks, dtype=torch.int32, device=dev)             # per-block counts | running totals
    target = n if until is None else min(int(until), n)
    blockend_ptr = blockcnt[blocks:].data_ptr()
    cur = 0
    batch = _ROUNDS_PER_CHECK
    while True:
        # small populations: peeling rounds are issued in growing batches between host checks, so that dozens of
        # fronts cost a handful of read-backs (a round past the target is cheap there; for large populations it
        # is not, and the batch stays at four)
        before = cur
        if n <= _PEEL_ALL_MAX_N and (cur > 0 or n <= _PEEL_ALL_DIRECT_N):
            # many fronts in a small population: three launches per round cost more than the round's work
            check(lib.evox_hip_nds_peel_all(obj.data_ptr(), ld, n, m, count.data_ptr(), rk.data_ptr(), cur,
                                            ranked if cur > 0 else 0, target, sizes.data_ptr(), info.data_ptr(), s),
                  "nds_peel_all")
            hb = buf[:2 + n].cpu()
            cur = int(hb[0])
            host = hb[2:2 + cur].cumsum(0)
            break
        check(lib.evox_hip_nds_rounds(obj.data_ptr(), ld, n, m, count.data_ptr(), rk.data_ptr(), front.data_ptr(),
                                      sizes[cur:].data_ptr(), blockcnt.data_ptr(), blockend_ptr, cur, batch, s),
              "nds_rounds")
        cur += batch
        done = sizes[:cur].cumsum(0)
        host = done.cpu()                                   # one small read-back per batch of rounds
        ranked = int(host[-1])
        if ranked >= target or cur >= n or ranked == (int(host[before - 1]) if before > 0 else 0):
            break
        if n <= _GROW_BATCH_MAX_N:
            batch = min(2 * batch, _MAX_ROUNDS_PER_CHECK)
    host_sizes = host.tolist()
    # the front at which the target was reached; everything later collapses into one index
    cut = next((k for k, c in enumerate(host_sizes) if c >= target), cur - 1)
    while cut > 0 and host_sizes[cut] == host_sizes[cut - 1]:
        cut -= 1                                            # trailing empty rounds
    rk64 = rk.to(torch.int64)
    last = cut + 1
    rk64 = torch.where((rk64 < 0) | (rk64 > cut), torch.full_like(rk64, last), rk64)
    out = torch.empty_like(rk64)
    out[order] = rk64
    if with_info:
        n_before = int(host_sizes[cut - 1]) if cut > 0 else 0
        return out, (cut, n_before, int(host_sizes[cut]) - n_before)
    return out


def crowding(vals: torch.Tensor) -> torch.Tensor:
    """Crowding distance of the rows of ``vals`` (count, m), all of one front: one stable sort per objective (the
    counting-rank launch up to 16384 members, the library's sort beyond) and one pass that reads each member's two
    neighbours."""
    from . import sort as _sort

    count, m = vals.shape
    soa = vals.float().T.contiguous()                                   # (m, count)
    order = torch.empty((m, count), dtype=torch.int64, device=vals.device)
    rnk = torch.empty((m, count), dtype=torch.int64, device=vals.device)
    lib, s = hip(), stream_of(soa)
    if count <= _sort._RANK_SORT_MAX:
        for k in range(m):
            check(lib.evox_hip_rank_sort(soa[k].data_ptr(), count, order[k].data_ptr(), rnk[k].data_ptr(), s), "rank_sort")
    else:
        # fronts beyond the counting sort's range: the radix sort per objective, order and rank together
        for k in range(m):
            if _sort.radix_sortable(soa[k]):
                order[k], rnk[k] = _sort.radix_argsort(soa[k], want_rank=True)
            else:
                order[k] = torch.argsort(soa[k], stable=True)
                rnk[k].scatter_(0, order[k], torch.arange(count, device=vals.device))
    dist = torch.empty(count, dtype=torch.float32, device=vals.device)
    check(lib.evox_hip_crowding(soa.data_ptr(), order.data_ptr(), rnk.data_ptr(), count, m, dist.data_ptr(), s),
          "crowding")
    return dist
