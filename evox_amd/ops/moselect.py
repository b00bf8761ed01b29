"""One-at-a-time removal / ranking loops of indicator and density based environmental selection
(IBEA removal, SPEA2 truncation, SRA stochastic ranking).

After each step the scores of the survivors change, so the loop length is inherent.  Native
implementations run the whole loop in one call: C++ on the host (``csrc/moselect_cpu.cpp``), one
persistent single-workgroup launch on the GPU (``csrc/moselect.hip``).  The ``*_reference``
functions are straightforward PyTorch versions used as oracles."""
from __future__ import annotations

import os

import torch

from .. import _native
from ._util import check, hip, on_gpu, stream_of
from .distance import cdist as _cdist

P, I, I64, F = _native.P, _native.I, _native.I64, _native.F
_native.register_cpu_api({
    "evox_cpu_ibea_remove": (None, [P, P, P, F, I64, I64, P]),
    "evox_cpu_spea2_truncate": (None, [P, I64, I64, P]),
    "evox_cpu_stochastic_ranking": (None, [P, P, P, F, I64, P]),
    "evox_cpu_stochastic_ranking_fresh": (None, [P, P, P, F, I64, P]),
})
_native.register_hip_api({
    "evox_hip_ibea_remove": (I, [P, P, P, F, I, I, P, P]),
    "evox_hip_spea2_truncate": (I, [P, I, I, I, P, P, P, P]),
    "evox_hip_stochastic_ranking": (I, [P, P, P, F, I, P, P, P]),
    "evox_hip_stochastic_ranking_fresh": (I, [P, P, P, F, I, P, P, P]),
    "evox_hip_stochastic_ranking_fresh_dev": (I, [P, P, P, P, I, P, P, P]),
    "evox_hip_stochastic_ranking_packed": (I, [P, P, P, F, P, I, P, P]),
    "evox_hip_stochastic_ranking_window_ws": (_native.I64, [I]),
    "evox_hip_stochastic_ranking_window": (I, [P, P, P, F, P, I, P, P, I, I, I, P]),
    "evox_hip_spea2_fitness": (I, [P, I, I, P, P, P, P]),
    "evox_hip_crowd_remove": (I, [P, I, I, P, P, P, P]),
    "evox_hip_ibea_fitness": (I, [P, I, I, F, P, P, P]),
    "evox_hip_ibea_remove_obj_max": (I64, [I]),
    "evox_hip_sra_indicators": (I, [P, I, I, I, P, P, P]),
    "evox_hip_bige_crowding": (I, [P, P, I, I, P, P, P]),
    "evox_hip_scatter_best": (I, [P, P, P, I64, I, I, P, P, P, P]),
    "evox_hip_ref_associate": (I, [P, I, P, I, I, P, P, P, P]),
    "evox_hip_knn_dist": (I, [P, I, I, I, P, P]),
    "evox_hip_knee_round": (I, [P, P, I, I, P, P, P]),
    "evox_hip_knee_state_round": (I, [P, P, I, I, P, I, P, P, I, P]),
    "evox_hip_knee_state_tail": (I, [P, P, I, P, I, P, P]),
    "evox_hip_ibea_remove_obj": (I, [P, I, I, P, F, P, I, P, P]),
    "evox_hip_ibea_remove_multi_max": (I64, [I]),
    "evox_hip_ibea_remove_obj_multi": (I, [P, I, I, P, F, P, I, P, P, P, I, P]),
    "evox_hip_ibea_lazy_max": (I64, [I]),
    "evox_hip_ibea_lazy_rem_floats": (I64, [I]),
    "evox_hip_ibea_lazy_ctrl_words": (I64, []),
    "evox_hip_ibea_lazy_rounds": (I, [P, I, I, P, F, P, I, P, P, P, I, I, I, P]),
    "evox_hip_ibea_lazy_cand_words": (I64, []),
    "evox_hip_spea2_lazy_max": (I64, [I]),
    "evox_hip_spea2_lazy_ctrl_words": (I64, []),
    "evox_hip_spea2_lazy_scratch_words": (I64, [I]),
    "evox_hip_spea2_lazy_rounds": (I, [P, I, I, P, P, P, I, P, P, I, I, P]),
})


def ibea_remove_reference(fitness, I_mat, C, kappa, n_remove: int) -> torch.Tensor:
    n = fitness.shape[0]
    alive = torch.ones(n, dtype=torch.bool, device=fitness.device)
    fit = fitness.clone()
    inf = torch.tensor(float("inf"), device=fitness.device)
    ar = torch.arange(n, device=fit.device)
    for _ in range(int(n_remove)):
        x = torch.argmin(torch.where(alive, fit, inf))
        fit = fit + torch.exp(-I_mat[x, :] / C[x] / kappa)
        alive = alive & (ar != x)
    return alive


def ibea_remove(fitness, I_mat, C, kappa, n_remove: int) -> torch.Tensor:
    """IBEA: repeatedly drop the individual with the smallest fitness and add back the indicator
    loss it caused to everyone else.  Returns the boolean survivor mask."""
    n = fitness.shape[0]
    fit = fitness.detach().clone().contiguous().float()
    I_mat, C = I_mat.contiguous().float(), C.contiguous().float()
    alive = torch.empty(n, dtype=torch.uint8, device=fit.device)
    if on_gpu(fit):
        check(hip().evox_hip_ibea_remove(fit.data_ptr(), I_mat.data_ptr(), C.data_ptr(), float(kappa), n, int(n_remove),
                                         alive.data_ptr(), stream_of(fit)), "ibea_remove")
    else:
        _native.cpu().evox_cpu_ibea_remove(fit.data_ptr(), I_mat.data_ptr(), C.data_ptr(), float(kappa), n,
                                           int(n_remove), alive.data_ptr())
    return alive.bool()


def ibea_streaming_supported(obj) -> bool:
    return on_gpu(obj) and obj.dim() == 2 and obj.dtype == torch.float32 and 2 <= obj.shape[1] <= 8


def ibea_fitness(obj, kappa):
    """``(F, C)`` of IBEA's additive ε-indicator fitness for normalised objectives ``obj`` (n, m):
    ``C_j = max_i |I(i, j)|``, ``F_j = 1 − Σ_i exp(−I(i, j) / C_j / κ)`` with ``I(i, j) = max_k (obj_i − obj_j)_k``,
    in two streaming passes without the n × n indicator matrix."""
    n, m = obj.shape
    obj = obj.contiguous()
    C = torch.empty(n, dtype=torch.float32, device=obj.device)
    fit = torch.empty(n, dtype=torch.float32, device=obj.device)
    check(hip().evox_hip_ibea_fitness(obj.data_ptr(), n, m, float(kappa), C.data_ptr(), fit.data_ptr(), stream_of(obj)),
          "ibea_fitness")
    return fit, C


_IBEA_MULTI_MIN_N = 4096         # from here on the (opt-in) multi-workgroup removal loop applies
_IBEA_MULTI_SPIN = 2_000_000     # polls a workgroup waits for its peers before it gives up (about a second)


def _ibea_multi(n: int, m: int) -> bool:
    return (os.environ.get("EVOX_AMD_IBEA_MULTI", "0") == "1" and n > _IBEA_MULTI_MIN_N
            and n <= int(hip().evox_hip_ibea_remove_multi_max(int(m))) and not torch.cuda.is_current_stream_capturing())


_IBEA_LAZY_MIN_N = 2048          # from here on the removal runs in chip-wide rounds
_IBEA_LAZY_K = 256               # members a round looks at (``csrc/ibea_lazy.hip::kK``)
_IBEA_LAZY_CHUNK = 64             # rounds issued between two looks at the progress
_IBEA_LAZY_MIN_RATE = 6.0        # removals per round below which the one-at-a-time loop is the cheaper way


def _ibea_lazy(n: int, m: int) -> bool:
    return (os.environ.get("EVOX_AMD_IBEA_LAZY", "1") == "1" and n >= _IBEA_LAZY_MIN_N
            and n <= int(hip().evox_hip_ibea_lazy_max(int(m))) and not torch.cuda.is_current_stream_capturing())


def ibea_remove_streaming_fits(n: int, m: int) -> bool:
    return (int(n) <= int(hip().evox_hip_ibea_remove_obj_max(int(m))) or _ibea_lazy(int(n), int(m))
            or _ibea_multi(int(n), int(m)))


def _remove_bad_first(n_remove: int, bad: torch.Tensor, n_bad: int, run_good) -> torch.Tensor:
    """Survivor mask of a removal among members some of which hold non-finite values (``bad``).  No round can select
    such a member and none counts it as removed, so a mask taken from the rounds would have the wrong size in either
    direction.  They are the worst members by definition and go first (highest index first when more are bad than must
    go); the rest of the removal is done among the good members by ``run_good(index, count)``."""
    keep = ~bad
    if n_bad >= n_remove:
        idx = torch.nonzero(bad).squeeze(1)
        keep[idx[:n_bad - n_remove]] = True               # more bad members than removals: the first ones stay
        return keep
    good = torch.nonzero(keep).squeeze(1)
    keep[good] = run_good(good, n_remove - n_bad)
    return keep


def ibea_remove_rounds(obj, fit, C, kappa, n_remove: int, stats: dict = None, flags: int = 0,
                       min_rate: float = None) -> torch.Tensor:
    """The removal loop in rounds (``csrc/ibea_lazy.hip``): fitness values only grow while members are removed, so the
    256 members with the smallest fitness can be played through on one wave for as long as their smallest stays below
    the 257th of the round's start, and the increments of everything removed are then added to the whole population by a
    chip-wide launch.  The survivors are those of the one-at-a-time loop.  The number of rounds depends on the data
    (≈ removals / 150 on converging populations): rounds are issued in chunks and the count of removed members is read
    back after each chunk (later rounds of a chunk are no-ops once the removal is complete).  A chunk that removed fewer
    than ``min_rate`` members per round hands the rest to the one-at-a-time loop.  ``flags``: 1 switches the sentinel
    rounds off, 2 the candidate lists (tests)."""
    n, m = obj.shape
    n_remove = int(n_remove)
    work = fit.detach().clone().contiguous()
    if n_remove <= 0:
        return work < float("inf")
    obj = obj.contiguous()
    C = C.contiguous()
    bad = ~(torch.isfinite(fit) & torch.isfinite(C) & torch.isfinite(obj).all(1))
    n_bad = int(bad.sum())
    if n_bad:
        return _remove_bad_first(n_remove, bad, n_bad, lambda good, k: ibea_remove_streaming(
            obj[good], fit[good].contiguous(), C[good].contiguous(), kappa, k))
    ctrl = torch.empty(int(hip().evox_hip_ibea_lazy_ctrl_words()), dtype=torch.int32, device=obj.device)
    rem = torch.empty(int(hip().evox_hip_ibea_lazy_rem_floats(m)), dtype=torch.float32, device=obj.device)
    cand = torch.empty(int(hip().evox_hip_ibea_lazy_cand_words()), dtype=torch.int32, device=obj.device)
    rounds = min(_IBEA_LAZY_CHUNK, max(4, int(n_remove / (0.6 * _IBEA_LAZY_K)) + 2))
    reset, done, worked, issued, fallback, alive = 1, 0, 0, 0, False, None
    while True:
        check(hip().evox_hip_ibea_lazy_rounds(obj.data_ptr(), n, m, C.data_ptr(), float(kappa), work.data_ptr(), n_remove,
                                              ctrl.data_ptr(), rem.data_ptr(), cand.data_ptr(), reset, rounds, int(flags),
                                              stream_of(obj)),
              "ibea_lazy_rounds")
        issued += rounds
        state = ctrl.tolist()
        if state[0] >= n_remove or state[0] == done:        # complete, or nothing left to remove
            break
        chunk_rate = (state[0] - done) / max(1, state[1] - worked)
        done, worked, reset = state[0], state[1], 0
        if chunk_rate < (_IBEA_LAZY_MIN_RATE if min_rate is None else min_rate) and n <= int(hip().evox_hip_ibea_remove_obj_max(int(m))):
            # rounds that remove only a handful each (everyone's fitness rises together, e.g. tight clusters of
            # near-copies) cost more than the one-at-a-time loop: it finishes the job from the current fitness values
            # (only the members still there go along: the loop's cost per removal grows with the number it holds)
            left = torch.nonzero(work < float("inf")).squeeze(1)
            sub_obj, sub_c, sub_fit = obj[left].contiguous(), C[left].contiguous(), work[left].contiguous()
            sub_alive = torch.empty(left.numel(), dtype=torch.uint8, device=obj.device)
            check(hip().evox_hip_ibea_remove_obj(sub_obj.data_ptr(), left.numel(), m, sub_c.data_ptr(), float(kappa),
                                                 sub_fit.data_ptr(), n_remove - done, sub_alive.data_ptr(), stream_of(obj)),
                  "ibea_remove_obj")
            alive = torch.zeros(n, dtype=torch.uint8, device=obj.device)
            alive[left] = sub_alive
            fallback = True
            break
        per_round = max(1.0, done / max(1, state[1]))
        rounds = min(_IBEA_LAZY_CHUNK, max(4, int(1.25 * (n_remove - done) / per_round) + 1))
    if stats is not None:
        stats.update(removed=state[0], rounds=state[1], tie_rounds=state[3], issued=issued, select_us=state[4] / 100.0,
                     simulate_us=state[5] / 100.0, bisections=state[6], sentinel_rounds=state[9], candidate_rounds=state[13], fallback=fallback)
    if not fallback and state[0] < min(n_remove, n):
        raise RuntimeError(f"IBEA removal: the rounds stopped after {state[0]} of {n_remove} removals on finite inputs")
    return alive.bool() if fallback else work < float("inf")


def ibea_remove_streaming(obj, fit, C, kappa, n_remove: int) -> torch.Tensor:
    """``ibea_remove`` with the indicator rows recomputed from the (normalised) objectives held on chip.

    ``EVOX_AMD_IBEA_MULTI=1``: beyond 4096 members the loop runs on eight compute units at once
    (``csrc/ibea.hip::ibea_remove_multi_kernel``: each workgroup owns a slice of the members, the arg-min of a removal is
    agreed on through one 64-bit word per workgroup); the survivors are the same as from the single-workgroup loop, its
    waits are bounded, and should one run out the call raises instead of returning a wrong mask.  Off by default: the
    agreement costs ≈ 2.5 µs per removal on this part (device-scope traffic between workgroups leaves the die), as much
    as the single compute unit needs for the whole removal — 29.4 against 31.7 ms per generation at population 10⁴."""
    n, m = obj.shape
    if _ibea_lazy(n, m) and not _ibea_multi(n, m):
        return ibea_remove_rounds(obj, fit, C, kappa, n_remove)
    obj = obj.contiguous()
    alive = torch.empty(n, dtype=torch.uint8, device=obj.device)
    if _ibea_multi(n, m):
        slots = torch.empty(16, dtype=torch.int64, device=obj.device)
        error = torch.empty(1, dtype=torch.int32, device=obj.device)
        check(hip().evox_hip_ibea_remove_obj_multi(obj.data_ptr(), n, m, C.data_ptr(), float(kappa), fit.data_ptr(),
                                                   int(n_remove), alive.data_ptr(), slots.data_ptr(), error.data_ptr(),
                                                   _IBEA_MULTI_SPIN, stream_of(obj)), "ibea_remove_obj_multi")
        if int(error):
            raise RuntimeError("IBEA removal: a workgroup waited in vain for its peers (the GPU is shared with work that "
                               "keeps them from running together?); set EVOX_AMD_IBEA_MULTI=0")
        return alive.bool()
    check(hip().evox_hip_ibea_remove_obj(obj.data_ptr(), n, m, C.data_ptr(), float(kappa), fit.data_ptr(), int(n_remove),
                                         alive.data_ptr(), stream_of(obj)), "ibea_remove_obj")
    return alive.bool()


def crowd_remove_reference(big_r, n_remove: int) -> torch.Tensor:
    """Survivor mask after ``n_remove`` times dropping the member j with the largest crowding degree
    ``1 − Π_{i alive} big_r[i, j]`` (``big_r`` ∈ [0, 1], symmetric, diagonal 1); the products are kept as
    log-sums plus zero counts and updated in O(n) per removal."""
    zero = big_r <= 0
    logs = torch.where(zero, torch.zeros_like(big_r), torch.log(big_r.clamp_min(1e-38)))
    log_sum, zero_cnt = logs.sum(0), zero.sum(0)
    alive = torch.ones(big_r.shape[0], dtype=torch.bool, device=big_r.device)
    for _ in range(int(n_remove)):
        crowd = torch.where(zero_cnt > 0, torch.ones_like(log_sum), 1 - torch.exp(log_sum))
        crowd = torch.where(alive, crowd, torch.full_like(crowd, -1.0))
        x = torch.argmax(crowd)
        alive[x] = False
        log_sum = log_sum - logs[x]
        zero_cnt = zero_cnt - zero[x].to(zero_cnt.dtype)
    return alive


def crowd_remove(big_r, n_remove: int) -> torch.Tensor:
    """``crowd_remove_reference`` as one persistent launch on the GPU."""
    if not (on_gpu(big_r) and big_r.dtype == torch.float32) or int(n_remove) <= 0:
        return crowd_remove_reference(big_r, n_remove)
    n = big_r.shape[0]
    zero = big_r <= 0
    logs = torch.log(big_r.clamp_min(1e-38))
    log_sum = torch.where(zero, torch.zeros_like(logs), logs).sum(0)
    zero_cnt = zero.sum(0).to(torch.int32)
    logs = torch.where(zero, torch.full_like(logs, float("nan")), logs).contiguous()      # NaN marks a zero factor
    alive = torch.empty(n, dtype=torch.uint8, device=big_r.device)
    check(hip().evox_hip_crowd_remove(logs.data_ptr(), n, int(n_remove), log_sum.data_ptr(), zero_cnt.data_ptr(),
                                      alive.data_ptr(), stream_of(logs)), "crowd_remove")
    return alive.bool()


def _masked_distances(obj, mask):
    n = obj.shape[0]
    dis = _cdist(obj.float(), obj.float())
    pair = mask[:, None] & mask[None, :] & ~torch.eye(n, dtype=torch.bool, device=obj.device)
    return torch.where(pair, dis, torch.full_like(dis, float("inf"))).contiguous()


def spea2_fitness_supported(obj) -> bool:
    return on_gpu(obj) and obj.dim() == 2 and obj.dtype == torch.float32 and 2 <= obj.shape[1] <= 8


def spea2_fitness(obj) -> torch.Tensor:
    """SPEA2 fitness (strength-weighted dominator sum + 1 / (2nd-nearest-neighbour distance + 2)) of the
    rows of ``obj`` in two streaming passes; no n × n matrix is formed."""
    n, m = obj.shape
    obj = obj.contiguous()
    scratch = torch.empty((2, n), dtype=torch.float32, device=obj.device)
    fit = torch.empty(n, dtype=torch.float32, device=obj.device)
    check(hip().evox_hip_spea2_fitness(obj.data_ptr(), n, m, scratch[0].data_ptr(), scratch[1].data_ptr(), fit.data_ptr(),
                                       stream_of(obj)), "spea2_fitness")
    return fit


def spea2_truncate_reference(obj, mask, n_remove: int) -> torch.Tensor:
    n = obj.shape[0]
    dis = _masked_distances(obj, mask)
    keep = mask.clone()
    ar = torch.arange(n, device=obj.device)
    for _ in range(int(n_remove)):
        x = torch.argmin(dis.amin(1))
        gone = ar == x
        keep = keep & ~gone
        dis = torch.where(gone[:, None] | gone[None, :], torch.full_like(dis, float("inf")), dis)
    return keep


_SPEA2_LAZY_MIN_N = 2048         # from here on the truncation runs in chip-wide rounds
_SPEA2_LAZY_K = 512              # points a round plays through (``csrc/spea2_lazy.hip::kKS``)


def _spea2_lazy(n: int, m: int) -> bool:
    return (os.environ.get("EVOX_AMD_SPEA2_LAZY", "1") == "1" and _SPEA2_LAZY_MIN_N <= n
            <= int(hip().evox_hip_spea2_lazy_max(int(m))) and not torch.cuda.is_current_stream_capturing())


def spea2_truncate_rounds(points, n_remove: int, stats: dict = None) -> torch.Tensor:
    """SPEA2 truncation of ``points`` (n, m), all of them candidates, in rounds (``csrc/spea2_lazy.hip``): nearest-neighbour
    distances only grow while points are removed, so the 512 points with the smallest distance are played through on one
    wave (each with the list of its 8 nearest live neighbours, collected by 512 workgroups at once) while their smallest
    stays below the 513th of the round's start; every live point whose neighbour went is then rescanned chip-wide.  No
    n × n distance matrix is formed, and the survivors are those of the one-at-a-time loop.  Returns the survivor mask."""
    n, m = points.shape
    n_remove = int(n_remove)
    points = points.contiguous().float()
    dev = points.device
    alive = torch.ones(n, dtype=torch.uint8, device=dev)
    if n_remove <= 0:
        return alive.bool()
    bad = ~torch.isfinite(points).all(1)
    n_bad = int(bad.sum())
    if n_bad:
        return _remove_bad_first(n_remove, bad, n_bad, lambda good, k: spea2_truncate(
            points[good], torch.ones(good.numel(), dtype=torch.bool, device=dev), k))
    rowmin = torch.empty(n, dtype=torch.float32, device=dev)
    nn = torch.empty(n, dtype=torch.int32, device=dev)
    ctrl = torch.empty(int(hip().evox_hip_spea2_lazy_ctrl_words()), dtype=torch.int32, device=dev)
    scratch = torch.empty(int(hip().evox_hip_spea2_lazy_scratch_words(n)) // 2 + 1, dtype=torch.int64, device=dev)
    rounds = min(_IBEA_LAZY_CHUNK, max(4, int(n_remove / (0.35 * _SPEA2_LAZY_K)) + 2))
    reset, done, issued = 1, 0, 0
    while True:
        check(hip().evox_hip_spea2_lazy_rounds(points.data_ptr(), n, m, alive.data_ptr(), rowmin.data_ptr(), nn.data_ptr(),
                                               n_remove, ctrl.data_ptr(), scratch.data_ptr(), reset, rounds, stream_of(points)),
              "spea2_lazy_rounds")
        issued += rounds
        state = ctrl.tolist()
        if state[0] >= n_remove or state[0] == done:        # complete, or nothing left to remove
            break
        done, reset = state[0], 0
        per_round = max(1.0, done / max(1, state[1]))
        rounds = min(_IBEA_LAZY_CHUNK, max(4, int(1.25 * (n_remove - done) / per_round) + 1))
    if stats is not None:
        stats.update(removed=state[0], rounds=state[1], tie_rounds=state[8], short_rounds=state[9], issued=issued)
    if state[0] < min(n_remove, n - 1):                       # the last point has no neighbour and always stays
        raise RuntimeError(f"SPEA2 truncation: the rounds stopped after {state[0]} of {n_remove} removals on finite inputs")
    return alive.bool()


def spea2_truncate(obj, mask, n_remove: int) -> torch.Tensor:
    """SPEA2 archive truncation among the rows selected by ``mask``: repeatedly drop the point
    whose nearest remaining neighbour is closest.  Returns the survivor mask (⊂ mask)."""
    n = obj.shape[0]
    if on_gpu(obj):
        if int(n_remove) <= 0:
            return mask.clone()
        # the loop only concerns the selected rows: compact them (host-synchronising; callers know the
        # count already) so that the matrix and the per-point state cover those rows only
        idx = torch.nonzero(mask).squeeze(1)
        sub = obj[idx].float()
        nv = sub.shape[0]
        if _spea2_lazy(nv, sub.shape[1]):
            keep = torch.zeros(n, dtype=torch.bool, device=obj.device)
            keep[idx] = spea2_truncate_rounds(sub, n_remove)
            return keep
        ld = (nv + 3) // 4 * 4
        dis = torch.full((nv, ld), float("inf"), dtype=torch.float32, device=obj.device)
        dis[:, :nv] = _cdist(sub, sub)
        dis.diagonal().fill_(float("inf"))
        rowmin, nn = dis.min(1)                           # full-chip reduction; the kernel keeps it current
        nn = nn.to(torch.int32)
        alive = torch.zeros(ld, dtype=torch.uint8, device=obj.device)
        alive[:nv] = 1
        check(hip().evox_hip_spea2_truncate(dis.data_ptr(), nv, ld, int(n_remove), alive.data_ptr(), rowmin.data_ptr(),
                                            nn.data_ptr(), stream_of(dis)), "spea2_truncate")
        keep = torch.zeros(n, dtype=torch.bool, device=obj.device)
        keep[idx] = alive[:nv].bool()
        return keep
    dis = _masked_distances(obj, mask)
    keep = mask.to(torch.uint8).contiguous()
    _native.cpu().evox_cpu_spea2_truncate(dis.data_ptr(), n, int(n_remove), keep.data_ptr())
    return keep.bool()


def stochastic_ranking_reference(I1, I2, u, pc):
    n = I1.shape[0]
    rank = list(range(n))
    a, b, uu = I1.tolist(), I2.tolist(), u.tolist()
    for _ in range((n + 1) // 2):
        swapped = False
        for j in range(n - 1):
            key = a if uu[j] < pc else b
            if key[rank[j]] < key[rank[j + 1]]:
                rank[j], rank[j + 1] = rank[j + 1], rank[j]
                swapped = True
        if not swapped:
            break
    return torch.tensor(rank, dtype=torch.int64, device=I1.device)


def coin_reference(seed: int, sweep: int, position: int, pc: float) -> bool:
    """The coin of comparison (sweep, position) of :func:`stochastic_ranking` with ``seed`` — a plain-Python copy of
    ``csrc/sr_coins.h`` for tests and documentation."""
    M = 0xFFFFFFFF

    def h(x):
        x ^= x >> 16
        x = (x * 0x7FEB352D) & M
        x ^= x >> 15
        x = (x * 0x846CA68B) & M
        x ^= x >> 16
        return x

    pc7 = min(128, max(0, int(pc * 128.0 + 0.5)))
    g = position >> 2
    word = h((((seed & M) + sweep * 0x9E3779B1) & M) ^ ((g * 0x85EBCA77 + 0x165667B1) & M))
    return ((word >> (8 * (position & 3))) & 0x7F) < pc7


def _dense_ranks(v: torch.Tensor) -> torch.Tensor:
    """1-based dense ranks in ascending order: equal values share a rank, so ``<`` on ranks is ``<`` on values (NaN keys
    count as the smallest value)."""
    from . import sort as _sort

    v = torch.where(torch.isnan(v), torch.full_like(v, float("-inf")), v)
    order = _sort.argsort(v)
    sv = v[order]
    new = torch.ones_like(sv, dtype=torch.int32)
    new[1:] = (sv[1:] != sv[:-1]).to(torch.int32)
    dense = torch.cumsum(new, 0, dtype=torch.int32)
    out = torch.empty_like(dense)
    out[order] = dense
    return out


def _stochastic_ranking_packed(I1, I2, u, pc, seed, pc_dev):
    """The blocked sort on one-word entries (``csrc/moselect.hip::stochastic_ranking_packed_kernel``): both keys are
    replaced by their dense ranks, packed into one 32-bit word per element; the kernel returns the words in their final
    order, and the element behind each word is found by matching two stable sorts by word (elements whose two keys are
    both equal never pass each other in the bubble sort, so the k-th occurrence of a word is the k-th such element in
    index order).  Same permutation as the float kernels for NaN-free keys."""
    from . import sort as _sort

    n = I1.shape[0]
    packed = (_dense_ranks(I1) << 16) | _dense_ranks(I2)
    out = torch.empty(n, dtype=torch.int32, device=I1.device)
    args = (packed.data_ptr(), u.data_ptr() if u is not None else None, seed.data_ptr() if seed is not None else None,
            float(pc) if pc_dev is None else 0.5, pc_dev.data_ptr() if pc_dev is not None else None, n, out.data_ptr())
    halo = _ranking_window_halo(n)
    if halo:
        # several workgroups, each stepping a window of the systolic schedule; one launch per `halo` macro steps
        ws = torch.empty(int(hip().evox_hip_stochastic_ranking_window_ws(n)), dtype=torch.int32, device=I1.device)
        import os
        check(hip().evox_hip_stochastic_ranking_window(*args, ws.data_ptr(), halo, int(os.environ.get("EVOX_AMD_SRA_WINDOW_BLOCK", "4")),
                                                       int(os.environ.get("EVOX_AMD_SRA_WINDOW_THREADS", "256")), stream_of(I1)), "stochastic_ranking_window")
    else:
        check(hip().evox_hip_stochastic_ranking_packed(*args, stream_of(I1)), "stochastic_ranking_packed")
    by_word = _sort.radix_argsort(packed, bits=(0, 32))            # elements, by word then index
    positions = _sort.radix_argsort(out, bits=(0, 32))             # final positions, by word then position
    rank = torch.empty(n, dtype=torch.int64, device=I1.device)
    rank[positions] = by_word
    return rank


def _ranking_window_halo(n: int) -> int:
    """Halo width of the multi-workgroup form of the packed sort, 0 for the single workgroup (``EVOX_AMD_SRA_WINDOW``: a
    halo width, or 0 to switch the form off; ``EVOX_AMD_SRA_WINDOW_THREADS``: 256, 512 or 1024 logical threads per window;
    ``EVOX_AMD_SRA_WINDOW_BLOCK``: 4 or 8 entries per logical thread).  Below 8192 entries the windows would be too few
    to pay for their overlap.  A macro step costs a workgroup barrier, so small windows win although they repeat more
    work and need more launches — 20 000 entries, one coin per comparison: 30.9 ms in one workgroup; windows of 1024
    threads (halo 256) 10.7 ms, of 512 threads (halo 128) 7.7 ms, of 256 threads (halo 96: 79 workgroups, 261 launches,
    three quarters of the work redundant) 6.5 ms; 8-entry threads are slower throughout
    (`profiles/sra_rank_timing_n20000.jsonl`)."""
    import os

    halo = int(os.environ.get("EVOX_AMD_SRA_WINDOW", "96"))
    return halo if n >= 8192 and halo > 0 else 0


def _packed_ranking_applies(I1) -> bool:
    import os

    n = I1.shape[0]
    return on_gpu(I1) and 4096 < n <= 24 * 1024 and os.environ.get("EVOX_AMD_SRA_PACKED", "1") != "0"


def stochastic_ranking(I1, I2, u, pc, seed: torch.Tensor = None) -> torch.Tensor:
    """SRA's stochastic bubble sort, ⌈n/2⌉ sweeps; a comparison ranks by ``I1`` when its coin says so and by ``I2``
    otherwise; larger values move to the front.

    ``seed=None``: position j compares by ``I1`` when ``u[j] < pc`` — the same coin in every sweep, the form of the
    reference (``algorithms/mo/sra.py:22-83``); the sort stops at the first sweep without a swap.
    ``seed`` (an int32 tensor with one element, on the device of the keys; ``u`` is ignored): every comparison draws its
    own coin — a counter-based hash of (seed, sweep, position), probability resolved to 1/128 (``csrc/sr_coins.h``) —
    which is the stochastic ranking of the SRA paper; all sweeps run."""
    n = I1.shape[0]
    I1, I2 = I1.contiguous().float(), I2.contiguous().float()
    rank = torch.empty(n, dtype=torch.int64, device=I1.device)
    if seed is not None and on_gpu(I1) and isinstance(pc, torch.Tensor) and pc.is_cuda:
        # probability and seed both on the device: the call reads no host value (it can be recorded into a HIP graph)
        seed = seed.reshape(-1)[:1].to(device=I1.device, dtype=torch.int32).contiguous()
        pcd = pc.reshape(-1)[:1].to(torch.float32).contiguous()
        if _packed_ranking_applies(I1):
            return _stochastic_ranking_packed(I1, I2, None, 0.5, seed, pcd)
        scratch = torch.empty(n, dtype=torch.int32, device=I1.device) if n > 40000 else None
        check(hip().evox_hip_stochastic_ranking_fresh_dev(I1.data_ptr(), I2.data_ptr(), seed.data_ptr(), pcd.data_ptr(), n,
                                                          rank.data_ptr(), scratch.data_ptr() if scratch is not None else None,
                                                          stream_of(I1)), "stochastic_ranking_fresh")
        return rank
    pc = float(pc)
    if seed is not None:
        seed = seed.reshape(-1)[:1].to(device=I1.device, dtype=torch.int32).contiguous()
        if _packed_ranking_applies(I1):
            return _stochastic_ranking_packed(I1, I2, None, pc, seed, None)
        if on_gpu(I1):
            scratch = torch.empty(n, dtype=torch.int32, device=I1.device) if n > 40000 else None
            check(hip().evox_hip_stochastic_ranking_fresh(I1.data_ptr(), I2.data_ptr(), seed.data_ptr(), pc, n, rank.data_ptr(),
                                                          scratch.data_ptr() if scratch is not None else None, stream_of(I1)),
                  "stochastic_ranking_fresh")
        else:
            _native.cpu().evox_cpu_stochastic_ranking_fresh(I1.data_ptr(), I2.data_ptr(), seed.data_ptr(), pc, n, rank.data_ptr())
        return rank
    u = u.contiguous().float()
    if _packed_ranking_applies(I1):
        return _stochastic_ranking_packed(I1, I2, u, pc, None, None)
    if on_gpu(I1):
        scratch = torch.empty(n, dtype=torch.int32, device=I1.device) if n > 40000 else None
        check(hip().evox_hip_stochastic_ranking(I1.data_ptr(), I2.data_ptr(), u.data_ptr(), pc, n, rank.data_ptr(),
                                                scratch.data_ptr() if scratch is not None else None, stream_of(I1)),
              "stochastic_ranking")
    else:
        _native.cpu().evox_cpu_stochastic_ranking(I1.data_ptr(), I2.data_ptr(), u.data_ptr(), pc, n, rank.data_ptr())
    return rank


def ref_associate(points: torch.Tensor, refs: torch.Tensor):
    """``(group, d1, d2)``: every point's reference direction by smallest perpendicular distance ``d2`` (ties: lowest
    index), with the distance along it (``d1``) — NSGA-III's niche association and θ-DEA's clustering.  Points with a NaN
    coordinate get direction 0 and ``d2 = +inf``.  On the GPU one streaming launch (``csrc/rvea.hip::ref_assoc_kernel``);
    the tensor formulation materialises the n × n_ref cosine and distance matrices."""
    n, m = points.shape
    if on_gpu(points) and 2 <= m <= 8 and points.dtype == torch.float32 and refs.dtype == torch.float32 and n > 0:
        vn = (refs / torch.linalg.vector_norm(refs, dim=-1, keepdim=True)).contiguous()
        p = points.contiguous()
        group = torch.empty(n, dtype=torch.int64, device=p.device)
        d1 = torch.empty(n, dtype=torch.float32, device=p.device)
        d2 = torch.empty(n, dtype=torch.float32, device=p.device)
        check(hip().evox_hip_ref_associate(p.data_ptr(), n, vn.data_ptr(), vn.shape[0], m, group.data_ptr(), d1.data_ptr(),
                                           d2.data_ptr(), stream_of(p)), "ref_associate")
        return group, d1, d2
    norm = torch.linalg.vector_norm(points, dim=-1, keepdim=True)
    cosine = ((points / norm) @ (refs / torch.linalg.vector_norm(refs, dim=-1, keepdim=True)).T).clamp(-1, 1)
    dist = norm * torch.sqrt((1 - cosine ** 2).clamp_min(0))
    dist = torch.where(torch.isnan(dist), torch.full_like(dist, float("inf")), dist)
    group = torch.argmin(dist, dim=1)
    ar = torch.arange(n, device=points.device)
    return group, (norm * cosine)[ar, group], dist[ar, group]


def scatter_best(values: torch.Tensor, target: torch.Tensor, n_sub: int, current: torch.Tensor):
    """``(best, winner)`` per sub-problem: entry (row, slot) offers ``values[row, slot]`` to sub-problem
    ``target[row, slot]``; a sub-problem takes the smallest offer below ``current``, ties to the smallest row; ``winner``
    is that row or -1.  One launch with a 64-bit atomic minimum per improving offer (``csrc/moselect.hip``)."""
    v = values.contiguous()
    t = target.contiguous().to(torch.int64)
    cur = current.contiguous()
    rows, slots = v.shape
    best = torch.empty(n_sub, dtype=torch.float32, device=v.device)
    winner = torch.empty(n_sub, dtype=torch.int64, device=v.device)
    keys = torch.empty(n_sub, dtype=torch.int64, device=v.device)
    check(hip().evox_hip_scatter_best(v.data_ptr(), t.data_ptr(), cur.data_ptr(), rows * slots, slots, n_sub, best.data_ptr(),
                                      winner.data_ptr(), keys.data_ptr(), stream_of(v)), "scatter_best")
    return best, winner


def knn_distances(f: torch.Tensor, k: int) -> torch.Tensor:
    """(n, k) ascending Euclidean distances from every row of ``f`` (n, m) to its k nearest other rows; streamed from
    the objective vectors on the GPU (``csrc/ibea.hip::knn_dist_kernel``), no n × n distance matrix."""
    n, m = f.shape
    if on_gpu(f) and 2 <= m <= 8 and 1 <= k <= 8 and f.dtype == torch.float32 and n > k:
        out = torch.empty((n, k), dtype=torch.float32, device=f.device)
        fc = f.contiguous()
        check(hip().evox_hip_knn_dist(fc.data_ptr(), n, m, k, out.data_ptr(), stream_of(fc)), "knn_dist")
        return out
    from .distance import cdist as _cdist
    dis = _cdist(f, f)
    return torch.topk(dis, k + 1, dim=1, largest=False, sorted=True).values[:, 1:]


def knee_suppress(f: torch.Tensor, order: torch.Tensor, R: torch.Tensor) -> torch.Tensor:
    """KnEA's knee selection on one front: walk the points in ``order``; a point that is still selected deselects every
    later point within ``R`` of it in every objective.  Returns the boolean selection in the row order of ``f``.

    Solved as the fixed point ``keep[q] = ¬∃ p before q: close(p, q) ∧ keep[p]`` — a few fully parallel rounds instead of
    one step per point (reference: the nested loops of ``algorithms/mo/knea.py:90-123``).  On the GPU a round is one
    streaming launch over the objective vectors; the tensor formulation builds the n × n neighbourhood matrix."""
    n, m = f.shape
    if on_gpu(f) and 2 <= m <= 8 and f.dtype == torch.float32:
        fs = f[order].contiguous()
        Rc = R.to(torch.float32).contiguous()
        if not torch.cuda.is_current_stream_capturing() and os.environ.get("EVOX_AMD_KNEE", "states") == "states":
            # three states per point (undecided / candidate for good / suppressed for good): a round only looks at the
            # undecided points, whose number — read back once per round to size the next launch — falls geometrically
            state = torch.zeros(n, dtype=torch.uint8, device=f.device)
            todo = torch.arange(n, device=f.device)
            while todo.numel() > 0:
                if todo.numel() <= 1024:
                    # few enough for one workgroup: it lists each point's close undecided predecessors once and runs
                    # the remaining rounds on chip (their fate no longer depends on the decided points)
                    # (first the points that the last round's new candidates suppress are taken out)
                    check(hip().evox_hip_knee_state_round(fs.data_ptr(), Rc.data_ptr(), n, m, todo.data_ptr(), todo.numel(),
                                                          state.data_ptr(), state.data_ptr(), 1, stream_of(fs)),
                          "knee_state_round")
                    check(hip().evox_hip_knee_state_tail(fs.data_ptr(), Rc.data_ptr(), m, todo.data_ptr(), todo.numel(),
                                                         state.data_ptr(), stream_of(fs)), "knee_state_tail")
                    break
                nxt = state.clone()
                check(hip().evox_hip_knee_state_round(fs.data_ptr(), Rc.data_ptr(), n, m, todo.data_ptr(), todo.numel(),
                                                      state.data_ptr(), nxt.data_ptr(), 0, stream_of(fs)), "knee_state_round")
                state = nxt
                todo = torch.nonzero(state == 0).squeeze(1)
            keep = torch.empty(n, dtype=torch.bool, device=f.device)
            keep[order] = state == 1
            return keep
        a = torch.ones(n, dtype=torch.uint8, device=f.device)
        b = torch.empty_like(a)
        for rounds in range(1, n + 1):
            check(hip().evox_hip_knee_round(fs.data_ptr(), Rc.data_ptr(), n, m, a.data_ptr(), b.data_ptr(), stream_of(fs)),
                  "knee_round")
            # rounds past the fixed point change nothing: the (host-synchronising) test runs every fourth round
            if rounds % 4 == 0 and torch.equal(a, b):
                break
            a, b = b, a
        keep = torch.empty(n, dtype=torch.bool, device=f.device)
        keep[order] = a.to(torch.bool)
        return keep
    close = ((f[:, None, :] - f[None, :, :]).abs() < R).all(-1)
    close = close & ~torch.eye(n, dtype=torch.bool, device=f.device)
    pos = torch.empty(n, dtype=torch.int64, device=f.device)
    pos[order] = torch.arange(n, device=f.device)
    before = ((pos[:, None] < pos[None, :]) & close).to(torch.float32)        # p suppresses q
    keep = torch.ones(n, dtype=torch.bool, device=f.device)
    for rounds in range(1, n + 1):
        new = ~((keep.to(torch.float32) @ before) > 0)
        if rounds % 4 == 0 and torch.equal(new, keep):
            break
        keep = new
    return keep


def bige_crowding(normed: torch.Tensor, pr: torch.Tensor, radius: torch.Tensor) -> torch.Tensor:
    """BiGE's crowding degree (sharing function over a niche of radius ``radius``, a 0-d tensor) from the normalised
    objectives ``normed`` (n, m) and the proximity ``pr`` (n), without the n × n distance and sharing matrices
    (``csrc/ibea.hip``).  Reference: ``algorithms/mo/bige.py:26-60``."""
    n, m = normed.shape
    if on_gpu(normed) and 2 <= m <= 8 and normed.dtype == torch.float32:
        f, p = normed.contiguous(), pr.contiguous().to(torch.float32)
        rad = radius.reshape(1).to(torch.float32).contiguous()
        cd = torch.empty(n, dtype=torch.float32, device=normed.device)
        check(hip().evox_hip_bige_crowding(f.data_ptr(), p.data_ptr(), n, m, rad.data_ptr(), cd.data_ptr(), stream_of(f)),
              "bige_crowding")
        return cd
    from .distance import cdist as _cdist
    dis = _cdist(normed, normed)
    ge = (pr[:, None] >= pr[None, :]).to(torch.float32)
    gt = (pr[:, None] > pr[None, :]).to(torch.float32)
    sh = ((dis < radius) * 0.5 * ((1 + ge + gt) * (1 - dis / radius))) ** 2
    return torch.sqrt((sh.sum(1) - torch.diagonal(sh)).clamp_min(0))


def sra_indicators(obj: torch.Tensor, lower_only: bool = False):
    """SRA's indicators from the objective vectors, without the n × n matrices (``csrc/ibea.hip``):
    ``I1_j = 1 − Σ_i exp(−max_k(f_i − f_j)_k / 0.05)`` and the shift-based density
    ``I2_j = min_i ‖f_j − max(f_i, f_j)‖`` over ``i ≠ j`` — or over ``i < j`` only (``lower_only``), which is what the
    reference computes (``algorithms/mo/sra.py:94-107``)."""
    n, m = obj.shape
    if on_gpu(obj) and 2 <= m <= 8 and obj.dtype == torch.float32:
        o = obj.contiguous()
        I1 = torch.empty(n, dtype=torch.float32, device=obj.device)
        I2 = torch.empty(n, dtype=torch.float32, device=obj.device)
        check(hip().evox_hip_sra_indicators(o.data_ptr(), n, m, int(bool(lower_only)), I1.data_ptr(), I2.data_ptr(),
                                            stream_of(o)), "sra_indicators")
        return I1, I2
    I1 = torch.zeros(n, dtype=obj.dtype, device=obj.device)
    I2 = torch.full((n,), float("inf"), dtype=obj.dtype, device=obj.device)
    step = max(1, (1 << 22) // max(n * m, 1))
    idx = torch.arange(n, device=obj.device)
    for s0 in range(0, n, step):
        a = obj[s0:s0 + step, None, :]                                   # points j of this block
        d = obj[None, :, :] - a                                          # f_i − f_j
        I1[s0:s0 + step] = 1.0 - torch.exp(-d.amax(-1) / 0.05).sum(1)
        sde = torch.linalg.vector_norm(d.clamp_min(0), dim=-1)
        jj = idx[s0:s0 + step, None]
        allowed = (idx[None, :] < jj) if lower_only else (idx[None, :] != jj)
        I2[s0:s0 + step] = torch.where(allowed, sde, torch.full_like(sde, float("inf"))).amin(1)
    return I1, I2
