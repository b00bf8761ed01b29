"""One native call per half of a CMA-ES generation (``csrc/cma_step.hip``).

``ask``  key split, Gaussian draws, ``X = m + σ Z Sᵀ``.
``tell`` rank sort, transposed factor of the selected rows, whitening, path and step-size update, rank-μ covariance
         update, factorisation.

The kernel sequences are those of ``ops.cma_bf16`` / ``ops.gemm`` / ``ops.cma``; issuing them from C++ removes the
per-kernel host cost that bounded the generation.  The decisions that need values from earlier generations (which
factorisation driver, how many iterations) stay with ``ops.cma.NewtonSchulzSqrt.plan``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

import torch

from .. import _native
from . import cma_bf16 as _b16
from ._util import check, hip, stream_of

P, I = _native.P, _native.I


class _Ask(ctypes.Structure):
    _fields_ = [("key", c_void_p), ("keys_out", c_void_p), ("z", c_void_p), ("s16", c_void_p), ("mean", c_void_p),
                ("sigma", c_void_p), ("x", c_void_p), ("ldn", c_int64), ("row_offset", c_int64), ("rows", c_int),
                ("n", c_int), ("draw", c_int), ("split", c_int), ("wait_side", c_int), ("prefetch", c_int),
                ("side", c_void_p), ("ev_main", c_void_p), ("ev_side", c_void_p), ("keys_next", c_void_p), ("z_next", c_void_p),
                ("count_in", c_void_p), ("count_out", c_void_p)]


class _Tell(ctypes.Structure):
    _fields_ = [("n", c_int), ("lam", c_int), ("mu", c_int), ("rb", c_int), ("ns_planes", c_int),
                ("ldn", c_int64), ("ldmu", c_int64), ("cov_ws_floats", c_int64), ("consts", c_float * 8),
                ("rowscale", c_void_p), ("yt", c_void_p), ("colsum", c_void_p), ("cov_ws", c_void_p), ("order", c_void_p),
                ("sort_scratch", c_void_p), ("step", c_void_p), ("whitened", c_void_p), ("planes", c_void_p), ("ns_f32", c_void_p), ("vec4", c_void_p),
                ("scal", c_void_p), ("scal2", c_void_p), ("lib_work", c_void_p), ("lib_eye", c_void_p),
                ("fitness", c_void_p), ("pop", c_void_p), ("ldpop", c_int64), ("mean", c_void_p), ("ps", c_void_p),
                ("pc", c_void_p), ("sigma", c_void_p), ("C", c_void_p), ("B", c_void_p), ("invsqrtC", c_void_p),
                ("iter_dev", c_void_p), ("ctrl_vec", c_void_p), ("prev_rec", c_void_p),
                ("mean_o", c_void_p), ("ps_o", c_void_p), ("pc_o", c_void_p), ("scalars_o", c_void_p), ("C_o", c_void_p),
                ("B_o", c_void_p), ("inv_o", c_void_p), ("b16_o", c_void_p), ("ctrl_vec_o", c_void_p), ("rec_o", c_void_p),
                ("factor", c_int), ("ns_iters", c_int), ("ns_power", c_int), ("ns_safe", c_int), ("ns_symmetric", c_int),
                ("ns_cond", c_int), ("ns_first_poly4", c_int), ("ns_tune", c_int), ("rec_host", c_void_p),
                ("sharded", c_int), ("gram", c_void_p), ("poison", c_void_p), ("gram_packed", c_int)]


class _Fast(ctypes.Structure):
    _fields_ = [("key", c_void_p), ("z_next", c_void_p), ("keys_next", c_void_p), ("count_in", c_void_p),
                ("count_next", c_void_p), ("row_offset", c_int64), ("rows", c_int), ("wide", c_int)]


_native.register_hip_api({
    "evox_hip_cma_tell_fast": (I, [P, P, P]),
    "evox_hip_rank_sort_scratch": (c_int64, [I]),
    "evox_hip_cma_ask": (I, [P, P]),
    "evox_hip_event_create": (c_void_p, []),
    "evox_hip_event_destroy": (None, [P]),
    "evox_hip_cma_tell": (I, [P, P]),
    "evox_hip_cma_select_local": (I, [P, I, I, I, I, P, P, P, P, P]),
    "evox_hip_cma_rank_select_local": (I, [P, I, I, I, I, I, P, P, P, P, P, P]),
    "evox_hip_rank_window_slices": (I, []),
    "evox_hip_cma_presorted_select_local": (I, [P, I, I, I, P, I, I, P, P, P, P, P, P]),
})


def select_local(ranking: torch.Tensor, lo: int, local: int, mu: int, keep: int, sqrt_w_padded: torch.Tensor,
                 overflow_out: torch.Tensor):
    """Row-sharded selection in one launch: ``(order, sqrt_w)`` of ``keep`` slots — this rank's rows with a global rank
    below ``mu`` (row index relative to ``lo``, in row order) and the square roots of their recombination weights, unused
    slots weighing zero; ``overflow_out`` (four floats) becomes ``[1, 0, 0, 0]`` when more than ``keep`` rows were selected,
    zeros otherwise."""
    dev = ranking.device
    order = torch.empty(keep, dtype=torch.int64, device=dev)
    w = torch.empty(keep, dtype=torch.float32, device=dev)
    check(hip().evox_hip_cma_select_local(ranking.data_ptr(), int(lo), int(local), int(mu), int(keep), sqrt_w_padded.data_ptr(),
                                          order.data_ptr(), w.data_ptr(), overflow_out.data_ptr(), stream_of(ranking)),
          "cma_select_local")
    return order, w


def presorted_select_local(msg: torch.Tensor, world: int, me: int, local: int, own_rank: torch.Tensor, mu: int, keep: int,
                           sqrt_w_padded: torch.Tensor, overflow_out: torch.Tensor):
    """:func:`select_local` from the pre-sorted fitness message of a row-sharded generation: ``msg`` is ``(2·world, local)``
    fp32 — row ``2q`` the fitness of process ``q``'s rows, row ``2q + 1`` the same values in ascending order — and
    ``own_rank`` the rank of every local row among the local rows.  The global rank of a local row is ``own_rank`` plus one
    binary search per other process, so no process sorts more than its own rows."""
    dev = msg.device
    if msg.dtype != torch.float32 or not msg.is_contiguous() or tuple(msg.shape) != (2 * world, local):
        raise ValueError(f"fitness message must be contiguous fp32 of shape {(2 * world, local)}, got {tuple(msg.shape)}")
    order = torch.empty(keep, dtype=torch.int64, device=dev)
    w = torch.empty(keep, dtype=torch.float32, device=dev)
    scratch = torch.empty(local, dtype=torch.int32, device=dev)
    check(hip().evox_hip_cma_presorted_select_local(msg.data_ptr(), int(world), int(me), int(local), own_rank.data_ptr(),
                                                    int(mu), int(keep), sqrt_w_padded.data_ptr(), order.data_ptr(),
                                                    w.data_ptr(), overflow_out.data_ptr(), scratch.data_ptr(),
                                                    stream_of(msg)), "cma_presorted_select_local")
    return order, w


def rank_select_local(fitness: torch.Tensor, lo: int, local: int, mu: int, keep: int, sqrt_w_padded: torch.Tensor,
                      overflow_out: torch.Tensor):
    """:func:`select_local` straight from the fitness of the whole population (fp32, contiguous; ``lo`` a multiple of 4):
    the ``local`` rows from ``lo`` on are ranked against all keys by counting — ``local x n`` compares, two launches in
    all — instead of sorting the whole vector on every rank."""
    dev = fitness.device
    order = torch.empty(keep, dtype=torch.int64, device=dev)
    w = torch.empty(keep, dtype=torch.float32, device=dev)
    scratch = torch.empty(int(hip().evox_hip_rank_window_slices()) * local, dtype=torch.int32, device=dev)
    check(hip().evox_hip_cma_rank_select_local(fitness.data_ptr(), int(fitness.shape[0]), int(lo), int(local), int(mu), int(keep),
                                               sqrt_w_padded.data_ptr(), order.data_ptr(), w.data_ptr(),
                                               overflow_out.data_ptr(), scratch.data_ptr(), stream_of(fitness)),
          "cma_rank_select_local")
    return order, w


class Generation:
    """Argument blocks of the two native calls for one CMA-ES instance and workspace."""

    def __init__(self, ws: "_b16.Workspace", rowscale: torch.Tensor, consts, ns):
        n, dev = ws.n, ws.device
        self.ws, self.ns = ws, ns
        self.rowscale = rowscale.contiguous()
        self.order = torch.empty(ws.lam, dtype=torch.int64, device=dev)
        self.sort_scratch = torch.empty(int(hip().evox_hip_rank_sort_scratch(ws.lam)), dtype=torch.int32, device=dev)
        self.vecs = torch.empty(2 * n, dtype=torch.float32, device=dev)         # step | whitened
        self.scal, self.scal2, self.vec4 = ns._gpu_scratch(n, dev, ws) if ns is not None else (None, None, None)
        self._ask = _Ask()
        self._ask.ldn, self._ask.n = ws.ldn, n
        t = self._tell = _Tell()
        t.n, t.lam, t.mu, t.rb, t.ns_planes = n, ws.lam, ws.mu, ws.rb, ws.ns_planes
        t.ldn, t.ldmu, t.cov_ws_floats = ws.ldn, ws.ldmu, ws.cov_ws.numel()
        for i, v in enumerate(consts):
            t.consts[i] = float(v)
        t.rowscale, t.yt, t.colsum, t.cov_ws = self.rowscale.data_ptr(), ws.Yt.data_ptr(), ws.colsum.data_ptr(), ws.cov_ws.data_ptr()
        t.order, t.step, t.whitened = self.order.data_ptr(), self.vecs.data_ptr(), self.vecs.data_ptr() + 4 * n
        t.sort_scratch = self.sort_scratch.data_ptr()
        if ns is not None:
            st = ws.ns()
            t.planes, t.ns_f32 = st["planes"].data_ptr(), st["f32"].data_ptr()
            t.vec4, t.scal, t.scal2 = self.vec4.data_ptr(), self.scal.data_ptr(), self.scal2.data_ptr()
        self._lib = None
        self._fast = _Fast()
        self.fast_calls = 0              # tells served by the folded launch sequence (``csrc/cma_fast.hip``)
        # (key tensor of the next ask, its split, index of the buffer with its draws, row window, made on the side
        # stream?, generation counter of the next ask | None)
        self._ahead = None
        self._window = None              # (rows, row offset) of the latest ask
        self._side = None
        self._events = None

    def __del__(self):
        try:
            if self._events is not None:
                for e in self._events:
                    hip().evox_hip_event_destroy(e)
        except Exception:
            pass

    def ask(self, key, s16, mean, sigma, rows: int, row_offset: int, prefetch: bool, count=None):
        """``(population, next key, count + 1 | None)`` — ``count``: the generation counter on the device (a 0-d fp32
        tensor), advanced by the kernel that splits the key.  With ``prefetch`` the split and the Gaussian draws of the NEXT generation
        are started on a side stream once this generation's product has been issued; the next ``ask`` uses them when
        it is called with the key tensor returned here (any other key is split and drawn on the spot: same values)."""
        a, ws = self._ask, self.ws
        dev = mean.device
        out = torch.empty((rows, ws.n), dtype=torch.float32, device=dev)
        ahead, self._ahead = self._ahead, None
        count_ahead = None
        if ahead is not None and ahead[0] is key and ahead[3] == (rows, row_offset):
            keys_out, zi = ahead[1], ahead[2]
            a.split = a.draw = 0
            a.wait_side = int(ahead[4])
            count_ahead = ahead[5] if len(ahead) > 5 else None
        else:
            keys_out, zi = torch.empty((2, 2), dtype=torch.int32, device=dev), ws._z_next
            a.split = a.draw = 1
            a.wait_side = 0
        ws._z_next = zi
        a.key, a.keys_out, a.z, a.s16 = key.data_ptr(), keys_out.data_ptr(), ws._Z[zi].data_ptr(), s16.data_ptr()
        a.mean, a.sigma, a.x = mean.data_ptr(), sigma.data_ptr(), out.data_ptr()
        a.rows, a.row_offset = rows, row_offset
        self._window = (rows, row_offset)
        count_out = None
        if count is not None and count_ahead is not None:
            count_out = count_ahead              # the tell that drew ahead advanced the counter as well
            a.count_in = a.count_out = None
        elif count is not None:
            count_out = torch.empty((), dtype=torch.float32, device=dev)
            a.count_in, a.count_out = count.data_ptr(), count_out.data_ptr()
        else:
            a.count_in = a.count_out = None
        next_key = keys_out[0]
        a.prefetch = 0
        if prefetch and not torch.cuda.is_current_stream_capturing():
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
                self._events = (hip().evox_hip_event_create(), hip().evox_hip_event_create())
                if not all(self._events):
                    raise RuntimeError("could not create the events of the draw prefetch")
                a.side, a.ev_main, a.ev_side = self._side.cuda_stream, self._events[0], self._events[1]
            keys_next = torch.empty((2, 2), dtype=torch.int32, device=dev)
            a.prefetch, a.keys_next, a.z_next = 1, keys_next.data_ptr(), ws._Z[zi ^ 1].data_ptr()
            self._ahead = (next_key, keys_next, zi ^ 1, (rows, row_offset), True)
        check(hip().evox_hip_cma_ask(ctypes.addressof(a), stream_of(mean)), "cma_ask")
        return out, next_key, count_out

    def _use_fast(self, reduced, factor, plan, n: int) -> bool:
        """The folded launch sequence applies to a generation (single-process, or the part of a row-sharded one behind the
        all-reduce when the rank-mu term arrives tile-packed) whose factorisation is ONE fifth-order step of the symmetric
        two-plane driver with the estimate scaling, from the third call on (the scale is the estimate the previous call
        left in its record)."""
        if reduced is not None and (reduced[1].dim() != 1 or os.environ.get("EVOX_AMD_CMA_FAST_SHARDED", "1") == "0"):
            return False
        return (bool(factor) and plan is not None and bool(plan["symmetric"]) and not plan.get("full") and plan["used"] == 1
                and not plan["use_safe"] and plan["calls"] >= 2 and self.ws.ns_planes == 2 and n % 4 == 0
                and os.environ.get("EVOX_AMD_CMA_FAST", "1") != "0")

    def draw_ahead(self, key, rows: int, row_offset: int) -> None:
        """Split ``key`` and draw the Gaussian numbers of the generation that the next ``ask(key, ...)`` will sample —
        now, on the current stream.  They depend on the key alone, so a caller that is about to wait for something else
        (the all-reduce of a multi-process generation) issues them first and lets them run underneath.  The next ``ask``
        uses them when it gets this very key tensor and the same row window; otherwise it draws on the spot (same values)."""
        if torch.cuda.is_current_stream_capturing():
            return
        ws = self.ws
        zi = ws._z_next ^ 1
        keys_next = torch.empty((2, 2), dtype=torch.int32, device=ws.device)
        stream = stream_of(key)
        check(hip().evox_hip_split(key.data_ptr(), keys_next.data_ptr(), 1, 2, stream), "split")
        check(hip().evox_hip_normal_bf16_pitched(keys_next.data_ptr() + 8, ws._Z[zi].data_ptr(), rows, ws.n, ws.ldn,
                                                 row_offset, stream), "normal_bf16")
        self._ahead = (key, keys_next, zi, (rows, row_offset), False)

    def tell(self, state, fitness, factor: bool, plan: dict | None, reduced=None):
        """Returns ``(mean, ps, pc, sigma, C, B, invsqrtC, b16, ctrl_vec, rec)``; the last five are ``None`` when the
        factorisation is skipped (``b16`` also when the fp32 library driver ran).

        ``reduced = (step_sum, gram, poison | None)`` (``gram``: (n, n), or flat = its tile-packed upper triangle): the row-sharded form — the selected rows have been reduced on every
        rank and summed over the ranks already; the call then starts at the whitening product (``fitness`` is unused)."""
        ws, t = self.ws, self._tell
        n, dev = ws.n, ws.device
        small = torch.empty(3 * n + 4 + n + 2 * n + 8, dtype=torch.float32, device=dev)
        mean_o, ps_o, pc_o = small[:n], small[n:2 * n], small[2 * n:3 * n]
        scalars = small[3 * n:4 * n + 4]
        ctrl_o = small[4 * n + 4:6 * n + 4].view(2, n)
        rec_o = small[6 * n + 4:6 * n + 11]
        C_o = torch.empty((n, n), dtype=torch.float32, device=dev)
        pop = state.population
        if reduced is None:
            t.sharded, t.gram, t.poison, t.gram_packed = 0, None, None, 0
            t.step = self.vecs.data_ptr()
            t.fitness, t.pop, t.ldpop = fitness.data_ptr(), pop.data_ptr(), pop.stride(0)
        else:
            step_sum, gram, poison = reduced
            t.sharded, t.gram, t.step = 1, gram.data_ptr(), step_sum.data_ptr()
            # a flat vector is the tile-packed upper triangle (``ops/cma_bf16.py::gram_packed``), a matrix the full square
            t.gram_packed = int(gram.dim() == 1)
            if gram.dim() == 1 and gram.numel() != _b16.packed_upper_floats(n):
                raise ValueError("packed Gram matrix of the wrong size")
            t.poison = poison.data_ptr() if poison is not None else None
        t.mean, t.ps, t.pc, t.sigma = state.mean.data_ptr(), state.ps.data_ptr(), state.pc.data_ptr(), state.sigma.data_ptr()
        t.C, t.B, t.invsqrtC, t.iter_dev = state.C.data_ptr(), state.B.data_ptr(), state.invsqrtC.data_ptr(), state.count_iter_dev.data_ptr()
        t.mean_o, t.ps_o, t.pc_o, t.scalars_o, t.C_o = mean_o.data_ptr(), ps_o.data_ptr(), pc_o.data_ptr(), scalars.data_ptr(), C_o.data_ptr()
        t.factor = int(bool(factor))
        B_o = inv_o = b16 = None
        if factor:
            B_o, inv_o = torch.empty_like(C_o), torch.empty_like(C_o)
            t.B_o, t.inv_o = B_o.data_ptr(), inv_o.data_ptr()
            t.ctrl_vec, t.prev_rec = state.ns_vec.data_ptr(), state.ns_res1.data_ptr()
            t.ctrl_vec_o, t.rec_o = ctrl_o.data_ptr(), rec_o.data_ptr()
            t.ns_iters, t.ns_power, t.ns_safe = plan["used"], plan["power"], int(plan["use_safe"])
            t.ns_symmetric, t.ns_cond = int(plan["symmetric"]) + int(plan.get("full", False)), int(plan["cond"])
            # the record of this call goes straight into a pinned slot of the iteration control (read two calls later)
            slot = self.ns.pinned_slot() if not torch.cuda.is_current_stream_capturing() else None
            t.rec_host = slot[2] if slot is not None else None
            if plan["symmetric"]:
                b16 = ws.factor_buffer()
                t.b16_o = b16.data_ptr()
                t.ns_planes = int(plan.get("planes") or ws.ns_planes)
                t.planes = ws.ns_planes_buffer(t.ns_planes).data_ptr()
            else:
                from . import blas as _blas
                if not _blas.available():
                    raise RuntimeError("the fp32 factorisation driver needs hipBLASLt")
                w = self.ns.library_workspace(n, dev)
                t.lib_work, t.lib_eye = w.work.data_ptr(), w.eye.data_ptr()
                t.ns_first_poly4, t.ns_tune = int(_blas._first_step_default() == "poly4"), _blas._tune()
                t.b16_o = None
        else:
            slot = None
        if self._use_fast(reduced, factor, plan, n):
            f = self._fast
            f.wide = int(os.environ.get("EVOX_AMD_CMA_WIDE", "1") == "1")
            f.key = f.count_in = f.count_next = None
            window = (ws.lam, 0) if reduced is None else self._window
            ahead = self._ahead
            if torch.cuda.is_current_stream_capturing() or os.environ.get("EVOX_AMD_CMA_DRAW_AHEAD", "1") == "0" or window is None:
                pass
            elif ahead is not None and ahead[0] is state.key and ahead[3] == window and len(ahead) == 5 and not ahead[4]:
                # the draws were issued underneath the all-reduce already: only the generation counter is left to advance
                count_next = torch.empty((), dtype=torch.float32, device=dev)
                f.count_in, f.count_next = state.count_iter_dev.data_ptr(), count_next.data_ptr()
                self._ahead = ahead + (count_next,)
            elif ahead is None:
                # the next generation's draws and key split ride along with the factorisation's products
                zi = ws._z_next ^ 1
                keys_next = torch.empty((2, 2), dtype=torch.int32, device=dev)
                count_next = torch.empty((), dtype=torch.float32, device=dev)
                f.key, f.z_next, f.keys_next = state.key.data_ptr(), ws._Z[zi].data_ptr(), keys_next.data_ptr()
                f.count_in, f.count_next = state.count_iter_dev.data_ptr(), count_next.data_ptr()
                f.rows, f.row_offset = window
                self._ahead = (state.key, keys_next, zi, window, False, count_next)
            self.fast_calls += 1
            check(hip().evox_hip_cma_tell_fast(ctypes.addressof(t), ctypes.addressof(f), stream_of(pop)), "cma_tell_fast")
        else:
            check(hip().evox_hip_cma_tell(ctypes.addressof(t), stream_of(pop)), "cma_tell")
        if slot is not None:
            self.ns.adopt(rec_o, slot)
        sigma = scalars[0]
        if not factor:
            return mean_o, ps_o, pc_o, sigma, C_o, None, None, None, None, None
        return mean_o, ps_o, pc_o, sigma, C_o, B_o, inv_o, b16, ctrl_o, rec_o
