"""Whole-step replay from a HIP graph.

The reference compiles ``workflow.step`` with a tracing JIT (``std_workflow.py:252-256``); this
framework's steps are short sequences of hand-written and library kernels, and what a compiler
would buy – removal of per-kernel launch latency – is obtained by recording the step once into
a HIP graph and replaying it.

``GraphedStep(fn)`` wraps a function ``state -> state`` (or ``state -> (state, aux)``) whose
tensors live on one GPU.  Two ways of feeding the recording:

``mode="ring"`` (default) – pointer-stable.  ``ring`` recordings (eight for states up to 64 MB, four up to 512 MB, else two) are chained: the first reads a set of static input
  buffers, every further one reads the outputs of its predecessor, and the last one ends (inside the recording) with
  a copy of its output state into the static input buffers, which are what it hands back.  The caller gets the
  recordings' own output tensors, not clones; as long as it passes them back unchanged (the normal
  ``state = step(state)`` loop) a replay is one graph launch — no copy-in, no clone-out — and the state is copied
  once every ``ring`` steps, inside a graph.  A state that is not the previous output (a loaded checkpoint, an edited
  field) is copied into the static buffers and the ring restarts.  Tensors handed out stay valid for ``ring − 1``
  further steps; keep a ``clone()`` of anything needed longer.

``mode="copy"`` – every call copies the input leaves into static buffers and hands out clones of the outputs: the
  state keeps value semantics, at the price of two small kernels per leaf and step.

In both modes:

* non-tensor leaves (Python ints, flags) are frozen into the recording.  Integer leaves that
  the step advances by a constant (generation counters) are advanced on the host at every
  replay; any other non-tensor change is refused at capture time, because the recording could
  not reproduce it;
* the function must not synchronise with the host (no ``.item()``, no ``bool(tensor)``) and
  must not change tensor shapes between calls – a new input structure triggers a re-capture.

Clearing memory inside a recording is done with fill kernels (``Tensor.zero_()``/``fill_``),
never with asynchronous memset nodes."""
from __future__ import annotations

from typing import Callable

import torch

from ..core import tree as _tree


class GraphCaptureError(RuntimeError):
    pass


def _signature(leaves):
    sig = []
    for l in leaves:
        if isinstance(l, torch.Tensor):
            sig.append(("t", tuple(l.shape), l.dtype, l.device))
        else:
            sig.append(("v", type(l)))
    return tuple(sig)


class GraphedStep:
    def __init__(self, fn: Callable, warmup: int = 2, clone_outputs: bool = True, host_counters=(), mode: str = "ring",
                 ring=None):
        """``host_counters``: tree paths (as reported by ``tree_paths``) of integer state leaves that
        the step advances by a constant and never uses in its device work – they are advanced on
        the host at replay; a ``(path, modulus)`` pair names a counter that wraps around.  Any other host value that
        changes makes the capture fail, because a branch or a coefficient derived from it would be frozen into the
        recording.  ``clone_outputs`` applies to ``mode="copy"``."""
        if mode not in ("ring", "copy"):
            raise ValueError("mode must be 'ring' or 'copy'")
        self.fn, self.warmup, self.clone_outputs = fn, warmup, clone_outputs
        # ring: recordings in the chain; None: chosen at capture from the size of the state (see ``_ring_length``)
        self.mode, self.ring = mode, (None if ring is None else max(1, int(ring)))
        # path -> modulus (None: plain counter); entries are paths or (path, modulus) pairs
        self.host_counters = {(c if isinstance(c, str) else c[0]): (None if isinstance(c, str) else int(c[1]))
                              for c in host_counters}
        self.graph = None
        self._sig = None
        self.replays = 0
        self.copies_in = 0          # replays that had to copy the caller's state into the static buffers

    _RING_BYTES = 4 << 30            # memory the recordings of one ring may hold together

    @staticmethod
    def _ring_length(state_bytes: int) -> int:
        """The state is copied back into the first recording's inputs once per turn of the ring — one copy node per leaf,
        each a few µs on the replay's critical path (SHADE at 50 000 individuals: ten leaves, 30 µs of a 130 µs
        generation with two recordings) — so the ring is long where memory allows: every recording keeps its own
        outputs and intermediates, a few times the state."""
        if state_bytes <= 64 << 20:
            return 8
        if state_bytes <= 512 << 20:
            return 4
        return 2

    # ------------------------------------------------------------------------------------
    def _record(self, graph, state, pool, feedback_to=None):
        import gc

        # The collector must not run inside a capture: collecting an older recording there (its graph, its pool's
        # blocks) issues calls that a capturing stream does not permit, and the process aborts.  Collect first,
        # keep the collector off while recording.
        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            return self._record_inner(graph, state, pool, feedback_to)
        finally:
            if was_enabled:
                gc.enable()

    def _record_inner(self, graph, state, pool, feedback_to=None):
        try:
            with torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"):
                out = self.fn(state)
                if feedback_to is not None:
                    out_state = out[0] if isinstance(out, tuple) else out
                    new = [l for l in _tree.tree_leaves(out_state) if isinstance(l, torch.Tensor)]
                    pairs = [(d, s) for d, s in zip(feedback_to, new) if d is not s and d.data_ptr() != s.data_ptr()]
                    # A state that shifts fields over generations (prev2 <- prev1 <- cur) hands back, as the new value
                    # of one leaf, the static input buffer of another leaf.  That buffer is itself a destination of
                    # this copy, so it is read from a snapshot: the result then does not depend on the leaf order.
                    static_ptrs = {d.data_ptr() for d in feedback_to}
                    pairs = [(d, s.clone() if s.data_ptr() in static_ptrs else s) for d, s in pairs]
                    if pairs:
                        torch._foreach_copy_([d for d, _ in pairs], [s for _, s in pairs])
        except GraphCaptureError:
            raise
        except Exception as e:      # a synchronising call inside the step invalidates the recording
            raise GraphCaptureError(f"the step cannot be recorded: {type(e).__name__}: {e}") from e
        return out

    def _host_deltas(self, leaves, st_leaves, paths):
        counters = {}
        for i, (old, new) in enumerate(zip(leaves, st_leaves)):
            if isinstance(new, torch.Tensor) or isinstance(old, torch.Tensor):
                continue
            if type(old) is type(new) and old == new:
                continue
            name = paths[i] if i < len(paths) else f"#{i}"
            if name in self.host_counters and isinstance(old, int) and isinstance(new, int) and not isinstance(old, bool):
                mod = self.host_counters[name]
                counters[i] = (new - old, None) if mod is None else ((new - old) % mod, mod)
            else:
                self.graph = None
                raise GraphCaptureError(f"host value {name!r} of the state changes during the step ({old!r} -> {new!r}); "
                                        "keep it in a device tensor to make the step replayable")
        return counters

    def _capture(self, state):
        leaves, treedef = _tree.tree_flatten(state)
        devices = {l.device for l in leaves if isinstance(l, torch.Tensor)}
        if not devices or any(d.type != "cuda" for d in devices) or len(devices) != 1:
            raise GraphCaptureError(f"graph capture needs all state tensors on one GPU, got {devices or 'no tensors'}")
        self.device = devices.pop()
        self._in_treedef = treedef
        self._tensor_pos = [i for i, l in enumerate(leaves) if isinstance(l, torch.Tensor)]
        self._static_in = [l.clone() if isinstance(l, torch.Tensor) else l for l in leaves]
        static_state = _tree.tree_unflatten(treedef, self._static_in)
        paths = [p for p, _ in _tree.tree_paths(state)]

        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.fn(static_state)
        torch.cuda.current_stream(self.device).wait_stream(side)

        if self.ring is None:
            self.ring = self._ring_length(sum(l.numel() * l.element_size() for l in leaves if isinstance(l, torch.Tensor)))
        n_rec = self.ring if self.mode == "ring" else 1
        pool = torch.cuda.graph_pool_handle()
        in_tensors = [self._static_in[i] for i in self._tensor_pos]
        self._graphs, self._outs, self._counter_sets = [], [], []
        cur_state, cur_leaves = static_state, list(self._static_in)
        j = 0
        while j < n_rec:
            graph = torch.cuda.CUDAGraph()
            last = self.mode == "ring" and j == n_rec - 1
            before = torch.cuda.memory_allocated(self.device)
            out = self._record(graph, cur_state, pool, feedback_to=in_tensors if last else None)
            if j == 0 and n_rec > 2:
                # every recording keeps its outputs and intermediates (a population can be many times the state): the
                # ring is shortened so that all of them stay within a few GiB
                grown = max(torch.cuda.memory_allocated(self.device) - before, 1)
                n_rec = self.ring = max(2, min(n_rec, self._RING_BYTES // grown))
            out_leaves, out_treedef = _tree.tree_flatten(out)
            out_state = out[0] if isinstance(out, tuple) else out
            st_leaves, st_treedef = _tree.tree_flatten(out_state)
            if st_treedef != treedef:
                raise GraphCaptureError("the step changes the structure of the state; capture a steady-state step")
            for a, b in zip(cur_leaves, st_leaves):
                if isinstance(a, torch.Tensor) != isinstance(b, torch.Tensor) or (
                        isinstance(a, torch.Tensor) and (a.shape != b.shape or a.dtype != b.dtype)):
                    raise GraphCaptureError("the step changes the shape or type of a state tensor; capture a steady-state step")
            self._counter_sets.append(self._host_deltas(cur_leaves, st_leaves, paths))
            self._graphs.append(graph)
            self._outs.append(out_leaves)
            self._out_treedef = out_treedef
            # The next recording reads this one's output tensors, but is recorded for the SAME host values: the
            # counters are advanced on the host at replay and, by contract, steer no device work; a host branch
            # (see ``graph_variant`` in the workflow) must be the same in every recording of the ring.
            cur_leaves = [n if isinstance(n, torch.Tensor) else o for o, n in zip(self._static_in, st_leaves)]
            cur_state = _tree.tree_unflatten(treedef, cur_leaves)
            j += 1
        self._n_state_leaves = len(cur_leaves)
        self.graph = self._graphs[0]
        self._static_out = self._outs[0]
        self._counters = self._counter_sets[0]
        self._next = 0                # recording to replay when the caller passes back what was handed out
        self._handed = None           # tensor leaves handed out by the last replay
        self._sig = _signature(leaves)

    # ------------------------------------------------------------------------------------
    def __call__(self, state):
        leaves = _tree.tree_leaves(state)
        if self.graph is None or _signature(leaves) != self._sig:
            self._capture(state)
        if self.mode == "copy":
            srcs = [leaves[i] for i in self._tensor_pos]
            dsts = [self._static_in[i] for i in self._tensor_pos]
            torch._foreach_copy_(dsts, srcs)
            self.graph.replay()
            self.replays += 1
            outs = [(l.clone() if self.clone_outputs else l) if isinstance(l, torch.Tensor) else l for l in self._static_out]
            for i, (delta, mod) in self._counters.items():   # state leaves come first in the output
                outs[i] = leaves[i] + delta if mod is None else (leaves[i] + delta) % mod
            return _tree.tree_unflatten(self._out_treedef, outs)

        handed = self._handed
        j = self._next
        if handed is None or any(leaves[i] is not h for i, h in zip(self._tensor_pos, handed)):
            # not the state this object handed out last: take it by value and restart the ring
            torch._foreach_copy_([self._static_in[i] for i in self._tensor_pos], [leaves[i] for i in self._tensor_pos])
            self.copies_in += 1
            j = 0
        self._graphs[j].replay()
        self.replays += 1
        outs = list(self._outs[j])
        if j == len(self._graphs) - 1:
            # the last recording of the ring has copied its state into the static input buffers: hand those out
            for i in self._tensor_pos:
                outs[i] = self._static_in[i]
        for i, (delta, mod) in self._counter_sets[j].items():
            outs[i] = leaves[i] + delta if mod is None else (leaves[i] + delta) % mod
        self._handed = [outs[i] for i in self._tensor_pos]
        self._next = (j + 1) % len(self._graphs)
        return _tree.tree_unflatten(self._out_treedef, outs)
