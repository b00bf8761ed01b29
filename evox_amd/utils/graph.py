"""Whole-step replay from a HIP graph.

The reference compiles ``workflow.step`` with a tracing JIT (``std_workflow.py:252-256``); this
framework's steps are short sequences of hand-written and library kernels, and what a compiler
would buy – removal of per-kernel launch latency – is obtained by recording the step once into
a HIP graph and replaying it.

``GraphedStep(fn)`` wraps a function ``state -> state`` (or ``state -> (state, aux)``) whose
tensors live on one GPU:

* tensor leaves of the input are copied into static buffers, the graph is replayed, and the
  output leaves are handed out as clones (``clone_outputs=False`` hands out the static output
  buffers themselves, valid until the next call);
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
    def __init__(self, fn: Callable, warmup: int = 2, clone_outputs: bool = True, host_counters=()):
        """``host_counters``: tree paths (as reported by ``tree_paths``) of integer state leaves that
        the step advances by a constant and never uses in its device work – they are advanced on
        the host at replay; a ``(path, modulus)`` pair names a counter that wraps around.  Any other host value that changes makes the capture fail, because
        a branch or a coefficient derived from it would be frozen into the recording."""
        self.fn, self.warmup, self.clone_outputs = fn, warmup, clone_outputs
        # path -> modulus (None: plain counter); entries are paths or (path, modulus) pairs
        self.host_counters = {(c if isinstance(c, str) else c[0]): (None if isinstance(c, str) else int(c[1]))
                              for c in host_counters}
        self.graph = None
        self._sig = None
        self.replays = 0

    # ------------------------------------------------------------------------------------
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

        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.fn(static_state)
        torch.cuda.current_stream(self.device).wait_stream(side)

        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self.fn(static_state)
        except GraphCaptureError:
            raise
        except Exception as e:      # a synchronising call inside the step invalidates the recording
            raise GraphCaptureError(f"the step cannot be recorded: {type(e).__name__}: {e}") from e
        self.graph = graph
        out_leaves, self._out_treedef = _tree.tree_flatten(out)
        self._static_out = out_leaves

        # host-side leaves: constant, or integer counters advanced by a fixed amount
        out_state = out[0] if isinstance(out, tuple) else out
        st_leaves, st_treedef = _tree.tree_flatten(out_state)
        if st_treedef != treedef:
            raise GraphCaptureError("the step changes the structure of the state; capture a steady-state step")
        self._n_state_leaves = len(st_leaves)
        self._counters = {}
        paths = [p for p, _ in _tree.tree_paths(state)]
        for i, (old, new) in enumerate(zip(leaves, st_leaves)):
            if isinstance(new, torch.Tensor) or isinstance(old, torch.Tensor):
                continue
            if type(old) is type(new) and old == new:
                continue
            name = paths[i] if i < len(paths) else f"#{i}"
            if name in self.host_counters and isinstance(old, int) and isinstance(new, int) and not isinstance(old, bool):
                mod = self.host_counters[name]
                self._counters[i] = (new - old, None) if mod is None else ((new - old) % mod, mod)
            else:
                self.graph = None
                raise GraphCaptureError(f"host value {name!r} of the state changes during the step ({old!r} -> {new!r}); "
                                        "keep it in a device tensor to make the step replayable")
        self._sig = _signature(leaves)

    # ------------------------------------------------------------------------------------
    def __call__(self, state):
        leaves = _tree.tree_leaves(state)
        if self.graph is None or _signature(leaves) != self._sig:
            self._capture(state)
        srcs = [leaves[i] for i in self._tensor_pos]
        dsts = [self._static_in[i] for i in self._tensor_pos]
        torch._foreach_copy_(dsts, srcs)
        self.graph.replay()
        self.replays += 1
        outs = [(l.clone() if self.clone_outputs else l) if isinstance(l, torch.Tensor) else l for l in self._static_out]
        for i, (delta, mod) in self._counters.items():   # state leaves come first in the output
            outs[i] = leaves[i] + delta if mod is None else (leaves[i] + delta) % mod
        out = _tree.tree_unflatten(self._out_treedef, outs)
        return out
