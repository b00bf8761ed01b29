"""Module system: ``Stateful`` objects hold hyper-parameters, their variables live in a State.

``Stateful.init(key)`` walks the module tree depth-first (sub-modules sorted by attribute
name, so node ids are deterministic and checkpoints are portable across runs), hands every
module its own PRNG key, calls ``setup`` and assembles the State tree.  ``use_state`` lets a
method written against *its own* state be called with the *root* state: it finds the
sub-state by node id, runs the method and grafts the returned state back.

API parity: reference ``src/evox/core/module.py:16-287``.
"""
from __future__ import annotations

import dataclasses
from functools import partial, wraps
from typing import Callable, Optional, Tuple

import torch

from . import tree as _tree
from .state import State


def _scatter_index(batch, index, new):
    if isinstance(batch, torch.Tensor):
        out = batch.clone()
        out[index] = new
        return out
    return new


def use_state(func: Callable, index: Optional[int] = None):
    """Adapt ``func(self, own_state, *args) -> (..., own_state)`` to take and return the state
    of any ancestor.  With ``index`` the method runs on slice ``index`` of a stacked module /
    state and the result is scattered back into the stack."""
    err = "the last return value must be a State, got {}"

    def wrapper(self, state: State, *args, **kwargs):
        if not isinstance(state, State):
            raise TypeError(f"the first argument must be a State, got {type(state)}")
        node_id = getattr(self, "_node_id", None)
        if node_id is None:
            raise ValueError(f"{self} is not initialized, did you forget to call `init`?")
        found = state.find_path_to(node_id, getattr(self, "_module_name", None))
        if found is None:
            raise ValueError(f"no state for module id {node_id} ({type(self).__name__}) under the given state")
        path, own = found

        if index is not None:
            sub_state = _tree.tree_map(lambda x: x[index], own)
            module = _tree.tree_map(lambda x: x[index], self)
        else:
            sub_state, module = own, self

        if hasattr(func, "__self__"):
            # bound method: `self` is already attached; a stacked call re-binds to the slice
            if index is None:
                result = func(sub_state, *args, **kwargs)
            else:
                result = func.__func__(module, sub_state, *args, **kwargs)
        else:
            result = func(module, sub_state, *args, **kwargs)

        if isinstance(result, tuple):
            if not isinstance(result[-1], State):
                raise TypeError(err.format(type(result[-1])))
            aux, new_own = result[:-1], result[-1]
        else:
            if not isinstance(result, State):
                raise TypeError(err.format(type(result)))
            aux, new_own = None, result

        if index is not None:
            new_own = _tree.tree_map(lambda b, n: _scatter_index(b, index, n), own, new_own)

        new_state = state.replace_by_path(path, new_own)
        return new_state if aux is None else (*aux, new_state)

    if hasattr(func, "__self__"):
        return wraps(func)(partial(wrapper, func.__self__))
    return wraps(func)(wrapper)


def jit_method(method: Callable) -> Callable:
    """No-op kept for source compatibility with code written for a tracing compiler.

    There is no tracer in this framework: hot paths are hand-written kernels and whole steps
    can be replayed from a HIP graph (``evox_amd.utils.graph.GraphedStep``)."""
    return method


def jit_class(cls):
    """No-op class decorator (see :func:`jit_method`)."""
    return cls


def jit(fn=None, **_kwargs):
    """No-op stand-in for a function-level jit decorator."""
    if fn is None:
        return lambda f: f
    return fn


class Stateful:
    """Base class of every module (algorithms, problems, workflows, optimizers, operators).

    Hyper-parameters are set in ``__init__``; variables are created in ``setup`` and live in
    the State returned by ``init``."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_node_id", None)
        object.__setattr__(self, "_module_name", None)

    def setup(self, key) -> State:
        return State()

    # -- tree construction ---------------------------------------------------------------
    def _submodules(self):
        subs = []
        if dataclasses.is_dataclass(self):
            for f in dataclasses.fields(self):
                attr = getattr(self, f.name, None)
                if isinstance(attr, Stateful):
                    subs.append((f.name, attr, f.metadata))
        else:
            for name, attr in vars(self).items():
                if not name.startswith("_") and isinstance(attr, Stateful):
                    subs.append((name, attr, {}))
        subs.sort(key=lambda s: s[0])
        return subs

    def _recursive_init(self, key, node_id: int, module_name, no_state: bool) -> Tuple[Optional[State], int]:
        from . import random as rnd

        object.__setattr__(self, "_node_id", node_id)
        object.__setattr__(self, "_module_name", module_name)
        child_states = {}
        for name, sub, meta in self._submodules():
            if key is None:
                subkey = None
            else:
                key, subkey = rnd.split(key)
            if meta.get("stack", False):
                # `sub` is a stacked module: every tensor field carries a leading batch axis.
                # All copies share one node id; their states are stacked leaf-wise.
                n = len(sub)
                first_id = node_id + 1
                _, node_id = sub._recursive_init(None, first_id, name, True)
                if not no_state:
                    subkeys = rnd.split(subkey, n) if subkey is not None else [None] * n
                    states = []
                    for i in range(n):
                        member = _tree.tree_map(lambda x: x[i], sub)
                        s, _ = Stateful._recursive_init(member, subkeys[i], first_id, name, False)
                        states.append(s)
                    child_states[name] = _tree.tree_map(lambda *xs: _stack_leaves(xs), *states)
            else:
                sub_state, node_id = sub._recursive_init(subkey, node_id + 1, name, no_state)
                if not no_state:
                    if not isinstance(sub_state, State):
                        raise TypeError("setup must return a State")
                    child_states[name] = sub_state
        if no_state:
            return None, node_id
        own = self.setup(key)
        if dataclasses.is_dataclass(own) and not isinstance(own, type):
            own = State.from_dataclass(own)
        if not isinstance(own, State):
            raise TypeError(f"{type(self).__name__}.setup must return a State or a dataclass instance")
        own = State._make(own._state_dict, child_states if child_states else State.EMPTY, self._node_id)
        return own, node_id

    def init(self, key=None, no_state: bool = False) -> Optional[State]:
        """Assign node ids to the module tree and (unless ``no_state``) build its State."""
        state, _ = self._recursive_init(key, 0, None, no_state)
        return state

    # -- stacked modules -----------------------------------------------------------------
    @classmethod
    def stack(cls, stateful_objs, axis: int = 0):
        """Group several modules of one kind.

        Dataclass modules are stacked leaf-wise into one module with a leading batch axis (the
        reference's behaviour, ``core/module.py:263-275``).  Ordinary modules are collected in
        a :class:`ModuleStack`, whose members keep their own hyper-parameters and are driven one
        after another by the containers that accept stacks."""
        if all(dataclasses.is_dataclass(obj) for obj in stateful_objs):
            return _tree.tree_map(lambda *xs: _stack_leaves(xs, axis), *stateful_objs)
        return ModuleStack(stateful_objs)

    def __len__(self) -> int:
        if not dataclasses.is_dataclass(self):
            raise TypeError("len() is only defined for (stacked) dataclass modules")
        return len(_tree.tree_leaves(self)[0])


def clone_module(module):
    """A copy of ``module`` with its own node ids: the module and every sub-module below it are shallow-copied (tensors
    and other hyper-parameters stay shared).  A plain ``copy.copy`` would share the sub-module objects, and with them
    their node ids, between the copies."""
    import copy

    new = copy.copy(module)
    for name, sub, _ in module._submodules():
        object.__setattr__(new, name, clone_module(sub))
    return new


def _stack_leaves(xs, axis: int = 0):
    x0 = xs[0]
    if isinstance(x0, torch.Tensor):
        return torch.stack(list(xs), dim=axis)
    return torch.stack([torch.as_tensor(x) for x in xs], dim=axis)


class ModuleStack(Stateful):
    """An ordered group of modules initialised as children ``m0, m1, …`` (each with its own
    PRNG key and state).  A module object listed twice is copied (with its sub-modules) so that every member
    has its own node ids."""

    def __init__(self, members):
        super().__init__()
        seen = set()
        self._names = []
        for i, m in enumerate(members):
            if id(m) in seen:
                m = clone_module(m)
            seen.add(id(m))
            name = f"m{i:04d}"
            setattr(self, name, m)
            self._names.append(name)

    def __len__(self) -> int:
        return len(self._names)

    def __getitem__(self, i):
        return getattr(self, self._names[i])

    def __iter__(self):
        return (getattr(self, n) for n in self._names)
