"""Immutable hierarchical state.

A ``State`` node holds this module's own fields (a dict or a tree dataclass instance), the
states of its sub-modules by attribute name, and the integer id of the module that owns it.
The state tree mirrors the module tree built by ``Stateful.init``; ``use_state`` navigates it
by node id.  Nodes are never mutated after construction: every update returns a new node that
shares the untouched children (and the untouched tensors) with the old one, so a "copy" costs a
few Python objects and no device memory traffic.

API parity: reference ``src/evox/core/state.py:28-334``.
"""
from __future__ import annotations

import dataclasses
import os
from pprint import pformat
from typing import Any, Dict, Optional, Tuple, Union

import torch

from . import tree as _tree
from .distributed import ShardingType

PathLike = Union[str, bytes, os.PathLike]


def _leaf_equal(a, b) -> bool:
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        if not (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor)):
            try:
                return bool(torch.as_tensor(a).cpu().eq(torch.as_tensor(b).cpu()).all())
            except Exception:
                return False
        return a.shape == b.shape and a.dtype == b.dtype and bool(torch.equal(a.cpu(), b.cpu()))
    try:
        import numpy as np

        if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
            return bool(np.array_equal(a, b))
    except ImportError:  # pragma: no cover
        pass
    return a == b


class State:
    """Immutable record of one module's variables plus the states of its sub-modules."""

    __slots__ = ("_state_dict", "_child_states", "_state_id")

    EMPTY: dict = {}

    def __init__(self, _dataclass=None, /, **kwargs) -> None:
        if _dataclass is not None:
            if not dataclasses.is_dataclass(_dataclass) or isinstance(_dataclass, type):
                raise TypeError("the positional argument of State must be a dataclass instance")
            if kwargs:
                raise TypeError("pass either a dataclass instance or keyword fields, not both")
            fields = _dataclass
        else:
            fields = kwargs
        object.__setattr__(self, "_state_dict", fields)
        object.__setattr__(self, "_child_states", State.EMPTY)
        object.__setattr__(self, "_state_id", None)

    # -- construction helpers ------------------------------------------------------------
    @classmethod
    def from_dataclass(cls, dataclass) -> "State":
        return cls(dataclass)

    @classmethod
    def _make(cls, fields, children, node_id) -> "State":
        obj = object.__new__(cls)
        object.__setattr__(obj, "_state_dict", fields)
        object.__setattr__(obj, "_child_states", children)
        object.__setattr__(obj, "_state_id", node_id)
        return obj

    def _set_state_dict_mut(self, state_dict) -> "State":
        object.__setattr__(self, "_state_dict", state_dict)
        return self

    def _set_child_states_mut(self, child_states: dict) -> "State":
        object.__setattr__(self, "_child_states", child_states)
        return self

    def _set_state_id_mut(self, state_id) -> "State":
        object.__setattr__(self, "_state_id", state_id)
        return self

    # -- functional updates --------------------------------------------------------------
    def replace(self, **kwargs) -> "State":
        """New state with some of this node's own fields replaced."""
        if dataclasses.is_dataclass(self._state_dict):
            fields = dataclasses.replace(self._state_dict, **kwargs)
        else:
            fields = {**self._state_dict, **kwargs}
        return State._make(fields, self._child_states, self._state_id)

    def update(self, **kwargs) -> "State":
        import warnings

        warnings.warn("update() is deprecated, use replace() instead", DeprecationWarning, stacklevel=2)
        return self.replace(**kwargs)

    def has_child(self, name: str) -> bool:
        return name in self._child_states

    def get_child_state(self, name: str) -> "State":
        return self._child_states[name]

    def query_state(self, name: str) -> "State":
        """``'a.b'`` -> the state of sub-module ``b`` of sub-module ``a``."""
        node = self
        for part in name.split("."):
            node = node.get_child_state(part)
        return node

    def replace_child(self, name: str, child_state: "State") -> "State":
        return State._make(self._state_dict, {**self._child_states, name: child_state}, self._state_id)

    def update_child(self, name: str, child_state: "State") -> "State":
        import warnings

        warnings.warn("update_child() is deprecated, use replace_child() instead", DeprecationWarning, stacklevel=2)
        return self.replace_child(name, child_state)

    # -- navigation by module id ---------------------------------------------------------
    def find_path_to(self, node_id: int, hint: Optional[str] = None):
        """``(path, state)`` of the node owned by module ``node_id`` or None.

        ``path`` is ``node_id`` for this node, else nested ``(child_name, sub_path)``.
        """
        if node_id == self._state_id:
            return node_id, self
        children = self._child_states
        if hint is not None and hint in children and children[hint]._state_id == node_id:
            return (hint, node_id), children[hint]
        for name, child in children.items():
            found = child.find_path_to(node_id)
            if found is not None:
                return (name, found[0]), found[1]
        return None

    def replace_by_path(self, path, new_state: "State") -> "State":
        if isinstance(path, int):
            if path != self._state_id:
                raise ValueError(f"path ends at node {path} but this is node {self._state_id}")
            return new_state
        if isinstance(path, tuple):
            name, rest = path
            return self.replace_child(name, self._child_states[name].replace_by_path(rest, new_state))
        raise ValueError("path must be an int or a (name, path) tuple")

    # -- field access --------------------------------------------------------------------
    def __getattr__(self, key: str) -> Any:
        if key.startswith("__") and key.endswith("__"):
            raise AttributeError(key)
        fields = object.__getattribute__(self, "_state_dict")
        if dataclasses.is_dataclass(fields):
            return getattr(fields, key)
        try:
            return fields[key]
        except KeyError:
            raise AttributeError(f"State has no field {key!r}; fields: {list(fields)}") from None

    def __getitem__(self, key: str) -> Any:
        return getattr(self, key)

    def __contains__(self, key: str) -> bool:
        fields = self._state_dict
        return hasattr(fields, key) if dataclasses.is_dataclass(fields) else key in fields

    def keys(self):
        f = self._state_dict
        return [x.name for x in dataclasses.fields(f)] if dataclasses.is_dataclass(f) else list(f.keys())

    def __setattr__(self, _key: str, _value: Any) -> None:
        raise TypeError("State is immutable")

    def __setitem__(self, _key: str, _value: Any) -> None:
        raise TypeError("State is immutable")

    def index(self, index) -> "State":
        """Apply ``x[index]`` to every leaf (used with stacked sub-modules)."""
        return _tree.tree_map(lambda x: x[index], self)

    # -- printing ------------------------------------------------------------------------
    def __repr__(self) -> str:
        kids = ",".join(f"{k!r}: {v!r}" for k, v in self._child_states.items())
        return f"State({self._state_dict!r}, {{{kids}}})"

    def __str__(self) -> str:
        return f"State{pformat(self.sprint_tree())}"

    def sprint_tree(self):
        return self._state_dict, {k: v.sprint_tree() for k, v in self._child_states.items()}

    # -- tree protocol -------------------------------------------------------------------
    def tree_flatten(self):
        return (self._state_dict, self._child_states), self._state_id

    @classmethod
    def tree_unflatten(cls, aux, children):
        fields, kids = children
        return cls._make(fields, kids, aux)

    def _tree_child_names(self):
        return ["fields", "children"]

    def __eq__(self, other) -> bool:
        """Structural equality of fields and children (node ids are not compared, so the state
        of a sub-module equals the state the same module builds on its own)."""
        if not isinstance(other, State):
            return NotImplemented
        la, da = _tree.tree_flatten(self._state_dict)
        lb, db = _tree.tree_flatten(other._state_dict)
        if da != db or not all(_leaf_equal(a, b) for a, b in zip(la, lb)):
            return False
        ka, kb = self._child_states, other._child_states
        return ka.keys() == kb.keys() and all(ka[k] == kb[k] for k in ka)

    __hash__ = None

    # -- device movement -----------------------------------------------------------------
    def to(self, device) -> "State":
        return _tree.tree_map(lambda x: x.to(device) if isinstance(x, torch.Tensor) else x, self)

    def clone(self) -> "State":
        return _tree.tree_map(lambda x: x.clone() if isinstance(x, torch.Tensor) else x, self)

    # -- checkpointing -------------------------------------------------------------------
    def save(self, path: PathLike, orbax: bool = True, **kwargs) -> None:
        """Write a checkpoint.  ``orbax=True`` (the default, name kept for API parity) writes a
        directory with a safetensors file and a JSON manifest; ``orbax=False`` writes one
        ``torch.save`` file of plain containers."""
        from .checkpoint import save_state

        save_state(self, path, directory=orbax, **kwargs)

    def load(self, path: PathLike, orbax: bool = True, **kwargs) -> "State":
        """Restore a checkpoint using this state as the structural template."""
        from .checkpoint import load_state

        return load_state(path, template=self, directory=orbax, **kwargs)

    @classmethod
    def load_from(cls, path: PathLike, **kwargs) -> "State":
        """Restore a directory checkpoint without a template (structure from the manifest)."""
        from .checkpoint import load_state

        return load_state(path, template=None, directory=True, **kwargs)


_tree.register_node(State, State.tree_flatten, State.tree_unflatten)


def _collect_sharding(obj, devices, out: list) -> None:
    if isinstance(obj, State):
        _collect_sharding(obj._state_dict, devices, out)
        for child in obj._child_states.values():
            _collect_sharding(child, devices, out)
    elif dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        for f in dataclasses.fields(obj):
            if f.metadata.get("static", False):
                continue
            kind = f.metadata.get("sharding", ShardingType.REPLICATED)
            n = len(_tree.tree_leaves(getattr(obj, f.name)))
            out.extend([kind.get_sharding(devices)] * n)
    elif isinstance(obj, dict):
        for v in obj.values():
            out.extend([ShardingType.REPLICATED.get_sharding(devices)] * len(_tree.tree_leaves(v)))


def get_state_sharding(state: State, devices=None):
    """A tree shaped like ``state`` whose leaves are placement descriptors, read from the
    ``pytree_field(sharding=...)`` metadata of dataclass states (dict states replicate)."""
    flat: list = []
    _collect_sharding(state, devices, flat)
    return _tree.tree_unflatten(_tree.tree_structure(state), flat)
