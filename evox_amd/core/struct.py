"""Dataclasses that are also tree nodes.

``@dataclass`` turns a class into a standard :mod:`dataclasses` dataclass and registers it
with :mod:`evox_amd.core.tree`: fields marked ``pytree_field(static=True)`` become part of the
tree structure (hyper-parameters, shapes), all others are children (tensors).  The same
decorator serves for state records (``CSOState``-like) and for modules whose hyper-parameters
should be stackable (reference: ``src/evox/core/pytree_dataclass.py:12-88``).
"""
from __future__ import annotations

import dataclasses
from typing import Any

from typing_extensions import dataclass_transform

from . import tree as _tree
from .distributed import ShardingType

_HIDDEN_SLOTS = ("_node_id", "_module_name")


def pytree_field(*, static: bool = False, stack: bool = False, sharding=ShardingType.REPLICATED, **kwargs):
    """``dataclasses.field`` carrying the framework's metadata.

    static   – the value is structure (hashable hyper-parameter), not a tensor child.
    stack    – the field holds a *stacked* sub-module (see ``Stateful.stack``).
    sharding – placement of the field when a state is distributed over GPUs.
    """
    meta = dict(kwargs.pop("metadata", {}) or {})
    meta.update(static=static, stack=stack, sharding=sharding)
    return dataclasses.field(metadata=meta, **kwargs)


def _field_table(cls):
    dyn, sta = [], []
    for f in dataclasses.fields(cls):
        (sta if f.metadata.get("static", False) else dyn).append(f.name)
    return tuple(dyn), tuple(sta) + _HIDDEN_SLOTS


def dataclass(cls=None, /, **kwargs):
    def wrap(c):
        c = dataclasses.dataclass(c, **kwargs)
        dyn, sta = _field_table(c)

        def flatten(obj):
            children = [getattr(obj, n, None) for n in dyn]
            aux = tuple(getattr(obj, n, None) for n in sta)
            return children, aux

        def unflatten(aux, children):
            obj = object.__new__(c)
            for n, v in zip(dyn, children):
                object.__setattr__(obj, n, v)
            for n, v in zip(sta, aux):
                object.__setattr__(obj, n, v)
            return obj

        _tree.register_node(c, flatten, unflatten)
        c._tree_child_names = lambda self: list(dyn)
        c.set_frozen_attr = lambda self, key, value: object.__setattr__(self, key, value)
        return c

    return wrap if cls is None else wrap(cls)


@dataclass_transform(field_specifiers=(pytree_field,), kw_only_default=True)
class PyTreeNode:
    """Subclassing registers the subclass as a tree dataclass."""

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__()
        dataclass(cls, **kwargs)


def is_tree_dataclass(obj: Any) -> bool:
    return dataclasses.is_dataclass(obj) and _tree.is_registered(type(obj))
