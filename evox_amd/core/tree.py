"""Minimal tree ("pytree") utilities over plain Python containers and registered node types.

The framework stores algorithm state as nested containers of ``torch.Tensor`` leaves.  This
module provides the flatten / unflatten / map primitives the rest of the code base relies on
(state checkpointing, stacking modules, HIP-graph capture of a step, sharding specs).

Containers: ``dict`` (insertion order), ``list``, ``tuple`` (incl. namedtuples), ``None`` (an
empty node) and any class registered with :func:`register_node`.  Everything else is a leaf.
"""
from __future__ import annotations

from typing import Any, Callable, Iterable, List, Sequence, Tuple

_REGISTRY: dict = {}


def register_node(cls, flatten: Callable, unflatten: Callable) -> None:
    """Register ``cls`` as an internal tree node.

    ``flatten(obj) -> (children: sequence, aux)`` and ``unflatten(aux, children) -> obj``.
    """
    _REGISTRY[cls] = (flatten, unflatten)


def is_registered(cls) -> bool:
    return cls in _REGISTRY


class TreeDef:
    """Structure of a tree with the leaves removed."""

    __slots__ = ("kind", "aux", "children", "num_leaves")

    def __init__(self, kind, aux, children):
        self.kind = kind
        self.aux = aux
        self.children = children
        self.num_leaves = 1 if kind == "leaf" else sum(c.num_leaves for c in children)

    def __eq__(self, other):
        return (
            isinstance(other, TreeDef)
            and self.kind == other.kind
            and self.aux == other.aux
            and self.children == other.children
        )

    def __hash__(self):
        return hash((str(self.kind), self.num_leaves))

    def __repr__(self):
        if self.kind == "leaf":
            return "*"
        name = self.kind if isinstance(self.kind, str) else self.kind.__name__
        return f"{name}({', '.join(map(repr, self.children))})"


_LEAF = TreeDef("leaf", None, ())


def _is_namedtuple(x) -> bool:
    return isinstance(x, tuple) and hasattr(x, "_fields")


def _flatten_into(obj, leaves: list, is_leaf) -> TreeDef:
    if is_leaf is not None and is_leaf(obj):
        leaves.append(obj)
        return _LEAF
    t = type(obj)
    if obj is None:
        return TreeDef("none", None, ())
    if t in _REGISTRY:
        children, aux = _REGISTRY[t][0](obj)
        return TreeDef(t, aux, tuple(_flatten_into(c, leaves, is_leaf) for c in children))
    if t is dict:
        keys = tuple(obj.keys())
        return TreeDef("dict", keys, tuple(_flatten_into(obj[k], leaves, is_leaf) for k in keys))
    if t is list:
        return TreeDef("list", None, tuple(_flatten_into(c, leaves, is_leaf) for c in obj))
    if _is_namedtuple(obj):
        return TreeDef("namedtuple", t, tuple(_flatten_into(c, leaves, is_leaf) for c in obj))
    if t is tuple:
        return TreeDef("tuple", None, tuple(_flatten_into(c, leaves, is_leaf) for c in obj))
    leaves.append(obj)
    return _LEAF


def tree_flatten(tree, is_leaf=None) -> Tuple[List[Any], TreeDef]:
    leaves: list = []
    treedef = _flatten_into(tree, leaves, is_leaf)
    return leaves, treedef


def _unflatten(treedef: TreeDef, it):
    k = treedef.kind
    if k == "leaf":
        return next(it)
    if k == "none":
        return None
    children = [_unflatten(c, it) for c in treedef.children]
    if k == "dict":
        return dict(zip(treedef.aux, children))
    if k == "list":
        return children
    if k == "tuple":
        return tuple(children)
    if k == "namedtuple":
        return treedef.aux(*children)
    return _REGISTRY[k][1](treedef.aux, children)


def tree_unflatten(treedef: TreeDef, leaves: Iterable):
    it = iter(leaves)
    out = _unflatten(treedef, it)
    return out


def tree_leaves(tree, is_leaf=None) -> List[Any]:
    return tree_flatten(tree, is_leaf)[0]


def tree_structure(tree, is_leaf=None) -> TreeDef:
    return tree_flatten(tree, is_leaf)[1]


def tree_map(fn: Callable, tree, *rest, is_leaf=None):
    leaves, treedef = tree_flatten(tree, is_leaf)
    others = []
    for r in rest:
        l, d = tree_flatten(r, is_leaf)
        if d != treedef:
            raise ValueError(f"tree_map: structure mismatch\n  {treedef}\n  {d}")
        others.append(l)
    return tree_unflatten(treedef, [fn(*xs) for xs in zip(leaves, *others)])


def tree_reduce(fn: Callable, tree, initializer=None):
    leaves = tree_leaves(tree)
    it = iter(leaves)
    acc = next(it) if initializer is None else initializer
    for x in it:
        acc = fn(acc, x)
    return acc


def tree_paths(tree, prefix: str = "") -> List[Tuple[str, Any]]:
    """``[(path, leaf)]`` with '/'-separated paths; used by the checkpoint manifest."""
    out: list = []

    def rec(obj, path):
        t = type(obj)
        if obj is None:
            return
        if t in _REGISTRY:
            children, _ = _REGISTRY[t][0](obj)
            names = getattr(obj, "_tree_child_names", None)
            names = names() if callable(names) else [str(i) for i in range(len(children))]
            for n, c in zip(names, children):
                rec(c, f"{path}/{n}" if path else str(n))
        elif t is dict:
            for k, v in obj.items():
                rec(v, f"{path}/{k}" if path else str(k))
        elif t in (list, tuple) or _is_namedtuple(obj):
            for i, v in enumerate(obj):
                rec(v, f"{path}/{i}" if path else str(i))
        else:
            out.append((path, obj))

    rec(tree, prefix)
    return out
