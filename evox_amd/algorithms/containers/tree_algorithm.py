"""One base algorithm per leaf of a parameter tree (e.g. per layer of a network).
Reference: ``src/evox/algorithms/containers/tree_algorithm.py:8-46``."""
from __future__ import annotations

from ...core.algorithm import Algorithm
from ...core.module import ModuleStack
from ...core.tree import tree_flatten, tree_map, tree_unflatten
from ._members import member_ask, member_tell


class FlattenParam:
    def __init__(self, dummy_input):
        self.shape_def = tree_map(lambda x: tuple(x.shape), dummy_input)

    def flatten(self, x):
        return tree_map(lambda t: t.reshape(t.shape[0], -1), x)

    def unflatten(self, x):
        leaves, treedef = tree_flatten(x)
        shapes, _ = tree_flatten(self.shape_def, is_leaf=lambda s: isinstance(s, tuple))
        return tree_unflatten(treedef, [t.reshape(-1, *s) for t, s in zip(leaves, shapes)])


class TreeAlgorithm(Algorithm):
    """``TreeAlgorithm(AlgoClass, initial_params, *arg_trees)`` builds ``AlgoClass(*leaf_args)``
    for every leaf position of the argument trees (which share ``initial_params``' structure);
    ``ask`` returns a tree of populations shaped ``(pop, *leaf_shape)`` and every member is
    told the same fitness."""

    def __init__(self, base_algorithm, initial_params, *args):
        super().__init__()
        self.flatten_param = FlattenParam(initial_params)
        built = tree_map(base_algorithm, *args)
        inner, self._treedef = tree_flatten(built, is_leaf=lambda x: isinstance(x, Algorithm))
        self.inner = ModuleStack(inner)

    def _ask(self, state, first):
        pops = []
        for m in self.inner:
            pop, state = member_ask(m, state, first)
            pops.append(pop)
        return self.flatten_param.unflatten(tree_unflatten(self._treedef, pops)), state

    def _tell(self, state, fitness, first):
        for m in self.inner:
            state = member_tell(m, state, fitness, first)
        return state

    def init_ask(self, state):
        return self._ask(state, True)

    def init_tell(self, state, fitness):
        return self._tell(state, fitness, True)

    def ask(self, state):
        return self._ask(state, False)

    def tell(self, state, fitness):
        return self._tell(state, fitness, False)
