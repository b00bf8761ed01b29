"""Helpers shared by the containers: drive member algorithms through the root state."""
from __future__ import annotations

import copy

from ...core.algorithm import has_init_ask, has_init_tell
from ...core.module import ModuleStack, Stateful, use_state


def as_stack(base, count=None, stacked=False) -> ModuleStack:
    """A ModuleStack from one algorithm (copied ``count`` times), a list, a ModuleStack, or –
    with ``stacked=True`` – a leaf-wise stacked dataclass module, which is split back into its
    ``count`` members."""
    import dataclasses
    from ...core.tree import tree_map
    if isinstance(base, ModuleStack):
        return base
    if isinstance(base, (list, tuple)):
        return ModuleStack(list(base))
    if stacked and dataclasses.is_dataclass(base):
        return ModuleStack([tree_map(lambda x: x[i], base) for i in range(count)])
    if isinstance(base, Stateful) and count is not None:
        return ModuleStack([copy.copy(base) for _ in range(count)])
    raise TypeError(f"cannot build a module stack from {type(base)}")


def member_ask(member, state, first):
    fn = member.init_ask if (first and has_init_ask(member)) else member.ask
    return use_state(fn)(state)


def member_tell(member, state, fitness, first):
    fn = member.init_tell if (first and has_init_tell(member)) else member.tell
    return use_state(fn)(state, fitness)
