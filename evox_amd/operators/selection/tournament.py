"""Tournament selection (reference: ``src/evox/operators/selection/tournament.py:10-69``)."""
from __future__ import annotations

from typing import Callable, Optional

import torch

from ...core import random as rnd


def _lex_argmin(keys: torch.Tensor) -> torch.Tensor:
    """Index of the lexicographically smallest row along dim 1 of ``keys`` (r, s, k); the LAST
    key column is the primary one (numpy ``lexsort`` convention)."""
    r, s, k = keys.shape
    order = torch.arange(s, device=keys.device).expand(r, s)
    for c in range(k):                      # stable sorts from least to most significant key
        col = torch.gather(keys[:, :, c], 1, order)
        order = torch.gather(order, 1, torch.argsort(col, dim=1, stable=True))
    return order[:, 0]


def tournament_single_fit(key, pop, fit, n_round, tournament_func=None, tournament_size=2):
    chosen = rnd.randint(key, (n_round, tournament_size), 0, fit.shape[0])
    cand = fit[chosen]
    if tournament_func is None or tournament_func is torch.argmin:
        winner = torch.argmin(cand, dim=1)
    else:
        winner = torch.stack([torch.as_tensor(tournament_func(c)) for c in cand]).to(chosen.device)
    index = chosen[torch.arange(n_round, device=chosen.device), winner]
    return pop[index], index


def tournament_multi_fit(key, pop, fit, n_round, tournament_func=None, tournament_size=2):
    chosen = rnd.randint(key, (n_round, tournament_size), 0, fit.shape[0])
    winner = _lex_argmin(fit[chosen])
    index = chosen[torch.arange(n_round, device=chosen.device), winner]
    return pop[index], index


class Tournament:
    """``n_round`` tournaments of ``tournament_size`` random individuals each; the winner is
    chosen by ``tournament_func`` (default arg-min of the fitness) or, with
    ``multi_objective=True``, lexicographically over several keys (last key most important)."""

    def __init__(self, n_round: int, tournament_func: Optional[Callable] = None, tournament_size: int = 2,
                 multi_objective: bool = False):
        self.n_round = n_round
        self.tournament_func = tournament_func
        self.tournament_size = tournament_size
        self.multi_obj = multi_objective

    def __call__(self, key, pop, *args):
        if self.multi_obj:
            fit = torch.stack([a.float() for a in args], dim=1) if len(args) > 1 else args[0]
            if fit.dim() == 1:
                fit = fit[:, None]
            return tournament_multi_fit(key, pop, fit, self.n_round, self.tournament_func, self.tournament_size)
        return tournament_single_fit(key, pop, args[0], self.n_round, self.tournament_func, self.tournament_size)
