"""Sorting helpers (argsort / top-k / rank / companion-array sort)."""
from __future__ import annotations

import torch


def sort_by_key(keys: torch.Tensor, *vals):
    """Sort ``keys`` ascending and permute every companion array the same way
    (reference: ``src/evox/algorithms/so/es_variants/sort_utils.py:5-12``)."""
    order = torch.argsort(keys, stable=True)
    return (keys[order], *[v[order] for v in vals])


def argsort(keys: torch.Tensor) -> torch.Tensor:
    return torch.argsort(keys, stable=True)
