"""Symmetric eigensolver for gfx950 (warm-started block Jacobi, ``csrc/jacobi_eigh.hip``)."""
from __future__ import annotations

from ._util import hip


def available() -> bool:
    try:
        getattr(hip(), "evox_hip_jacobi_eigh")
        return True
    except AttributeError:
        return False


def jacobi_eigh(C, B_prev=None):
    raise NotImplementedError
