"""Fixtures from the reference's ``tests/test_metrics.py:7-59``."""
from math import isclose

import torch

import evox_amd as ex
from evox_amd.metrics import GD, HV, IGD, GDPlus, IGDPlus


def test_gd_and_igd():
    pf = torch.tensor([[0, 5], [1, 4], [2, 3], [3, 2], [4, 1], [5.5, 0]])
    objs = torch.tensor([[0, 6], [5, 8], [4.3, 2]])
    assert isclose(float(GD(pf)(objs)), 2.5669618, rel_tol=1e-4)
    assert isclose(float(GDPlus(pf)(objs)), 2.5669618, rel_tol=1e-4)
    assert isclose(float(IGD(pf)(objs)), 1.7367444, rel_tol=1e-4)
    assert isclose(float(IGDPlus(pf)(objs)), 1.6073387, rel_tol=1e-4)
    assert isclose(float(GD(pf)(objs)), float(IGD(objs)(pf)), abs_tol=1e-4)


def test_hv_monte_carlo():
    key = ex.PRNGKey(0)
    ref = torch.tensor([-1.0, -1.0])
    objs = torch.tensor([[1.0, 9], [2, 2], [3, 1]])
    for method in ("bounding_cube", "each_cube"):
        assert isclose(float(HV(ref, 100_000, method)(key, objs)), 25, rel_tol=1e-2)
    ref = torch.tensor([10, 9, 8.0])
    objs = torch.tensor([[0.1, 7, 0.3], [5.5, 0, 2.3], [-0.1, 1, -1], [3, 2, 5.5]])
    for method in ("bounding_cube", "each_cube"):
        assert isclose(float(HV(ref, 100_000, method)(key, objs)), 753, rel_tol=1e-2)


def test_hv_exact():
    from evox_amd.ops import hv
    ref = torch.tensor([10, 9, 8.0])
    objs = torch.tensor([[0.1, 7, 0.3], [5.5, 0, 2.3], [-0.1, 1, -1], [3, 2, 5.5]])
    exact = float(hv.hypervolume_exact(objs, ref))
    assert isclose(exact, 753, rel_tol=1e-2)
