"""Host-side behaviour of the modules whose fast paths are GPU kernels: on a machine without a GPU
every one of them must take its tensor formulation (or refuse loudly), with the documented semantics."""
import pytest
import torch

from evox_amd.core import random as rnd
from evox_amd.ops import blas, es, nds, sort, variation


def test_argsort_with_rank_is_stable_and_consistent():
    keys = torch.tensor([3.0, 1.0, 2.0, 1.0, float("inf"), -1.0, 2.0])
    order, rank = sort.argsort_with_rank(keys)
    assert order.tolist() == [5, 1, 3, 2, 6, 0, 4]                    # ties keep their input order
    assert rank[order].tolist() == list(range(7))
    assert sort.argsort(keys).tolist() == order.tolist()
    k, v = sort.sort_by_key(keys, torch.arange(7))
    assert k.tolist() == sorted(keys.tolist()) and v.tolist() == order.tolist()


def test_native_blas_is_gpu_only():
    assert blas.available() == torch.cuda.is_available() or not torch.cuda.is_available()
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            blas.gemm(torch.zeros(4, 4), torch.zeros(4, 4))


def test_rank_crowding_select_matches_the_sort_formulation_on_host():
    g = torch.Generator().manual_seed(3)
    fit = torch.rand(400, 3, generator=g)
    k = 150
    idx = nds.rank_crowding_select(fit, k)
    rank = nds.non_dominated_rank(fit, until=k)
    ref = nds._truncate_from_rank(fit, rank, k)
    assert torch.equal(idx, ref)
    # survivors are whole fronts plus the most isolated members of the cut front
    worst = int(rank[idx].max())
    assert bool((rank[idx] <= worst).all()) and int((rank < worst).sum()) <= k
    chosen = set(idx.tolist())
    assert all(i in chosen for i in torch.nonzero(rank < worst).flatten().tolist())


def test_cast_population_keeps_shape_and_values():
    center = torch.linspace(-1, 1, 10)
    pop = es.perturb(rnd.PRNGKey(0), center, 0.1, 8, mirrored=True)
    half = es.cast_population(pop, torch.bfloat16)
    assert half.dtype == torch.bfloat16 and half.shape == pop.shape
    assert torch.allclose(half.float(), pop, atol=1e-2)
    # mirrored layout: [+z; −z] around the centre
    assert torch.allclose(pop[:4] + pop[4:], 2 * center.expand(4, -1), atol=1e-6)


def test_fused_variation_is_not_claimed_on_host():
    x = torch.rand(10, 4)
    assert not variation.sbx_pm_supported(x, torch.zeros(4), torch.ones(4))
