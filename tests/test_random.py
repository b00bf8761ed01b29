"""Counter-based PRNG: determinism, shard-invariance, statistics, CPU/GPU bit-exactness."""
import math

import pytest
import torch

from evox_amd.core import random as rnd


def test_keys_and_split_are_deterministic():
    k = rnd.PRNGKey(7)
    assert torch.equal(k, rnd.PRNGKey(7))
    assert not torch.equal(k, rnd.PRNGKey(8))
    ks = rnd.split(k, 4)
    assert ks.shape == (4, 2) and len({tuple(r.tolist()) for r in ks}) == 4
    assert torch.equal(ks, rnd.split(k, 4))
    assert torch.equal(rnd.split(k, 2), ks[:2])
    batch = rnd.split(ks, 3)
    assert batch.shape == (4, 3, 2)
    assert torch.equal(batch[2], rnd.split(ks[2], 3))
    assert not torch.equal(rnd.fold_in(k, 1), rnd.fold_in(k, 2))


@pytest.mark.parametrize("draw", ["uniform", "normal", "bits"])
def test_offset_windows_tile_the_logical_tensor(draw):
    k = rnd.PRNGKey(11)
    fn = getattr(rnd, draw)
    full = fn(k, (10, 7))
    for r0, r1 in [(0, 3), (3, 4), (4, 10), (1, 9)]:
        part = fn(k, (r1 - r0, 7), offset=r0 * 7)
        assert torch.equal(part, full[r0:r1])


def test_moments():
    k = rnd.PRNGKey(2024)
    n = 1 << 20
    u = rnd.uniform(k, (n,))
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 2e-3 and abs(float(u.var()) - 1 / 12) < 1e-3
    z = rnd.normal(k, (n,))
    assert abs(float(z.mean())) < 4e-3 and abs(float(z.std()) - 1) < 4e-3
    assert abs(float((z ** 3).mean())) < 2e-2 and abs(float((z ** 4).mean()) - 3) < 6e-2
    assert float(z.abs().max()) < 6.0
    # no correlation between successive draws of the same stream
    assert abs(float((z[:-1] * z[1:]).mean())) < 4e-3
    i = rnd.randint(k, (n,), 3, 13)
    assert int(i.min()) == 3 and int(i.max()) == 12
    counts = torch.bincount(i - 3, minlength=10).float() / n
    assert float((counts - 0.1).abs().max()) < 3e-3


def test_normal_matches_inverse_cdf_quantiles():
    z = rnd.normal(rnd.PRNGKey(5), (200000,)).sort().values
    for q in (0.01, 0.1, 0.5, 0.9, 0.99):
        expect = math.sqrt(2) * torch.erfinv(torch.tensor(2 * q - 1)).item()
        assert abs(float(z[int(q * z.numel())]) - expect) < 2e-2


def test_permutation_choice_and_friends():
    k = rnd.PRNGKey(3)
    p = rnd.permutation(k, 50)
    assert sorted(p.tolist()) == list(range(50))
    x = torch.arange(20.0).reshape(10, 2)
    px = rnd.permutation(k, x)
    assert sorted(px[:, 0].tolist()) == x[:, 0].tolist()
    c = rnd.choice(k, 30, (10,), replace=False)
    assert len(set(c.tolist())) == 10
    w = torch.tensor([0.0, 0.0, 1.0, 0.0])
    assert set(rnd.choice(k, 4, (20,), p=w).tolist()) == {2}
    assert set(rnd.choice(k, 4, (1,), replace=False, p=w).tolist()) == {2}
    bp = rnd.batched_permutation(k, 5, 9)
    assert bp.shape == (5, 9) and all(sorted(r.tolist()) == list(range(9)) for r in bp)
    g = rnd.geometric(k, 0.25, (20000,))
    assert int(g.min()) >= 1 and abs(float(g.float().mean()) - 4.0) < 0.15
    ca = rnd.cauchy(k, (20001,))
    assert abs(float(ca.median())) < 0.05
    cat = rnd.categorical(k, torch.log(torch.tensor([0.2, 0.8])), (20000,))
    assert abs(float(cat.float().mean()) - 0.8) < 0.02


@pytest.mark.gpu
def test_gpu_draws_are_bit_identical_to_cpu(cuda):
    k = rnd.PRNGKey(99)
    kg = k.to(cuda)
    for shape, off in [((1000, 33), 0), ((517,), 3), ((64, 64), 4096), ((5,), 1)]:
        assert torch.equal(rnd.bits(k, shape, offset=off), rnd.bits(kg, shape, offset=off).cpu())
        assert torch.equal(rnd.uniform(k, shape, offset=off), rnd.uniform(kg, shape, offset=off).cpu())
        assert torch.equal(rnd.normal(k, shape, offset=off), rnd.normal(kg, shape, offset=off).cpu())
        assert torch.equal(rnd.randint(k, shape, -5, 77, offset=off), rnd.randint(kg, shape, -5, 77, offset=off).cpu())
    assert torch.equal(rnd.split(k, 9), rnd.split(kg, 9).cpu())
    assert torch.equal(rnd.fold_in(k, 12345), rnd.fold_in(kg, 12345).cpu())
