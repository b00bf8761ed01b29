"""Spatial articulated-body engine and the ant task (``problems/neuroevolution/reinforcement_learning/spatial_physics.py``).

The reference reaches ``ant`` through Brax (``src/evox/problems/neuroevolution/reinforcement_learning/brax.py``,
``tests/visualize/test_brax.py:8``), which is not importable here: the dynamics are checked against mechanics —
conservation of energy, linear and angular momentum of free trees, agreement with the planar engine on a mechanism that
moves in a plane, a body at rest on the ground carrying its weight.  Parity with Brax's own trajectories is unpinned."""
import math

import pytest
import torch

import evox_amd as ex
from evox_amd.problems.neuroevolution.reinforcement_learning import planar_physics as pp
from evox_amd.problems.neuroevolution.reinforcement_learning import spatial_physics as sp
from evox_amd.problems.neuroevolution.reinforcement_learning import Brax, make_tensor_env


def _random_tree(gravity):
    bodies = [sp.Body("root", -1, 2.0, com=(0.1, -0.05, 0.02), inertia=(0.05, 0.08, 0.03)),
              sp.Body("a", 0, 1.0, com=(0.2, 0.0, 0.1), inertia=(0.01, 0.02, 0.015), joint_pos=(0.3, 0.1, 0.0), axis=(0.0, 1.0, 0.2)),
              sp.Body("b", 1, 0.7, com=(0.0, 0.15, 0.0), inertia=(0.004, 0.002, 0.004), joint_pos=(0.4, 0.0, 0.2), axis=(1.0, 0.0, 0.0)),
              sp.Body("c", 0, 1.3, com=(-0.2, 0.1, 0.0), inertia=(0.02, 0.02, 0.03), joint_pos=(-0.2, 0.0, 0.1), axis=(0.3, 0.3, 1.0)),
              sp.Body("d", 3, 0.5, com=(0.0, 0.0, -0.2), inertia=(0.006, 0.006, 0.001), joint_pos=(-0.3, 0.2, 0.0), axis=(0.0, 1.0, 0.0))]
    return sp.SpatialEngine(sp.SpatialModel(bodies, gravity=gravity, dt=1e-4, substeps=1, max_velocity=1e9))


def _random_state(eng, batch, seed):
    g = torch.Generator().manual_seed(seed)
    pos = torch.randn(batch, 3, generator=g, dtype=torch.float64)
    quat = torch.nn.functional.normalize(torch.randn(batch, 4, generator=g, dtype=torch.float64), dim=1)
    theta = torch.randn(batch, eng.nj, generator=g, dtype=torch.float64)
    vel = torch.randn(batch, eng.ndof, generator=g, dtype=torch.float64) * 1.5
    return pos, quat, theta, vel


def test_free_trees_conserve_energy_and_both_momenta():
    eng = _random_tree(0.0)
    st = _random_state(eng, 6, 0)
    lin0, ang0, e0 = eng.momenta_energy(*st)
    for _ in range(3000):
        st = eng.step_tensor(*st, None)
    lin, ang, e = eng.momenta_energy(*st)
    assert float((lin - lin0).abs().max()) < 2e-3 and float((ang - ang0).abs().max()) < 2e-3
    assert float(((e - e0) / e0.abs()).abs().max()) < 2e-3
    assert float((st[1].norm(dim=1) - 1).abs().max()) < 1e-12
    # under gravity: energy stays, the vertical momentum falls by weight x time, the horizontal one stays
    eng = _random_tree(9.81)
    st = _random_state(eng, 6, 1)
    lin0, _, e0 = eng.momenta_energy(*st)
    for _ in range(2000):
        st = eng.step_tensor(*st, None)
    lin, _, e = eng.momenta_energy(*st)
    total = float(eng.mass.sum())
    assert float(((e - e0) / e0.abs().clamp_min(1.0)).abs().max()) < 3e-3
    assert float((lin[:, :2] - lin0[:, :2]).abs().max()) < 2e-3
    assert torch.allclose(lin[:, 2] - lin0[:, 2], torch.full((6,), -total * 9.81 * 0.2, dtype=torch.float64), atol=5e-3)


def test_a_mechanism_that_moves_in_a_plane_matches_the_planar_engine():
    """Free-floating three-link chain with all hinge axes along −y, started in the x–z plane: the generalized
    accelerations equal those of the planar engine (whose angles run counter-clockwise in x–z, i.e. about −y)."""
    l0, l1, l2, m0, m1, m2 = 0.7, 0.5, 0.4, 2.0, 1.0, 0.6
    planar = pp.PlanarEngine(pp.PlanarModel(
        [pp.Link("a", -1, l0, m0, joint_at=0.0), pp.Link("b", 0, l1, m1, attach=l0), pp.Link("c", 1, l2, m2, attach=l1)]))
    for n in ("anc", "rest_abs", "W_com", "W_con", "mass", "M_const", "lo", "hi", "damping", "stiffness", "gear", "lengths"):
        setattr(planar, n, getattr(planar, n).double())
    rod = lambda m, l: (1e-9, m * l * l / 12, m * l * l / 12)
    spatial = sp.SpatialEngine(sp.SpatialModel(
        [sp.Body("a", -1, m0, com=(l0 / 2, 0, 0), inertia=rod(m0, l0)),
         sp.Body("b", 0, m1, com=(l1 / 2, 0, 0), inertia=rod(m1, l1), joint_pos=(l0, 0, 0), axis=(0, -1, 0)),
         sp.Body("c", 1, m2, com=(l2 / 2, 0, 0), inertia=rod(m2, l2), joint_pos=(l1, 0, 0), axis=(0, -1, 0))]))
    g = torch.Generator().manual_seed(3)
    B = 16
    q = torch.randn(B, 5, generator=g, dtype=torch.float64)
    qd = torch.randn(B, 5, generator=g, dtype=torch.float64)
    want = planar.acceleration(q, qd, None)                                    # (ẍ, z̈, φ̈₀, θ̈₁, θ̈₂)
    pos = torch.stack([q[:, 0], torch.zeros(B, dtype=torch.float64), q[:, 1]], dim=1)
    half = q[:, 2] / 2                                                          # rotation by φ₀ about −y
    quat = torch.stack([torch.cos(half), torch.zeros_like(half), -torch.sin(half), torch.zeros_like(half)], dim=1)
    zero = torch.zeros(B, dtype=torch.float64)
    vel = torch.stack([qd[:, 0], zero, qd[:, 1], zero, -qd[:, 2], zero, qd[:, 3], qd[:, 4]], dim=1)
    got = spatial.acceleration(pos, quat, q[:, 3:], vel, None)
    assert torch.allclose(got[:, 0], want[:, 0], atol=1e-8) and torch.allclose(got[:, 2], want[:, 1], atol=1e-8)
    assert torch.allclose(-got[:, 4], want[:, 2], atol=1e-8) and torch.allclose(got[:, 6:], want[:, 3:], atol=1e-8)
    assert float(got[:, [1, 3, 5]].abs().max()) < 1e-9                         # nothing leaves the plane


def test_ant_has_the_published_sizes_stays_finite_and_comes_to_rest_on_its_feet():
    env = make_tensor_env("ant")
    assert (env.obs_dim, env.act_dim) == (27, 8) and env.engine.ndof == 14
    st, o = env.reset(ex.PRNGKey(1), 6, torch.device("cpu"))
    assert o.shape == (6, 27)
    gen = torch.Generator().manual_seed(2)
    for t in range(120):
        st, o, r, d = env.step(st, torch.rand(6, 8, generator=gen) * 4 - 2)
        assert o.shape == (6, 27) and r.shape == (6,) and d.dtype == torch.bool
        assert bool(torch.isfinite(o).all()) and bool(torch.isfinite(r).all()), t
    st, _ = env.reset(ex.PRNGKey(3), 2, torch.device("cpu"))
    for _ in range(500):                                                        # (the geared joints settle with a time constant of a second)
        st, o, r, d = env.step(st, torch.zeros(2, 8))
    assert not bool(d.any()) and float(st["vel"].abs().max()) < 0.15
    eng = env.engine
    c = eng.on(torch.device("cpu"))
    R, x, _ = eng.kinematics(st["pos"], st["quat"], st["theta"])
    pts = x[:, c["con_body"]] + torch.einsum("bpij,pj->bpi", R[:, c["con_body"]], c["con_local"])
    assert float(pts[..., 2].min()) > -0.02
    fn = (eng.model.contact_stiffness * (-pts[..., 2]).clamp_min(0)).sum(1)
    assert torch.allclose(fn, torch.full_like(fn, float(c["mass"].sum()) * eng.model.gravity), rtol=0.06)
    assert float(st["pos"][:, 2].min()) > 0.26                                  # on the four feet (ankles at their limit), not on the belly (0.25)


def test_brax_ant_rollout_and_visualisation():
    """The reference's visualisation test drives ``ant`` with a random policy (``tests/visualize/test_brax.py``)."""
    def policy(w, obs):
        return torch.tanh(obs @ w)

    prob = Brax(policy, "ant", 15, 2)
    w = 0.1 * torch.randn(3, 27, 8, generator=torch.Generator().manual_seed(0))
    fit, _ = prob.evaluate(prob.setup(ex.PRNGKey(0)), w)
    assert fit.shape == (3,) and bool(torch.isfinite(fit).all())
    frames = prob.visualize(ex.PRNGKey(0), w[0], output_type="rgb_array", width=120, height=90)
    assert len(frames) == 16 and frames[0].shape == (90, 120, 3)
    assert int((frames[0] == (30, 60, 140)).all(-1).sum()) > 30


@pytest.mark.gpu
def test_spatial_step_kernel_matches_the_tensor_formulation():
    """``csrc/spatial.hip`` (one environment per lane) against ``SpatialEngine.step_tensor`` in fp64 from the same states:
    a free tree of five bodies with skew axes, and the ant along a rollout under random controls (feet touching down
    and lifting, ankles at their limits) — every environment step agrees to fp32 rounding."""
    dev = torch.device("cuda")
    eng = _random_tree(9.81)
    eng.model.dt, eng.model.substeps = 2e-3, 4
    st = tuple(t.float().to(dev) for t in _random_state(eng, 200, 5))
    for t in range(20):
        got = eng.step(*st, None)
        want = eng.step_tensor(*(x.double() for x in st), None)
        for g, w in zip(got, want):
            assert float(((g.double() - w) / (1 + w.abs())).abs().max()) < 1e-4, t
        st = got
    env = make_tensor_env("ant")
    e = env.engine
    s, _ = env.reset(ex.PRNGKey(11).to(dev), 256, dev)
    st = (s["pos"], s["quat"], s["theta"], s["vel"])
    gen = torch.Generator().manual_seed(12)
    for t in range(100):
        a = (torch.rand(256, 8, generator=gen) * 2 - 1).to(dev)
        got = e.step(*st, a)
        assert got[0].data_ptr() != st[0].data_ptr()
        want = e.step_tensor(*(x.double() for x in st), a.double())
        for g, w in zip(got, want):
            assert float(((g.double() - w) / (1 + w.abs())).abs().max()) < 2e-3, t
        st = got


@pytest.mark.gpu
def test_ant_fused_rollout_step_matches_step_plus_bookkeeping_and_records():
    """``Ant.step_accumulate`` (one launch, in place) against ``step`` plus the rollout's own bookkeeping from the same
    states; and OpenES over MLP policies on the ant replays from a graph with the generations of eager issue."""
    from evox_amd import algorithms, workflows
    from evox_amd.models import MLP

    dev = torch.device("cuda")
    env = make_tensor_env("ant")
    st, _ = env.reset(ex.PRNGKey(21).to(dev), 512, dev)
    total = torch.zeros(512, device=dev)
    done = torch.zeros(512, dtype=torch.bool, device=dev)
    gen = torch.Generator().manual_seed(22)
    for t in range(60):
        a = (torch.rand(512, 8, generator=gen) * 3 - 1.5).to(dev)
        before = {k: v.clone() for k, v in st.items()}
        st2, obs2, reward, step_done = env.step(before, a)
        want_total = total + torch.where(done, torch.zeros_like(reward), reward)
        want_done = done | step_done
        st, obs, total, done = env.step_accumulate(st, a, total, done)
        assert torch.allclose(obs, obs2, rtol=1e-4, atol=1e-4), (t, float((obs - obs2).abs().max()))
        assert torch.allclose(total, want_total, rtol=1e-4, atol=1e-3), t
        assert int((done != want_done).sum()) <= 2, t
        total, done = want_total.clone(), want_done.clone()
        st = {k: v.clone() for k, v in st2.items()}
    net = MLP([27, 32, 8])

    def run(graph):
        prob = Brax(net.batched_apply, "ant", 30, 1)
        algo = algorithms.OpenES(net.init(ex.PRNGKey(1), device=dev), 256, 0.05, 0.1)
        wf = workflows.StdWorkflow(algo, prob, opt_direction="max", graph_step=graph)
        state = wf.init(ex.PRNGKey(2).to(dev))
        for _ in range(8):
            state = wf.step(state)
        return wf, state.get_child_state("algorithm").center.clone()

    we, ce = run(False)
    wg, cg = run(True)
    assert wg.graph_step is True and wg._graphed
    assert torch.equal(ce, cg)
