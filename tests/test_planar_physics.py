"""Planar articulated-body engine and the tasks on it (``problems/neuroevolution/reinforcement_learning/planar_physics.py``).

The reference reaches these tasks through Brax (``src/evox/problems/neuroevolution/reinforcement_learning/brax.py``), which is
not importable here: the dynamics are checked against mechanics instead — closed-form cart-pole accelerations, energy
conservation of free chains, a body at rest on the ground carrying its weight — and the tasks against their published
observation / action sizes and termination rules.  Parity with Brax's own trajectories is unpinned."""
import math

import pytest
import torch

import evox_amd as ex
from evox_amd.problems.neuroevolution.reinforcement_learning import planar_physics as pp
from evox_amd.problems.neuroevolution.reinforcement_learning import Brax, make_tensor_env


def test_cart_pole_accelerations_match_the_closed_form():
    """Cart (mass M) with a uniform pole (mass m, length l) under a force F: the textbook equations of motion."""
    M, m, l, g = 2.0, 0.5, 0.8, 9.81
    links = [pp.Link("cart", -1, 0.0, M, inertia=0.0), pp.Link("pole", 0, l, m, rest=math.pi / 2)]
    eng = pp.PlanarEngine(pp.PlanarModel(links, free_base=(True, False, False), actuators=[(0, 1.0)]))
    gen = torch.Generator().manual_seed(0)
    q = torch.zeros(64, 4, dtype=torch.float64)
    qd = torch.zeros(64, 4, dtype=torch.float64)
    q[:, 3] = torch.rand(64, generator=gen, dtype=torch.float64) * 2 - 1          # pole angle from the vertical (counter-clockwise)
    qd[:, 0] = torch.randn(64, generator=gen, dtype=torch.float64)
    qd[:, 3] = torch.randn(64, generator=gen, dtype=torch.float64)
    F = torch.randn(64, 1, generator=gen, dtype=torch.float64) * 5
    for name in ("anc", "rest_abs", "W_com", "W_con", "mass", "M_const", "lo", "hi", "damping", "stiffness", "gear", "lengths"):
        setattr(eng, name, getattr(eng, name).double())
    qdd = eng.acceleration(q, qd, F)
    # Lagrangian with θ counter-clockwise from the vertical, centre of the pole at (x − (l/2) sin θ, (l/2) cos θ):
    #   (M + m) ẍ − (m l / 2) cos θ θ̈ + (m l / 2) sin θ θ̇² = F
    #   (m l² / 3) θ̈ − (m l / 2) cos θ ẍ − (m g l / 2) sin θ = 0
    th, thd = q[:, 3], qd[:, 3]
    a11, a12 = torch.full_like(th, M + m), -(m * l / 2) * torch.cos(th)
    a22 = torch.full_like(th, m * l * l / 3)
    b1 = F[:, 0] - (m * l / 2) * torch.sin(th) * thd ** 2
    b2 = (m * g * l / 2) * torch.sin(th)
    det = a11 * a22 - a12 * a12
    xdd, thdd = (b1 * a22 - a12 * b2) / det, (a11 * b2 - a12 * b1) / det
    assert torch.allclose(qdd[:, 0], xdd, atol=1e-9) and torch.allclose(qdd[:, 3], thdd, atol=1e-9)
    assert float(qdd[:, 1:3].abs().max()) == 0.0                                    # locked coordinates stay put


def test_free_chains_conserve_energy_and_momentum():
    """A floating three-link tree (a branch on the root) tumbling under gravity without actuation, damping, limits or
    contact: total energy and horizontal momentum stay put to the integrator's order."""
    links = [pp.Link("a", -1, 0.7, 2.0, rest=0.3, joint_at=0.2), pp.Link("b", 0, 0.5, 1.0, rest=1.0, attach=0.7),
             pp.Link("c", 0, 0.4, 0.7, rest=-2.0, attach=0.1, joint_at=0.1), pp.Link("d", 1, 0.3, 0.4, rest=0.5, attach=0.5)]
    eng = pp.PlanarEngine(pp.PlanarModel(links, dt=2e-4, substeps=1))
    for name in ("anc", "rest_abs", "W_com", "W_con", "mass", "M_const", "lo", "hi", "damping", "stiffness", "gear", "lengths"):
        setattr(eng, name, getattr(eng, name).double())
    gen = torch.Generator().manual_seed(1)
    q = torch.randn(8, 6, generator=gen, dtype=torch.float64)
    q[:, 1] += 5.0
    qd = torch.randn(8, 6, generator=gen, dtype=torch.float64) * 2
    e0 = eng.energy(q, qd)

    def px(q, qd):
        c = eng.on(q.device)
        _, u = eng.directions(q)
        up = torch.stack([-u[..., 1], u[..., 0]], dim=-1)
        ad = qd[:, 2:] @ c["anc"].T
        vc = qd[:, None, :2] + torch.einsum("pk,bk,bkc->bpc", c["W_com"], ad, up)
        return (c["mass"] * vc[..., 0]).sum(-1)

    p0 = px(q, qd)
    for _ in range(2000):
        q, qd = eng.step(q, qd, None)
    assert float(((eng.energy(q, qd) - e0) / e0.abs()).abs().max()) < 2e-3
    assert float((px(q, qd) - p0).abs().max()) < 2e-3                          # of momenta of order 10


@pytest.mark.parametrize("name,obs,act", [("inverted_pendulum", 4, 1), ("inverted_double_pendulum", 8, 1), ("reacher", 11, 2),
                                          ("swimmer", 8, 2), ("hopper", 11, 3), ("walker2d", 17, 6), ("halfcheetah", 17, 6)])
def test_tasks_have_the_published_sizes_and_stay_finite_under_random_controls(name, obs, act):
    env = make_tensor_env(name)
    assert (env.obs_dim, env.act_dim) == (obs, act)
    key = ex.PRNGKey(3)
    st, o = env.reset(key, 16, torch.device("cpu"))
    assert o.shape == (16, obs)
    gen = torch.Generator().manual_seed(4)
    total = torch.zeros(16)
    for t in range(150):
        a = torch.rand(16, act, generator=gen) * 2 - 1
        st, o, r, d = env.step(st, a * 3)                      # beyond the control range: clipped by the task
        assert o.shape == (16, obs) and r.shape == (16,) and d.dtype == torch.bool
        assert bool(torch.isfinite(o).all()) and bool(torch.isfinite(r).all()), (name, t)
        total += r
    assert float(st["q"].abs().max()) < 1e3


def test_bodies_come_to_rest_on_the_ground_carrying_their_weight():
    """Hopper, walker and cheetah dropped from their initial pose with zero controls: after four seconds nothing has passed
    through the floor by more than a centimetre, and (cheetah, which stands on its spring-loaded legs) the contact
    forces balance the weight."""
    for name in ("hopper", "walker2d", "halfcheetah"):
        env = make_tensor_env(name)
        st, _ = env.reset(ex.PRNGKey(0), 2, torch.device("cpu"))
        zero = torch.zeros(2, env.act_dim)
        for _ in range(500):
            st, o, r, d = env.step(st, zero)
        s, e = env.engine.link_segments(st["q"])
        assert float(torch.minimum(s[..., 1], e[..., 1]).min()) > -0.01, name
        assert bool(torch.isfinite(st["q"]).all())
    eng = env.engine
    c = eng.on(torch.device("cpu"))
    _, u = eng.directions(st["q"])
    pos_z = st["q"][:, None, 1] + torch.einsum("pk,bk->bp", c["W_con"], u[..., 1])
    fn = (eng.model.contact_stiffness * (-pos_z).clamp_min(0)).sum(1)
    weight = float(c["mass"].sum()) * eng.model.gravity
    assert float(st["qd"].abs().max()) < 0.05                                   # at rest
    assert torch.allclose(fn, torch.full_like(fn, weight), rtol=0.02)


def test_inverted_pendulum_balances_under_a_linear_controller_and_falls_without():
    env = make_tensor_env("inverted_pendulum")
    dev = torch.device("cpu")
    st, o = env.reset(ex.PRNGKey(1), 8, dev)
    alive = torch.ones(8, dtype=torch.bool)
    for _ in range(300):
        # observation = (x, θ, ẋ, θ̇) with θ counter-clockwise: the pole leaning to −x (θ > 0) needs the cart pushed to −x
        a = -(0.5 * o[:, 0] + 1.0 * o[:, 2]) * 0.1 - (40.0 * o[:, 1] + 6.0 * o[:, 3]) * 0.1
        st, o, r, d = env.step(st, a[:, None])
        alive &= ~d
    assert bool(alive.all())
    st, o = env.reset(ex.PRNGKey(1), 8, dev)
    fell = torch.zeros(8, dtype=torch.bool)
    for _ in range(300):
        st, o, r, d = env.step(st, torch.zeros(8, 1))
        fell |= d
    assert bool(fell.all())


def test_swimmer_moves_by_undulating_and_reacher_reward_is_the_distance():
    env = make_tensor_env("swimmer")
    st, o = env.reset(ex.PRNGKey(2), 4, torch.device("cpu"))
    x0 = st["q"][:, 0].clone()
    for t in range(400):
        ph = 2 * math.pi * t * env.frame_dt / 1.0
        a = torch.tensor([math.sin(ph), math.sin(ph - 2.0)]).expand(4, 2)
        st, o, r, d = env.step(st, a)
    assert float((st["q"][:, 0] - x0).abs().min()) > 0.2                       # a travelling wave propels the body
    assert not bool(d.any())
    env = make_tensor_env("reacher")
    st, o = env.reset(ex.PRNGKey(5), 6, torch.device("cpu"))
    st2, o2, r, d = env.step(st, torch.zeros(6, 2))
    tip = env.engine.points(st2["q"], env._tip)[:, 0]
    assert torch.allclose(r, -(tip - st2["target"]).norm(dim=1), atol=1e-6)
    assert torch.allclose(o2[:, 8:10], tip - st2["target"], atol=1e-6) and float(o2[:, 10].abs().max()) == 0.0
    assert float(st["target"].norm(dim=1).max()) <= 0.2 + 1e-6


def test_brax_problem_evaluates_a_population_on_a_planar_task_and_openes_improves_it():
    """The reference's entry point (``Brax(policy, env_name, max_episode_length, num_episodes)``) on a physics task: a
    linear policy on the inverted pendulum, improved by OpenES within a few generations."""
    from evox_amd import algorithms, workflows

    def policy(w, obs):                         # w: (pop, 4), obs: (pop, episodes, 4)
        return (obs * w[:, None, :]).sum(-1, keepdim=True)

    prob = Brax(policy, "inverted_pendulum", 100, 2)
    algo = algorithms.OpenES(torch.zeros(4), 32, 0.5, 0.3)
    wf = workflows.StdWorkflow(algo, prob, opt_direction="max")
    state = wf.init(ex.PRNGKey(7))
    for _ in range(12):
        state = wf.step(state)
    center = state.get_child_state("algorithm").center
    fitness, _ = prob.evaluate(prob.setup(ex.PRNGKey(9)), center[None])
    base, _ = prob.evaluate(prob.setup(ex.PRNGKey(9)), torch.zeros(1, 4))
    assert float(base[0]) < 60.0                                  # an idle cart drops the pole within a second
    assert float(fitness[0]) > float(base[0]) + 10.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(pp.PLANAR_TASKS))
def test_step_kernel_matches_the_tensor_formulation(name):
    """``csrc/planar.hip`` (one environment per lane, all sub-steps in one launch) against ``PlanarEngine.step_tensor`` in
    fp64 from the same states: along a rollout under random controls — contacts made and broken, joint limits hit —
    every single environment step agrees to fp32 rounding."""
    dev = torch.device("cuda")
    env = make_tensor_env(name)
    eng = env.engine
    ref = pp.PlanarEngine(eng.model)
    for n in ("anc", "rest_abs", "W_com", "W_con", "mass", "M_const", "lo", "hi", "damping", "stiffness", "gear", "lengths"):
        setattr(ref, n, getattr(ref, n).double())
    st, _ = env.reset(ex.PRNGKey(11).to(dev), 256, dev)
    q, qd = st["q"], st["qd"]
    gen = torch.Generator().manual_seed(12)
    worst = 0.0
    for t in range(120):
        a = (torch.rand(256, env.act_dim, generator=gen) * 2 - 1).to(dev) * env.ctrl_range
        q1, qd1 = eng.step(q, qd, a)
        assert q1.data_ptr() != q.data_ptr()
        q2, qd2 = ref.step_tensor(q.double(), qd.double(), a.double())
        scale_q, scale_v = 1.0 + q2.abs(), 1.0 + qd2.abs()
        err = max(float(((q1.double() - q2) / scale_q).abs().max()), float(((qd1.double() - qd2) / scale_v).abs().max()))
        worst = max(worst, err)
        assert err < 2e-3, (name, t, err)
        q, qd = q1, qd1
    assert worst < 2e-3
    # the environment's own step takes the kernel, finite and of the right shapes
    st, o, r, d = env.step(st, torch.zeros(256, env.act_dim, device=dev))
    assert o.shape == (256, env.obs_dim) and bool(torch.isfinite(o).all())


def test_visualize_draws_the_mechanism():
    """``visualize`` of a planar task renders the links (reference ``brax.py:99-133`` renders the physics scene): frames
    with dark link pixels above the ground line, an HTML page that carries the segments of every time step."""
    def policy(w, obs):
        return torch.tanh(obs @ w)                       # (pop, episodes, obs) @ (pop, obs, act)

    prob = Brax(policy, "hopper", 20, 1)
    w = 0.1 * torch.randn(11, 3, generator=torch.Generator().manual_seed(0))
    frames = prob.visualize(ex.PRNGKey(0), w, output_type="rgb_array", width=160, height=120)
    assert len(frames) == 21 and frames[0].shape == (120, 160, 3) and frames[0].dtype.name == "uint8"
    dark = (frames[0] == (30, 60, 140)).all(-1)
    assert int(dark.sum()) > 100 and int(dark[: int(120 * 0.85)].sum()) > 0.9 * int(dark.sum())       # the body, above the ground
    page = prob.visualize(ex.PRNGKey(0), w, output_type="HTML")
    assert page.startswith("<!DOCTYPE html>") and '"seg"' in page and "scene(t)" in page
    obs, rew, done, shots = prob.rollout(ex.PRNGKey(0), w, snapshots=True)
    assert len(shots) == obs.shape[0] == 21 and tuple(shots[0].shape) == (4, 2, 2)
    assert len(prob.rollout(ex.PRNGKey(0), w)) == 3


@pytest.mark.gpu
def test_policy_search_on_a_physics_task_records_into_a_graph():
    """OpenES over MLP policies on the hopper, whole generation (sampling, 40 policy + physics steps, update) replayed
    from a HIP graph: the same generations as eager issue, and the search improves the return."""
    from evox_amd import algorithms, workflows
    from evox_amd.models import MLP

    dev = torch.device("cuda")
    net = MLP([11, 16, 3])

    def run(graph):
        prob = Brax(net.batched_apply, "hopper", 40, 1)
        algo = algorithms.OpenES(net.init(ex.PRNGKey(1), device=dev), 256, 0.05, 0.1)
        wf = workflows.StdWorkflow(algo, prob, opt_direction="max", graph_step=graph)
        state = wf.init(ex.PRNGKey(2).to(dev))
        for _ in range(8):
            state = wf.step(state)
        return wf, state.get_child_state("algorithm").center.clone()

    we, ce = run(False)
    wg, cg = run(True)
    assert wg.graph_step is True and wg._graphed is not None
    assert torch.allclose(ce, cg, atol=1e-5)
    prob = Brax(net.batched_apply, "hopper", 40, 1)
    start, _ = prob.evaluate(prob.setup(ex.PRNGKey(3).to(dev)), net.init(ex.PRNGKey(1), device=dev)[None])
    end, _ = prob.evaluate(prob.setup(ex.PRNGKey(3).to(dev)), ce[None])
    assert bool(torch.isfinite(end).all()) and float(end[0]) >= float(start[0]) - 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(pp.PLANAR_TASKS))
def test_fused_rollout_step_matches_step_plus_bookkeeping(name):
    """``step_accumulate`` on the GPU (physics, observation, reward, termination and the rollout's accumulation in one
    launch, in place) against ``step`` followed by the loop's own bookkeeping, from the same states, along a rollout
    with controls beyond the task's range: observations, accumulated returns and done flags agree."""
    dev = torch.device("cuda")
    env = make_tensor_env(name)
    st, _ = env.reset(ex.PRNGKey(21).to(dev), 512, dev)
    total = torch.zeros(512, device=dev)
    done = torch.zeros(512, dtype=torch.bool, device=dev)
    gen = torch.Generator().manual_seed(22)
    ended = 0
    for t in range(80):
        a = (torch.rand(512, env.act_dim, generator=gen) * 3 - 1.5).to(dev) * env.ctrl_range
        before = {k: v.clone() for k, v in st.items()}
        total_before, done_before = total.clone(), done.clone()
        st2, obs2, reward, step_done = env.step(before, a)
        want_total = total_before + torch.where(done_before, torch.zeros_like(reward), reward)
        want_done = done_before | step_done
        st, obs, total, done = env.step_accumulate(st, a, total, done)
        assert obs.shape == (512, env.obs_dim)
        assert torch.allclose(obs, obs2, rtol=1e-4, atol=1e-4), (name, t, float((obs - obs2).abs().max()))
        assert torch.allclose(total, want_total, rtol=1e-4, atol=1e-3), (name, t)
        # a termination threshold crossed within rounding may flip one flag; never more than a handful
        assert int((done != want_done).sum()) <= 2, (name, t)
        done = want_done.clone()
        total = want_total.clone()
        st = {k: v.clone() for k, v in st2.items()}
        ended = int(done.sum())
    if name in ("hopper", "walker2d", "inverted_pendulum", "inverted_double_pendulum"):
        assert ended > 0                                     # random controls make some of them fall: the rule was exercised
