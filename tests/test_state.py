"""State tree / module system semantics (modelled on the reference's tests/test_state.py)."""
import pytest
import torch

import evox_amd as ex
from evox_amd import State, Stateful, use_state
from evox_amd.core.tree import tree_map


class Leaf(Stateful):
    def setup(self, key):
        return State(c=42)

    def run(self, state):
        return [2, 7, 1], state.replace(c=state.c * 2)

    def check(self, state):
        assert state.c == 84
        return state


class Middle(Stateful):
    def __init__(self):
        super().__init__()
        self.leaf = Leaf()

    def setup(self, key):
        return State(d=[3, 1, 4, 1, 5, 9, 2, 6])

    def run(self, state):
        e, state = use_state(self.leaf.run)(state)
        return e + [8, 2, 8], state.replace(d=3.1415926)

    def check(self, state):
        assert state.d == 3.1415926
        return state


class Root(Stateful):
    def __init__(self):
        super().__init__()
        self.a, self.b = 123, 456
        self.leaf = Leaf()
        self.middle = Middle()

    def setup(self, key):
        return State(attr_a=self.a, attr_b=self.b)

    def run(self, state):
        e1, state = use_state(self.leaf.run)(state)
        e2, state = use_state(self.middle.run)(state)
        return e1 + e2, state.replace(attr_a=state.attr_a + 2, attr_b=state.attr_b - 3)

    def check(self, state):
        assert state.attr_a == 125 and state.attr_b == 453
        state = use_state(self.middle.check)(state)
        return use_state(self.leaf.check)(state)


def test_init_matches_standalone_substates():
    root, middle, leaf = Root(), Middle(), Leaf()
    rs = root.init(ex.PRNGKey(123))
    ms = middle.init(ex.PRNGKey(456))
    ls = leaf.init(ex.PRNGKey(789))
    assert rs.get_child_state("leaf") == ls
    assert rs.get_child_state("middle") == ms
    assert ms.get_child_state("leaf") == ls
    magic, rs = root.run(rs)
    assert magic == [2, 7, 1, 2, 7, 1, 8, 2, 8]
    root.check(rs)


def test_repr_and_str():
    state = Root().init(ex.PRNGKey(456))
    assert repr(state) == (
        "State({'attr_a': 123, 'attr_b': 456}, {'leaf': State({'c': 42}, {}),"
        "'middle': State({'d': [3, 1, 4, 1, 5, 9, 2, 6]}, {'leaf': State({'c': 42}, {})})})"
    )
    assert str(state) == (
        "State({'attr_a': 123, 'attr_b': 456},\n"
        " {'leaf': ({'c': 42}, {}),\n"
        "  'middle': ({'d': [3, 1, 4, 1, 5, 9, 2, 6]}, {'leaf': ({'c': 42}, {})})})"
    )


def test_query_and_immutability():
    state = Root().init(ex.PRNGKey(0))
    assert state.query_state("leaf").c == 42
    assert state.query_state("middle.leaf").c == 42
    assert state.query_state("middle").d == [3, 1, 4, 1, 5, 9, 2, 6]
    with pytest.raises(TypeError):
        state.attr_a = 1
    with pytest.raises(TypeError):
        state["attr_a"] = 1
    s2 = state.replace(attr_a=7)
    assert s2.attr_a == 7 and state.attr_a == 123


def test_tree_roundtrip():
    state = Root().init(ex.PRNGKey(0))
    assert state == tree_map(lambda x: x, state)


def test_node_ids_are_deterministic():
    a, b = Root(), Root()
    sa, sb = a.init(ex.PRNGKey(1)), b.init(ex.PRNGKey(1))
    assert a.leaf._node_id == b.leaf._node_id == 1
    assert a.middle._node_id == 2 and a.middle.leaf._node_id == 3
    assert sa.find_path_to(3)[0] == ("middle", ("leaf", 3))


def test_uninitialised_module_raises():
    with pytest.raises(ValueError):
        use_state(Leaf().run)(State(c=1))


def test_dataclass_state_and_static_fields():
    @ex.dataclass
    class Rec:
        x: torch.Tensor
        n: int = ex.pytree_field(static=True)

    s = State(Rec(x=torch.ones(3), n=5))
    assert s.n == 5
    s2 = s.replace(x=torch.zeros(3))
    assert s2.x.sum() == 0 and s.x.sum() == 3
    leaves = ex.tree.tree_leaves(s)
    assert len(leaves) == 1


def test_stacked_modules_and_indexed_use_state():
    @ex.dataclass
    class Counter(Stateful):
        step: torch.Tensor

        def setup(self, key):
            return State(total=torch.zeros(()), key=key)

        def bump(self, state):
            return state.total + self.step, state.replace(total=state.total + self.step)

    class Holder(Stateful):
        def __init__(self):
            super().__init__()
            self.counters = Stateful.stack([Counter(step=torch.tensor(float(i + 1))) for i in range(3)])

    @ex.dataclass
    class DHolder(Stateful):
        counters: Counter = ex.pytree_field(stack=True)

        def setup(self, key):
            return State()

    h = DHolder(counters=Stateful.stack([Counter(step=torch.tensor(float(i + 1))) for i in range(3)]))
    state = h.init(ex.PRNGKey(3))
    sub = state.get_child_state("counters")
    assert sub.total.shape == (3,) and sub.key.shape == (3, 2)
    assert len(h.counters) == 3
    val, state = use_state(Counter.bump, index=1)(h.counters, state)
    assert float(val) == 2.0
    assert state.get_child_state("counters").total.tolist() == [0.0, 2.0, 0.0]


def test_checkpoint_roundtrip(tmp_path):
    import evox_amd.algorithms as alg

    pso = alg.PSO(lb=torch.full((3,), -1.0), ub=torch.full((3,), 1.0), pop_size=8)
    state = pso.init(ex.PRNGKey(5))
    state.save(tmp_path / "ckpt")
    restored = state.load(tmp_path / "ckpt")
    assert restored == state
    free = State.load_from(tmp_path / "ckpt")
    assert free == state
    state.save(tmp_path / "ckpt.pt", orbax=False)
    assert state.load(tmp_path / "ckpt.pt", orbax=False) == state


def test_checkpoint_resume_continues_stream(tmp_path):
    import evox_amd.algorithms as alg
    from evox_amd import problems, workflows

    def make():
        return workflows.StdWorkflow(alg.PSO(lb=torch.full((4,), -5.0), ub=torch.full((4,), 5.0), pop_size=16),
                                     problems.numerical.Sphere())

    wf = make()
    s = wf.init(ex.PRNGKey(9))
    for _ in range(3):
        s = wf.step(s)
    s.save(tmp_path / "c")
    for _ in range(3):
        s = wf.step(s)
    wf2 = make()
    t = wf2.init(ex.PRNGKey(1234)).load(tmp_path / "c")
    for _ in range(3):
        t = wf2.step(t)
    assert t == s


def _cma_resume_case(tmp_path, device, precision):
    import evox_amd.algorithms as alg
    from evox_amd import problems, workflows

    def make():
        algo = alg.CMAES(center_init=torch.full((16,), 2.0, device=device), init_stdev=1.0, pop_size=32,
                         decomposition="newton_schulz", precision=precision)
        return workflows.StdWorkflow(algo, problems.numerical.Rosenbrock())

    wf = make()
    s = wf.init(ex.PRNGKey(11))
    for _ in range(7):
        s = wf.step(s)
    s.save(tmp_path / "c")
    for _ in range(9):
        s = wf.step(s)
    # a fresh process would start like this: new objects, nothing carried over but the checkpoint
    wf2 = make()
    t = wf2.init(ex.PRNGKey(99)).load(tmp_path / "c")
    for _ in range(9):
        t = wf2.step(t)
    a, b = s.get_child_state("algorithm"), t.get_child_state("algorithm")
    assert int(a.ns_iters) == int(b.ns_iters) and int(a.ns_calls) == int(b.ns_calls)
    assert torch.equal(a.mean, b.mean) and torch.equal(a.C, b.C) and torch.equal(a.sigma, b.sigma)
    assert t == s


def test_cmaes_newton_schulz_resume_equals_uninterrupted(tmp_path):
    """The iteration control of the factorisation (count, safe-scaling countdown, power vectors, lagged residual
    records) is part of the state: a run resumed from a checkpoint is bit-identical to the uninterrupted one."""
    _cma_resume_case(tmp_path, "cpu", "fp32")


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cmaes_newton_schulz_resume_equals_uninterrupted_gpu(precision, tmp_path):
    _cma_resume_case(tmp_path, "cuda", precision)
