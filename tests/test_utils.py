"""``TreeAndVector`` round trips (reference ``tests/test_utils.py:9-36``), optimisers, rank
helpers, aggregation functions and the Gaussian-process operators."""
import pytest
import torch

import evox_amd as ex
from evox_amd import utils
from evox_amd.core.tree import tree_leaves, tree_map
from evox_amd.utils import optimizers as ox


def assert_tree_close(a, b):
    for x, y in zip(tree_leaves(a), tree_leaves(b)):
        torch.testing.assert_close(x, y)


def test_tree_and_vector():
    tree = {
        "layer1": torch.arange(10).reshape(2, 5),
        "layer2": {
            "layer2_1": torch.arange(10).reshape(2, 5),
            "layer2_2": torch.arange(8).reshape(2, 2, 2),
            "layer2_3": torch.arange(9).reshape(1, 1, 1, 9),
        },
        "layer3": [torch.ones((7, 7), dtype=torch.int64), torch.zeros((8, 8), dtype=torch.int64)],
    }
    tv = utils.TreeAndVector(tree)
    assert_tree_close(tree, tv.to_tree(tv.to_vector(tree)))
    batched = tree_map(lambda x, y: torch.stack([x, y], dim=0), tree, tree)
    assert_tree_close(batched, tv.batched_to_tree(tv.batched_to_vector(batched)))


def test_rank_helpers():
    x = torch.tensor([3.0, 1.0, 2.0, 10.0])
    assert utils.rank(x).tolist() == [2, 0, 1, 3]
    torch.testing.assert_close(utils.rank_based_fitness(x), torch.tensor([2 / 3 - 0.5, -0.5, 1 / 3 - 0.5, 0.5]))
    v, k = utils.min_by([torch.tensor([1.0, 2.0]), torch.tensor([3.0])], [torch.tensor([5.0, 0.5]), torch.tensor([0.7])])
    assert float(v) == 2.0 and float(k) == 0.5


def test_pairwise_and_domination():
    a = torch.tensor([[0.0, 0.0], [1.0, 1.0]])
    b = torch.tensor([[1.0, 0.0], [0.0, 2.0], [1.0, 1.0]])
    torch.testing.assert_close(utils.pairwise_euclidean_dist(a, b), torch.cdist(a, b))
    assert utils.pairwise_manhattan_dist(a, b)[0].tolist() == [1.0, 2.0, 2.0]
    assert utils.pairwise_chebyshev_dist(a, b)[0].tolist() == [1.0, 2.0, 1.0]
    dom = utils.dominate_relation(a, b)
    assert dom.tolist() == [[True, True, True], [False, False, False]]
    assert utils.parse_opt_direction("max") == -1
    assert utils.parse_opt_direction(["min", "max"]).tolist() == [1, -1]
    f = utils.compose(lambda x: x + 1, lambda x: x * 2)
    assert f(3) == 8


def test_aggregation_functions():
    f = torch.tensor([[1.0, 2.0], [3.0, 1.0]])
    w = torch.tensor([[0.5, 0.5], [0.2, 0.8]])
    z = torch.zeros(2)
    torch.testing.assert_close(utils.AggregationFunction("weighted_sum")(f, w), torch.tensor([1.5, 1.4]))
    torch.testing.assert_close(utils.AggregationFunction("tchebycheff")(f, w, z), torch.tensor([1.0, 0.8]))
    assert utils.AggregationFunction("pbi")(f, w, z).shape == (2,)
    assert utils.AggregationFunction("modified_tchebycheff")(f, w, z).shape == (2,)
    assert utils.AggregationFunction("tchebycheff_norm")(f, w, z, torch.ones(2) * 4).shape == (2,)


def test_optimizers_minimise_quadratic():
    for make in (lambda: ox.sgd(0.1), lambda: ox.adam(0.1), lambda: ox.rmsprop(0.05), lambda: ox.sgd(0.05, momentum=0.9)):
        opt = make()
        p = torch.tensor([3.0, -2.0])
        st = opt.init(p)
        for _ in range(300):
            upd, st = opt.update(2 * p, st, p)
            p = ox.apply_updates(p, upd)
        assert p.abs().max() < 0.05


def test_gp_regression_and_classification():
    from evox_amd.operators.gaussian_process import Gaussian, GPClassification, GPRegression, Linear
    torch.manual_seed(0)
    x = torch.linspace(-3, 3, 40)[:, None]
    y = torch.sin(x) + 0.05 * torch.randn_like(x)
    model = GPRegression(likelihood=Gaussian(num_datapoints=40), num_iters=150)
    model.fit(x, y, ox.adam(0.02))
    _, mu, std = model.predict(x)
    assert (mu - torch.sin(x)[:, 0]).abs().max() < 0.2
    assert model.history[-1] < model.history[0]
    lin = GPRegression(kernel=Linear(), likelihood=Gaussian(num_datapoints=40), num_iters=50)
    lin.fit(x, 2 * x, ox.adam(0.01))
    _, mu, _ = lin.predict(torch.tensor([[1.0], [2.0]]))
    torch.testing.assert_close(mu, torch.tensor([2.0, 4.0]), atol=0.2, rtol=0.1)
    xc = torch.randn(60, 2)
    yc = (xc[:, 0] > 0).float()
    clf = GPClassification(num_iters=150)
    clf.fit(xc, yc)
    _, p, _ = clf.predict(xc)
    assert ((p > 0.5).float() == yc).float().mean() > 0.9


def test_graphed_step_refuses_cpu_state():
    from evox_amd.utils.graph import GraphCaptureError, GraphedStep
    import pytest
    step = GraphedStep(lambda s: s)
    with pytest.raises(GraphCaptureError):
        step(ex.State(x=torch.zeros(3)))


def test_every_registered_hip_entry_point_is_exported():
    """The op wrappers declare their kernels' C entry points; each must exist in the built library
    (catches a kernel that was declared in Python but never made it into a .hip file)."""
    import ctypes
    import importlib
    import os

    import pytest
    from evox_amd import _native
    for mod in ("gemm", "es", "_nds_hip", "moselect", "fitness", "swarm"):
        importlib.import_module(f"evox_amd.ops.{mod}")
    importlib.import_module("evox_amd.core.random")
    path = os.path.join(os.path.dirname(_native.__file__), "libevox_hip.so")
    if not os.path.exists(path):
        pytest.skip("libevox_hip.so has not been built")
    lib = ctypes.CDLL(path)
    missing = [name for name in _native._HIP_API if not hasattr(lib, name)]
    assert not missing, missing


def test_rank_launcher_relays_rank0_and_fails_fast(tmp_path):
    """``utils/launch.py`` (what ``bench.py --gpus N`` uses): N child processes with the rendezvous environment, rank 0's
    output relayed, and a failing rank takes its peers down instead of leaving them waiting."""
    import importlib.util
    import os
    import time

    import evox_amd

    spec = importlib.util.spec_from_file_location(
        "_launch", os.path.join(os.path.dirname(evox_amd.__file__), "utils", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    ok = tmp_path / "ok.py"
    ok.write_text("import os\nif os.environ['RANK'] == '0':\n    print('world', os.environ['WORLD_SIZE'], os.environ['MASTER_ADDR'])\n")
    assert launch.launch_ranks(str(ok), [], 3) == 0
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(60)\n")
    t0 = time.time()
    assert launch.launch_ranks(str(bad), [], 3) == 7
    assert time.time() - t0 < 30


@pytest.mark.gpu
def test_graph_ring_length_follows_state_size_and_memory():
    """Pointer-stable replay chains eight recordings for a small state; a step whose outputs are large (every recording
    keeps its own) gets a shorter ring, and replays stay correct across several turns of the ring."""
    from evox_amd.core.state import State
    from evox_amd.utils.graph import GraphedStep

    dev = torch.device("cuda")

    def small(state):
        return state.replace(x=state.x * 1.5 + 1.0, n=state.n + 1)

    g = GraphedStep(small, host_counters=("fields/n",))
    s = State(x=torch.ones(1000, device=dev), n=0)
    for _ in range(21):
        s = g(s)
    torch.cuda.synchronize()
    ref = torch.ones((), dtype=torch.float64)
    for _ in range(21):
        ref = ref * 1.5 + 1.0
    assert g.ring == 8 and s.n == 21
    torch.testing.assert_close(s.x[0].double().cpu(), ref, rtol=1e-5, atol=0)

    def big(state):                                          # 1.5 GiB of auxiliary output per recording
        return state.replace(x=state.x + 1.0), torch.zeros(3 * 2 ** 27, device=dev) + state.x[0]

    g2 = GraphedStep(big)
    s2 = State(x=torch.zeros(8, device=dev))
    for _ in range(7):
        s2, aux = g2(s2)
    torch.cuda.synchronize()
    assert g2.ring == 2 and float(s2.x[0]) == 7.0 and float(aux[5]) == 6.0
