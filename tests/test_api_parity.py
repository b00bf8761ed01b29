"""Every public name of the reference's API surface resolves in evox_amd."""
import pytest

import evox_amd as ex

TOP = ["State", "Stateful", "use_state", "jit_class", "jit_method", "Algorithm", "Problem", "Workflow", "Monitor",
       "has_init_ask", "has_init_tell", "dataclass", "pytree_field", "PyTreeNode", "get_state_sharding"]
ALGOS = ["DE", "ODE", "JaDE", "SaDE", "SHADE", "CoDE", "CMAES", "SepCMAES", "IPOPCMAES", "BIPOPCMAES", "AMaLGaM",
         "IndependentAMaLGaM", "ARS", "ASEBO", "CR_FM_NES", "DES", "ESMC", "GuidedES", "MAES", "LMMAES", "XNES",
         "SeparableNES", "Noise_reuse_es", "OpenES", "PersistentES", "PGPE", "RMES", "SNES", "LES", "PSO", "CSO", "CLPSO",
         "DMSPSOEL", "FIPS", "SLPSOGS", "SLPSOUS", "SwmmPSO", "FSPSO", "NSGA2", "NSGA3", "RVEA", "RVEAa", "MOEAD", "MOEADDRA",
         "MOEADM2M", "EAGMOEAD", "IBEA", "BCEIBEA", "SPEA2", "HypE", "KnEA", "BiGE", "GDE3", "SRA", "TDEA", "LMOCSO", "IMMOEA",
         "ClusterdAlgorithm", "RandomMaskAlgorithm", "TreeAlgorithm", "VectorizedCoevolution", "Coevolution"]
NUMERICAL = (["Ackley", "Rastrigin", "Sphere", "Griewank", "Rosenbrock", "Schwefel", "ackley_func", "rastrigin_func",
              "sphere_func", "griewank_func", "rosenbrock_func", "schwefel_func", "ZDT1", "ZDT2", "ZDT3", "ZDT4", "ZDT6",
              "CEC2022TestSuit"] + [f"DTLZ{i}" for i in range(1, 8)] + [f"LSMOP{i}" for i in range(1, 10)]
             + [f"MaF{i}" for i in range(1, 16)] + [f"F{i}_CEC2022" for i in range(1, 13)])
UTILS = ["min_by", "pairwise_func", "euclidean_dist", "manhattan_dist", "chebyshev_dist", "pairwise_euclidean_dist",
         "pairwise_manhattan_dist", "pairwise_chebyshev_dist", "pair_max", "cos_dist", "cal_max", "dominate_relation",
         "new_dist_mat", "compose", "rank", "rank_based_fitness", "OptaxWrapper", "TreeAndVector", "parse_opt_direction",
         "frames2gif", "AggregationFunction"]
OPERATORS = {
    "": ["non_dominated_sort", "crowding_distance", "crowding_distance_sort"],
    "selection": ["NonDominate", "non_dominate", "ReferenceVectorGuided", "ref_vec_guided", "Tournament", "TopkFit", "topk_fit",
                  "RouletteWheelSelection", "UniformRand", "uniform_rand", "select_rand_pbest"],
    "crossover": ["DifferentialEvolve", "differential_evolve", "de_diff_sum", "de_bin_cross", "de_exp_cross", "de_arith_recom",
                  "SimulatedBinary", "simulated_binary", "SBXCrossover", "sbx", "UniformRand", "OnePoint"],
    "mutation": ["Polynomial", "polynomial", "Bitflip", "bitflip", "Gaussian", "gaussian"],
    "sampling": ["UniformSampling", "LatinHypercubeSampling", "GridSampling"],
    "gaussian_process": ["GPRegression", "GPClassification"],
}


@pytest.mark.parametrize("name", TOP)
def test_top_level(name):
    assert hasattr(ex, name)


@pytest.mark.parametrize("name", ALGOS)
def test_algorithms(name):
    assert hasattr(ex.algorithms, name), name


@pytest.mark.parametrize("name", NUMERICAL)
def test_numerical_problems(name):
    assert hasattr(ex.problems.numerical, name), name


def test_other_packages():
    for name in ("Brax", "Gym", "EnvPool", "TensorflowDataset"):
        assert hasattr(ex.problems.neuroevolution, name)
    for name in ("C10MOP", "CitySegMOP", "IN1kMOP"):
        assert hasattr(ex.problems.evoxbench, name)
    for name in ("StdWorkflow", "RayDistributedWorkflow", "ShardedWorkflow", "IslandWorkflow"):
        assert getattr(ex.workflows, name) is not None
    for name in ("EvalMonitor", "PopMonitor", "StdSOMonitor", "StdMOMonitor", "EvoXVisMonitor"):
        assert getattr(ex.monitors, name) is not None
    for name in ("HV", "GD", "GDPlus", "IGD", "IGDPlus", "gd", "gd_plus", "igd", "igd_plus"):
        assert hasattr(ex.metrics, name)
    for name in UTILS:
        assert hasattr(ex.utils, name), name
    from evox_amd.vis_tools import plot
    for name in ("plot_dec_space", "plot_obj_space_1d", "plot_obj_space_2d", "plot_obj_space_3d"):
        assert hasattr(plot, name)


@pytest.mark.parametrize("pkg", sorted(OPERATORS))
def test_operators(pkg):
    mod = ex.operators if pkg == "" else getattr(ex.operators, pkg)
    missing = [n for n in OPERATORS[pkg] if not hasattr(mod, n)]
    assert not missing, (pkg, missing)
