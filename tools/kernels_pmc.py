"""Runs every hand-written kernel a few times at its benchmark shape (target program of the
hardware-counter passes; see tools/gpu_pmc.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EVOX_AMD_CMA_SAMPLE"] = "fused"
from evox_amd.core import random as rnd
from evox_amd.ops import gemm, fitness, nds, es
from evox_amd.models import MLP

dev = torch.device("cuda")
key = rnd.PRNGKey(0).to(dev)
lam, n, mu = 10000, 1000, 5000
B = torch.linalg.qr(torch.randn(n, n, device=dev))[0].contiguous(); D = torch.rand(n, device=dev) + 0.5
mean = torch.zeros(n, device=dev); sigma = torch.tensor(1.0, device=dev)
for _ in range(3):
    x = gemm.cma_sample(key, mean, sigma, B, D, lam)                        # cma_sample (Philox prologue, fp32 MFMA)
    z = rnd.normal(key, (lam, n))                                            # draw_kernel
    g = gemm.matmul_nt(z, (B * D).contiguous())                              # gemm_f32 (fp32 MFMA)
    f = fitness.ackley(x)                                                    # so_fitness_kernel
    order = torch.argsort(f)
    w = torch.softmax(torch.randn(mu, device=dev), 0); pc = torch.randn(n, device=dev); C = torch.eye(n, device=dev)
    c2 = gemm.cma_covariance_update(C, x, order, mean, sigma, w, pc, 0.9, 1e-6, 5e-3)   # cma_cov (gathered MFMA SYRK)
    S, step = gemm.cma_elite_factor(x, order, mean, sigma, w.sqrt())         # gather_scale_kernel
    r = nds.non_dominated_rank(torch.rand(100000, 3, device=dev))            # nds_count / nds_peel
    net = MLP([244, 1024, 720, 17])
    pop = es.perturb(key, net.init(rnd.PRNGKey(1)).to(dev), 0.05, 1024, mirrored=True)   # es_perturb_kernel
    a = net.batched_apply(pop, torch.randn(1024, 1, 244, device=dev))        # mlp_forward_kernel
    gsum = es.weighted_noise_sum(key, torch.randn(1024, device=dev), net.n_params)      # noise_wsum_kernel
# kernels added with the native factorisation path and the multi-objective / DE fusions
from evox_amd import problems
from evox_amd.ops import blas, de_adapt, sort, variation, _nds_hip
ws = blas.NSWorkspace(n, dev)
Cm = torch.eye(n, device=dev) + 0.01 * torch.randn(n, n, device=dev); Cm = (Cm + Cm.T) / 2 + 0.3 * torch.eye(n, device=dev)
lb20, ub20 = torch.full((20,), -100.0, device=dev), torch.full((20,), 100.0, device=dev)
cec = problems.numerical.CEC2022TestSuit.create(1)
dtlz = problems.numerical.DTLZ2(m=3)
for _ in range(3):
    blas.ns_sqrt(Cm, ws, 4)                                                  # ns_matvec / ns_scale / ns_residual / ns_finish
    o, rk = sort.argsort_with_rank(torch.randn(10000, device=dev))           # rank_sort_kernel
    base = gemm.cma_covariance_base(Cm, pc, torch.tensor(0.99, device=dev), 1e-6)   # cov_base_kernel
    px = torch.rand(100000, 12, device=dev)
    off = variation.sbx_pm(key, key, px, torch.zeros(12, device=dev), torch.ones(12, device=dev))   # sbx_pm_kernel
    fz, _ = dtlz.evaluate(None, px)                                          # dtlz_kernel
    fr = torch.rand(8000, 3, device=dev)
    cd = _nds_hip.crowding(fr)                                               # crowding_kernel (+3 rank sorts)
    pop20 = torch.rand(50000, 20, device=dev) * 200 - 100
    fc, _ = cec.evaluate(None, pop20)                                        # shift_rotate_kernel
    Fv, CRv = de_adapt.draw_params(key, key, key, torch.full((100,), 0.5, device=dev), torch.full((100,), 0.5, device=dev), 50000)
    de_adapt.select_and_adapt(pop20, pop20 + 1, fc + 1, fc, Fv, CRv, torch.full((100,), 0.5, device=dev),
                              torch.full((100,), 0.5, device=dev), strict=True, weight_by_gain=True, mode="shade")
# selection-loop, indicator, distance and reference-vector kernels
from evox_amd.algorithms.mo import indicator, nsga3
from evox_amd.ops import distance, moselect, linalg_small, swarm
from evox_amd.operators.selection import rvea_selection
from evox_amd.core import random as _rnd
obj4k = torch.rand(4000, 3, device=dev)
for _ in range(2):
    d = distance.cdist(torch.rand(20000, 3, device=dev), torch.rand(10000, 3, device=dev))     # cdist_small_kernel
    d = distance.cdist(torch.rand(2048, 100, device=dev), torch.rand(2048, 100, device=dev))    # cdist_kernel
    keep = indicator.ibea_survivors(obj4k, 0.05, 2000)                       # ibea_fitness_kernel, ibea_remove_reg_kernel
    sf = indicator.spea2_fitness(obj4k)                                      # spea2_density_kernel, spea2_raw_kernel
    keep = moselect.spea2_truncate(obj4k, sf < 2, 1000)                      # spea2_truncate_kernel
    I1, I2, u = torch.rand(4000, device=dev), torch.rand(4000, device=dev), torch.rand(4000, device=dev)
    rk = moselect.stochastic_ranking(I1, I2, u, 0.45)                        # stochastic_ranking_kernel
    dd = distance.cdist(obj4k, obj4k); dd.fill_diagonal_(float("inf"))
    br = (dd / torch.sort(dd, dim=1).values[:, 2].mean()).clamp_max(1.0)
    keep = moselect.crowd_remove(br, 1500)                                   # crowd_remove_kernel
    v = torch.rand(2000, 3, device=dev) + 0.01
    xs, fs = rvea_selection.ref_vec_guided(torch.rand(4000, 12, device=dev), obj4k, v, 0.3)   # rvea_assoc_kernel
    slots = nsga3._niche_fill(torch.randint(0, 5, (2000,), device=dev), torch.randint(0, 5, (2000,), device=dev),
                              torch.tensor(900, device=dev))                 # niche_fill_kernel
    x3, ok3 = linalg_small.solve_small(torch.rand(3, 3, device=dev) + torch.eye(3, device=dev), torch.ones(3, device=dev))
    zb = _rnd.normal(key, (lam, n), dtype=torch.bfloat16)                    # normal_bf16_kernel
    keys50k = torch.rand(50000, device=dev) * 600 + 400
    top = sort.select_indices(keys50k, 2500)                                 # select_grid_kernel (co-operating workgroups, keys in registers)
    top = sort.select_indices(torch.rand(1000000, device=dev), 50000)        # ms_hist / ms_count / ms_emit kernels
    fz1, _ = problems.numerical.ZDT1(n=30).evaluate(None, torch.rand(100000, 30, device=dev))   # zdt_kernel
    fc1, _ = cec.evaluate(None, pop20)                                       # cec_simple_kernel<20>
    r1 = nds.non_dominated_rank(torch.rand(800, 3, device=dev))             # nds_small_kernel (one launch)
    r2 = nds.non_dominated_rank(torch.rand(4000, 2, device=dev))            # four rounds, then nds_peel_all_kernel
# kernels of the sweep non-dominated sort, the fused CEC 2022 evaluation, SRA, and the bf16 CMA-ES generation
from evox_amd import algorithms as _A, workflows as _W
for _ in range(2):
    r3 = nds.non_dominated_rank(torch.rand(200000, 3, device=dev), until=100000)   # sweep_merge / scan / chunks / query
    for fnum in (3, 7, 9, 12):
        fcx, _ = problems.numerical.CEC2022TestSuit.create(fnum).evaluate(None, pop20)   # cec_full_kernel<20>
    i1, i2 = moselect.sra_indicators(torch.rand(20000, 3, device=dev), lower_only=True)   # sra_indicators_kernel
    rk2 = moselect.stochastic_ranking(i1, i2, torch.rand(20000, device=dev), 0.45)        # stochastic_ranking_blocked_kernel<24>
algo = _A.CMAES(center_init=torch.full((n,), 2.0, device=dev), init_stdev=1.0, pop_size=lam, decomposition="newton_schulz",
                precision="bf16")
wf = _W.StdWorkflow(algo, problems.numerical.Ackley())
st = wf.init(rnd.PRNGKey(1))
for _ in range(12):                                                          # gemm_nt_kernel family, cma_bf16 / cma_step kernels
    st = wf.step(st)
# kernels of the row-sharded generation, the one-pass MaF / LSMOP evaluators and the streaming indicator / association /
# replacement kernels
from evox_amd.workflows import ShardedWorkflow as _SW
algo2 = _A.CMAES(center_init=torch.full((n,), 2.0, device=dev), init_stdev=1.0, pop_size=lam, decomposition="newton_schulz",
                 precision="bf16")
wf2 = _SW(algo2, problems.numerical.Ackley())
st2 = wf2.init(rnd.PRNGKey(1))
for _ in range(6):                      # presorted_rank / select_local / gram_pack / cov_from_packed kernels
    st2 = wf2.step(st2)
for _ in range(2):
    for name in ("LSMOP1", "LSMOP7", "MaF3", "MaF10", "MaF12", "MaF13", "MaF9"):
        pr = getattr(problems.numerical, name)(m=3)
        fo, _ = pr.evaluate(None, torch.rand(10000, pr.d, device=dev))       # grouped / maf / maf_wfg / maf9 kernels
    f3 = torch.rand(20000, 3, device=dev)
    cd = moselect.bige_crowding(f3, f3.sum(1), torch.tensor(0.05, device=dev))            # bige_crowding_kernel
    kn = moselect.knn_distances(f3, 3)                                                    # knn_dist_kernel
    kp = moselect.knee_suppress(f3[:4000], torch.randperm(4000, device=dev), torch.full((3,), 0.05, device=dev))   # knee_round_kernel
    gp, d1, d2 = moselect.ref_associate(f3, torch.rand(10000, 3, device=dev) + 0.01)      # ref_assoc_kernel
    bw = moselect.scatter_best(torch.randn(10000, 1000, device=dev), torch.randint(0, 10000, (10000, 1000), device=dev),
                               10000, torch.zeros(10000, device=dev) - 1.0)               # scatter_best_kernel
    rk3 = moselect.stochastic_ranking(i1, i2, None, 0.45, seed=torch.tensor([7], dtype=torch.int32, device=dev))   # fresh coins
# removal loops in chip-wide rounds (select / simulate / apply; select / lists / play / rescan)
front = torch.randn(20000, 3, device=dev).abs()
front = front / front.norm(dim=1, keepdim=True) * (1 + 0.05 * torch.rand(20000, 1, device=dev))
front = (front - front.min()) / (front.max() - front.min())
for _ in range(2):
    fit20k, c20k = moselect.ibea_fitness(front, 0.05)
    keep = moselect.ibea_remove_rounds(front, fit20k, c20k, 0.05, 10000)                  # ibea_lazy_select / apply kernels
    keep = moselect.spea2_truncate_rounds(front, 10000)                                   # spea2_lazy_* / flag / scan / finish kernels
# the one-launch two-objective sort and the three-state knee suppression
from evox_amd.ops import _nds_hip as _nh
for _ in range(2):
    r2d = _nh.rank(torch.rand(20000, 2, device=dev), 10000)                               # nds2_scan_kernel
    kp2 = moselect.knee_suppress(front, torch.randperm(20000, device=dev), torch.full((3,), 0.01, device=dev))   # knee_state / knee_tail
# selection with the keys in registers, the blocked DE launch, the swarm moves, the windowed stochastic ranking, the
# one-workgroup factorisations and the physics steps
from evox_amd import algorithms as _AL
from evox_amd.ops import linalg_small as _ls, swarm as _sw
from evox_amd.problems.neuroevolution.reinforcement_learning.tensor_env import make_tensor_env
for _ in range(2):
    top = sort.select_indices(torch.rand(12000, device=dev), 600)                         # select_reg_kernel
    top = sort.select_indices(torch.rand(50000, device=dev), 2500)                        # select_grid_kernel
    shade = _AL.SHADE(torch.full((10,), -5.0, device=dev), torch.full((10,), 5.0, device=dev), 256).blocked(100)
    sst = shade.init(rnd.PRNGKey(3).to(dev))
    tr, sst = shade.ask(sst)                                                              # de_trial_kernel, 100 blocks
    xs, vs = torch.rand(1024, 100, device=dev), torch.rand(1024, 100, device=dev)
    lo, hi = torch.zeros(100, device=dev), torch.ones(100, device=dev)
    nx, nv = _sw.attract(xs, vs, lo, hi, [(xs, torch.randint(0, 1024, (1024,), device=dev), 1.5, key, 0),
                                          (xs.mean(0), "all", 0.2, key, 102400)], inertia=0.6)      # swarm_attract_kernel
    nbr = torch.randint(0, 1024, (1024, 5), device=dev)
    nx, nv = _sw.fips_move(key, xs, vs, xs, nbr, torch.ones(1024, 5, device=dev), torch.rand(1024, 5, device=dev) + 0.1,
                           torch.full((1024,), 0.73, device=dev), torch.full((1024,), 4.1, device=dev), lo, hi, 4.1)   # fips_move_kernel
    rk4 = moselect.stochastic_ranking(torch.rand(20000, device=dev), torch.rand(20000, device=dev), None, 0.45,
                                      seed=torch.tensor([9], dtype=torch.int32, device=dev))       # stochastic_ranking_window_kernel
    A100 = torch.randn(100, 100, device=dev); C100 = A100 @ A100.T / 100 + 0.1 * torch.eye(100, device=dev)
    ev, V = _ls.eigh_small(C100, None, spd=True)                                          # cholesky_lds + eigh_spd_kernel
    ev, V = _ls.eigh_small(C100 * 1.01, V, spd=True)                                      # warm start
    net_s = MLP([11, 32, 32, 3])
    a_s = net_s.batched_apply(torch.randn(16384, net_s.n_params, device=dev) * 0.1, torch.randn(16384, 1, 11, device=dev))   # mlp_small_kernel
    for task in ("hopper", "ant"):
        env = make_tensor_env(task)
        est, ob = env.reset(rnd.PRNGKey(5).to(dev), 4096, dev)
        tot, dn = torch.zeros(4096, device=dev), torch.zeros(4096, dtype=torch.bool, device=dev)
        for _ in range(3):
            est, ob, tot, dn = env.step_accumulate(est, torch.rand(4096, env.act_dim, device=dev) * 2 - 1, tot, dn)   # planar_step / spatial_step kernels
torch.cuda.synchronize()
print("done")
