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
torch.cuda.synchronize()
print("done")
