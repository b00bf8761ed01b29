"""Micro-benchmark of the CMA-ES sampling / covariance building blocks at the headline shape."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evox_amd.core import random as rnd
from evox_amd.ops import gemm, cma

dev = torch.device("cuda")
lam, n, mu = 10000, 1000, 5000
key = rnd.PRNGKey(0).to(dev)
mean = torch.zeros(n, device=dev); sigma = torch.tensor(1.0, device=dev)
B = torch.linalg.qr(torch.randn(n, n, device=dev))[0].contiguous(); D = torch.rand(n, device=dev) + 0.5


def timeit(name, fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps * 1e6
    print(json.dumps({"op": name, "us": round(dt, 1)}), flush=True)
    return out

BD = (B * D).contiguous()            # x = z @ BD^T
ref = timeit("fused philox+mfma sample (fp32)", lambda: gemm.cma_sample(key, mean, sigma, B, D, lam))
z = timeit("philox normal 1e7", lambda: rnd.normal(key, (lam, n)))
timeit("torch fp32 addmm z@BD^T", lambda: torch.addmm(mean, z, BD.T))
timeit("own fp32 mfma gemm z@BD^T", lambda: gemm.matmul_nt(z, BD))
x32 = mean + sigma * (z @ BD.T)
print("fused vs materialised max err", float((ref - x32).abs().max()))
zb, BDb = z.bfloat16(), BD.bfloat16()
timeit("bf16 cast z", lambda: z.bfloat16())
timeit("torch bf16 gemm", lambda: zb @ BDb.T)
def bf16x3():
    zh = z.bfloat16(); zl = (z - zh.float()).bfloat16()
    ah = BD.bfloat16(); al = (BD - ah.float()).bfloat16()
    acc = (zh @ ah.T).float()
    acc += (zh @ al.T).float(); acc += (zl @ ah.T).float()
    return acc
x3 = timeit("bf16x3 split gemm (3 products)", bf16x3)
xd = (z.double() @ BD.double().T)
print("rel err fp32 addmm", float(((z @ BD.T).double() - xd).abs().max() / xd.abs().max()))
print("rel err bf16x3   ", float((x3.double() - xd).abs().max() / xd.abs().max()))
print("rel err bf16     ", float(((zb @ BDb.T).double() - xd).abs().max() / xd.abs().max()))
# covariance
pop = ref; fit = torch.rand(lam, device=dev); order = torch.argsort(fit)
w = torch.softmax(torch.randn(mu, device=dev), 0); pc = torch.randn(n, device=dev); C = torch.eye(n, device=dev)
timeit("fused gathered cov update (fp32)", lambda: gemm.cma_covariance_update(C, pop, order, mean, sigma, w, pc, 0.9, 1e-6, 5e-3))
y = ((pop[order[:mu]] - mean) / sigma)
timeit("gather+center y", lambda: (pop[order[:mu]] - mean) / sigma)
timeit("torch fp32 (y^T*w)@y", lambda: (y.T * w) @ y)
yb = y.bfloat16(); ywb = (y.T * w).bfloat16()
timeit("torch bf16 (y^T*w)@y", lambda: ywb @ yb)
timeit("argsort 1e4", lambda: torch.argsort(fit))
timeit("torch fp32 1000^3 matmul", lambda: C @ BD)
Cb = BD.bfloat16()
timeit("torch bf16 1000^3 matmul", lambda: Cb @ Cb)
