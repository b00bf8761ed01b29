"""Host dispatch cost and device time of the native hipBLASLt path against the tensor library's
matmul for the GEMM shapes of the CMA-ES generation (λ=10⁴, n=10³, μ=5·10³)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evox_amd.ops import blas

dev = torch.device("cuda")
torch.manual_seed(0)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return t_issue / reps * 1e6, t_all / reps * 1e6, e0.elapsed_time(e1) / reps * 1e3


print("available:", blas.available(), flush=True)
shapes = [("sample NT 10000x1000x1000", 10000, 1000, 1000, False, True),
          ("gram   TN 1000x1000x5000", 1000, 1000, 5000, True, False),
          ("ns     NN 1000x1000x1000", 1000, 1000, 1000, False, False)]
for name, M, N, K, ta, tb in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    out = torch.empty((M, N), device=dev)
    ref = (a.T if ta else a).double() @ (b.T if tb else b).double()
    got = blas.gemm(a, b, trans_a=ta, trans_b=tb)
    err = ((got.double() - ref).abs().max() / ref.abs().max()).item()
    terr = (((a.T if ta else a) @ (b.T if tb else b)).double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"{name}: rel err native {err:.2e} torch {terr:.2e}", flush=True)
    info = blas.plan_info(torch.float32, ta, tb, M, N, K, a.stride(0), b.stride(0), N, N)
    print("   plan candidates/best us:", info, flush=True)
    r_n = timeit(lambda: blas.gemm(a, b, trans_a=ta, trans_b=tb, out=out))
    r_t = timeit(lambda: torch.mm(a.T if ta else a, b.T if tb else b, out=out))
    print(f"   native: issue {r_n[0]:.1f} us  wall {r_n[1]:.1f} us  device {r_n[2]:.1f} us", flush=True)
    print(f"   torch : issue {r_t[0]:.1f} us  wall {r_t[1]:.1f} us  device {r_t[2]:.1f} us", flush=True)
    if name.startswith("sample"):
        sig = torch.tensor(1.5, device=dev); mean = torch.randn(N, device=dev)
        r_b = timeit(lambda: blas.gemm(a, b, trans_a=ta, trans_b=tb, out=out, alpha=sig, bias=mean))
        print(f"   native with device alpha + bias: issue {r_b[0]:.1f} wall {r_b[1]:.1f} device {r_b[2]:.1f}", flush=True)
        ab, bb = a.bfloat16(), b.bfloat16()
        r_n = timeit(lambda: blas.gemm(ab, bb, trans_a=ta, trans_b=tb, out=out))
        r_t = timeit(lambda: torch.mm(ab, bb.T, out_dtype=torch.float32))
        print(f"   bf16 native: issue {r_n[0]:.1f} wall {r_n[1]:.1f} device {r_n[2]:.1f};  torch: issue {r_t[0]:.1f} wall {r_t[1]:.1f} device {r_t[2]:.1f}", flush=True)

# Newton–Schulz driver
n = 1000
for cond in (1.2, 1e3):
    q, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64))
    ev = torch.logspace(0, -torch.log10(torch.tensor(cond)).item(), n, dtype=torch.float64)
    C64 = (q * ev) @ q.T
    C = C64.float().to(dev)
    for safe in (False, True):
        ws = blas.NSWorkspace(n, dev)
        for iters in (4, 8, 14):
            s, zi = blas.ns_sqrt(C, ws, iters, safe=safe)
            sc = ws.scal.cpu()
            e1 = ((s.double().cpu() @ s.double().cpu() - C64).norm() / C64.norm()).item()
            e2 = ((zi.double().cpu() @ C64 @ zi.double().cpu() - torch.eye(n, dtype=torch.float64)).norm() / n ** 0.5).item()
            print(f"ns cond={cond:g} safe={safe} iters={iters}: sqrt err {e1:.2e} whiten err {e2:.2e} s={sc[13]:.4f} "
                  f"res={sc[14]:.2e} fallback={sc[15]:.0f} lam_true={ev.max():.3f}", flush=True)
        r = timeit(lambda: blas.ns_sqrt(C, ws, 4, safe=safe), reps=30)
        print(f"   ns_sqrt(iters=4): issue {r[0]:.1f} us wall {r[1]:.1f} us device {r[2]:.1f} us", flush=True)

from evox_amd.ops import sort
for n in (1000, 10000, 16384):
    k = torch.randn(n, device=dev)
    r1 = timeit(lambda: sort.argsort(k))
    r2 = timeit(lambda: torch.argsort(k, stable=True))
    print(f"argsort n={n}: rank kernel issue {r1[0]:.1f} device {r1[2]:.1f} us; torch issue {r2[0]:.1f} device {r2[2]:.1f} us", flush=True)
