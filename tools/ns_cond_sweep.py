"""Cost of the Newton–Schulz factorisation as the covariance becomes ill-conditioned (n = 1000 by default).

For every condition number a covariance ``C = Q diag(λ) Qᵀ`` (log-uniform spectrum between 1 and cond, random
orthogonal Q) is factorised repeatedly through the controller CMA-ES uses (``ops.cma.NewtonSchulzSqrt.step``) until
the iteration count has settled; reported are the settled count, which driver ran (bf16 planes with mirrored products, with plain
products on two or three planes, or the fp32 library products), the time of one factorisation, its residual ``‖Z·Y − I‖_F/√n`` measured in fp64 on the host, and
for comparison the time of ``torch.linalg.eigh`` on the same matrix.

    python tools/ns_cond_sweep.py [--n 1000] [--planes 2] > gpurun_out/ns_cond_sweep.txt
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from evox_amd.ops import cma as _cma            # noqa: E402
from evox_amd.ops import cma_bf16 as _b16       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--planes", type=int, default=2)
    ap.add_argument("--settle", type=int, default=40)
    ap.add_argument("--timed", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda")
    n = args.n
    g = torch.Generator().manual_seed(5)
    Q, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))
    print(f"# n={n} planes={args.planes}  columns: cond  iterations  driver  ms/factorisation  residual(fp64)  eigh_ms")
    for cond in (1.0, 3.0, 10.0, 30.0, 1e2, 1e3, 1e4, 1e5, 1e6):
        lam = torch.pow(torch.tensor(cond, dtype=torch.float64), torch.linspace(0, 1, n, dtype=torch.float64))
        C64 = (Q * lam) @ Q.T
        C = C64.to(torch.float32).to(dev)
        ns = _cma.NewtonSchulzSqrt(planes=args.planes, bf16_copy=True)
        ctrl = ns.init_control(n, dev)
        ws = _b16.Workspace(n, 16, 8, dev, ns_planes=max(args.planes, 2))
        eye = torch.eye(n, device=dev)
        prev = (eye, eye)
        sym0 = 0
        for i in range(args.settle + args.timed):
            if i == args.settle:
                torch.cuda.synchronize()
                sym0 = ns.symmetric_calls
                t0 = time.perf_counter()
            B, inv, _b, ctrl = ns.step(C, ctrl, prev=prev, ws=ws)
            prev = (B, inv)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.timed * 1e3
        Y, Z = B.double().cpu(), inv.double().cpu()
        res = float(torch.linalg.matrix_norm(Z @ Y - torch.eye(n, dtype=torch.float64)) / n ** 0.5)
        acc = float(torch.linalg.matrix_norm(Y @ Y - C64) / torch.linalg.matrix_norm(C64))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            torch.linalg.eigh(C)
        torch.cuda.synchronize()
        eigh_ms = (time.perf_counter() - t0) / 3 * 1e3
        sym = int(ctrl["ns_sym"])          # driver of the last call: 0 library, 1 mirrored planes, 2 / 3 plain products on 2 / 3 planes
        driver = {0: "fp32-library", 1: "bf16-planes-symmetric", 2: "bf16-planes-plain-2", 3: "bf16-planes-plain-3"}[sym]
        print(f"{cond:9.0e}  {int(ctrl['ns_iters']):3d}  {driver:22s}  {ms:8.3f}  inv-residual {res:.2e}  sqrt-error {acc:.2e}  eigh {eigh_ms:7.2f}",
              flush=True)


if __name__ == "__main__":
    main()
