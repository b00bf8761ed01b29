"""Newton-Schulz plane driver with plain (un-mirrored) iteration products against the mirrored form and the fp32 library
driver, by conditioning: residuals in fp64 and time per factorisation.
usage: ns_full_check.py [n]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evox_amd.ops import cma_bf16 as b16, blas

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
Q, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))
ws = b16.Workspace(n, 8, 4, dev)
eye = torch.eye(n, dtype=torch.float64, device=dev)
for cond in (1.5, 10, 30, 1e2, 1e3, 1e4, 1e5, 1e6):
    ev = torch.logspace(0, -torch.log10(torch.tensor(float(cond))).item(), n, dtype=torch.float64)
    C64 = ((Q * ev) @ Q.T).to(dev)
    C64 = 0.5 * (C64 + C64.T)
    C = C64.float()
    row = {"n": n, "cond": cond}
    for name in ("mirrored", "full", "full3", "library"):
        best = None
        for iters in (1, 2, 3, 4, 5, 6, 8, 10, 12, 14, 16, 20, 24):
            try:
                if name == "library":
                    w = blas.NSWorkspace(n, dev)
                    S, Z = blas.ns_sqrt(C, w, iters, power_steps=8, safe=False)
                else:
                    S, Z, _ = b16.ns_sqrt(C, ws, iters, power_steps=8, full=name.startswith("full"), planes=3 if name == "full3" else 2)
            except Exception as e:                                        # noqa: BLE001
                best = {"error": str(e)[:80]}
                break
            S64, Z64 = S.double(), Z.double()
            r_sq = float((S64 @ S64 - C64).norm() / C64.norm())
            r_inv = float((Z64 @ S64 - eye).norm() / n ** 0.5)
            if not (r_inv == r_inv):
                continue
            if best is None or r_inv < best["inv_residual"] * 0.5:
                best = {"iters": iters, "sqrt_residual": r_sq, "inv_residual": r_inv}
            if r_inv < 5e-4 and best.get("iters_to_tol") is None:
                best["iters_to_tol"] = iters
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    if name == "library":
                        blas.ns_sqrt(C, w, iters, power_steps=1, safe=False)
                    else:
                        b16.ns_sqrt(C, ws, iters, power_steps=1, full=name.startswith("full"), planes=3 if name == "full3" else 2)
                torch.cuda.synchronize()
                best["ms_at_tol"] = (time.perf_counter() - t0) / 20 * 1e3
        row[name] = best
    print(json.dumps(row), flush=True)
