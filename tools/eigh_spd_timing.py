"""Time of the one-workgroup eigendecompositions at order n: general form (rotations of C V with V accumulated) against the
positive definite form (rotations of the Cholesky factor), cold and from a nearby start; Cholesky alone.
usage: eigh_spd_timing.py [n ...]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evox_amd.ops import linalg_small as ls

dev = torch.device("cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for n in [int(x) for x in sys.argv[1:]] or [32, 64, 100, 128, 160, 188]:
    for cond in (1.5, 1e3):
        g = torch.Generator().manual_seed(n)
        Q = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))[0]
        lam = torch.logspace(0, torch.log10(torch.tensor(cond)).item(), n, dtype=torch.float64)
        C64 = (Q * lam) @ Q.T
        Z = (Q * lam.sqrt()) @ torch.randn(n, 3 * n, generator=g, dtype=torch.float64)
        C = C64.float().to(dev); C2 = (0.95 * C64 + 0.05 * (Z @ Z.T) / (3 * n)).float().to(dev)
        sw = torch.zeros(1, dtype=torch.int32, device=dev)
        ev, B = ls.eigh_small(C, sort=False, spd=True, sweeps_out=sw); cold_sweeps = int(sw)
        row = {"n": n, "cond": cond,
               "general_cold_ms": t(lambda: ls.eigh_small(C, sort=False)),
               "general_warm_ms": t(lambda: ls.eigh_small(C2, B0=B, sort=False)),
               "spd_cold_ms": t(lambda: ls.eigh_small(C, sort=False, spd=True)), "spd_cold_sweeps": cold_sweeps}
        if n <= 134:
            row["spd_warm_ms"] = t(lambda: ls.eigh_small(C2, B0=B, sort=False, spd=True, sweeps_out=sw)); row["spd_warm_sweeps"] = int(sw)
        row["cholesky_ms"] = t(lambda: ls.cholesky_small(C)); row["cholesky_library_ms"] = t(lambda: torch.linalg.cholesky(C), 5)
        row["host_lapack_ms"] = t(lambda: [x.to(dev) for x in torch.linalg.eigh(C.cpu())], 5)
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
