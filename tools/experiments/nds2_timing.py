"""Two-objective non-dominated sort: pair-test rounds against the one-launch prefix-minimum scan."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from evox_amd.ops import _nds_hip

dev = torch.device("cuda")


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


for n, kind in ((2000, "cloud"), (4000, "cloud"), (10000, "cloud"), (20000, "cloud"), (32768, "cloud"), (20000, "zdt")):
    g = torch.Generator().manual_seed(n)
    if kind == "cloud":
        x = torch.rand(n, 2, generator=g).to(dev)
    else:                                              # a converging ZDT1-like population: few fronts
        t = torch.rand(n, generator=g)
        x = torch.stack([t, 1 - t.sqrt() + 0.02 * torch.rand(n, generator=g)], dim=1).to(dev)
    for until in (None, n // 2):
        os.environ["EVOX_AMD_NDS_M2_SCAN"] = "0"
        t_old, r_old = timed(lambda: _nds_hip.rank(x, until))
        os.environ["EVOX_AMD_NDS_M2_SCAN"] = "1"
        t_new, r_new = timed(lambda: _nds_hip.rank(x, until))
        print(json.dumps({"n": n, "kind": kind, "until": until, "fronts": int(r_new.max()) + 1, "pair_test_ms": t_old,
                          "scan_ms": t_new, "equal": bool(torch.equal(r_old, r_new))}), flush=True)
