"""Non-dominated sort timings: pair-test kernels against the sweep path, N x m random objectives.

    python tools/nds_timing.py > gpurun_out/nds_timing.txt
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evox_amd.ops import _nds_hip  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


dev = torch.device("cuda")
os.environ["EVOX_AMD_NDS_SWEEP_M2"] = "1"
print("# n  m  population  until  pair_tests_ms  sweep_ms")
for m in (2, 3):
    for n in (8192, 16384, 32768, 65536, 131072, 200000, 400000):
        for kind in ("uniform", "dtlz-like"):
            g = torch.Generator().manual_seed(n)
            x = torch.rand(n, m, generator=g)
            if kind == "dtlz-like":
                x = x / x.norm(dim=1, keepdim=True) * (1 + 0.3 * torch.rand(n, 1, generator=g))
            x = x.to(dev)
            for until in (None, n // 2):
                os.environ["EVOX_AMD_NDS_SWEEP_MIN"] = str(1 << 30)
                a = timed(lambda: _nds_hip.rank(x, until), reps=3 if n > 100000 else 5)
                os.environ["EVOX_AMD_NDS_SWEEP_MIN"] = "1"
                b = timed(lambda: _nds_hip.rank(x, until))
                print(f"{n:7d} {m} {kind:10s} until={str(until):7s} {a:9.3f} {b:9.3f}", flush=True)
