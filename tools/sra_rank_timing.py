"""Time of SRA's stochastic ranking at n keys: three-word entries against packed one-word entries in one workgroup and in windows over several workgroups, one coin per position
against one per comparison.  usage: sra_rank_timing.py [n]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evox_amd.ops import moselect

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
I1, I2, u = (torch.rand(n, generator=g).to(dev) for _ in range(3))
seed = torch.tensor([5], dtype=torch.int32, device=dev)
for packed, window, block, wg in (("0", "0", "8", "1024"), ("1", "0", "8", "1024"), ("1", "256", "8", "1024"),
                                  ("1", "128", "4", "1024"), ("1", "256", "4", "1024"), ("1", "384", "4", "1024"),
                                  ("1", "64", "4", "512"), ("1", "128", "4", "512"), ("1", "192", "4", "512"),
                                  ("1", "32", "4", "256"), ("1", "64", "4", "256"), ("1", "96", "4", "256"), ("1", "64", "8", "256")):
    os.environ["EVOX_AMD_SRA_PACKED"] = packed
    os.environ["EVOX_AMD_SRA_WINDOW"] = window
    os.environ["EVOX_AMD_SRA_WINDOW_BLOCK"] = block
    os.environ["EVOX_AMD_SRA_WINDOW_THREADS"] = wg
    for coins in ("per_position", "per_comparison"):
        f = (lambda: moselect.stochastic_ranking(I1, I2, u, 0.45)) if coins == "per_position" else \
            (lambda: moselect.stochastic_ranking(I1, I2, None, 0.45, seed=seed))
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        print(json.dumps({"n": n, "packed": packed == "1", "window_halo": int(window), "block": int(block), "window_threads": int(wg), "coins": coins, "ms": (time.perf_counter() - t0) / 3 * 1e3}), flush=True)
