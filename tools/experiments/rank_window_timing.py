"""Device time of ranking one rank's rows against the whole fitness vector (counting) against a full sort, for the
population sizes of 1, 2, 4 and 8 ranks of 10 000 rows each."""
import torch
from evox_amd.ops import cma_native as nat, sort as _sort


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


local = 10000
for world in (1, 2, 4, 8):
    n = local * world
    fit = torch.randn(n, device="cuda")
    mu = n // 2
    padded = torch.cat([torch.rand(mu, device="cuda"), torch.zeros(n - mu, device="cuda")])
    flags = torch.zeros(4, device="cuda")
    keep = 5632
    lo = (world - 1) * local
    t_count = timed(lambda: nat.rank_select_local(fit, lo, local, mu, keep, padded, flags))
    t_sort = timed(lambda: _sort.argsort_with_rank(fit))
    print(f"world {world}: n={n}  count+select {t_count:.1f} us   full sort with rank {t_sort:.1f} us", flush=True)
