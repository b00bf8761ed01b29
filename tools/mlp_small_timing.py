"""Time per call of the per-individual MLP forward for small policies on a mirrored population: the one-lane-per-individual
kernel on the transposed population (all rows / pairs from the first half) against one workgroup per pair."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex
from evox_amd.models import MLP
from evox_amd.ops import es

dev = torch.device("cuda")


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


for sizes in ((11, 32, 32, 3), (17, 32, 32, 6), (27, 32, 32, 8), (27, 64, 64, 8)):
    net = MLP(list(sizes))
    center = net.init(ex.PRNGKey(3)).to(dev)
    for P in (1024, 4096, 16384, 65536):
        pop = es.perturb(ex.PRNGKey(4).to(dev), center, 0.1, P, mirrored=True)
        obs = torch.randn(P, 1, sizes[0], device=dev)
        row = {"sizes": sizes, "pop": P}
        for name, small, mirror in (("small_all_rows_us", "1", "0"), ("small_pairs_us", "1", "1"), ("workgroup_per_pair_us", "0", "1")):
            os.environ["EVOX_AMD_MLP_SMALL"], os.environ["EVOX_AMD_MLP_MIRROR"] = small, mirror
            if hasattr(pop, "_evox_wt"):
                del pop._evox_wt
            row[name] = round(timed(lambda: net.batched_apply(pop, obs)), 2)
        print(json.dumps(row), flush=True)
