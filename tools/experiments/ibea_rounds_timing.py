"""IBEA removal: the one-at-a-time loop on one compute unit against the round-wise form (same survivors)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from evox_amd.ops import moselect as sel

dev = torch.device("cuda")


def population(n, m, seed, duplicates=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    d = torch.randn(n, m, generator=g).abs()
    d = d / d.norm(dim=1, keepdim=True)
    rad = 1 + torch.where(torch.arange(n) < n // 2, 0.02, 0.15) * torch.randn(n, generator=g).abs()
    obj = d * rad[:, None]
    if duplicates == "flood":                      # 3000 clones of one point among distinct ones
        obj[n // 3: n // 3 + 3000] = obj[7]
    elif duplicates:
        obj = (obj * 8).round() / 8
    obj = (obj - obj.min()) / (obj.max() - obj.min())
    return obj.to(dev)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


for n, m, dup in ((4000, 3, False), (20000, 3, False), (20000, 2, False), (20000, 5, False), (8000, 3, True), (20000, 3, "flood"), (30000, 3, False)):
    obj = population(n, m, n + m, dup)
    fit, C = sel.ibea_fitness(obj, 0.05)
    stats = {}
    t_new, a_new = timed(lambda: sel.ibea_remove_rounds(obj, fit, C, 0.05, n // 2, stats))
    row = {"n": n, "m": m, "duplicates": dup, "rounds_ms": t_new, **stats}
    if n <= int(sel.hip().evox_hip_ibea_remove_obj_max(m)):
        os.environ["EVOX_AMD_IBEA_LAZY"] = "0"
        t_old, a_old = timed(lambda: sel.ibea_remove_streaming(obj, fit, C, 0.05, n // 2), reps=2)
        os.environ["EVOX_AMD_IBEA_LAZY"] = "1"
        row.update(single_cu_ms=t_old, same_survivors=bool(torch.equal(a_new, a_old)), survivors=int(a_new.sum()))
    print(json.dumps(row), flush=True)
