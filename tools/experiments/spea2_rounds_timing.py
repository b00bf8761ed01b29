"""SPEA2 truncation: the one-at-a-time loop over a distance matrix against the round-wise form (same survivors)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from evox_amd.ops import moselect as sel

dev = torch.device("cuda")


def front(n, m, seed, kind):
    g = torch.Generator(device="cpu").manual_seed(seed)
    d = torch.randn(n, m, generator=g).abs()
    d = d / d.norm(dim=1, keepdim=True)
    if kind == "clones":
        d[n // 3: n // 3 + n // 4] = d[5]
    elif kind == "grid":
        d = (d * 16).round() / 16
    return d.to(dev)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


cases = [(3000, 3, "front"), (6000, 2, "front"), (8000, 3, "clones"), (5000, 3, "grid"), (20000, 3, "front"), (20000, 5, "front")]
for n, m, kind in cases:
    pts = front(n, m, n + m, kind)
    mask = torch.ones(n, dtype=torch.bool, device=dev)
    stats = {}
    print(json.dumps({"case": [n, m, kind], "step": "rounds"}), flush=True)
    t_new, a_new = timed(lambda: sel.spea2_truncate_rounds(pts, n // 2, stats))
    row = {"n": n, "m": m, "kind": kind, "rounds_ms": t_new, **stats}
    print(json.dumps({"case": [n, m, kind], "step": "matrix"}), flush=True)
    os.environ["EVOX_AMD_SPEA2_LAZY"] = "0"
    t_old, a_old = timed(lambda: sel.spea2_truncate(pts, mask, n // 2), reps=1)
    os.environ["EVOX_AMD_SPEA2_LAZY"] = "1"
    row.update(matrix_ms=t_old, same_survivors=bool(torch.equal(a_new, a_old)), differ=int((a_new != a_old).sum()),
               survivors=int(a_new.sum()))
    print(json.dumps(row), flush=True)
