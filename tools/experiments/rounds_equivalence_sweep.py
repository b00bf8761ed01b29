"""Randomised equivalence sweep: round-wise IBEA removal / SPEA2 truncation / two-objective sort / knee suppression against
their one-at-a-time or pair-test forms on many random shapes (prints one line per family; exits non-zero on a mismatch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from evox_amd.ops import moselect as sel, _nds_hip

dev = torch.device("cuda")
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
bad = 0


def pts(n, m, style):
    x = torch.rand(n, m, generator=g)
    if style == 1:                                   # a front with noise
        x = x / x.norm(dim=1, keepdim=True) * (1 + 0.1 * torch.rand(n, 1, generator=g))
    elif style == 2:                                 # quantised: many exact ties and duplicates
        x = (x * 12).round() / 12
    elif style == 3:                                 # a flood of clones
        k = int(torch.randint(n // 10, n // 2, (1,), generator=g))
        x[torch.randperm(n, generator=g)[:k]] = x[0].clone()
    return x.to(dev)


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for case in range(n_cases):
    m = int(torch.randint(2, 9, (1,), generator=g))
    n = int(torch.randint(2100, 15000 if m <= 4 else 8100, (1,), generator=g))      # sizes the one-at-a-time IBEA kernel takes
    style = case % 4
    x = pts(n, m, style)
    obj = (x - x.min()) / (x.max() - x.min())
    rem = int(torch.randint(1, n - 1, (1,), generator=g))
    fit, C = sel.ibea_fitness(obj, 0.05)
    ok1 = True
    if n <= int(sel.hip().evox_hip_ibea_remove_obj_max(m)):          # the one-at-a-time kernel takes this size
        os.environ["EVOX_AMD_IBEA_LAZY"] = "0"; want = sel.ibea_remove_streaming(obj, fit, C, 0.05, rem)
        os.environ["EVOX_AMD_IBEA_LAZY"] = "1"; got = sel.ibea_remove_streaming(obj, fit, C, 0.05, rem)
        ok1 = torch.equal(got, want)
    else:
        got = sel.ibea_remove_streaming(obj, fit, C, 0.05, rem)
        ok1 = int(got.sum()) == n - rem
    mask = torch.ones(n, dtype=torch.bool, device=dev)
    os.environ["EVOX_AMD_SPEA2_LAZY"] = "0"; want = sel.spea2_truncate(x, mask, min(rem, n - 2))
    os.environ["EVOX_AMD_SPEA2_LAZY"] = "1"; got = sel.spea2_truncate(x, mask, min(rem, n - 2))
    ok2 = torch.equal(got, want)
    x2 = x[:, :2].contiguous(); n2 = max(n, 4200)
    x2 = pts(n2, 2, style)
    until = None if case % 2 else int(torch.randint(1, n2, (1,), generator=g))
    os.environ["EVOX_AMD_NDS_M2_SCAN"] = "0"; want = _nds_hip.rank(x2, until)
    os.environ["EVOX_AMD_NDS_M2_SCAN"] = "1"; got = _nds_hip.rank(x2, until)
    ok3 = torch.equal(got, want)
    order = torch.randperm(n, generator=g).to(dev); R = torch.full((m,), 0.02 + 0.1 * float(torch.rand(1, generator=g)), device=dev)
    os.environ["EVOX_AMD_KNEE"] = "rounds"; want = sel.knee_suppress(x, order, R)
    os.environ["EVOX_AMD_KNEE"] = "states"; got = sel.knee_suppress(x, order, R)
    ok4 = torch.equal(got, want)
    print(f"case {case}: n={n} m={m} style={style} remove={rem} ibea={ok1} spea2={ok2} nds2={ok3} knee={ok4}", flush=True)
    bad += (not ok1) + (not ok2) + (not ok3) + (not ok4)
print("mismatches", bad)
sys.exit(1 if bad else 0)
