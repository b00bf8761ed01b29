"""Replays the random stream of rounds_equivalence_sweep.py up to one case and looks at where the two SPEA2 truncations part."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from evox_amd.ops import moselect as sel
from evox_amd.ops.distance import cdist

dev = torch.device("cuda")
g = torch.Generator().manual_seed(1234)
target = int(sys.argv[1])


def pts(n, m, style):
    x = torch.rand(n, m, generator=g)
    if style == 1:
        x = x / x.norm(dim=1, keepdim=True) * (1 + 0.1 * torch.rand(n, 1, generator=g))
    elif style == 2:
        x = (x * 12).round() / 12
    elif style == 3:
        k = int(torch.randint(n // 10, n // 2, (1,), generator=g))
        x[torch.randperm(n, generator=g)[:k]] = x[0].clone()
    return x.to(dev)


for case in range(target + 1):
    n = int(torch.randint(2100, 15000, (1,), generator=g)); m = int(torch.randint(2, 6, (1,), generator=g))
    style = case % 4
    x = pts(n, m, style)
    rem = int(torch.randint(1, n - 1, (1,), generator=g))
    n2 = max(n, 4200)
    x2 = pts(n2, 2, style)
    until = None if case % 2 else int(torch.randint(1, n2, (1,), generator=g))
    order = torch.randperm(n, generator=g); _ = torch.rand(1, generator=g)
print("case", target, "n", n, "m", m, "style", style, "remove", rem, "distinct points", int(torch.unique(x, dim=0).shape[0]))
mask = torch.ones(n, dtype=torch.bool, device=dev)
rem = min(rem, n - 2)
# first removal count at which the two forms differ (bisection over the number of removals)
lo, hi = 0, rem
while hi - lo > 1:
    mid = (lo + hi) // 2
    os.environ["EVOX_AMD_SPEA2_LAZY"] = "0"; a = sel.spea2_truncate(x, mask, mid)
    os.environ["EVOX_AMD_SPEA2_LAZY"] = "1"; b = sel.spea2_truncate(x, mask, mid)
    if torch.equal(a, b): lo = mid
    else: hi = mid
os.environ["EVOX_AMD_SPEA2_LAZY"] = "0"; a = sel.spea2_truncate(x, mask, hi)
os.environ["EVOX_AMD_SPEA2_LAZY"] = "1"; st = {}; b = sel.spea2_truncate_rounds(x, hi, st)
da, db = torch.nonzero(a & ~b).flatten().tolist(), torch.nonzero(b & ~a).flatten().tolist()
print("first difference at removal", hi, "kept only by matrix form:", da, "kept only by rounds:", db, st)
os.environ["EVOX_AMD_SPEA2_LAZY"] = "0"; prev = sel.spea2_truncate(x, mask, lo)          # state before the differing removal
D = cdist(x, x); D.fill_diagonal_(float("inf"))
D = torch.where(prev[None, :] & prev[:, None], D, torch.full_like(D, float("inf")))
rm = D.min(1).values
for i in set(da + db):
    print(" point", i, "alive before", bool(prev[i]), "rowmin", float(rm[i]), "bits", hex(rm[i].view(torch.int32).item() & 0xffffffff), "coords", x[i].tolist())
best = torch.argmin(torch.where(prev, rm, torch.full_like(rm, float("inf"))))
print(" expected removal (smallest rowmin, smallest index among equals):", int(torch.nonzero(rm == rm[best]).flatten()[0]), "rowmin", float(rm[best]),
      "points with that rowmin:", int((rm == rm[best]).sum()))
# nearest-neighbour distances of all points: the rescan kernels against the distance matrix, bit for bit
from evox_amd.ops._util import hip, check, stream_of
pts_ = x.contiguous().float()
alive = torch.ones(n, dtype=torch.uint8, device=dev)
rowmin = torch.empty(n, dtype=torch.float32, device=dev); nn = torch.empty(n, dtype=torch.int32, device=dev)
ctrl = torch.empty(int(hip().evox_hip_spea2_lazy_ctrl_words()), dtype=torch.int32, device=dev)
scratch = torch.empty(int(hip().evox_hip_spea2_lazy_scratch_words(n)) // 2 + 1, dtype=torch.int64, device=dev)
check(hip().evox_hip_spea2_lazy_rounds(pts_.data_ptr(), n, m, alive.data_ptr(), rowmin.data_ptr(), nn.data_ptr(), 0,
                                       ctrl.data_ptr(), scratch.data_ptr(), 1, 1, stream_of(pts_)), "rounds")
D0 = cdist(x, x); D0.fill_diagonal_(float("inf"))
ref = D0.min(1).values
diff = torch.nonzero(rowmin != ref).flatten()
print("rowmin differs from the matrix at", int(diff.numel()), "of", n, "points")
for i in diff[:5].tolist():
    j = int(nn[i])
    print("  point", i, "rescan", float(rowmin[i]), hex(rowmin[i].view(torch.int32).item()), "matrix", float(ref[i]), hex(ref[i].view(torch.int32).item()),
          "nn", j, "matrix entry to nn", hex(D0[i, j].view(torch.int32).item()), "and back", hex(D0[j, i].view(torch.int32).item()))
import numpy as np, itertools
i = 458; j = int(nn[i])
a = x[i].cpu().numpy().astype(np.float32); b = x[j].cpu().numpy().astype(np.float32)
print("pair", i, j, a.tolist(), b.tolist())
def chain(order, fused):
    acc = np.float32(0)
    for t, k in enumerate(order):
        d = np.float32(a[k] - b[k])
        if fused:
            acc = np.float32(np.float64(d) * np.float64(d) + np.float64(acc))
        else:
            acc = np.float32(np.float32(d * d) + acc)
    return np.float32(np.sqrt(np.float64(acc)))
for order in ((0, 1, 2, 3), (3, 2, 1, 0)):
    for fused in (True, False):
        v = chain(order, fused)
        print("  order", order, "fused", fused, hex(v.view(np.int32)))
