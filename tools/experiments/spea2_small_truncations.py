import sys, time, torch, json
sys.path.insert(0, "/root/repo")
from evox_amd.ops import moselect as sel
dev = torch.device("cuda")
for n, rem in ((10500, 500), (12000, 2000), (2500, 500), (4000, 2000), (10100, 100)):
    g = torch.Generator().manual_seed(n)
    d = torch.randn(n, 3, generator=g).abs(); d = (d / d.norm(dim=1, keepdim=True)).to(dev)
    st = {}
    sel.spea2_truncate_rounds(d, rem, st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): sel.spea2_truncate_rounds(d, rem, st)
    torch.cuda.synchronize()
    print(json.dumps({"n": n, "remove": rem, "ms": (time.perf_counter() - t0) / 5 * 1e3, **st}), flush=True)
