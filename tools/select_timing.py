"""Device time of the selection kernels (registers / co-operating workgroups / one workgroup / five launches) and of the library's top-k."""
import json
import sys

import torch

from evox_amd.ops import sort as _sort


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    for n in (1000, 4096, 10000, 20000, 50000, 65536, 131072, 262144):
        for kind in ("uniform", "clustered"):
            x = torch.rand(n, device="cuda") if kind == "uniform" else torch.rand(n, device="cuda") * 1e-3 + 700.0
            k = max(1, n // 20)
            row = {"n": n, "k": k, "keys": kind}
            for route in ("registers", "grid", "one_workgroup", "wide"):
                if route == "registers" and n > 65536 or route == "grid" and n > 262144:
                    continue
                _sort._SELECT_ROUTE = route
                row[route + "_us"] = round(timed(lambda: _sort.select_indices(x, k)), 2)
            row["topk_us"] = round(timed(lambda: torch.topk(x, k, largest=False)), 2)
            print(json.dumps(row), file=out, flush=True)


if __name__ == "__main__":
    main()
