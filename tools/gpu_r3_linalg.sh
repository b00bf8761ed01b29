#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_ops_gpu.py -x -q --timeout 120 --timeout-method thread -k "small_eigh or small_svd" > $out/linalg_tests.log 2>&1 || { tail -40 $out/linalg_tests.log; exit 1; }
tail -3 $out/linalg_tests.log
timeout -k 10 300 python - <<'PY' > $out/linalg_timing.txt 2>&1
import time, torch
from evox_amd.ops import linalg_small as ls
dev = torch.device("cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for n in (32, 64, 100, 128, 160, 192, 256, 384):
    a = torch.randn(n, 3 * n, device=dev); C = a @ a.T / (3 * n)
    ev, B = ls.eigh_small(C)
    C2 = C + 0.005 * C * torch.rand_like(C); C2 = 0.5 * (C2 + C2.T)
    cold = t(lambda: ls.eigh_small(C, sort=False)); warm = t(lambda: ls.eigh_small(C2, B0=B, sort=False))
    lib = t(lambda: torch.linalg.eigh(C), 5); host = t(lambda: [x.to(dev) for x in torch.linalg.eigh(C.cpu())], 5)
    chol = t(lambda: ls.cholesky_small(C + torch.eye(n, device=dev))); chol_lib = t(lambda: torch.linalg.cholesky(C + torch.eye(n, device=dev)), 5)
    print(f"n={n}: jacobi eigh cold {cold:.3f} ms, warm {warm:.3f} ms, rocSOLVER {lib:.3f} ms, host LAPACK {host:.3f} ms | cholesky {chol:.3f} ms, library {chol_lib:.3f} ms", flush=True)
for n in (50, 100):
    S = torch.randn(n, n, device=dev) * 0.1
    print(f"n={n}: expm kernel {t(lambda: ls.expm_small(S)):.3f} ms, library {t(lambda: torch.linalg.matrix_exp(S), 5):.3f} ms")
X = torch.randn(50, 100, device=dev)
print(f"svd 50x100: kernel {t(lambda: ls.svd_vt_small(X)):.3f} ms, host {t(lambda: torch.linalg.svd(X.cpu(), full_matrices=False)[2].to(dev), 5):.3f} ms")
print(f"qr 1000x8: kernel {t(lambda: ls.qr_q_small(torch.randn(1000, 8, device=dev))):.3f} ms, library {t(lambda: torch.linalg.qr(torch.randn(1000, 8, device=dev)), 5):.3f} ms")
stack = torch.stack([C] * 100)
print(f"batched eigh 100 x n=384?? skipped; 100 x n=100:", end=" ")
a = torch.randn(100, 100, 300, device=dev); Cs = a @ a.transpose(1, 2) / 300
print(f"kernel {t(lambda: ls.eigh_small(Cs, sort=False), 5):.3f} ms, host {t(lambda: torch.linalg.eigh(Cs.cpu()), 3):.3f} ms")
PY
cat $out/linalg_timing.txt
timeout -k 10 500 python tools/graph_survey.py > $out/graph_survey.log 2>&1; tail -45 $out/graph_survey.log | cut -c1-200
