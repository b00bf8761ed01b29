"""One-off probe: time the stock torch/rocSOLVER building blocks of the CMA-ES headline config."""
import time, torch, sys
dev = torch.device("cuda:0")
print(torch.__version__, torch.cuda.get_device_name(0), flush=True)
def bench(name, fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/n
    print(f"{name}: {dt*1e3:.3f} ms", flush=True)
n=1000; lam=10000; mu=5000
A=torch.randn(n,n,device=dev); C=A@A.T/n+torch.eye(n,device=dev)
bench("eigh fp32 n=1000", lambda: torch.linalg.eigh(C))
C64=C.double()
bench("eigh fp64 n=1000", lambda: torch.linalg.eigh(C64), n=3, warm=1)
bench("cholesky fp32", lambda: torch.linalg.cholesky(C))
Z=torch.randn(lam,n,device=dev); B=torch.randn(n,n,device=dev)
bench("randn 1e7", lambda: torch.randn(lam,n,device=dev))
bench("gemm fp32 10000x1000x1000", lambda: Z@B)
Zb=Z.bfloat16(); Bb=B.bfloat16()
bench("gemm bf16 10000x1000x1000", lambda: Zb@Bb)
Y=Z[:mu]; w=torch.rand(mu,device=dev)
bench("syrk fp32 Yt W Y", lambda: (Y.T*w)@Y)
bench("gemm fp32 1000^3", lambda: B@B)
f=torch.randn(lam,device=dev)
bench("argsort 1e4", lambda: torch.argsort(f))
bench("sort 1e5", lambda: torch.sort(torch.randn(100000,device=dev)))
def ackley(x):
    return -20*torch.exp(-0.2*torch.sqrt((x*x).mean(1)))-torch.exp(torch.cos(2*3.14159265*x).mean(1))+20+2.718281828
bench("ackley eager", lambda: ackley(Z))
bench("tiny op launch", lambda: f.add_(1.0), n=1000)
for k in (4,):
    s=[torch.cuda.Stream() for _ in range(k)]
# svd/qr
bench("qr 1000", lambda: torch.linalg.qr(B))
# graph capture sanity
g=torch.cuda.CUDAGraph(); x=torch.zeros(1000,device=dev)
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): x.add_(1)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(50): x.add_(1)
bench("graph 50 tiny ops", lambda: g.replay(), n=100)
try:
    import ctypes
    print(torch.backends.cuda.preferred_linalg_library())
except Exception as e: print(e)
