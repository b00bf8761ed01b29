"""One-off probe: symmetric eigendecomposition of a 1000x1000 fp32 matrix with the available back ends."""
import time, torch
dev = torch.device("cuda")
n = 1000
A = torch.randn(n, n, device=dev); C = A @ A.T / n + torch.eye(n, device=dev)
def bench(name, fn, reps=5):
    try:
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t) / reps * 1e3:.2f} ms", flush=True)
    except Exception as e:
        print(f"{name}: failed {type(e).__name__}: {str(e)[:120]}", flush=True)
print("default linalg library:", torch.backends.cuda.preferred_linalg_library())
bench("eigh default", lambda: torch.linalg.eigh(C))
for lib in ("magma", "cusolver"):
    try:
        torch.backends.cuda.preferred_linalg_library(lib)
        bench(f"eigh {lib}", lambda: torch.linalg.eigh(C))
    except Exception as e:
        print(lib, "unavailable:", str(e)[:100])
Cc = C.cpu()
t = time.perf_counter(); torch.linalg.eigh(Cc); print(f"eigh cpu: {(time.perf_counter() - t) * 1e3:.1f} ms")
bench("eigh fp64 default", lambda: torch.linalg.eigh(C.double()))
bench("svd", lambda: torch.linalg.svd(C))

# rocSOLVER's Jacobi solver on (a) C itself, (b) C rotated into an approximate eigenbasis (warm start)
import ctypes, os
lib_dir = os.path.join(os.path.dirname(torch.__file__), "lib")
rb = ctypes.CDLL(os.path.join(lib_dir, "librocblas.so"))
rs = ctypes.CDLL(os.path.join(lib_dir, "librocsolver.so"))
h = ctypes.c_void_p()
assert rb.rocblas_create_handle(ctypes.byref(h)) == 0
rb.rocblas_set_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
rb.rocblas_set_stream(h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
rs.rocsolver_ssyevj.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_void_p]
resid = torch.zeros(1, device=dev); nsw = torch.zeros(1, dtype=torch.int32, device=dev)
W = torch.empty(n, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
def syevj(M, max_sweeps=20, abstol=1e-6):
    A_ = M.clone()
    st = rs.rocsolver_ssyevj(h, 252, 211, 121, n, A_.data_ptr(), n, ctypes.c_float(abstol), resid.data_ptr(), max_sweeps,
                             nsw.data_ptr(), W.data_ptr(), info.data_ptr())
    assert st == 0, st
    return A_
bench("syevj cold", lambda: syevj(C))
print("  sweeps", int(nsw), "residual", float(resid))
evals, B = torch.linalg.eigh(C)
C2 = C + 0.005 * (lambda G: G @ G.T / n)(torch.randn(n, n, device=dev))          # next generation's matrix
def warm():
    A2 = B.T @ C2 @ B
    V = syevj((A2 + A2.T) / 2)
    return B @ V.T
bench("syevj warm (rotated by previous eigenvectors)", warm)
print("  sweeps", int(nsw), "residual", float(resid))
