"""Which stage of the Gaussian transform differs between host and device?"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evox_amd import _native
from evox_amd.core import random as rnd
n = 1 << 16
k = rnd.PRNGKey(99)
cpu = _native.cpu(); hip = _native.hip()
cpu.evox_cpu_debug_parts.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
hip.evox_hip_debug_parts.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
a = torch.empty(n, 4)
cpu.evox_cpu_debug_parts(k.data_ptr(), a.data_ptr(), n)
kg = k.cuda(); b = torch.empty(n, 4, device="cuda")
hip.evox_hip_debug_parts(kg.data_ptr(), b.data_ptr(), n, None)
torch.cuda.synchronize()
b = b.cpu()
for j, name in enumerate(["log", "sqrt", "sin", "cos"]):
    d = (a[:, j] != b[:, j])
    print(name, "mismatches", int(d.sum()), "max abs", float((a[:, j] - b[:, j]).abs().max()))
    if d.any():
        i = int(d.nonzero()[0])
        print("   e.g.", i, a[i, j].item(), b[i, j].item(), a[i, j].view(torch.int32).item(), b[i, j].view(torch.int32).item())
