"""In-tree build of the native libraries.

Two shared objects are produced next to this file:

* ``libevox_cpu.so``  – host twins (bit-exact Philox draws, reference sorts) built with g++.
* ``libevox_hip.so``  – every gfx950 kernel in ``csrc/*.hip`` built with hipcc
  (``--offload-arch=gfx950``; cross-compiles without a GPU present).

Both expose a plain C ABI and are loaded with ctypes (see ``__init__.py``), so the build
needs neither the PyTorch headers nor a JIT cache and takes seconds per file.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = HERE / "build"
CPU_LIB = HERE / "libevox_cpu.so"
HIP_LIB = HERE / "libevox_hip.so"

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CXX = os.environ.get("CXX") or shutil.which("g++") or "g++"

HIP_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fvisibility=hidden",
    "-Wno-unused-result",
    "-Wno-unused-value",
    # keep hipcc from forming packed-fp32 VALU ops out of scalar float code
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
]
CPU_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-march=x86-64-v3"]
# EVOX_AMD_SANITIZE=address,undefined builds the HOST library with those sanitizers (run the CPU tests
# with LD_PRELOAD=$(g++ -print-file-name=libasan.so)); device code is never instrumented.
_SAN = os.environ.get("EVOX_AMD_SANITIZE", "").strip()
if _SAN:
    CPU_FLAGS = [f for f in CPU_FLAGS if f != "-O3"] + ["-O1", "-g", "-fno-omit-frame-pointer", f"-fsanitize={_SAN}"]


def _newer(target: Path, sources) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(s).stat().st_mtime <= t for s in sources)


def cpu_sources():
    return sorted(CSRC.glob("*_cpu.cpp"))


def hip_sources():
    return sorted(CSRC.glob("*.hip"))


def headers():
    return sorted(CSRC.glob("*.h"))


def build_cpu(force: bool = False, verbose: bool = False) -> Path:
    srcs = cpu_sources()
    if not force and _newer(CPU_LIB, srcs + headers()):
        return CPU_LIB
    tmp = CPU_LIB.with_suffix(f".tmp{os.getpid()}.so")
    cmd = [CXX, *CPU_FLAGS, *map(str, srcs), "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, CPU_LIB)
    return CPU_LIB


def _compile_hip(src: Path, verbose: bool) -> Path:
    OBJ.mkdir(exist_ok=True)
    obj = OBJ / (src.stem + ".o")
    if _newer(obj, [src] + headers()):
        return obj
    cmd = [HIPCC, *HIP_FLAGS, "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"hipcc failed on {src.name}")
    if verbose:
        noise = "packed-fp32-ops"
        err = "\n".join(l for l in r.stderr.splitlines() if noise not in l)
        if err.strip():
            sys.stderr.write(err + "\n")
    return obj


class _BuildLock:
    """Inter-process lock around a rebuild: the ranks of a multi-process job import the package at the same time, and
    only one of them may compile (the others then find the library fresh)."""

    def __init__(self, name: str):
        self.path = HIP_LIB.parent / name
        self.fd = None

    def __enter__(self):
        import fcntl

        self.fd = open(self.path, "w")
        fcntl.flock(self.fd, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl

        fcntl.flock(self.fd, fcntl.LOCK_UN)
        self.fd.close()


def build_hip(force: bool = False, verbose: bool = False, jobs: int | None = None) -> Path:
    srcs = hip_sources()
    if not force and _newer(HIP_LIB, srcs + headers()):
        return HIP_LIB
    with _BuildLock(".build_hip.lock"):
        if not force and _newer(HIP_LIB, srcs + headers()):
            return HIP_LIB                   # another process built it while this one waited
        return _build_hip_locked(srcs, force, verbose, jobs)


def _build_hip_locked(srcs, force: bool, verbose: bool, jobs) -> Path:
    if force and OBJ.exists():
        shutil.rmtree(OBJ)
    jobs = jobs or min(8, max(1, (os.cpu_count() or 2) // 2), int(os.environ.get("MAX_JOBS", "8")))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile_hip(s, verbose), srcs))
    tmp = HIP_LIB.with_suffix(f".tmp{os.getpid()}.so")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, HIP_LIB)
    return HIP_LIB


def have_hipcc() -> bool:
    return Path(HIPCC).exists() or shutil.which(HIPCC) is not None


def build_all(force: bool = False, verbose: bool = True):
    out = [build_cpu(force, verbose)]
    if have_hipcc():
        out.append(build_hip(force, verbose))
    elif verbose:
        print("hipcc not found: skipping libevox_hip.so", flush=True)
    return out


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("native build ok")
