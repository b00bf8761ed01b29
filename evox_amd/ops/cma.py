"""The three heavy steps of covariance-matrix adaptation.

``sample``            X = m + σ · (Z ∘ d) · Bᵀ with Z ~ N(0, I) drawn from the Philox stream
                      of ``key`` inside the GEMM (Z is never written to memory).
``covariance_update`` C' = α·C + c₁·pc pcᵀ + cμ · Yᵀ diag(w) Y with Y gathered from the
                      population through the fitness order, centred and scaled on the fly.
``eigh``              symmetric eigendecomposition C = B diag(d²) Bᵀ (+ C^{-1/2}).

Each has a PyTorch reference (host path / numerics oracle) and a gfx950 implementation.
Reference formulas: ``src/evox/algorithms/so/es_variants/cma_es.py:128-196``.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from .. import _native
from ..core import random as rnd
from ._util import check, hip, on_gpu, stream_of  # noqa: F401


# --------------------------------------------------------------------------------------
# sampling
# --------------------------------------------------------------------------------------
def sample_reference(key, mean, sigma, B, D, pop_size: int, row_offset: int = 0):
    n = mean.shape[0]
    z = rnd.normal(key, (pop_size, n), offset=row_offset * n)
    return mean + sigma * ((z if D is None else z * D) @ B.T)


def sample(key, mean, sigma, B, D, pop_size: int, precision: str = "fp32", row_offset: int = 0):
    """``pop_size`` rows starting at logical row ``row_offset`` of the generation's sample.
    ``D=None`` means ``B`` already is the full factor (``B Bᵀ = C``), saving the column scaling."""
    if on_gpu(mean):
        from . import gemm

        return gemm.cma_sample(key, mean, sigma, B, D, pop_size, precision, row_offset)
    return sample_reference(key, mean, sigma, B, D, pop_size, row_offset)


# --------------------------------------------------------------------------------------
# rank-mu covariance update
# --------------------------------------------------------------------------------------
def covariance_update_reference(C, pop, order, mean, sigma, weights, pc, alpha, c1, cmu):
    mu = weights.shape[0]
    y = (pop[order[:mu]] - mean) / sigma
    return alpha * C + c1 * torch.outer(pc, pc) + cmu * ((y.T * weights) @ y)


def covariance_update(C, pop, order, mean, sigma, weights, pc, alpha, c1, cmu, precision: str = "fp32"):
    """``alpha`` may be a 0-d tensor (it depends on ``hsig``)."""
    if on_gpu(C):
        from . import gemm

        return gemm.cma_covariance_update(C, pop, order, mean, sigma, weights, pc, alpha, c1, cmu, precision)
    return covariance_update_reference(C, pop, order, mean, sigma, weights, pc, alpha, c1, cmu)


def use_factor_form() -> bool:
    """GPU covariance update through the explicit factor S (gather kernel + library GEMM), the
    fastest measured form; ``EVOX_AMD_CMA_COV=fused`` selects the hand-written gathered-operand
    MFMA kernel instead."""
    import os

    return os.environ.get("EVOX_AMD_CMA_COV", "factor") == "factor" and os.environ.get("EVOX_AMD_GEMM", "hip") != "torch"


def elite_factor(pop, order, mean, sigma, sqrt_w, precision: str = "fp32"):
    from . import gemm

    return gemm.cma_elite_factor(pop, order, mean, sigma, sqrt_w, precision)


def covariance_from_factor(C, S, pc, alpha, c1, cmu):
    from . import gemm

    return gemm.cma_covariance_from_factor(C, S, pc, alpha, c1, cmu)


# --------------------------------------------------------------------------------------
# eigendecomposition
# --------------------------------------------------------------------------------------
def _host_max() -> int:
    """Largest matrix order whose dense factorisations take the host round trip (``EVOX_AMD_HOST_LINALG_MAX``).
    Measured on an MI355X box: the device solvers are latency-bound for small matrices — a symmetric
    eigendecomposition takes 1.35 ms at n = 100 on the GPU against 0.30 ms for copy to the host, LAPACK and copy
    back (n = 300: 5.5 vs 2.8 ms; even at n = 500), a thin SVD 2.6 vs 0.43 ms."""
    return int(os.environ.get("EVOX_AMD_HOST_LINALG_MAX", "384"))


def _device_jacobi(C) -> bool:
    """Orders whose symmetric eigendecomposition runs in the one-workgroup Jacobi kernels (``csrc/linalg_small.hip``): up
    to ``EVOX_AMD_DEVICE_EIGH_MAX`` (default 134, the largest order whose warm-started positive definite form works in
    LDS: order 100 takes 0.33–0.54 ms there from the previous eigenvectors against 0.37 ms for copy to the host, LAPACK
    and copy back, without stalling the stream; beyond 134 the host round trip is faster in eager issue) — and up to 384
    while a generation is being recorded into a graph, where a host round trip is not possible."""
    from . import linalg_small as _ls

    n = C.shape[-1]
    if not _ls.device_linalg(C, n):
        return False
    return n <= int(os.environ.get("EVOX_AMD_DEVICE_EIGH_MAX", "134")) or torch.cuda.is_current_stream_capturing()


def eigh_reference(C, B_prev: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(B, evals)`` of the symmetric ``C``.  ``B_prev``: eigenvectors of a nearby matrix (warm start of the device
    kernel; ignored by the library routes)."""
    C = torch.triu(C) + torch.triu(C, 1).T
    if _device_jacobi(C):
        from . import linalg_small as _ls

        evals, B = _ls.eigh_small(C, B_prev, spd=True)         # every caller decomposes a covariance matrix
        return B.contiguous(), evals
    if C.is_cuda and C.shape[-1] <= _host_max():
        evals, B = torch.linalg.eigh(C.cpu())
        return B.to(C.device, non_blocking=True), evals.to(C.device, non_blocking=True)
    evals, B = torch.linalg.eigh(C)
    return B, evals


def small_svd_vt(X: torch.Tensor) -> torch.Tensor:
    """``Vᵀ`` of the thin SVD of ``X`` (k, m): one-sided Jacobi on the device for k ≤ 384 ≤ m-independent sizes
    (``csrc/linalg_small.hip``), the library otherwise (small problems through the host, see ``_host_max``)."""
    from . import linalg_small as _ls

    if X.dim() == 2 and X.shape[0] <= X.shape[1] and _ls.device_linalg(X, X.shape[0]):
        return _ls.svd_vt_small(X)[0]
    if X.is_cuda and max(X.shape) <= _host_max():
        return torch.linalg.svd(X.cpu(), full_matrices=False)[2].to(X.device, non_blocking=True)
    return torch.linalg.svd(X, full_matrices=False)[2]


def decompose(C, B_prev: Optional[torch.Tensor] = None, method: Optional[str] = None):
    """``(B, D, invsqrtC)`` with ``C = B diag(D²) Bᵀ`` and ``invsqrtC = B diag(1/D) Bᵀ`` from the vendor dense
    symmetric solver behind ``torch.linalg.eigh`` (small problems on a GPU go through the host, see ``_host_max``).

    This is the exact-semantics path of the reference (``algorithms/so/es_variants/cma_es.py:143-166``); it costs
    tens of milliseconds at n = 1000 on the GPU, which is why CMA-ES defaults to :class:`NewtonSchulzSqrt` there.
    Small orders on a GPU run in the one-workgroup Jacobi kernel, warm-started from ``B_prev`` (the previous
    generation's eigenvectors: the covariance moves slowly, one or two sweeps suffice).  ``method`` is accepted for call
    compatibility and ignored."""
    B, evals = eigh_reference(C, B_prev)
    D = torch.sqrt(evals.clamp_min(1e-30))
    invsqrtC = (B / D) @ B.T
    return B, D, invsqrtC


# --------------------------------------------------------------------------------------
# matrix square roots without an eigendecomposition
# --------------------------------------------------------------------------------------
class NewtonSchulzSqrt:
    """``C^{1/2}`` and ``C^{-1/2}`` of a symmetric positive definite matrix by the coupled
    Newton–Schulz iteration (GEMMs only, quadratically convergent)::

        Y₀ = C / s,  Z₀ = I,   T = (3I − Z Y) / 2,   Y ← Y T,   Z ← T Z
        Y → (C/s)^{1/2},  Z → (C/s)^{-1/2}          (s ≈ λmax)

    The object holds the policy (tolerance, iteration bounds); everything that changes from call to call —
    the iteration count, the safe-scaling countdown, the power-iteration vector and the residual records of
    the last two calls — is an explicit *control* dictionary that the caller carries (CMA-ES keeps it in its
    ``State``, so checkpoints, replicas on several ranks and resumed runs all see the same values).

    The iteration count follows the conditioning of ``C`` without stalling the GPU and without depending on
    host timing: the call of generation g reads the residual that generation g − 2 left behind (copied to
    pinned memory when it was produced; by now that copy has long completed, the wait on its event is a
    formality that makes the decision a pure function of the run's history).

    ``planes``: 0 runs the products in fp32 through the vendor library; 2 or 3 runs them on the bf16 matrix
    cores from that many split planes per operand (``ops/cma_bf16.py``), the form CMA-ES uses with bf16 sampling.
    """

    def __init__(self, tol: float | None = None, min_iters: int = 2, max_iters: int = 40, start_iters: int = 6,
                 bf16_copy: bool = False, planes: int = 0, initial_cond: float | None = None):
        # Two bf16 planes carry 16 significant bits: the residual of such an iteration bottoms out near 1e-5, and
        # its consumers (a sampling factor rounded to bf16, relative error 4e-3; the whitening of the path update)
        # do not see a residual of a few 1e-4.
        self.tol = (5e-4 if planes == 2 else 2e-6) if tol is None else tol
        if planes:
            min_iters = min(min_iters, 1)     # the fifth-order first step alone can meet the tolerance when C is near a multiple of I
        self.min_iters, self.max_iters, self.start_iters = min_iters, max_iters, start_iters
        self.bf16_copy = bf16_copy        # also emit the square root rounded to bf16 (fp32 library driver)
        self.planes = int(planes)
        # condition number the caller vouches for during the first two calls, before any estimate exists (CMA-ES starts
        # from C = I); None: unknown, the first calls take the fp32 library products
        self.initial_cond = initial_cond
        self.last_residual = None
        self.fallbacks = 0                # times the guarded outputs passed the previous factors on
        self._eye = None
        self._ws = None                   # scratch of the fp32 library driver
        self._pinned = {}                 # id(record tensor) -> (weakref, event, pinned copy)
        self._ring, self._ring_next = [], 0
        self._graph_scal = None
        self._scratch = None
        self.symmetric_calls = 0          # calls served by the symmetric bf16 driver
        self._ctrl = None                 # control of the stand-alone use ``ns(C)``

    # -- legacy attribute: tests and tools pin the iteration count through it --------------------------
    @property
    def iters(self):
        return self.start_iters if self._ctrl is None else self._ctrl["ns_iters"]

    @iters.setter
    def iters(self, v):
        self.start_iters = int(v)
        if self._ctrl is not None:
            self._ctrl["ns_iters"] = int(v)

    # -- control ---------------------------------------------------------------------------------------
    # record of one call (7 floats, ``scal[9:16]`` of the native drivers):
    #   0 cond estimate from below, 1 cond bound from above, 2 1/√s, 3 √s, 4 scale s ≈ λmax, 5 residual, 6 failed
    REC = 7
    SYM_COND_MAX = 30.0      # the symmetric-half products are used while the cond bound of two calls ago is below
    # beyond it the plane driver multiplies whole products in its iterations (``full``): with two planes (16 significant
    # bits) up to TWO_PLANE_COND_MAX — measured at n = 1000: C^{-1/2} to 1.8e-4 at cond 1e3, 3.6e-3 at 1e4 —, with three
    # planes (24 bits) up to PLANE_COND_MAX (3.2e-4 at 1e4, 6.7e-3 at 1e5); beyond that the fp32 products of the vendor
    # library (``tools/ns_full_check.py``)
    TWO_PLANE_COND_MAX = 2.0e3
    PLANE_COND_MAX = float(os.environ.get("EVOX_AMD_NS_PLANE_COND_MAX", "2e4"))
    COND_EVERY = 4           # calls between refreshes of the condition estimates
    COND_TRUST = 3.0         # the estimate from below times this factor may stand in for a looser bound from above

    def init_control(self, n: int, device) -> dict:
        g = torch.Generator().manual_seed(0x5EED)
        v = torch.randn(2, n, generator=g)
        zero = torch.zeros(self.REC, dtype=torch.float32, device=device)
        return {"ns_iters": int(self.start_iters), "ns_safe": 0, "ns_calls": 0, "ns_used1": 0, "ns_used2": 0, "ns_sym": 0,
                "ns_vec": (v / v.norm(dim=1, keepdim=True)).to(device), "ns_res1": zero, "ns_res2": zero.clone()}

    # ns_sym: which GPU driver the last call used (1: symmetric bf16 planes); a recorded step keeps using it
    CONTROL_FIELDS = ("ns_iters", "ns_safe", "ns_calls", "ns_used1", "ns_used2", "ns_sym", "ns_vec", "ns_res1", "ns_res2")

    def _policy(self, iters: int, safe: int, r: float, failed: bool, used: int):
        self.last_residual = r
        if failed or not (r == r) or r >= 0.25:
            # the estimate-scaled iteration did not contract: back to the provably convergent scaling for a while
            self.fallbacks += int(failed)
            safe = 64
        # convergence is quadratic: one more iteration squares the residual, one fewer takes its root — raise by
        # as many as the tolerance needs, lower only when the root still passes
        if not (r == r) or r > self.tol:
            more = 1 if r * r < self.tol else (2 if r < 0.19 else 4)
            iters = min(self.max_iters, max(iters, used) + (more if r == r else 4))
        elif r ** 0.5 < 0.25 * self.tol and used <= iters:
            iters = max(self.min_iters, used - 1)
        return max(self.min_iters, min(self.max_iters, iters)), safe

    def _record_value(self, rec: torch.Tensor):
        """The record of a call on the host (list of ``REC`` floats)."""
        ent = self._pinned.pop(id(rec), None)
        if ent is not None and ent[0]() is rec:
            _, ev, host = ent
            if ev is not None:
                ev.synchronize()
            return [float(x) for x in host]
        return [float(x) for x in rec.detach().cpu()]

    def pinned_slot(self):
        """``(host tensor, event, device-visible address)`` of the next slot of the pinned record ring: a native driver
        writes its record there directly (no copy command) and hands the slot back through :meth:`adopt`."""
        import ctypes

        if len(self._ring) < 4:
            host = torch.empty(self.REC, dtype=torch.float32, pin_memory=True)
            dev_ptr = ctypes.c_void_p()
            from .. import _native
            _native.register_hip_api({"evox_hip_host_device_ptr": (_native.I, [_native.P, _native.P])})
            check(hip().evox_hip_host_device_ptr(host.data_ptr(), ctypes.byref(dev_ptr)), "hipHostGetDevicePointer")
            self._ring.append((host, torch.cuda.Event(), dev_ptr.value))
        slot = self._ring[self._ring_next]
        self._ring_next = (self._ring_next + 1) % 4
        return slot

    def adopt(self, rec: torch.Tensor, slot) -> None:
        """``rec``'s values are (being) written into ``slot`` by work already issued on the current stream."""
        import weakref

        host, ev, _ = slot
        ev.record()
        self._pinned[id(rec)] = (weakref.ref(rec), ev, host)
        self._adopted = id(rec)

    def _publish(self, rec: torch.Tensor):
        """Start the copy of a fresh record to pinned memory (read two calls later)."""
        import weakref

        if len(self._pinned) > 8:          # records of abandoned states
            self._pinned = {k: e for k, e in self._pinned.items() if e[0]() is not None}
        if getattr(self, "_adopted", None) == id(rec) and id(rec) in self._pinned:
            return                             # the native driver has written this record into its slot already
        if rec.is_cuda:
            host, ev, _ = self.pinned_slot()
            host.copy_(rec, non_blocking=True)
            ev.record()
            self._pinned[id(rec)] = (weakref.ref(rec), ev, host)

    def needs_more_iterations(self) -> bool:
        """For recorded (graph-replayed) use: read the residual the last replay left on the
        device (one synchronisation) and raise the iteration count if it misses the tolerance."""
        if self._graph_scal is None:
            return False
        r = float(self._graph_scal[5])
        self.last_residual = r
        if not (r == r) or r > self.tol:
            self.start_iters = min(self.max_iters, self.iters + (4 if (r != r or r > 1e-2) else 2))
            return True
        return False

    # -- drivers ---------------------------------------------------------------------------------------
    def _library_ok(self, C) -> bool:
        if not (C.is_cuda and C.dtype == torch.float32 and os.environ.get("EVOX_AMD_NS", "native") == "native"):
            return False
        from . import blas as _blas
        return _blas.available()

    def _torch_iteration(self, C, iters):
        n = C.shape[0]
        eye = self._eye
        if eye is None or eye.shape[0] != n or eye.device != C.device or eye.dtype != C.dtype:
            eye = self._eye = torch.eye(n, device=C.device, dtype=C.dtype)
        # scale towards the centre of the spectrum when that is provably safe (λmax < 2s needs
        # ‖C‖∞ < 2s), else by the norm bound: a centred spectrum saves iterations
        s = torch.maximum(torch.diagonal(C).mean(), C.abs().sum(1).max() / 1.9)
        Y = C / s
        # first iteration with Z₀ = I written out: T = (3I − Y)/2, Z₁ = T (saves two GEMMs)
        T = torch.add(eye, Y, alpha=-0.5).add_(eye, alpha=0.5) if iters > 0 else eye
        Z = T
        if iters > 0:
            Y = Y @ T
        # the multiplication ORDER below is the numerically stable coupled form
        for _ in range(iters - 1):
            T = torch.addmm(eye, Z, Y, beta=1.5, alpha=-0.5)
            Y = Y @ T
            Z = T @ Z
        # I − Z Y evaluated before the last update equals 2 (I − T): one iteration later the
        # residual is its square (quadratic convergence), so this bounds the final error
        res = 2.0 * torch.linalg.matrix_norm(T - eye) / (n ** 0.5)
        res = res * res
        rs = torch.sqrt(s)
        z = torch.zeros((), dtype=torch.float32, device=C.device)
        rec = torch.stack([z, z, (1 / rs).to(torch.float32), rs.to(torch.float32), s.to(torch.float32),
                           res.to(torch.float32), z])
        return 0.5 * (Y + Y.T) * rs, 0.5 * (Z + Z.T) / rs, rec

    def _gpu_scratch(self, n, device, ws):
        """Scalar blocks and power vectors shared by the two GPU drivers: (scal, scal2, vec4) with vec4 laid out as
        [v | scratch | v2 | scratch2]."""
        sc = self._scratch
        if sc is None or sc[2].numel() != 4 * n or sc[2].device != device:
            from . import blas as _blas
            k = int(hip().evox_hip_ns_scal_floats())
            sc = self._scratch = (torch.zeros(k, dtype=torch.float32, device=device),
                                  torch.zeros(k, dtype=torch.float32, device=device),
                                  torch.zeros(4 * n, dtype=torch.float32, device=device))
        return sc

    def plan(self, ctrl: dict, capturing: bool, gpu: bool, have_ws: bool) -> dict:
        """Host decisions of one factorisation from the control (and the record of two calls ago): iteration count,
        scaling mode, which GPU driver, power steps, whether the condition estimates are refreshed."""
        iters, safe = int(ctrl["ns_iters"]), int(ctrl["ns_safe"])
        calls = int(ctrl["ns_calls"])
        lagged = None
        if not capturing and calls >= 2 and int(ctrl["ns_used2"]) > 0:
            lagged = self._record_value(ctrl["ns_res2"])
            iters, safe = self._policy(iters, safe, lagged[5], lagged[6] != 0.0, int(ctrl["ns_used2"]))
        use_safe = safe > 0
        if use_safe and not capturing:
            safe -= 1
        full, nplanes = False, self.planes
        if capturing:
            symmetric = bool(int(ctrl.get("ns_sym", 0))) and bool(self.planes and have_ws and gpu)
            full = symmetric and int(ctrl.get("ns_sym", 0)) >= 2
            nplanes = 3 if int(ctrl.get("ns_sym", 0)) == 3 else self.planes
        else:
            # The bound from above (an ∞-norm) overshoots by up to √n when the eigenvectors are far from the axes;
            # the estimate from below (a power vector carried from call to call) is tight after a few refreshes.
            # Three times the estimate is trusted when it is the smaller one — a wrong decision shows up as a
            # residual over tolerance and sends the next calls through the safe scaling and the library products.
            cond = 0.0
            if lagged is None and calls < 2 and self.initial_cond is not None:
                cond = float(self.initial_cond)
            if lagged is not None and lagged[1] > 0.0:
                cond = lagged[1] if not lagged[0] > 0.0 else min(lagged[1], self.COND_TRUST * lagged[0])
            planes_ok = bool(self.planes and have_ws and gpu)
            symmetric = planes_ok and 0.0 < cond <= self.SYM_COND_MAX and not use_safe
            if planes_ok and not symmetric and (cond <= 0.0 or cond <= self.PLANE_COND_MAX) and (calls >= 2 or self.initial_cond is not None):
                # own kernels off the best case too: the plane driver with plain products (also under the safe scaling)
                symmetric = full = True
                # an unknown conditioning (no bound yet, or the safe scaling after a failed call) takes the wide form
                nplanes = self.planes if (0.0 < cond <= self.TWO_PLANE_COND_MAX and not use_safe) else 3
        self.symmetric_calls += int(symmetric and not full)
        self.full_plane_calls = getattr(self, "full_plane_calls", 0) + int(full)
        used = max(1, iters)
        return {"iters": iters, "safe": safe, "use_safe": use_safe, "symmetric": symmetric, "full": full, "planes": nplanes, "used": used,
                "calls": calls,
                "power": 8 if calls == 0 else 1,
                "cond": bool(self.planes and gpu and (calls % self.COND_EVERY == 0 or capturing))}

    def commit(self, ctrl: dict, plan: dict, rec: torch.Tensor, new_vec: torch.Tensor, capturing: bool) -> dict:
        """The control after the call whose record is ``rec``."""
        if not capturing:
            self._publish(rec)
        if capturing:
            # a recording freezes the host values: only the call counter moves (the replay advances it on the host)
            return dict(ctrl, ns_calls=plan["calls"] + 1, ns_vec=new_vec, ns_res1=rec, ns_res2=ctrl["ns_res1"])
        return {"ns_iters": plan["iters"], "ns_safe": plan["safe"], "ns_calls": plan["calls"] + 1, "ns_used1": plan["used"],
                "ns_used2": int(ctrl["ns_used1"]), "ns_sym": int(plan["symmetric"]) + int(plan.get("full", False)) + int(plan.get("full", False) and plan.get("planes") == 3),
                "ns_vec": new_vec,
                "ns_res1": rec,
                "ns_res2": ctrl["ns_res1"]}

    def library_workspace(self, n: int, device):
        from . import blas as _blas
        if self._ws is None or self._ws.n != n or self._ws.work.device != device:
            self._ws = _blas.NSWorkspace(n, device)
        return self._ws

    def step(self, C: torch.Tensor, ctrl: dict, prev=None, ws=None):
        """One factorisation.  Returns ``(sqrtC, invsqrtC, sqrtC_bf16 | None, new control)``.
        ``prev``: the previous ``(sqrtC, invsqrtC)``, passed on by the GPU drivers when an iteration fails to
        contract.  ``ws``: the ``ops.cma_bf16.Workspace`` of the caller (required when ``planes`` > 0).

        With ``planes`` > 0 two GPU drivers exist.  The symmetric one multiplies only the upper triangle of every
        product on the bf16 matrix cores and mirrors it; forcing the iterates symmetric changes the first-order
        error propagation of the coupled iteration, which then amplifies rounding errors by about (√cond − 1)/2 per
        step, so it is used while the condition bound of the covariance (from the factorisation of two calls ago)
        is below ``SYM_COND_MAX``.  Beyond that the same kernels multiply whole products in the iterations (every
        iterate kept in both orientations; ``csrc/blas.hip::ns_sqrt_planes``) up to ``PLANE_COND_MAX``, and only
        beyond that the full fp32 products of the vendor library run."""
        n = C.shape[0]
        capturing = C.is_cuda and torch.cuda.is_current_stream_capturing()
        gpu_native = C.is_cuda and C.dtype == torch.float32 and os.environ.get("EVOX_AMD_NS", "native") == "native"
        pl = self.plan(ctrl, capturing, gpu_native, ws is not None)
        b16 = None
        if gpu_native and (pl["symmetric"] or self._library_ok(C)):
            scal, scal2, vec4 = self._gpu_scratch(n, C.device, ws)
            vec4.view(2, 2 * n)[:, :n].copy_(ctrl["ns_vec"])
            if pl["symmetric"]:
                from . import cma_bf16 as _b
                sq, inv, b16 = _b.ns_sqrt(C, ws, pl["used"], scal, vec4, power_steps=pl["power"], safe=pl["use_safe"], prev=prev,
                                          full=pl["full"], planes=pl["planes"])
            else:
                from . import blas as _blas
                w = self.library_workspace(n, C.device)
                w.fresh, w.vec, w.scal = False, vec4, scal
                sq, inv = _blas.ns_sqrt(C, w, pl["used"], power_steps=pl["power"], safe=pl["use_safe"], prev=prev,
                                        bf16_copy=self.bf16_copy)
            if pl["cond"]:
                # the condition estimates move slowly and are read with a lag anyway: refreshed every few calls
                from . import blas as _blas
                _blas.ns_cond(inv, vec4[2 * n:], scal, scal2)
            elif self.planes:
                scal[9:11].copy_(ctrl["ns_res1"][:2])
            rec = scal[9:16].clone()
            new_vec = vec4.view(2, 2 * n)[:, :n].clone()
            if capturing:
                self._graph_scal = scal[9:16]
        else:
            sq, inv, rec = self._torch_iteration(C, pl["used"])
            new_vec = ctrl["ns_vec"]
        return sq, inv, b16, self.commit(ctrl, pl, rec, new_vec, capturing)

    def __call__(self, C: torch.Tensor, prev=None):
        """Stand-alone use: the control is kept on the object."""
        if self._ctrl is None or self._ctrl["ns_vec"].shape[1] != C.shape[0] or self._ctrl["ns_vec"].device != C.device:
            self._ctrl = self.init_control(C.shape[0], C.device)
        sq, inv, b16, self._ctrl = self.step(C, self._ctrl, prev)
        if b16 is not None:
            sq._evox_bf16 = b16
        if C.is_cuda and not torch.cuda.is_current_stream_capturing():
            # stand-alone callers read ``last_residual`` right away
            self.last_residual = self._record_value(self._ctrl["ns_res1"])[5]
        return sq, inv
