from __future__ import annotations

import math
from typing import Callable, Sequence

import torch

from ..core import random as rnd

_ACT = {"tanh": torch.tanh, "relu": torch.relu, "identity": lambda x: x, "sigmoid": torch.sigmoid,
        "gelu": torch.nn.functional.gelu}


def mlp_param_count(sizes: Sequence[int], bias: bool = True) -> int:
    return sum(i * o + (o if bias else 0) for i, o in zip(sizes[:-1], sizes[1:]))


def batched_mlp_forward(flat: torch.Tensor, obs: torch.Tensor, sizes: Sequence[int], activation="tanh",
                        final_activation="tanh", bias: bool = True, compute_dtype=None) -> torch.Tensor:
    """Per-individual MLP forward.

    flat : (P, n_params) parameters, layer after layer, each ``[W (in×out, row-major), b (out)]``
    obs  : (P, E, in) – E observations per individual (or (P, in))
    →      (P, E, out) (or (P, out))

    Each layer is one strided batched GEMM reading the weights in place (``as_strided`` view of
    ``flat``), so the only HBM traffic is one pass over the parameters per call: the op is
    bandwidth-bound at ≈ bytes(flat) / HBM rate, which bf16 storage halves."""
    squeeze = obs.dim() == 2
    h = obs[:, None, :] if squeeze else obs
    P = flat.shape[0]
    if _small_ok(flat, h, sizes, activation, final_activation):
        out = _forward_small(flat, h, sizes, activation, final_activation, bias)
        return out[:, 0, :] if squeeze else out
    if _hip_ok(flat, h, sizes, activation, final_activation):
        out = _forward_hip(flat, h, sizes, activation, final_activation, bias)
        return out[:, 0, :] if squeeze else out
    act = _ACT[activation] if isinstance(activation, str) else activation
    fin = _ACT[final_activation] if isinstance(final_activation, str) else final_activation
    dt = compute_dtype or flat.dtype
    h = h.to(dt)
    off = 0
    n_layers = len(sizes) - 1
    stride0 = flat.stride(0)
    for li, (i, o) in enumerate(zip(sizes[:-1], sizes[1:])):
        W = flat.as_strided((P, i, o), (stride0, o, 1), flat.storage_offset() + off)
        off += i * o
        if bias:
            b = flat.as_strided((P, 1, o), (stride0, o, 1), flat.storage_offset() + off)
            off += o
            h = torch.baddbmm(b.to(dt), h, W.to(dt))
        else:
            h = torch.bmm(h, W.to(dt))
        h = fin(h) if li == n_layers - 1 else act(h)
    return h[:, 0, :] if squeeze else h


_ACT_CODE = {"identity": 0, "tanh": 1, "relu": 2, "sigmoid": 3}


def _small_ok(flat, obs, sizes, activation, final_activation) -> bool:
    """Small networks (every width <= 64, e.g. the 32-unit policies of locomotion tasks) take ``mlp_small_kernel``: one
    lane per individual on a transposed copy of the population (``EVOX_AMD_MLP_SMALL=0`` switches it off, ``=1`` forces
    it wherever it can run).  Where it wins was measured in rollouts on an MI355X (`tools/mlp_small_timing.py`,
    `tools/planar_es_timing.py`): from 8192 (individual, observation) pairs on always — the one-workgroup-per-pair kernel
    takes 55 us there against 19-24 us, 213 against 38-56 us at 65 536 —; below that only for 32-unit networks whose
    population fits the L2 (28 MB) and so stays there between the time steps of a rollout (hopper, 4096 policies: 3.02 ->
    2.81 ms per generation; the ant's 36 MB: 11.1 -> 12.3 ms, so it keeps the other kernel)."""
    import os

    mode = os.environ.get("EVOX_AMD_MLP_SMALL", "auto")
    if not (flat.is_cuda and obs.is_cuda and flat.dim() == 2 and flat.dtype in (torch.float32, torch.bfloat16)
            and max(sizes) <= 64 and 1 <= len(sizes) - 1 <= 8
            and flat.shape[0] * flat.shape[1] < 2 ** 31 and flat.shape[0] * obs.shape[1] < 2 ** 31
            and isinstance(activation, str) and isinstance(final_activation, str)
            and activation in _ACT_CODE and final_activation in _ACT_CODE and mode != "0"):
        return False
    pairs = flat.shape[0] * obs.shape[1]
    if mode == "1":
        return pairs >= 64
    return pairs >= 8192 or (pairs >= 1024 and max(sizes) <= 32 and flat.shape[0] * flat.shape[1] * 4 <= 28 * 2 ** 20)


def _forward_small(flat, obs, sizes, activation, final_activation, bias):
    """``csrc/mlp.hip::mlp_small_kernel``.  The kernel reads the population transposed — (n_params, P), so that the lanes
    of a wavefront are neighbouring individuals and every weight load is coalesced.  The transposed copy is made once
    per population tensor (a rollout calls the policy once per time step with the same tensor) and kept on it."""
    import ctypes
    import os

    from .. import _native
    from ..ops._util import check, hip, stream_of
    P_, I_, I64_ = _native.P, _native.I, _native.I64
    _native.register_hip_api({"evox_hip_mlp_small": (I_, [P_, I64_, P_, P_, P_, I_, I_, P_, P_, I_, I_, P_])})
    P, E = flat.shape[0], obs.shape[1]
    n_layers = len(sizes) - 1
    # a mirrored population (marked by ``ops.es.perturb``) is evaluated pair-wise from its first half and the centre
    tag = getattr(flat, "_evox_mirror", None)
    center = None
    # (pairs halve the bytes read but also the lanes at work: they pay from 32 768 pairs on — 75 -> 38 us at 65 536)
    mirror_mode = os.environ.get("EVOX_AMD_MLP_MIRROR", "auto")
    if tag is not None and mirror_mode != "0" and (mirror_mode == "1" or P * E >= 32768):
        c, half = tag
        if (P == 2 * half and c.dtype == torch.float32 and c.is_contiguous() and c.device == flat.device
                and c.numel() == flat.shape[1]):
            center = c
    cached = getattr(flat, "_evox_wt", None)
    if cached is None or cached[0] != flat._version or cached[2] != (center is not None):
        rows = flat[: P // 2] if center is not None else flat
        wt = rows.detach().to(torch.float32).t().contiguous()
        flat._evox_wt = cached = (flat._version, wt, center is not None)
    wt = cached[1]
    x = obs.to(torch.float32).contiguous()
    out = torch.empty((P, E, sizes[-1]), dtype=torch.float32, device=flat.device)
    c_sizes = (ctypes.c_int * (n_layers + 1))(*[int(s) for s in sizes])
    acts = [_ACT_CODE[activation]] * (n_layers - 1) + [_ACT_CODE[final_activation]]
    c_acts = (ctypes.c_int * n_layers)(*acts)
    check(hip().evox_hip_mlp_small(wt.data_ptr(), wt.stride(0), center.data_ptr() if center is not None else None, x.data_ptr(),
                                   out.data_ptr(), P, E, ctypes.addressof(c_sizes), ctypes.addressof(c_acts), n_layers,
                                   int(bool(bias)), stream_of(flat)), "mlp_small")
    return out


def _hip_ok(flat, obs, sizes, activation, final_activation) -> bool:
    return (flat.is_cuda and obs.is_cuda and flat.dtype in (torch.float32, torch.bfloat16) and flat.stride(1) == 1
            and obs.shape[1] in (1, 2, 4) and isinstance(activation, str) and isinstance(final_activation, str)
            and activation in _ACT_CODE and final_activation in _ACT_CODE and 1 <= len(sizes) - 1 <= 8
            and obs.shape[1] * (2 * ((max(sizes) + 3) // 4 * 4) + 256) * 4 <= 64 * 1024)


def _forward_hip(flat, obs, sizes, activation, final_activation, bias):
    """Fused kernel (``csrc/mlp.hip``): one workgroup per individual streams its parameters once."""
    import ctypes

    from .. import _native
    from ..ops._util import check, hip, stream_of
    P_, I_, I64_ = _native.P, _native.I, _native.I64
    _native.register_hip_api({"evox_hip_mlp_forward": (I_, [P_, I_, I64_, P_, P_, I_, I_, P_, P_, I_, I_, P_]),
                              "evox_hip_mlp_forward_mirrored": (I_, [P_, I_, I64_, P_, I_, P_, P_, I_, I_, P_, P_, I_, I_, P_])})
    P, E = flat.shape[0], obs.shape[1]
    n_layers = len(sizes) - 1
    x = obs.to(torch.float32).contiguous()
    out = torch.empty((P, E, sizes[-1]), dtype=torch.float32, device=flat.device)
    c_sizes = (ctypes.c_int * (n_layers + 1))(*[int(s) for s in sizes])
    acts = [_ACT_CODE[activation]] * (n_layers - 1) + [_ACT_CODE[final_activation]]
    c_acts = (ctypes.c_int * n_layers)(*acts)
    mirror = _mirror_of(flat, E, sizes)
    if mirror is not None:
        # mirrored population: each workgroup evaluates a (+, −) pair from one pass over the + row
        if flat.dtype == torch.bfloat16:
            # bf16 rows: a bf16 copy of the centre (made once per population) halves the centre traffic;
            # 2·bf16(c) − bf16(w₊) is as close to c − σz as a stored bf16 row would be
            c16 = getattr(flat, "_evox_mirror_c16", None)
            if c16 is None:
                c16 = mirror.to(torch.bfloat16)
                flat._evox_mirror_c16 = c16
            mirror = c16
        check(hip().evox_hip_mlp_forward_mirrored(
            flat.data_ptr(), 1 if flat.dtype == torch.bfloat16 else 0, flat.stride(0), mirror.data_ptr(),
            1 if mirror.dtype == torch.bfloat16 else 0, x.data_ptr(),
            out.data_ptr(), P, E, ctypes.addressof(c_sizes), ctypes.addressof(c_acts), n_layers, int(bool(bias)),
            stream_of(flat)), "mlp_forward_mirrored")
        return out
    check(hip().evox_hip_mlp_forward(flat.data_ptr(), 1 if flat.dtype == torch.bfloat16 else 0, flat.stride(0), x.data_ptr(),
                                     out.data_ptr(), P, E, ctypes.addressof(c_sizes), ctypes.addressof(c_acts), n_layers,
                                     int(bool(bias)), stream_of(flat)), "mlp_forward")
    return out


def _mirror_of(flat, E, sizes):
    """The centre of a mirrored population (``ops.es.perturb`` marks the tensors it returns) when the
    pair form of the kernel applies to it, else None."""
    import os

    tag = getattr(flat, "_evox_mirror", None)
    if tag is None or os.environ.get("EVOX_AMD_MLP_MIRROR", "1") == "0":
        return None
    center, half = tag
    if (flat.shape[0] != 2 * half or E not in (1, 2) or center.dtype != torch.float32 or not center.is_contiguous()
            or center.device != flat.device or center.numel() != flat.shape[1]
            or 2 * E * (2 * ((max(sizes) + 3) // 4 * 4) + 256) * 4 > 64 * 1024):
        return None
    return center


class MLP:
    """Functional MLP over flat parameter vectors.

    >>> net = MLP([obs_dim, 64, 64, act_dim])
    >>> center = net.init(key)                      # (n_params,)
    >>> actions = net.batched_apply(pop, obs)       # pop (P, n_params), obs (P, E, obs_dim)
    """

    def __init__(self, sizes: Sequence[int], activation="tanh", final_activation="tanh", bias=True):
        self.sizes = tuple(int(s) for s in sizes)
        self.activation = activation
        self.final_activation = final_activation
        self.bias = bias
        self.n_params = mlp_param_count(self.sizes, bias)

    def init(self, key, device=None, dtype=torch.float32) -> torch.Tensor:
        """LeCun-normal weights, zero biases."""
        parts = []
        for li, (i, o) in enumerate(zip(self.sizes[:-1], self.sizes[1:])):
            w = rnd.normal(rnd.fold_in(key, li), (i * o,)) / math.sqrt(i)
            parts.append(w)
            if self.bias:
                parts.append(torch.zeros(o, device=w.device))
        v = torch.cat(parts).to(dtype)
        return v.to(device) if device is not None else v

    def apply(self, params: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """One parameter vector, x (..., in)."""
        lead = x.shape[:-1]
        out = batched_mlp_forward(params[None, :], x.reshape(1, -1, x.shape[-1]), self.sizes, self.activation,
                                  self.final_activation, self.bias)
        return out.reshape(*lead, self.sizes[-1])

    def batched_apply(self, pop: torch.Tensor, obs: torch.Tensor) -> torch.Tensor:
        return batched_mlp_forward(pop, obs, self.sizes, self.activation, self.final_activation, self.bias)

    __call__ = batched_apply

    def to_tree(self, params: torch.Tensor):
        """Split a flat vector (or a (P, n) population) into ``[{"w", "b"}, …]`` views."""
        out, off = [], 0
        lead = params.shape[:-1]
        for i, o in zip(self.sizes[:-1], self.sizes[1:]):
            layer = {"w": params[..., off:off + i * o].reshape(*lead, i, o)}
            off += i * o
            if self.bias:
                layer["b"] = params[..., off:off + o]
                off += o
            out.append(layer)
        return out
