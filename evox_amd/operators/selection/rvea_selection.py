"""Reference-vector-guided selection (Cheng et al. 2016, RVEA).  Behaviour of the reference's
``src/evox/operators/selection/rvea_selection.py:8-54``: rows of the result that belong to a
reference vector without associated solutions are NaN."""
from __future__ import annotations

import torch

from ... import _native
from ...ops._util import check as _check
from ...ops._util import hip as _hip
from ...ops._util import stream_of as _stream_of
from ...utils.common import cos_dist

_native.register_hip_api({"evox_hip_rvea_select": (_native.I, [_native.P, _native.I, _native.P, _native.I, _native.I, _native.F,
                                                              _native.P, _native.P, _native.P]),
                          "evox_hip_rvea_select_dev": (_native.I, [_native.P, _native.I, _native.P, _native.I, _native.I,
                                                                  _native.P, _native.P, _native.P, _native.P])})


def _fused_ok(f, v) -> bool:
    return f.is_cuda and f.dtype == torch.float32 and v.dtype == torch.float32 and 2 <= f.shape[1] <= 8


def _select_fused(x, f, obj, v, theta):
    """Association, APD and the per-vector arg-min in two streaming launches (``csrc/rvea.hip``)."""
    n, m = f.shape
    nv = v.shape[0]
    vn = (v / torch.linalg.vector_norm(v, dim=-1, keepdim=True)).contiguous()
    gamma = torch.empty(nv, dtype=torch.float32, device=f.device)
    best = torch.full((nv,), -1, dtype=torch.int64, device=f.device)          # all ones = no solution yet
    if isinstance(theta, torch.Tensor):
        # θ on the device (one float, e.g. a function of a device-side generation counter): no host value enters the launch
        th = theta.reshape(1).to(device=f.device, dtype=torch.float32)
        _check(_hip().evox_hip_rvea_select_dev(obj.contiguous().data_ptr(), n, vn.data_ptr(), nv, m, th.data_ptr(),
                                               gamma.data_ptr(), best.data_ptr(), _stream_of(f)), "rvea_select")
    else:
        _check(_hip().evox_hip_rvea_select(obj.contiguous().data_ptr(), n, vn.data_ptr(), nv, m, float(theta), gamma.data_ptr(),
                                           best.data_ptr(), _stream_of(f)), "rvea_select")
    empty = best == -1
    next_ind = torch.where(empty, torch.zeros_like(best), best & 0xFFFFFFFF)
    nan = f.new_full((), float("nan"))
    return torch.where(empty[:, None], nan, x[next_ind]), torch.where(empty[:, None], nan, f[next_ind])


def ref_vec_guided(x, f, v, theta):
    n, m = f.shape
    nv = v.shape[0]
    fmin = torch.where(torch.isnan(f), torch.full_like(f, float("inf")), f).amin(0)
    obj = (f - fmin).clamp_min(1e-32)
    if _fused_ok(f, v):
        return _select_fused(x, f, obj, v, theta)
    return _ref_vec_guided_dense(x, f, obj, v, theta)


def _ref_vec_guided_dense(x, f, obj, v, theta):
    """The same selection with the n × nv angle and APD matrices."""
    n, m = f.shape
    nv = v.shape[0]
    cosine = cos_dist(v, v)
    cosine = cosine.masked_fill(torch.eye(nv, dtype=torch.bool, device=v.device), 0.0).clamp(0, 1)
    gamma = torch.acos(cosine).amin(1)                                  # (nv,)
    angle = torch.acos(cos_dist(obj, v).clamp(0, 1))                    # (n, nv)
    nan_row = torch.isnan(obj).any(1)
    angle_for_assoc = torch.where(torch.isnan(angle), torch.full_like(angle, float("inf")), angle)
    associate = torch.argmin(angle_for_assoc, dim=1)
    associate = torch.where(nan_row, torch.full_like(associate, -1), associate)
    member = associate[:, None] == torch.arange(nv, device=f.device)[None, :]     # (n, nv)
    norm = torch.sqrt((obj * obj).sum(1))                                          # (n,)
    apd = (1 + m * theta * angle / gamma[None, :]) * norm[:, None]
    apd = torch.where(member, apd, torch.full_like(apd, float("inf")))
    apd = torch.where(torch.isnan(apd), torch.full_like(apd, float("inf")), apd)
    next_ind = torch.argmin(apd, dim=0)                                            # (nv,)
    empty = ~member.any(0)
    nan = f.new_full((), float("nan"))
    next_x = torch.where(empty[:, None], nan, x[next_ind])
    next_f = torch.where(empty[:, None], nan, f[next_ind])
    return next_x, next_f


class ReferenceVectorGuided:
    """Reference vector guided environmental selection."""

    def __call__(self, x, f, v, theta):
        return ref_vec_guided(x, f, v, theta)
