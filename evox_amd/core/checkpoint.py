"""Checkpoint / resume of a State tree.

Directory format (default)::

    <path>/manifest.json      tree layout: node ids, child names, field names, dataclass type
                              names, static fields, non-tensor leaves (JSON values)
    <path>/tensors.safetensors  every tensor leaf keyed by its '/'-separated tree path

The PRNG keys live inside the state, so a resumed run continues the same random streams.
Writes go to a temporary sibling first and are renamed into place, so a crash never leaves a
half-written checkpoint under the final name.  Loading executes nothing from the file
(safetensors + JSON).  Reference behaviour: ``src/evox/core/state.py:264-301``.

Several ranks (``save_state_sharded`` / ``load_state_sharded``): rank 0 writes the manifest and the replicated
tensors once; every rank writes the row-local tensors it owns (the fields a workflow declares as sharded along
dim 0) to ``rank00000.safetensors``, ``rank00001.safetensors``, ….  All ranks write into one temporary directory,
meet at a barrier, and rank 0 renames it into place.  A checkpoint written by W ranks loads on W' ranks: the
row-local tensors are concatenated in rank order and cut again for the loading job.
"""
from __future__ import annotations

import dataclasses
import json
import os
import shutil
from pathlib import Path
from typing import Any, Optional

import torch

from . import tree as _tree

FORMAT_VERSION = 1


def _encode(obj, tensors: dict, path: str):
    from .state import State

    if isinstance(obj, State):
        return {
            "__kind__": "state",
            "id": obj._state_id,
            "fields": _encode(obj._state_dict, tensors, f"{path}/fields" if path else "fields"),
            "children": {k: _encode(v, tensors, f"{path}/children/{k}" if path else f"children/{k}")
                         for k, v in obj._child_states.items()},
        }
    if isinstance(obj, torch.Tensor):
        tensors[path] = obj.detach().contiguous().cpu()
        return {"__kind__": "tensor", "key": path}
    if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        cls = type(obj)
        return {
            "__kind__": "dataclass",
            "type": f"{cls.__module__}:{cls.__qualname__}",
            "fields": {f.name: _encode(getattr(obj, f.name), tensors, f"{path}/{f.name}")
                       for f in dataclasses.fields(obj)},
        }
    if isinstance(obj, dict):
        return {"__kind__": "dict", "items": {str(k): _encode(v, tensors, f"{path}/{k}") for k, v in obj.items()}}
    if isinstance(obj, (list, tuple)):
        return {"__kind__": "list" if isinstance(obj, list) else "tuple",
                "items": [_encode(v, tensors, f"{path}/{i}") for i, v in enumerate(obj)]}
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return {"__kind__": "value", "value": obj}
    try:
        import numpy as np

        if isinstance(obj, np.ndarray):
            tensors[path] = torch.from_numpy(np.ascontiguousarray(obj))
            return {"__kind__": "ndarray", "key": path}
        if isinstance(obj, np.generic):
            return {"__kind__": "value", "value": obj.item()}
    except ImportError:  # pragma: no cover
        pass
    raise TypeError(f"cannot checkpoint leaf of type {type(obj)} at {path!r}")


def _resolve_dataclass(name: str):
    import importlib
    import sys

    mod, _, qual = name.partition(":")
    # only classes of modules that the program has already imported and that are registered
    # tree dataclasses are instantiated
    m = sys.modules.get(mod)
    if m is None and mod.startswith("evox_amd"):
        m = importlib.import_module(mod)
    obj = m
    if obj is not None:
        for part in qual.split("."):
            obj = getattr(obj, part, None)
            if obj is None:
                break
    if obj is not None and isinstance(obj, type) and dataclasses.is_dataclass(obj):
        return obj
    return None


def _decode(node, tensors, device):
    from .state import State

    kind = node["__kind__"]
    if kind == "state":
        fields = _decode(node["fields"], tensors, device)
        children = {k: _decode(v, tensors, device) for k, v in node["children"].items()}
        return State._make(fields, children if children else State.EMPTY, node["id"])
    if kind == "tensor":
        t = tensors[node["key"]]
        return t.to(device) if device is not None else t
    if kind == "ndarray":
        return tensors[node["key"]].numpy()
    if kind == "dataclass":
        vals = {k: _decode(v, tensors, device) for k, v in node["fields"].items()}
        cls = _resolve_dataclass(node["type"])
        if cls is None:
            return vals
        obj = object.__new__(cls)
        for k, v in vals.items():
            object.__setattr__(obj, k, v)
        return obj
    if kind == "dict":
        return {k: _decode(v, tensors, device) for k, v in node["items"].items()}
    if kind == "list":
        return [_decode(v, tensors, device) for v in node["items"]]
    if kind == "tuple":
        return tuple(_decode(v, tensors, device) for v in node["items"])
    if kind == "value":
        return node["value"]
    raise ValueError(f"unknown manifest node kind {kind!r}")


def save_state(state, path, directory: bool = True, extra: Optional[dict] = None) -> None:
    path = Path(path).resolve()
    tensors: dict = {}
    manifest = {"format": FORMAT_VERSION, "tree": _encode(state, tensors, ""), "extra": extra or {}}
    tmp = path.with_name(path.name + f".tmp{os.getpid()}")
    if directory:
        from safetensors.torch import save_file

        if tmp.exists():
            shutil.rmtree(tmp)
        tmp.mkdir(parents=True)
        # safetensors refuses aliased storage: every entry gets its own buffer
        save_file({k: v.clone() for k, v in tensors.items()}, str(tmp / "tensors.safetensors"))
        (tmp / "manifest.json").write_text(json.dumps(manifest))
        if path.exists():
            shutil.rmtree(path)
        os.replace(tmp, path)
    else:
        path.parent.mkdir(parents=True, exist_ok=True)
        torch.save({"manifest": manifest, "tensors": tensors}, str(tmp))
        os.replace(tmp, path)


def load_state(path, template=None, directory: bool = True, device=None):
    path = Path(path).resolve()
    if directory:
        from safetensors.torch import load_file

        manifest = json.loads((path / "manifest.json").read_text())
        tensors = load_file(str(path / "tensors.safetensors"))
    else:
        blob = torch.load(str(path), weights_only=True)
        manifest, tensors = blob["manifest"], blob["tensors"]
    if manifest.get("format") != FORMAT_VERSION:
        raise ValueError(f"unsupported checkpoint format {manifest.get('format')}")
    restored = _decode(manifest["tree"], tensors, device)
    if template is None:
        return restored
    # pour the restored leaves into the template's structure; tensors follow the template's
    # device and dtype
    t_leaves, t_def = _tree.tree_flatten(template)
    r_leaves, r_def = _tree.tree_flatten(restored)
    if len(t_leaves) != len(r_leaves):
        raise ValueError(
            f"checkpoint has {len(r_leaves)} leaves but the template has {len(t_leaves)}: "
            "the module tree that wrote it differs from the one loading it"
        )
    out = []
    for t, r in zip(t_leaves, r_leaves):
        if isinstance(t, torch.Tensor) and isinstance(r, torch.Tensor):
            if tuple(t.shape) != tuple(r.shape):
                raise ValueError(f"shape mismatch on restore: {tuple(r.shape)} vs template {tuple(t.shape)}")
            out.append(r.to(device=device if device is not None else t.device, dtype=t.dtype))
        else:
            out.append(r)
    return _tree.tree_unflatten(t_def, out)


def _rank_file(rank: int) -> str:
    return f"rank{rank:05d}.safetensors"


def save_state_sharded(state, path, local_keys, rank: int, world: int, barrier=None, extra: Optional[dict] = None) -> None:
    """Collective: every rank calls it with the same ``path`` (a directory on a file system all ranks see).

    ``local_keys``: '/'-separated tree paths (as in the manifest, e.g. ``children/algorithm/fields/population``) of
    the tensors that hold only this rank's rows; everything else is taken to be replicated and written by rank 0."""
    from safetensors.torch import save_file

    path = Path(path).resolve()
    local_keys = set(local_keys)
    tensors: dict = {}
    manifest = {"format": FORMAT_VERSION, "tree": _encode(state, tensors, ""), "extra": extra or {},
                "sharded": {"world": int(world), "local_keys": sorted(local_keys)}}
    missing = local_keys - set(tensors)
    if missing:
        raise KeyError(f"row-local keys not found in the state tree: {sorted(missing)}")
    tmp = path.with_name(path.name + ".tmp-sharded")
    if rank == 0:
        if tmp.exists():
            shutil.rmtree(tmp)
        tmp.mkdir(parents=True)
    if barrier is not None:
        barrier()
    save_file({k: tensors[k].clone() for k in sorted(local_keys)} or {"__empty__": torch.zeros(1)}, str(tmp / _rank_file(rank)))
    if rank == 0:
        save_file({k: v.clone() for k, v in tensors.items() if k not in local_keys}, str(tmp / "tensors.safetensors"))
        (tmp / "manifest.json").write_text(json.dumps(manifest))
    if barrier is not None:
        barrier()
    if rank == 0:
        if path.exists():
            shutil.rmtree(path)
        os.replace(tmp, path)
    if barrier is not None:
        barrier()


def load_state_sharded(path, rank: int, world: int, template=None, device=None):
    """This rank's view of a checkpoint written by ``save_state_sharded`` (any writer world size)."""
    from safetensors.torch import load_file

    path = Path(path).resolve()
    manifest = json.loads((path / "manifest.json").read_text())
    if manifest.get("format") != FORMAT_VERSION:
        raise ValueError(f"unsupported checkpoint format {manifest.get('format')}")
    info = manifest.get("sharded")
    if info is None:
        raise ValueError("not a sharded checkpoint; use load_state")
    tensors = load_file(str(path / "tensors.safetensors"))
    saved_world, keys = int(info["world"]), list(info["local_keys"])
    if saved_world == world:
        mine = load_file(str(path / _rank_file(rank)))
        for k in keys:
            tensors[k] = mine[k]
    elif keys:
        parts = [load_file(str(path / _rank_file(r))) for r in range(saved_world)]
        for k in keys:
            full = torch.cat([p[k] for p in parts], dim=0)
            n = full.shape[0]
            if n % world:
                raise ValueError(f"{k}: {n} rows cannot be spread evenly over {world} ranks")
            per = n // world
            tensors[k] = full[rank * per:(rank + 1) * per].clone()
    restored = _decode(manifest["tree"], tensors, device)
    if template is None:
        return restored
    t_leaves, t_def = _tree.tree_flatten(template)
    r_leaves, _ = _tree.tree_flatten(restored)
    if len(t_leaves) != len(r_leaves):
        raise ValueError(f"checkpoint has {len(r_leaves)} leaves but the template has {len(t_leaves)}")
    out = [r.to(device=device if device is not None else t.device, dtype=t.dtype)
           if isinstance(t, torch.Tensor) and isinstance(r, torch.Tensor) else r for t, r in zip(t_leaves, r_leaves)]
    return _tree.tree_unflatten(t_def, out)
