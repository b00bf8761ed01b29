"""Writer of the EvoXVis log format: Apache Arrow IPC record batches with one row per
generation (generation number, raw fitness bytes, optional raw population bytes, duration and
named metrics; dtypes and population size in the schema metadata).
Format parity: reference ``src/evox/monitors/evoxvis_monitor.py:10-221``.

Device tensors are copied to host memory when recorded; rows are buffered and flushed every
``batch_size`` generations."""
from __future__ import annotations

import tempfile
import time
import warnings
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from ..core.monitor import Monitor

_NP_NAMES = {torch.float32: "float32", torch.float64: "float64", torch.float16: "float16", torch.int32: "int32",
             torch.int64: "int64", torch.bool: "bool", torch.uint8: "uint8", torch.int8: "int8", torch.int16: "int16"}


def _host_bytes(t):
    if isinstance(t, torch.Tensor):
        if t.dtype == torch.bfloat16:
            t = t.to(torch.float32)
        return t.detach().cpu().contiguous().numpy().tobytes(), _NP_NAMES.get(t.dtype, str(t.dtype))
    a = np.ascontiguousarray(t)
    return a.tobytes(), str(a.dtype)


class EvoXVisMonitor(Monitor):
    def __init__(self, base_filename: str, out_dir: Optional[str] = None, out_type: str = "file", batch_size: int = 64,
                 compression: Optional[str] = None):
        import pyarrow as pa
        self._pa = pa
        if out_type not in ("file", "stream"):
            raise ValueError(f"out_type must be 'file' or 'stream', got {out_type!r}")
        base = Path(tempfile.gettempdir()) / "evox" if out_dir is None else Path(out_dir)
        base.mkdir(parents=True, exist_ok=True)
        i = 0
        while (base / f"{base_filename}_{i}.arrow").exists():
            i += 1
        self.path = str(base / f"{base_filename}_{i}.arrow")
        self.sink = pa.OSFile(self.path, "wb")
        self.out_type = out_type
        self.comp_alg = compression
        self.batch_size = batch_size
        self.generation_counter = 0
        self.rows = {"generation": [], "fitness": [], "population": None, "duration": [], "metrics": None}
        self.population_size = None
        self.population_dtype = None
        self.fitness_dtype = None
        self.timestamp = None
        self.ref_time = None
        self.metric_names = None
        self.ec_schema = None
        self.writer = None
        self.is_closed = False

    def hooks(self):
        return ["pre_step", "post_ask", "post_eval"]

    def pre_step(self, _state):
        self.record_time()

    def post_ask(self, _state, cand_sol):
        if isinstance(cand_sol, torch.Tensor):
            self.record_pop(cand_sol)

    def post_eval(self, _state, _cand_sol, _transformed, fitness):
        self.record_fit(fitness)

    # ---------------------------------------------------------------------------------------
    def record_pop(self, population, transform=None):
        if self.rows["population"] is None:
            self.rows["population"] = []
        raw, self.population_dtype = _host_bytes(population)
        self.rows["population"].append(raw)

    def record_fit(self, fitness, metrics=None, transform=None):
        raw, self.fitness_dtype = _host_bytes(fitness)
        self.rows["generation"].append(self.generation_counter)
        self.rows["fitness"].append(raw)
        self.population_size = int(fitness.shape[0])
        self.generation_counter += 1
        if metrics:
            if self.rows["metrics"] is None:
                self.rows["metrics"] = []
            self.rows["metrics"].append({k: float(v) for k, v in metrics.items()})
        ready = len(self.rows["fitness"])
        if self.rows["population"] is not None:
            ready = min(ready, len(self.rows["population"]))
        if ready >= self.batch_size:
            self._write_batch(ready)

    def record_time(self, transform=None):
        if self.timestamp is None:
            self.timestamp = time.time()
            self.ref_time = time.monotonic()
        self.rows["duration"].append(time.monotonic() - self.ref_time)

    def _open_writer(self):
        pa, rows = self._pa, self.rows
        fields = [("generation", pa.uint64()), ("fitness", pa.binary(len(rows["fitness"][0])))]
        meta = {"population_size": str(self.population_size), "fitness_dtype": self.fitness_dtype}
        if rows["population"]:
            fields.append(("population", pa.binary(len(rows["population"][0]))))
            meta["population_dtype"] = self.population_dtype
        if self.timestamp is not None:
            fields.append(("duration", pa.float64()))
            meta["begin_time"] = str(self.timestamp)
        if rows["metrics"]:
            self.metric_names = list(rows["metrics"][0].keys())
            fields += [(name, pa.float64()) for name in self.metric_names]
            meta["metrics"] = "_".join(self.metric_names)
        self.ec_schema = pa.schema(fields, metadata=meta)
        opts = pa.ipc.IpcWriteOptions(compression=self.comp_alg)
        make = pa.ipc.new_file if self.out_type == "file" else pa.ipc.new_stream
        self.writer = make(self.sink, self.ec_schema, options=opts)

    def _write_batch(self, count):
        rows = self.rows
        if count == 0 or not rows["generation"]:
            return
        if self.writer is None:
            self._open_writer()
        names = set(self.ec_schema.names)
        cols = [rows["generation"][:count], rows["fitness"][:count]]
        rows["generation"], rows["fitness"] = rows["generation"][count:], rows["fitness"][count:]
        if "population" in names:
            cols.append(rows["population"][:count])
            rows["population"] = rows["population"][count:]
        if "duration" in names:
            cols.append(rows["duration"][:count])
            rows["duration"] = rows["duration"][count:]
        if self.metric_names:
            for name in self.metric_names:
                cols.append([m[name] for m in rows["metrics"][:count]])
            rows["metrics"] = rows["metrics"][count:]
        self.writer.write_batch(self._pa.record_batch(cols, schema=self.ec_schema))

    def flush(self):
        self._write_batch(len(self.rows["fitness"]))

    def close(self, flush=True):
        if self.is_closed:
            return
        self.is_closed = True
        if flush:
            self.flush()
        if self.writer is not None:
            self.writer.close()
        self.sink.close()

    def __del__(self):
        if not getattr(self, "is_closed", True):
            warnings.warn("The monitor is not correctly closed. Please close the monitor with `close`.")
            self.close(flush=False)
