"""A distributed workflow launched from a single driver process.
Role parity: reference ``src/evox/workflows/distributed.py:21-388`` (``RayDistributedWorkflow``:
Ray actors, fitness exchanged through the object store, a supervisor actor in between).

The driver spawns ``num_workers`` worker processes (``multiprocessing`` *spawn* context – a
worker never inherits an initialised GPU runtime).  Every worker builds the same algorithm and
problem from a ``cloudpickle`` payload with the same seed, joins one ``torch.distributed`` group
(``nccl`` = RCCL when ``options["devices"]`` assigns GPUs, ``gloo`` on CPU) and runs a
:class:`~evox_amd.parallel.workflows.ShardedWorkflow`: each evaluates its slice of the
population and the fitness is all-gathered *between the workers* – nothing flows through the
driver except the step command and, for the driver-side monitors, rank 0's copy of the hook
arguments (with the full candidate set, row-aligned with the fitness — the reference passes ``None`` there,
``distributed.py:110-113``).  ``async_dispatch`` steps may be in flight before the driver waits.

``metrics`` (name → ``f(fitness)``) are evaluated on rank 0 on every generation's whole-population fitness and
delivered with the step's result: ``latest_metrics`` / ``metric_history``.  ``global_fit_transform`` is applied on
every worker to the gathered fitness before ``fit_transforms``.  (The reference declares both parameters and never
reads them, ``distributed.py:262-266``.)

A worker that raises or dies does not leave the job hanging: the driver waits on all pipes at once, and on the
first error — or a worker process found dead — kills the remaining workers (exact processes it started) and
raises ``RuntimeError`` with the worker's traceback."""
from __future__ import annotations

import multiprocessing as mp
import os
import socket
import warnings
from collections import deque
from multiprocessing.connection import wait as _wait_pipes
from typing import Callable, Dict, List, Optional, Union

import torch

from ..core.monitor import HOOK_NAMES
from ..core.state import State
from ..core.workflow import Workflow
from ..utils.common import parse_opt_direction


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _HookRecorder:
    """Worker-side monitor that queues the arguments of the hooks the driver listens to."""

    def __init__(self, hook_names):
        self._hooks = list(hook_names)
        self.queue = []
        for name in self._hooks:
            setattr(self, name, self._make(name))

    def _make(self, name):
        def record(state, *args):
            self.queue.append((name, tuple(a.detach().cpu() if isinstance(a, torch.Tensor) else a for a in args)))
        return record

    def hooks(self):
        return self._hooks

    def set_opt_direction(self, _):
        pass

    def drain(self):
        out, self.queue = self.queue, []
        return out


def _worker_main(rank, world, port, conn, payload, options):
    import cloudpickle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.set_num_threads(max(1, int(options.get("num_cpus", 1))))
    devices = options.get("devices")
    import torch.distributed as dist
    if devices:
        device = torch.device(devices[rank % len(devices)])
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from ..core.module import use_state
    from .workflows import ShardedWorkflow
    spec = cloudpickle.loads(payload)
    recorder = _HookRecorder(spec["hooks"] if rank == 0 else [])
    metrics = spec.get("metrics") or {}
    metric_values = {}
    if rank == 0 and metrics:
        def _metrics_hook(state, cands, tcands, fitness):
            for name, fn in metrics.items():
                v = fn(fitness)
                metric_values[name] = v.detach().cpu() if isinstance(v, torch.Tensor) else v

        class _MetricMonitor:
            def hooks(self):
                return ["post_eval"]

            def set_opt_direction(self, _):
                pass

            post_eval = staticmethod(_metrics_hook)
        extra_monitors = [_MetricMonitor()]
    else:
        extra_monitors = []
    wf = ShardedWorkflow(spec["algorithm"], spec["problem"], monitors=([recorder] if rank == 0 else []) + extra_monitors,
                         opt_direction=spec["opt_direction"], candidate_transforms=spec["sol_transforms"],
                         fitness_transforms=spec["fit_transforms"], global_fitness_transforms=spec.get("global_fit_transforms", ()),
                         monitor_candidates=spec.get("monitor_candidates", "local"))
    state = None
    try:
        while True:
            cmd, arg = conn.recv()
            if cmd == "init":
                state = wf.init(arg.to(device) if device.type == "cuda" else arg)
                state = wf.enable_multi_devices(state) if hasattr(wf, "enable_multi_devices") else state
                conn.send(("ok", None))
            elif cmd == "step":
                state = wf.step(state)
                conn.send(("ok", (recorder.drain(), dict(metric_values)) if rank == 0 else None))
            elif cmd == "valid":
                # the problem switched to its validation split on a throw-away state; every worker evaluates its rows
                # of the current candidates, the fitness is gathered between the workers
                vstate = use_state(wf.problem.valid)(state, metric=arg)
                pop, vstate = use_state(wf.algorithm.ask)(vstate)
                n = pop.shape[0]
                lo, hi = wf.slice_bounds(n, rank, world)
                part = pop[lo:hi]
                for t in wf.candidate_transforms:
                    part = t(part)
                fit, _ = use_state(wf.problem.evaluate)(vstate, part)
                full = wf._gather_fitness(fit, n, world)
                conn.send(("ok", full.detach().cpu() if rank == 0 else None))
            elif cmd == "fail":                                  # test hook: a worker that raises mid-run
                if rank == arg:
                    raise RuntimeError("injected worker failure")
                conn.send(("ok", None))
            elif cmd == "sample":
                cands, _ = wf.algorithm.ask(state.get_child_state("algorithm"))
                conn.send(("ok", cands.detach().cpu() if rank == 0 else None))
            elif cmd == "state":
                conn.send(("ok", state.to("cpu") if rank == 0 else None))
            elif cmd == "close":
                conn.send(("ok", None))
                break
    except Exception as e:                                      # reported to the driver
        import traceback
        conn.send(("err", f"{e!r}\n{traceback.format_exc()}"))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


class ProcessDistributedWorkflow(Workflow):
    def __init__(self, algorithm, problem, num_workers: int, monitors=(), opt_direction: Union[str, List[str]] = "min",
                 metrics: Optional[Dict[str, Callable]] = None, options: dict = {}, sol_transforms: List[Callable] = (),
                 fit_transforms: List[Callable] = (), global_fit_transform: List[Callable] = (), async_dispatch: int = 4,
                 monitor=None):
        super().__init__()
        import cloudpickle
        self.monitors = list(monitors)
        if monitor is not None:
            warnings.warn("`monitor` is deprecated, use the `monitors` parameter with a list of monitors instead",
                          DeprecationWarning)
            self.monitors = [monitor]
        self.registered_hooks = {h: [] for h in HOOK_NAMES}
        for m in self.monitors:
            for h in m.hooks():
                self.registered_hooks[h].append(m)
        self.opt_direction = parse_opt_direction(opt_direction)
        for m in self.monitors:
            m.set_opt_direction(self.opt_direction)
        remote_hooks = [h for h in HOOK_NAMES if self.registered_hooks[h] and h not in ("pre_step", "post_step")]
        needs_cands = any(h in remote_hooks for h in ("post_eval", "pre_tell"))
        self._problem_has_valid = callable(getattr(problem, "valid", None))
        self.metrics = dict(metrics or {})
        self.latest_metrics: Dict[str, object] = {}
        self.metric_history: Dict[str, list] = {k: [] for k in self.metrics}
        payload = cloudpickle.dumps(dict(algorithm=algorithm, problem=problem, opt_direction=opt_direction,
                                         sol_transforms=list(sol_transforms), fit_transforms=list(fit_transforms),
                                         global_fit_transforms=list(global_fit_transform), metrics=self.metrics,
                                         hooks=remote_hooks, monitor_candidates="full" if needs_cands else "local"))
        self.num_workers = num_workers
        self.async_dispatch = max(1, async_dispatch)
        self._inflight = deque()
        ctx = mp.get_context("spawn")
        port = _free_port()
        self._conns, self._procs = [], []
        for rank in range(num_workers):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker_main, args=(rank, num_workers, port, child, payload, dict(options)), daemon=True)
            p.start()
            child.close()
            self._conns.append(parent)
            self._procs.append(p)

    # ------------------------------------------------------------------------------------
    def _send_all(self, cmd, arg=None):
        for c in self._conns:
            c.send((cmd, arg))

    def _abort(self, why: str):
        """Kill every worker this driver started and raise."""
        for p in self._procs:
            if p.is_alive():
                p.kill()
        for p in self._procs:
            p.join(timeout=5)
        for c in self._conns:
            try:
                c.close()
            except OSError:
                pass
        self._conns, self._procs = [], []
        self._inflight.clear()
        raise RuntimeError(f"distributed worker failed: {why}")

    def _recv_all(self, poll_s: float = 0.5):
        """One reply from every worker.  Waits on all pipes at once: the first error, closed pipe or dead process
        aborts the job instead of blocking on a peer that waits in a collective for the failed one."""
        out = [None] * len(self._conns)
        pending = {c: i for i, c in enumerate(self._conns)}
        while pending:
            ready = _wait_pipes(list(pending), timeout=poll_s)
            for c in ready:
                i = pending[c]
                try:
                    status, value = c.recv()
                except (EOFError, OSError):
                    self._abort(f"worker {i} closed its pipe (exit code {self._procs[i].exitcode})")
                if status == "err":
                    self._abort(f"worker {i}: {value}")
                out[i] = value
                del pending[c]
            if not ready:
                for c, i in pending.items():
                    p = self._procs[i]
                    if not p.is_alive() and not c.poll():
                        self._abort(f"worker {i} died (exit code {p.exitcode})")
        return out

    def _replay(self, state, calls):
        for hook, args in calls or []:
            for m in self.registered_hooks[hook]:
                getattr(m, hook)(state, *args)

    def _collect_one(self, state):
        self._inflight.popleft()
        calls, metric_values = self._recv_all()[0]
        self._replay(state, calls)
        if metric_values:
            self.latest_metrics = metric_values
            for k, v in metric_values.items():
                self.metric_history.setdefault(k, []).append(v)

    def setup(self, key):
        self._send_all("init", key.cpu())
        self._recv_all()
        return State()

    def step(self, state: State, block: bool = False):
        for m in self.registered_hooks["pre_step"]:
            m.pre_step(state)
        self._send_all("step")
        self._inflight.append(1)
        while len(self._inflight) >= self.async_dispatch or (block and self._inflight):
            self._collect_one(state)
        for m in self.registered_hooks["post_step"]:
            m.post_step(state)
        return state

    def flush(self, state: State = None):
        """Wait for every dispatched step (and deliver its monitor calls)."""
        while self._inflight:
            self._collect_one(state)

    def sample(self, state: State):
        self.flush(state)
        self._send_all("sample")
        return self._recv_all()[0], state

    def valid(self, state: State, metric: str = "loss"):
        """Fitness of the current candidates on the problem's validation split (``problem.valid(state, metric)``);
        the workers' states are left untouched.  Reference: ``workflows/distributed.py:145-153, 381-383``."""
        if not self._problem_has_valid:
            raise NotImplementedError("the problem has no valid(state, metric) method")
        self.flush(state)
        self._send_all("valid", metric)
        return self._recv_all()[0], state

    def get_full_state(self):
        self.flush()
        self._send_all("state")
        return self._recv_all()[0]

    def close(self):
        if not self._conns:
            return
        try:
            self.flush()
            self._send_all("close")
            self._recv_all()
        except (RuntimeError, EOFError, BrokenPipeError, OSError):
            pass
        if not self._procs:
            return
        for p in self._procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
        self._conns, self._procs = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


RayDistributedWorkflow = ProcessDistributedWorkflow
DistributedWorkflow = ProcessDistributedWorkflow
