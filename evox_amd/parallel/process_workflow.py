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
arguments.  ``async_dispatch`` steps may be in flight before the driver waits."""
from __future__ import annotations

import multiprocessing as mp
import os
import socket
import warnings
from collections import deque
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
    from .workflows import ShardedWorkflow
    spec = cloudpickle.loads(payload)
    recorder = _HookRecorder(spec["hooks"] if rank == 0 else [])
    wf = ShardedWorkflow(spec["algorithm"], spec["problem"], monitors=[recorder] if rank == 0 else [],
                         opt_direction=spec["opt_direction"], candidate_transforms=spec["sol_transforms"],
                         fitness_transforms=spec["fit_transforms"])
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
                conn.send(("ok", recorder.drain() if rank == 0 else None))
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
        payload = cloudpickle.dumps(dict(algorithm=algorithm, problem=problem, opt_direction=opt_direction,
                                         sol_transforms=list(sol_transforms), fit_transforms=list(fit_transforms),
                                         hooks=remote_hooks))
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

    def _recv_all(self):
        out = []
        for c in self._conns:
            status, value = c.recv()
            if status == "err":
                raise RuntimeError(f"distributed worker failed: {value}")
            out.append(value)
        return out

    def _replay(self, state, calls):
        for hook, args in calls or []:
            for m in self.registered_hooks[hook]:
                getattr(m, hook)(state, *args)

    def _collect_one(self, state):
        self._inflight.popleft()
        self._replay(state, self._recv_all()[0])

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
