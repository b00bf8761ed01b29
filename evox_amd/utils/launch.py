"""Start a script as N ranks on this node — one child process per GPU, rendezvous on 127.0.0.1.

Standard library only, and loadable by file path (``bench.py`` and ``benchmarks/run.py`` do so before anything
imports torch): the parent never touches a GPU, the children are started, never exec'ed into.  A rank that fails
takes the others down with it (exact PIDs), so a dead rank cannot leave its peers waiting in a collective; an overall
time limit does the same for a job in which every rank waits (a rank hung in the communicator's set-up), and the last
lines every rank wrote to its standard error are reported with the failure.
"""
from __future__ import annotations

import collections
import os
import socket
import subprocess
import sys
import threading
import time


def free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def default_timeout() -> float:
    """Overall limit of a self-launched job in seconds (``EVOX_AMD_LAUNCH_TIMEOUT``, default 900; 0 = none)."""
    try:
        return float(os.environ.get("EVOX_AMD_LAUNCH_TIMEOUT", "900"))
    except ValueError:
        return 900.0


def visible_gpus() -> int | None:
    """Number of GPUs the children will see, without initialising one here (device files of the compute driver, cut
    down by ``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES``); ``None`` when it cannot be told."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    try:
        nodes = [d for d in os.listdir("/dev/dri") if d.startswith("renderD")]
    except OSError:
        return None
    return len(nodes) or None


def _tail_reader(stream, tail: collections.deque, echo):
    for line in iter(stream.readline, ""):
        tail.append(line)
        if echo is not None:
            echo.write(line)
            echo.flush()
    stream.close()


def launch_ranks(script: str, argv, n: int, poll_s: float = 0.2, timeout_s: float | None = None, tail_lines: int = 20,
                 term_grace_s: float = 5.0) -> int:
    """Run ``script argv...`` as ``n`` ranks; relay rank 0's standard output; return the worst exit code.

    ``timeout_s`` (default :func:`default_timeout`): when the job is still running after that long every rank is
    terminated (then killed after ``term_grace_s``) and 124 is returned.  On any failure the last ``tail_lines`` lines
    of each rank's standard error are written to this process's standard error, labelled by rank."""
    timeout_s = default_timeout() if timeout_s is None else float(timeout_s)
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs, tails, readers = [], [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        p = subprocess.Popen([sys.executable, os.path.abspath(script), *argv], env=env,
                             stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        procs.append(p)
        tail = collections.deque(maxlen=tail_lines)
        tails.append(tail)
        t = threading.Thread(target=_tail_reader, args=(p.stderr, tail, sys.stderr if r == 0 else None), daemon=True)
        t.start()
        readers.append(t)

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def stop(live):
        for q in live:
            procs[q].terminate()
        deadline = time.monotonic() + term_grace_s
        for q in live:
            try:
                procs[q].wait(timeout=max(0.0, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                procs[q].kill()

    worst = 0
    timed_out = False
    live = set(range(n))
    t_end = time.monotonic() + timeout_s if timeout_s > 0 else None
    while live:
        for r in list(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and worst == 0:
                worst = rc
                stop(live)              # the peers of a failed rank would wait for it forever
        if live and t_end is not None and time.monotonic() > t_end:
            timed_out = True
            stop(live)
            for r in live:
                procs[r].poll()
            live.clear()
            worst = worst or 124
        if live:
            time.sleep(poll_s)
    reader.join(timeout=5)
    for t in readers:
        t.join(timeout=2)
    out = "".join(c for c in chunks if c)
    if out:
        sys.stdout.write(out)
        sys.stdout.flush()
    if worst != 0:
        why = f"timed out after {timeout_s:.0f} s" if timed_out else f"failed with exit code {worst}"
        sys.stderr.write(f"[launch] {n}-rank job {why}; last lines of each rank's stderr:\n")
        for r, tail in enumerate(tails):
            rc = procs[r].returncode
            sys.stderr.write(f"[launch] --- rank {r} (exit {rc}) ---\n")
            for line in tail:
                sys.stderr.write(f"[launch] r{r}| {line}" if line.endswith("\n") else f"[launch] r{r}| {line}\n")
        sys.stderr.flush()
    return worst
