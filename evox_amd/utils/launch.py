"""Start a script as N ranks on this node — one child process per GPU, rendezvous on 127.0.0.1.

Standard library only, and loadable by file path (``bench.py`` and ``benchmarks/run.py`` do so before anything
imports torch): the parent never touches a GPU, the children are started, never exec'ed into.  A rank that fails
takes the others down with it (exact PIDs), so a dead rank cannot leave its peers waiting in a collective.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time


def free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(script: str, argv, n: int, poll_s: float = 0.2) -> int:
    """Run ``script argv...`` as ``n`` ranks; relay rank 0's standard output; return the worst exit code."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    worst = 0
    live = set(range(n))
    while live:
        for r in list(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and worst == 0:
                worst = rc
                for q in live:          # the peers of a failed rank would wait for it forever
                    procs[q].terminate()
        time.sleep(poll_s)
    reader.join(timeout=5)
    out = "".join(c for c in chunks if c)
    if out:
        sys.stdout.write(out)
        sys.stdout.flush()
    return worst
