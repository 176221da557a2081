"""One process per GPU on a single node, started from a parent that never touches the GPU.

The reference is single-device (``device = "cuda:0"`` at
/root/reference/vits_text_to_speech.py:47); here a node runs one rank per MI355X on disjoint
utterance batches.  ``run_ranks`` starts the ranks as fresh child processes with the usual
``torch.distributed`` environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT),
relays rank 0's standard output and fails if any rank fails.  The parent must call it before
any HIP call of its own: a process that has initialised the GPU is never replaced or forked.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    """Environment of one rank: ranks are local to the node, so LOCAL_RANK == RANK."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                "LOCAL_WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC, needed by RCCL on this driver
    return env


def needs_spawn(gpus: int, environ=None) -> bool:
    """True when this process was asked for several GPUs and is not itself a rank."""
    environ = os.environ if environ is None else environ
    return gpus > 1 and "RANK" not in environ


def run_ranks(argv, world: int, port: int | None = None, python: str | None = None,
              timeout: float | None = None, stdout=None) -> int:
    """Start ``world`` copies of ``python argv...`` as ranks 0..world-1 and wait for them.

    Rank 0's standard output is copied to ``stdout`` (default: this process's); the other
    ranks' output is dropped, every rank's standard error is inherited.  Returns 0 when every
    rank exited 0, otherwise the first non-zero exit status; when one rank fails (or the
    timeout expires) the remaining ranks are terminated by their own process ids.
    """
    port = free_port() if port is None else port
    python = sys.executable if python is None else python
    stdout = sys.stdout if stdout is None else stdout
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([python, *argv], env=rank_env(r, world, port),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      text=True))
    chunks = []                         # rank 0's output is drained as it comes: a full pipe would block the rank
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    try:
        pending = set(range(world))
        while pending and rc == 0:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code if code > 0 else 128 - code
            if deadline is not None and time.monotonic() > deadline:
                rc = 124
            if pending and rc == 0:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    out = "".join(chunks)
    if out:
        stdout.write(out)
        stdout.flush()
    return rc
