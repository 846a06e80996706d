"""One process per GPU without an external launcher.

`python bench.py --gpus N` (and tools/bench_train.py, run.py) are normally started by
`python -m torch.distributed.run --nproc-per-node N ...`, which sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.
When those are absent and more than one GPU is asked for, the entry point calls `self_launch` FIRST -- before any HIP
call, before `import torch` has been asked anything about devices -- and becomes a plain parent: it starts N children of the
same command line with the rendezvous environment set, lets them write to its own stdout / stderr (rank 0 prints the one JSON
line), and returns the worst child exit code.  No `os.exec*` anywhere: a process that has initialised the GPU must never be
replaced, and the parent here never initialises it.

The reference has no counterpart (one process, one device: /root/reference/run.py:37-58, models/base.py:16).
This module imports nothing but the standard library on purpose.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def needs_self_launch(n_gpus, env=None):
    """True when this process was NOT started by a launcher (no WORLD_SIZE) and more than one rank is wanted."""
    env = os.environ if env is None else env
    return n_gpus > 1 and "WORLD_SIZE" not in env


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_env(rank, world, port, base=None):
    """the variables `torch.distributed.run` would have set for local rank `rank` of a one-node job"""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MDIE_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // world)))
    return env


def self_launch(argv, world, grace_s=20.0):
    """Run `sys.executable argv...` as `world` ranks; returns the worst exit code (a signal death counts as 128 + signal).
    When one rank fails the others get `grace_s` seconds to notice (a broken collective usually ends them) and are then
    terminated -- each by its own PID."""
    port = free_port()
    procs = [subprocess.Popen([sys.executable] + list(argv), env=child_env(r, world, port)) for r in range(world)]
    worst, failed_at = 0, None
    try:
        while any(p.poll() is None for p in procs):
            for p in procs:
                rc = p.poll()
                if rc is not None and rc != 0 and failed_at is None:
                    failed_at = time.monotonic()
            if failed_at is not None and time.monotonic() - failed_at > grace_s:
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                failed_at = time.monotonic() + 1e9     # terminate once; the loop ends when they are gone
            time.sleep(0.05)
    except KeyboardInterrupt:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGINT)
        for p in procs:
            p.wait()
        return 130
    for p in procs:
        rc = p.returncode
        worst = max(worst, rc if rc >= 0 else 128 - rc)
    return worst
