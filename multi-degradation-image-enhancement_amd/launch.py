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


def init_or_exit(init, *args, **kwargs):
    """`init(*args, **kwargs)` (torch.distributed.init_process_group) for a self-launched rank: when the rendezvous port handed out by
    free_port() was taken by somebody else before rank 0 could bind it, exit with EADDRINUSE_EXIT so that the parent retries on a fresh
    port (self_launch); every other error propagates."""
    try:
        return init(*args, **kwargs)
    except Exception as e:      # torch raises DistNetworkError (a RuntimeError) with the errno's text
        text = str(e).lower()
        if os.environ.get("MDIE_SELF_LAUNCHED") == "1" and ("eaddrinuse" in text or "address already in use" in text):
            sys.exit(EADDRINUSE_EXIT)
        raise


EADDRINUSE_EXIT = 98      # a rank whose rendezvous port was taken between free_port() and its bind exits with this code (errno EADDRINUSE)


def self_launch(argv, world, grace_s=20.0, kill_after_s=10.0, _retried=False):
    """Run `sys.executable argv...` as `world` ranks; returns the worst exit code (a signal death counts as 128 + signal).
    When one rank fails the others get `grace_s` seconds to notice (a broken collective usually ends them), are then terminated
    -- each by its own PID -- and, if a rank ignores that for `kill_after_s` more seconds (blocked inside a collective or behind a hung
    kernel, SIGTERM is not delivered to Python), killed; the return code is then non-zero whatever the survivors report.
    free_port() closes its socket before the children bind the port, so another job can take it in between: when EVERY rank that failed
    did so with EADDRINUSE_EXIT (entry points map the rendezvous error to it) the launch is repeated once on a fresh port."""
    port = free_port()
    procs = [subprocess.Popen([sys.executable] + list(argv), env=child_env(r, world, port)) for r in range(world)]
    worst, failed_at, terminated_at, forced = 0, None, None, False
    try:
        while any(p.poll() is None for p in procs):
            now = time.monotonic()
            if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
                failed_at = now
            if failed_at is not None and terminated_at is None and now - failed_at > grace_s:
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                terminated_at, forced = now, True
            if terminated_at is not None and now - terminated_at > kill_after_s:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                terminated_at = now + 1e9      # (SIGKILL cannot be ignored: the loop ends when the kernel has reaped them)
            time.sleep(0.05)
    except KeyboardInterrupt:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGINT)
        for p in procs:
            p.wait()
        return 130
    codes = [p.returncode for p in procs]
    failed = [rc for rc in codes if rc != 0 and not (forced and rc < 0)]      # (ranks this function signalled are not the cause)
    if failed and not _retried and all(rc == EADDRINUSE_EXIT for rc in failed):
        return self_launch(argv, world, grace_s, kill_after_s, _retried=True)
    for rc in codes:
        worst = max(worst, rc if rc >= 0 else 128 - rc)
    return max(worst, 1) if forced else worst
