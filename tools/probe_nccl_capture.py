#!/usr/bin/env python3
"""What makes ProcessGroupNCCL's watchdog abort a process that captures RCCL collectives into a hipGraph (round 5: two unexplained deaths;
round 6: `HIP error: operation not permitted on an event last recorded in a capturing stream`, raised from WorkNCCL::isCompleted() on the watchdog
thread, gpurun_out/r06zz/train_ddp1_bf16_b8_256.err).  One-rank "nccl" group, each case in a child process of its own:
  A  capture a graph with an all_reduce, replay, sleep 1 s (several watchdog polls), destroy
  B  an EAGER all_reduce immediately followed by a capture whose host side takes ~0.5 s (collectives inside), sleep, destroy
  C  as B with torch.cuda.synchronize() + 0.5 s of sleep between the eager collective and the capture (the watchdog has polled the eager Work away)
  D  as B without any collective inside the capture
python tools/probe_nccl_capture.py            -> runs A..D, prints one line per case"""
import os, subprocess, sys, time

CASES = "ABCD"


def child(case):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mdie_amd import launch as LA
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{LA.free_port()}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    x = torch.ones(1 << 20, device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    time.sleep(0.5)                      # everything so far has left the watchdog's list
    if case in "BCD":
        dist.all_reduce(x)               # an eager Work, enqueued for the watchdog ...
        if case == "C":
            torch.cuda.synchronize()
            time.sleep(0.5)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(10):
                x.mul_(1.0)
                if case != "D":
                    w = dist.all_reduce(x, async_op=True)
                    w.wait()
                if case in "BCD":
                    time.sleep(0.05)     # ... a capture that takes the host a while: the watchdog polls during it
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    time.sleep(1.0)
    del g
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print(f"CASE-{case}-OK", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in list(CASES):
        child(sys.argv[1])
        sys.exit(0)
    for c in CASES:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=300)
        ok = f"CASE-{c}-OK" in r.stdout and r.returncode == 0
        err = [l for l in r.stderr.splitlines() if "HIP error" in l or "what():" in l or "Aborted" in l or "Error" in l]
        print(f"case {c}: {'ok' if ok else 'DIED rc ' + str(r.returncode)}  {err[:2]}", flush=True)
