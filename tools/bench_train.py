#!/usr/bin/env python3
"""Training-step timing (BASELINE configs[2] shape: blur-style loss, 512x512, batch 8 per GPU), eager and as one hipGraph.
  python tools/bench_train.py [bf16|fp16|fp32] [B] [S] [loss terms, e.g. charbonnier:1,ssim:0.5] [eager|graph|both]
  (`--gpus N` anywhere on the line: N ranks, one per GPU, started by this script itself; or under torchrun: one rank per GPU, bucketed RCCL all-reduce; the graph then holds forward + loss + backward and the
   exchange + Adam step follow the replay.  MDIE_DDP_SINGLE=1: a ONE-rank "nccl" group on a single GPU -- the whole
   hook -> bucket -> RCCL all-reduce -> finish path on HIP without a second device; the eager mode then also prints the step with
   the exchange left out and with the exchange AFTER backward: overlapped vs exposed communication time)

Roofline model of the step (stated, not measured): FLOPs = 3 x the forward's (forward, input-gradient and weight-gradient
GEMMs of every convolution; `mdie_cdan_flops`); HBM bytes = 3 x the forward's fused-schedule activation bytes
(`mdie_cdan_algorithmic_bytes`: each of the three GEMM families reads its operands and writes its result once) + 16 bytes
per parameter per step (fp32 weight read, gradient write + read, Adam's two moments read + written, weight written)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REHEARSE = "--rehearse" in sys.argv      # the same launch / rendezvous / exchange / teardown plumbing on gloo with a stub step: no GPU (tests/test_bench_sharding_cpu.py)
if REHEARSE:
    sys.argv.remove("--rehearse")
    os.environ["MDIE_BENCH_TRAIN_REHEARSE"] = "1"
REHEARSE = REHEARSE or os.environ.get("MDIE_BENCH_TRAIN_REHEARSE") == "1"
if "--gpus" in sys.argv:      # `tools/bench_train.py --gpus N ...` with no launcher: become the parent of N ranks before anything touches the GPU
    _i = sys.argv.index("--gpus")
    _n = int(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
    from mdie_amd import launch as _LA
    if _LA.needs_self_launch(_n):
        sys.exit(_LA.self_launch([os.path.abspath(__file__)] + sys.argv[1:], _n))
import torch


def rehearse():
    """`tools/bench_train.py --gpus N --rehearse [prec] [B] [S]`: every rank joins a gloo group, runs GradBuckets over a tiny stand-in
    network for three steps -- hooks, five-bucket all-reduce, finish -- checks that all ranks end with the same averaged gradients, and
    leaves through host.shutdown_distributed: what can break on argument plumbing, rendezvous or teardown breaks here, without a GPU."""
    import torch.distributed as dist
    from mdie_amd import host as H
    from mdie_amd import train as T
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    prec, B, S = (args + ["bf16", "8", "512"])[0] if args else "bf16", int(args[1]) if len(args) > 1 else 8, int(args[2]) if len(args) > 2 else 512
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 3, 3, padding=1))
    buckets = T.GradBuckets(net.parameters(), n_buckets=2)
    try:
        for i in range(3):
            torch.manual_seed(100 + rank + 10 * i)
            net.zero_grad(set_to_none=True)
            net(torch.rand(2, 3, 16, 16)).square().mean().backward()
            buckets.finish()
        flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same = all(torch.equal(g, gathered[0]) for g in gathered)
        nb = len(buckets.buckets)
    finally:
        H.shutdown_distributed(None, buckets)
    if rank == 0:
        print(f"train[rehearse,{prec}] B={B}x{world} {S}x{S}: gloo, stub step | exchange: hooks, {nb} buckets, ranks agree: {same} | teardown: ok", flush=True)
    sys.exit(0 if same else 1)


if REHEARSE:
    rehearse()
from models.cdan import CDAN
from mdie_amd import host as H
from mdie_amd import lib as L
from mdie_amd import train as T
from mdie_amd import synthetic as P

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 512
spec = sys.argv[4] if len(sys.argv) > 4 else "charbonnier:1,ssim:0.5"
modes = {"both": ["eager", "graph"]}.get(sys.argv[5] if len(sys.argv) > 5 else "both", [sys.argv[5]] if len(sys.argv) > 5 else [])
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist = None
if world > 1 or os.environ.get("MDIE_DDP_SINGLE") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        from mdie_amd import launch as _LA2
        os.environ["MASTER_PORT"] = str(_LA2.free_port())
    from mdie_amd import launch as _LA3
    _LA3.init_or_exit(dist.init_process_group, "nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
x, t = P.lowlight_batch(100 + rank, B, S, S)
x, t = x.cuda(), t.cuda()
losses = H.build_losses({"enabled": True, "terms": [{"name": s.split(":")[0], "weight": float(s.split(":")[1])} for s in spec.split(",")]})
esz = 4 if prec == "fp32" else 2
flops = 3.0 * L.lib.mdie_cdan_flops(B, S, S)
byts = 3.0 * L.lib.mdie_cdan_algorithmic_bytes(B, S, S, esz) + 16.0 * 3585663
peak_tf = 157.3 if prec == "fp32" else 2500.0

variants = [(m, "overlap") for m in modes]
if dist is not None and os.environ.get("MDIE_DDP_CAPTURE", "0") not in ("1", "auto"):
    # the exchange INSIDE the captured step is opt-in: torch's NCCL watchdog can abort a process that captures collectives (train.CapturedStep)
    variants = [v for v in variants if v != ("graph", "overlap")]
if dist is not None and "eager" in modes:
    variants += [("eager", "none"), ("eager", "after")]     # the same step without the exchange, and with it after backward (nothing overlapped)
if dist is not None and "graph" in modes:
    variants += [("graph", "none"), ("graph", "after")]     # round 4's form: the five collectives behind the replay
for mode, comm in variants:
    torch.manual_seed(42)
    net = CDAN(precision=prec).cuda().train()
    scaler = torch.amp.GradScaler("cuda", enabled=prec == "fp16")
    in_graph = mode == "graph" and comm == "overlap"         # the collectives are branches of the captured step (train.CapturedStep, buckets=)
    whole = mode == "graph" and (dist is None or in_graph or comm == "none") and not scaler.is_enabled()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=whole, fused=os.environ.get("ADAM_FUSED", "1") == "1")
    buckets = T.GradBuckets(net.parameters()) if (dist is not None and comm != "none") else None
    if buckets is not None and comm == "after":
        buckets.remove()
    if mode == "graph":
        cap = T.CapturedStep(net, losses, opt if whole else None, x, t, scale_fn=scaler.scale if scaler.is_enabled() else None,
                             buckets=buckets if in_graph else None)

        def step():
            v = cap(x, t)
            if not whole:
                if buckets is not None and not in_graph:
                    buckets.exchange()
                scaler.step(opt)
                scaler.update()
            return v[-1]
    else:
        def step():
            opt.zero_grad(set_to_none=True)
            total, _ = losses(net(x), t)
            scaler.scale(total).backward()
            if buckets is not None:
                buckets.finish() if comm == "overlap" else buckets.exchange()
            scaler.step(opt)
            scaler.update()
            return total

    for _ in range(3):
        l = step()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        l = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    if buckets is not None:
        form = "in-graph" if (mode == "graph" and comm == "overlap") else "hooks" if comm == "overlap" else "after backward (not overlapped)"
        note = f" | exchange: {form}, {len(buckets.buckets)} buckets, {buckets.copies_in} gradient copies into buckets over {n + 3} steps"
    else:
        note = " | no gradient exchange" if dist is not None else ""
    if rank == 0:
        print(f"train[{prec},{mode}] B={B}x{world} {S}x{S} loss={spec}: {dt*1e3:.2f} ms/step, {B*world/dt:.1f} img/s, loss {l.item():.4f} | "
              f"model: {flops/1e9:.0f} GFLOP, {byts/1e9:.2f} GB per rank-step -> {flops/dt/1e12:.0f} TFLOP/s = {flops/dt/1e12/peak_tf:.3f} of the {prec} MFMA peak, "
              f"{byts/dt/1e9:.0f} GB/s = {byts/dt/8e12:.3f} of 8 TB/s{note}", flush=True)
    # this variant's graph (captured collectives of the communicator under world > 1) and buckets go BEFORE the next variant and before the
    # group: host.shutdown_distributed's order
    cap = step = None
    H.shutdown_distributed(None, buckets, destroy=False)
if dist is not None:
    H.shutdown_distributed()
