#!/usr/bin/env python3
"""Headline benchmark: images/sec of the CDAN forward (config/low_light.json network) at
256x256, bf16 storage / fp32 accumulate, batch 32 per GPU, inputs resident in HBM.

  python bench.py [--gpus N --steps K --warmup W]        (N > 1 with no launcher: starts its own N ranks, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one forward pass of the hot path over one synthetic batch.  Inference shards by
batch with no data-path collective, so N ranks run N independent batches
("weak" scaling); the only communication is the timing barrier / max-over-ranks.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      algorithmic HBM bytes of the forward / measured kernel time vs 8 TB/s
  cpu_baseline  the CPU oracle (port of the reference path) timed on this host's cores
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # kernel arguments in device memory (mdie_amd/__init__.py says why); before the runtime initialises

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TF = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}


def cpu_baseline(sd, x_cpu, seconds_budget=16.0):
    """The oracle (a port of the reference's PyTorch CPU path) on this host's cores: the whole batch of the GPU run
    (B = 32 at 256x256, 1 warm-up + timed forwards, median) on every core this process may use, plus a
    one-thread figure on a 2-image sample (a one-thread pass over 32 images alone would take most of a minute)."""
    from oracle import cdan_oracle as O
    # the cores this process is GRANTED (affinity mask and cgroup quota: a one-GPU box sees 256 CPUs and owns 16; more threads only oversubscribe)
    from mdie_amd import host as _host
    cores = int(os.environ.get("MDIE_CPU_THREADS", min(_host.cpu_share(), 64)))

    def timed(x, threads, budget, max_reps, min_reps=1):
        torch.set_num_threads(threads)
        with torch.no_grad():
            t0 = time.perf_counter()
            ref = O.cdan_forward(sd, x)          # warm-up (also the parity reference)
            warm = time.perf_counter() - t0
            reps = max(min_reps, min(max_reps, int(budget / max(warm, 1e-3)) - 1))
            times = []
            for _ in range(reps):
                t0 = time.perf_counter()
                O.cdan_forward(sd, x)
                times.append(time.perf_counter() - t0)
        return ref, sorted(times)[len(times) // 2], reps

    ref, med, reps = timed(x_cpu, cores, seconds_budget, 3, min_reps=3)     # >= 3 timed forwards, whatever the host's pace
    _, med1, reps1 = timed(x_cpu[:2], 1, 6.0, 2)
    torch.set_num_threads(cores)
    return ref, {"value": round(x_cpu.shape[0] / med, 3), "unit": "images/sec", "cores": cores, "kind": "port",
                 "sample": f"{reps} timed fp32 forwards of {x_cpu.shape[0]}x3x{x_cpu.shape[2]}x{x_cpu.shape[3]} "
                           f"(the GPU batch itself: same synthetic low-light images), median, after 1 warm-up",
                 "one_thread": {"value": round(2 / med1, 3), "unit": "images/sec", "cores": 1,
                                "sample": f"{reps1} timed fp32 forwards of 2x3x{x_cpu.shape[2]}x{x_cpu.shape[3]}, median, after 1 warm-up"}}


def source_sha16():
    """hash of the kernel sources + C header, comments and white space removed: ties a committed rocprofv3 traffic file to
    the CODE it was measured on (a reworded comment does not make a measurement stale)"""
    import glob
    import hashlib
    import re
    h = hashlib.sha256()
    pk = os.path.join(ROOT, "multi-degradation-image-enhancement_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(pk, "*.hip")) + glob.glob(os.path.join(pk, "*.hpp")) + [os.path.join(ROOT, "include", "mdie.h")]):
        with open(f, encoding="utf-8", errors="replace") as fh:
            t = fh.read()
        t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
        t = re.sub(r"//[^\n]*", "", t)
        t = re.sub(r"\s+", " ", t)
        h.update(os.path.basename(f).encode() + b"\0" + t.encode())
    with open(os.path.join(pk, "Makefile"), encoding="utf-8") as fh:      # the compiler flags are part of the code (lines that are not comments)
        h.update(b"Makefile\0" + " ".join(l.strip() for l in fh if l.strip() and not l.lstrip().startswith("#")).encode())
    return h.hexdigest()[:16]


def timed_loop(fn, dev, n, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / n


def max_over_ranks(elapsed, dist, device):
    """Whole-job time of a step loop = the slowest rank's (the contract's max over ranks)."""
    if dist is None:
        return elapsed
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def fence(dev, dist):
    """barrier + device synchronisation on both sides of the timed region (the contract's bracket)"""
    if dev is not None and dev.type == "cuda":
        torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    if dev is not None and dev.type == "cuda":
        torch.cuda.synchronize(dev)


def timed_region(run, steps, dev, dist):
    """EXACTLY `steps` calls of `run` between two fences; returns the slowest rank's wall time"""
    fence(dev, dist)
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    fence(dev, dist)
    return max_over_ranks(time.perf_counter() - t0, dist, dev)


def rehearse_main(args, rank, world):
    """`--rehearse`: the multi-rank plumbing of this file (self-launch or torchrun environment, rendezvous, per-rank batches, fences,
    max over ranks, rank 0's single JSON line) with the `gloo` backend on CPU and a STUB step (a scaled copy of the rank's batch --
    neither the engine nor the oracle).  For the CPU test suite and for checking an N-rank invocation on a box without N GPUs; the line
    says `"rehearsal": true` and its value means nothing."""
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mdie_amd import synthetic as P
    B, S = args.batch, args.size
    x, _ = rank_inputs(P, rank, B, S)
    y = torch.empty_like(x)
    dev = torch.device("cpu")

    def step():
        torch.mul(x, 0.5, out=y)
    for _ in range(args.warmup):
        step()
    elapsed = timed_region(step, args.steps, dev, dist)
    sums = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    if dist is not None:
        dist.all_gather(sums, y.double().sum().reshape(1))
    else:
        sums = [y.double().sum().reshape(1)]
    if rank == 0:
        print(json.dumps({"metric": "REHEARSAL (gloo, CPU stub step) of: images/sec @256x256 bf16 (low_light CDAN)", "rehearsal": True,
                          "value": round(world * B * args.steps / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"stub step on {B}x3x{S}x{S} per rank", "global_batch": B * world,
                                     "parallelism": f"batch-parallel x{world}, no collective",
                                     "launcher": "self-launched children" if os.environ.get("MDIE_SELF_LAUNCHED") == "1" else "external (torchrun environment)",
                                     "distinct_rank_batches": len({round(float(v), 6) for v in sums})}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def rank_inputs(P, rank, B, S):
    """Every rank owns its own batch (weak scaling, no data-path collective)."""
    return P.lowlight_batch(1000 + rank, B, S, S)


ROUTED_TASKS = ["blur", "color_distortion", "high_light", "jpeg", "low_contrast", "low_light", "motion_blur", "noise", "pixelation"]


def routed_labels(step, n, seed=0):
    """the stub router's output for global batch `step`: the same list on every rank (seeded), a different grouping every step"""
    g = torch.Generator().manual_seed(seed * 1000003 + step)
    return [ROUTED_TASKS[i] for i in torch.randint(0, len(ROUTED_TASKS), (n,), generator=g).tolist()]


def routed_main(args, rank, world, dev, dist, P):
    """BASELINE configs[3]: classifier-routed mixed degradations.  One step = one global batch of --batch x N images, each image run with its
    task's weights by ONE launch chain per rank (engine.RoutedEngine); no data-path collective.  --routed-shard slices (default): every
    rank holds all nine weight sets (9 x 7 MB) and takes an equal contiguous slice of the batch whatever the labels; tasks: the TASKS are
    dealt to ranks, a rank holds only its tasks' weight sets and an image follows its task (engine.routed_shard).  Labels are
    a stub router's (seeded, a different grouping every step, 16 distinct batches cycled); the router network itself is timed by
    tools/bench_configs.py."""
    from mdie_amd import engine as E
    B, S, n_lists = args.batch * world, args.size, 16
    by_task = args.routed_shard == "tasks"
    eng = E.RoutedEngine(dev, args.precision)
    mine_tasks = [t for i, t in enumerate(sorted(ROUTED_TASKS)) if i % world == rank] if by_task else sorted(ROUTED_TASKS)
    for t in mine_tasks:
        eng.load_task(t, P.make_state_dict(100 + ROUTED_TASKS.index(t)))
    x_all, _ = P.lowlight_batch(2000, B, S, S)
    batches = []
    for k in range(n_lists):
        labels = routed_labels(k, B)
        idx = E.routed_shard(labels, rank, world, ROUTED_TASKS) if by_task else E.routed_slice(B, rank, world)
        batches.append((x_all[idx].to(dev), [labels[i] for i in idx]))
    it = [0]

    def step():
        xb, lb = batches[it[0] % n_lists]
        it[0] += 1
        if len(lb):
            eng.forward(xb, lb)

    with torch.no_grad():
        for _ in range(max(args.warmup, n_lists)):
            step()

        elapsed = timed_region(step, args.steps, dev, dist)
    if rank == 0:
        print(json.dumps({"metric": "images/sec @256x256 bf16 (classifier-routed mixed degradations, 9 weight sets)", "value": round(B * args.steps / elapsed, 2),
                          "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, n_lists), "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                          "config": {"workload": f"BASELINE configs[3]: all config/*.json tasks as 9 seeded weight sets, {S}x{S}, global batch {B} labelled by a stub router "
                                                 f"(a different grouping every step), {args.precision} storage + fp32 accumulate",
                                     "global_batch": B,
                                     "parallelism": (f"tasks dealt to {world} rank(s), images follow their task, no collective" if by_task
                                                     else f"{world} rank(s), each holds all 9 weight sets and an equal slice of the batch, no collective"),
                                     "launch": "eager, ONE launch chain per rank-batch: every kernel looks up its image's weight set (mdie_cdan_fwd_desc.blob_delta)",
                                     "runtime": getattr(args, "runtime", None)}}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--routed-shard", choices=["slices", "tasks"], default="slices", help="--workload routed under N ranks: equal slices of the batch (default) or tasks dealt to ranks")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="enqueue launches eagerly instead of replaying a hipGraph (= --launch eager)")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "eager"],
                    help="how a step is enqueued: one hipGraph replay, the ~40 eager launches of one host call, or (auto) whichever "
                         "of the two ran the untimed warmup steps faster on this box")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-batch", type=int, default=32, help="images of the GPU batch the CPU baseline runs (SURVEY.md 8d: the whole batch)")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed side measurements (fp32 / fp16 paths, nn.Module boundary)")
    ap.add_argument("--workload", default="forward", choices=["forward", "routed"],
                    help="forward: the headline (BASELINE configs[1]); routed: BASELINE configs[3] -- 9 task weight sets, a global batch of "
                         "--batch x N images labelled by a stub router, every rank runs the images of the tasks it owns (no collective)")
    ap.add_argument("--rehearse", action="store_true", help="gloo / CPU rehearsal of the multi-rank plumbing with a stub step (no GPU, no engine)")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it: this process becomes the parent of N ranks (one per GPU) BEFORE
    # anything here touches the GPU, relays their output and exits with the worst rank's code (mdie_amd/launch.py)
    from mdie_amd import launch as LA
    if LA.needs_self_launch(args.gpus):
        sys.exit(LA.self_launch([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world      # under a launcher the launcher's world size is the truth
    if args.rehearse:
        return rehearse_main(args, rank, world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from mdie_amd import host as _host
    numa = _host.bind_to_gpu_numa(local)  # the launching thread next to its GPU (two-socket hosts; MDIE_NUMA_BIND=0 turns it off)
    args.runtime = {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG"), "bound_to_gpu_numa_node": bool(numa)}
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        LA.init_or_exit(dist.init_process_group, "nccl", device_id=dev)

    from mdie_amd import lib as L
    from models.cdan import CDAN
    from mdie_amd import synthetic as P  # synthetic-input recipe + seeded checkpoint (bit-identical to the oracle's generator)

    if args.workload == "routed":
        return routed_main(args, rank, world, dev, dist, P)

    B, S = args.batch, args.size
    sd = P.make_state_dict(42)
    net = CDAN(precision=args.precision)
    net.load_state_dict(sd, strict=True)
    net = net.eval().to(dev)
    x_cpu, clean_cpu = rank_inputs(P, rank, B, S)
    x = x_cpu.to(dev)
    y = torch.empty_like(x)
    eng = net._engine(dev)

    def step():
        eng.forward(x, out=y)

    if args.no_graph:
        args.launch = "eager"
    graph = None
    with torch.no_grad():
        # untimed warm-up: the host TIMES the bit-identical forms of encoder.conv4 / the dense1 branch for this batch shape (an explicit call:
        # forward() itself never times or synchronises anything and would run the static default form)
        if os.environ.get("MDIE_BENCH_TUNE", "1") != "0":
            eng.tune(x)
        step()
        torch.cuda.synchronize(dev)
        if args.launch != "eager":
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
        run = graph.replay if graph is not None else step
        if args.launch == "auto":
            # Both forms enqueue the same launches; which one keeps the GPU busier depends on the box (graph replay pays a
            # few tens of us of inter-node latency per step, eager launching needs a host that stays ahead of the GPU).
            # The warmup steps are run in each form and the faster one is timed.
            def time_form(fn):
                n = max(5, args.warmup)
                for _ in range(2):
                    fn()
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize(dev)
                return (time.perf_counter() - t) / n
            # (three alternating rounds, best of each: the two forms are 1 % apart and a single round of 10 steps is noisier)
            t_graph = t_eager = float("inf")
            for _ in range(3):
                t_graph, t_eager = min(t_graph, time_form(graph.replay)), min(t_eager, time_form(step))
            if t_eager < t_graph:
                run, graph = step, None
        else:
            for _ in range(args.warmup):
                run()

        elapsed = timed_region(run, args.steps, dev, dist)

        # ---- roofline: per-launch HIP events on the launch stream (instrumented mode, eager) ------------------
        prof = {}
        if rank == 0:
            # every launch reports its kind and ITS share of the SURVEY 8d model (mdie_launch_info, from the engine's own launch
            # list): a kind's bytes are the bytes of launches that exist (the folded 1x1 transition of decoder.final_dense is
            # booked with the four 3x3 launches that do its work)
            reps = 5
            for _ in range(reps):
                _, extras = eng.forward(x, out=y, profile=True)
                for (kind, ms), (_, by, fl) in zip(extras["launches"], extras["launch_info"]):
                    k = "cbam" if kind.startswith("cbam") else kind
                    p_ = prof.setdefault(k, [0, 0.0, 0.0, 0.0])
                    p_[0] += 1
                    p_[1] += ms
                    p_[2] += by
                    p_[3] += fl
            for k in prof:
                prof[k] = (prof[k][0] // reps, prof[k][1] / reps, prof[k][2] / reps, prof[k][3] / reps)

    # under world > 1 every rank reports the form ITS box timed as fastest (the forms are bit-identical; boxes differ in which one wins)
    forms = None
    if dist is not None:
        forms = [None] * world
        dist.all_gather_object(forms, getattr(eng, "tuned", {"form": eng.form(B, S, S), "untimed": True}).get("form"))
    if rank != 0:
        if dist is not None:
            graph = run = None
            _host.shutdown_distributed()      # (no graph of this process holds a collective; the order is the library's one teardown order all the same)
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    esz = 4 if args.precision == "fp32" else 2
    alg_bytes = L.lib.mdie_cdan_algorithmic_bytes(B, S, S, esz)
    flops = L.lib.mdie_cdan_flops(B, S, S)
    kernel_ms = sum(v[1] for v in prof.values())
    model_bytes = sum(v[2] for v in prof.values())
    # the per-launch model must sum to the whole-forward figure (asserted in tests/test_gpu_parity.py); here a mismatch -- an ablation
    # build that skips launches, a new fused path booked wrongly -- is REPORTED in the line, it does not throw a finished measurement away
    model_mismatch = None if abs(model_bytes - alg_bytes) <= 1e-6 * alg_bytes else {"per_launch_sum": model_bytes, "mdie_cdan_algorithmic_bytes": alg_bytes}
    per_kernel = {}
    for k, (n, ms, b, f) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        per_kernel[k] = {"launches": n, "ms": round(ms, 4), "alg_GB": round(b / 1e9, 4), "GBps": round(b / ms / 1e6, 1) if ms else None,
                         "hbm_frac": round(b / ms / 1e6 / HBM_PEAK_GBS, 4) if ms else None,
                         "TFLOPs": round(f / ms / 1e9, 1) if ms and f else None}
    dom = max(prof.items(), key=lambda kv: kv[1][1])[0] if prof else None
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms else None
    # HBM bytes per step from the committed rocprofv3 PMC passes of this exact workload (tools/traffic_from_pmc.py;
    # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) -- bench.py cannot run the profiler itself
    traffic, traffic_note = None, "no profiles/traffic_*.json for this workload"
    tfile = os.path.join(ROOT, "profiles", f"traffic_{args.precision}_b{B}_{S}.json")
    if os.path.exists(tfile):
        with open(tfile) as f:
            tj = json.load(f)
        if tj.get("kernel_source_sha16") == source_sha16():
            traffic, traffic_note = tj.get("hbm_bytes_per_step"), f"{os.path.basename(tfile)} (rocprofv3 PMC passes on this source, sha16 {tj.get('kernel_source_sha16')})"
        else:   # never echo counters measured on other kernels
            traffic_note = (f"{os.path.basename(tfile)} was measured on kernel source {tj.get('kernel_source_sha16')}, this tree is {source_sha16()}: "
                            f"stale, not reported (was {tj.get('hbm_bytes_per_step')})")
    roofline = {"bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                # the same bytes over the timed step itself (side branches overlapped, launch gaps included) instead of the serial kernel time
                "frac_step": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_note,
                "kernel": "cdan_forward (all launches of one step)", "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_step": alg_bytes, "flops_per_step": flops,
                "mfma_frac": round(flops / (kernel_ms * 1e-3) / 1e12 / MFMA_PEAK_TF[args.precision], 4) if kernel_ms else None,
                "dominant_kernel": dom, "per_kernel": per_kernel}
    if model_mismatch is not None:
        roofline["model_mismatch"] = model_mismatch

    out = {"metric": "images/sec @256x256 bf16 (low_light CDAN)", "value": round(value, 2), "unit": "images/sec",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
           "config": {"workload": f"config/low_light.json CDAN forward (eval), {S}x{S}, batch {B}/GPU, {args.precision} storage + fp32 accumulate, "
                                  f"seeded random-init weights, synthetic low-light images resident in HBM",
                      "global_batch": B * world, "parallelism": f"batch-parallel x{world}, no collective",
                      "launch": (f"eager (one host call, {sum(v[0] for v in prof.values())} launches)" if graph is None else "hipGraph replay") + (", chosen in warmup" if args.launch == "auto" else ""),
                      "runtime": args.runtime,
                      # encoder.conv4 runs in one of three bit-identical forms, timed once per shape in the first (untimed) step: how long it
                      # holds its CUs while the DenseBlock branches wait for them (mdie_amd/engine.py: CdanEngine.tune)
                      "conv4_form": getattr(eng, "tuned", {"form": eng.form(B, S, S), "untimed": True}),
                      **({"conv4_form_per_rank": forms} if forms is not None else {})},
           "roofline": roofline}

    if not args.no_extra and world == 1:
        # side measurements, outside the timed region: the other storage types of the same workload and the cost of going
        # through nn.Module.forward (state_dict fingerprint walk + output allocation per call) instead of the C ABI call above
        extra = {}
        with torch.no_grad():
            t_mod = timed_loop(lambda: net(x), dev, 60, warm=10)   # (long enough for the launch queue to fill: 20 calls read 8 % slow)
            extra["module_forward_bf16" if args.precision == "bf16" else f"module_forward_{args.precision}"] = {
                "images_per_sec": round(B / t_mod, 1), "ms_per_step": round(t_mod * 1e3, 4),
                "what": "net(x) through models.cdan.CDAN.forward (eager launches), same batch"}
            for prec in ("fp16", "fp32", "bf16"):
                if prec == args.precision:
                    continue
                net.precision = prec
                e2 = net._engine(dev)
                y2 = torch.empty_like(x)
                t2 = timed_loop(lambda: e2.forward(x, out=y2), dev, 10 if prec == "fp32" else 20)
                extra[prec] = {"images_per_sec": round(B / t2, 1), "ms_per_step": round(t2 * 1e3, 4), "_y": y2}
            net.precision = args.precision
            # two batches in flight (two engines / workspaces on two streams, a serving loop's steady state): the tail of one
            # batch -- thin final_dense layers -- overlaps the head of the next.  NOT `value`, which keeps one batch in flight.
            from mdie_amd import engine as EG
            engs = [EG.CdanEngine(dev, args.precision).load(sd) for _ in range(2)]
            streams = [torch.cuda.Stream(dev) for _ in range(2)]
            ys = [torch.empty_like(x) for _ in range(2)]
            it = [0]

            def two():
                k = it[0] & 1
                it[0] += 1
                with torch.cuda.stream(streams[k]):
                    engs[k].forward(x, out=ys[k])
            for s_ in streams:
                s_.wait_stream(torch.cuda.current_stream(dev))
            t_if = timed_loop(two, dev, 40, warm=6)
            assert torch.equal(ys[0], y) and torch.equal(ys[1], y), "overlapped batches must reproduce the single-stream output"
            extra["two_batches_in_flight"] = {"images_per_sec": round(B / t_if, 1), "ms_per_step": round(t_if * 1e3, 4)}
            del engs, ys
        out["extra"] = extra

    if not args.no_cpu and world == 1:   # the CPU leg is a single-GPU-run feature (rank 0 at N=1 only)
        nb = min(args.cpu_batch, B)
        ref, cb = cpu_baseline(sd, x_cpu[:nb])
        yc = y[:nb].float().cpu()
        err = ((yc - ref).abs().max() / ref.abs().max()).item()
        # PSNR / SSIM side by side (torchmetrics-default formulas of utils/metrics_factory.py, computed by the HIP
        # metrics kernel on the GPU): engine output vs CPU-path output, and each against the clean targets
        from mdie_amd import pipeline as PL
        clean_d, ref_d = clean_cpu[:nb].to(dev), ref.to(dev)
        m_gc, m_gt, m_ct = (PL.psnr_ssim(a, b).cpu().tolist() for a, b in ((y[:nb], ref_d), (y[:nb], clean_d), (ref_d, clean_d)))
        for prec, e in out.get("extra", {}).items():      # the other storage types against the same CPU output
            if "_y" in e:
                ye = e.pop("_y")[:nb].float().cpu()
                e["max_abs_err_over_max_vs_cpu"] = round(((ye - ref).abs().max() / ref.abs().max()).item(), 7)
        cb["parity"] = {"max_abs_err_over_max": round(err, 6),
                        "gpu_vs_cpu": {"psnr_db": round(m_gc[0], 2), "ssim": round(m_gc[1], 5)},
                        "gpu_vs_clean": {"psnr_db": round(m_gt[0], 2), "ssim": round(m_gt[1], 5)},
                        "cpu_vs_clean": {"psnr_db": round(m_ct[0], 2), "ssim": round(m_ct[1], 5)}}
        out["cpu_baseline"] = cb
    for e in out.get("extra", {}).values():
        e.pop("_y", None)
    print(json.dumps(out))
    if dist is not None:
        graph = run = None
        _host.shutdown_distributed()


if __name__ == "__main__":
    main()
