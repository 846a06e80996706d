"""world_size-2 rehearsal (gloo, CPU) of bench.py's multi-rank logic: ranks own different
batches (no data-path collective) and the reported time is the max over ranks."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """a rendezvous port the kernel hands out (a fixed 29500-range port collides with other torch jobs on a shared host)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle import params as P
    t = bench.max_over_ranks(1.0 + rank, dist, "cpu")
    x, _ = bench.rank_inputs(P, rank, 2, 16)
    sums = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sums, x.double().sum().reshape(1))
    q.put((rank, t, [float(s) for s in sums]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, t, sums in res:
        assert t == 2.0                      # max over ranks, identical on every rank
        assert sums[0] != sums[1]            # ranks hold different batches


def _bucket_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mdie_amd.train import GradBuckets
    torch.manual_seed(0)                       # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(12, 20), torch.nn.ReLU(), torch.nn.Linear(20, 7), torch.nn.Linear(7, 3))
    buckets = GradBuckets(net.parameters(), n_buckets=3)
    g = torch.Generator().manual_seed(100 + rank)   # different shard per rank
    x = torch.randn(5, 12, generator=g)
    net(x).pow(2).mean().backward()
    local = [p.grad.clone() for p in net.parameters()]
    buckets.finish()
    gathered = []
    for p, l in zip(net.parameters(), local):
        parts = [torch.zeros_like(l) for _ in range(world)]
        dist.all_gather(parts, l)
        gathered.append(torch.allclose(p.grad, sum(parts) / world, atol=1e-7))
    # a second step reuses the buckets
    net.zero_grad()
    net(x).pow(2).mean().backward()
    buckets.finish()
    q.put((rank, all(gathered), len(buckets.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_two_ranks():
    """SURVEY.md 8e / fixture c-4 semantics: after the exchange every rank holds the MEAN of the per-shard gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, nb in res:
        assert ok and nb >= 2


def _loader_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mdie_amd import host as H
    r, w, _ = H.init_distributed("gloo")          # what Model.__init__ does under torchrun (RCCL on a GPU node)
    assert (r, w) == (rank, world) and dist.is_initialized()
    data = torch.arange(10).float().reshape(10, 1)
    loader = H.make_dataloader(torch.utils.data.TensorDataset(data), {"batch_size": 2, "shuffle": True, "num_workers": 0})
    loader.sampler.set_epoch(0)
    seen = sorted(int(v) for (b,) in loader for v in b.reshape(-1))
    q.put((rank, seen))
    dist.barrier()
    dist.destroy_process_group()


def test_driver_shards_the_dataset_under_torchrun_env():
    """`run.py` under torchrun: every rank joins the group and reads a disjoint shard; together they cover the dataset"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loader_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len(res[0]) == len(res[1]) == 5 and sorted(res[0] + res[1]) == list(range(10))


def _ddp_fixture_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from mdie_amd.train import GradBuckets
    from models.cdan import CDAN
    z = np.load(os.path.join(ROOT, "tests", "golden", "ddp_2shard_32.npz"))
    net = CDAN()                                  # the 140 parameters of the real network (CPU tensors: only the exchange runs here)
    named = dict(net.named_parameters())
    buckets = GradBuckets(net.parameters())       # the production bucketing
    mine = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"g{rank}:")}
    # deliver this rank's per-shard reference gradients through autograd, so the post-accumulate hooks fire as in a real backward
    sum((named[k] * g).sum() for k, g in mine.items()).backward()
    assert all(torch.equal(named[k].grad, g) for k, g in mine.items())
    buckets.finish()
    ok = True
    for k in mine:
        mean = (torch.from_numpy(z["g0:" + k]).double() + torch.from_numpy(z["g1:" + k]).double()) / 2
        ok = ok and torch.allclose(named[k].grad.double(), mean, rtol=0, atol=1e-7 * float(mean.abs().max()) + 1e-12)
    untouched = [k for k, p in named.items() if k not in mine and p.grad is not None and float(p.grad.abs().max()) != 0.0]
    q.put((rank, ok, len(buckets.buckets), untouched))
    dist.barrier()
    dist.destroy_process_group()


def test_gradbuckets_deliver_the_reference_two_shard_mean():
    """fixture c-4 (tests/golden/ddp_2shard_32.npz, produced by the reference): two ranks hold the reference's per-shard
    gradients of the real parameter set; after GradBuckets' bucketed all-reduce every rank holds their mean."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_fixture_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, nb, untouched in res:
        assert ok and nb == 5 and not untouched     # tail..decoder.conv2 | decoder.conv1 | CBAMs, encoder dense blocks | encoder.conv4 | encoder.conv3..1


def _sink_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mdie_amd.train as T

    class Scale(torch.autograd.Function):
        """y = x * w with a sink-aware backward: dw is WRITTEN into the tensor train._gout hands out (the parameter's slice of its
        all-reduce bucket while a GradBuckets is active) -- what every training Function of mdie_amd/train.py does"""
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x)
            ctx.wparam = w
            return x * w

        @staticmethod
        def backward(ctx, dy):
            (x,) = ctx.saved_tensors
            dw = T._gout(ctx.wparam, ctx.wparam.shape, dy.device)
            torch.sum(dy * x, dim=0, out=dw)
            return None, dw

    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (7, 300, 5, 64)]
    frozen_bias = torch.nn.Parameter(torch.zeros(9))         # receives an exactly-zero gradient (a bias in front of a batch-statistic BatchNorm)
    buckets = T.GradBuckets(params + [frozen_bias], n_buckets=2)
    g = torch.Generator().manual_seed(100 + rank)
    xs = [torch.randn(4, p.numel(), generator=g) for p in params]
    out = {}
    for step in range(2):                                    # the second step reuses the slices (zero_grad(set_to_none=True) in between)
        for p in params + [frozen_bias]:
            p.grad = None
        loss = sum(Scale.apply(x, p).sum() for x, p in zip(xs, params))
        loss = loss + (frozen_bias * 0).sum()                # autograd's own node: a gradient that is NOT a bucket view
        loss.backward()
        views_before = [buckets.is_view(p, p.grad) for p in params]
        buckets.finish()
        local = [x.sum(0) for x in xs]
        means = []
        for l in local:
            parts = [torch.zeros_like(l) for _ in range(world)]
            dist.all_gather(parts, l)
            means.append(sum(parts) / world)
        out[step] = (all(views_before), all(buckets.is_view(p, p.grad) for p in params + [frozen_bias]),
                     all(torch.allclose(p.grad, m, atol=1e-6) for p, m in zip(params, means)), float(frozen_bias.grad.abs().max()))
    copies_two_steps = buckets.copies_in
    # ---- one parameter, two gradients in ONE backward (the network applied twice in a graph, weight sharing): `.grad` is still None at
    # the second producer while the first gradient -- the slice -- waits un-summed in autograd's input buffer.  The second sighting must
    # get a fresh tensor (GradBuckets.claim); writing the slice again would make autograd add the slice to itself (2 x the second term)
    for p in params + [frozen_bias]:
        p.grad = None
    xa, xb = xs[1], torch.randn(4, params[1].numel(), generator=g)
    buckets.trace = []
    (Scale.apply(xa, params[1]).sum() + 3.0 * Scale.apply(xb, params[1]).sum() + sum(Scale.apply(x, p).sum() for x, p in zip(xs, params))).backward()
    trace = list(buckets.trace)
    buckets.trace = None
    twice_local = 2 * xa.sum(0) + 3.0 * xb.sum(0)
    twice_ok_before = torch.allclose(params[1].grad, twice_local, atol=1e-5)
    buckets.finish()
    parts = [torch.zeros_like(twice_local) for _ in range(world)]
    dist.all_gather(parts, twice_local)
    twice_ok_after = torch.allclose(params[1].grad, sum(parts) / world, atol=1e-5) and buckets.is_view(params[1], params[1].grad)
    # ---- overlap: the first bucket's collective is issued while later gradients are still to come (parameters are bucketed in reverse
    # registration order; the last-registered parameters' gradients arrive first)
    first_launch = next(i for i, e in enumerate(trace) if e[0] == "launch")
    last_grad = max(i for i, e in enumerate(trace) if e[0] == "grad")
    # ---- outside an engine-run backward nothing is claimed: a direct call gets a fresh tensor, never bucket memory
    params[0].grad = None
    direct = T._gout(params[0], params[0].shape, params[0].device)
    q.put((rank, out, copies_two_steps, (twice_ok_before, twice_ok_after, first_launch < last_grad, len([e for e in trace if e[0] == "launch"]),
                                         not buckets.is_view(params[0], direct))))
    buckets.close()
    assert T._GRAD_SINK is None
    dist.barrier()
    dist.destroy_process_group()


def test_gradients_are_views_of_the_flat_buckets_two_ranks():
    """GradBuckets as the gradient SINK (the step being sharded: models/model.py:154-166): a sink-aware backward writes dW
    into the parameter's bucket slice and autograd adopts the view -- `.grad` is bucket memory before and after the exchange, the
    only gradient ever copied in is the one an ordinary autograd node produced, and every rank ends with the two-shard mean."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sink_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out, copies, (twice_before, twice_after, launched_before_last_grad, n_launch, direct_is_fresh) in res:
        for step in (0, 1):
            views_before, views_after, mean_ok, zero_max = out[step]
            assert views_before and views_after and mean_ok and zero_max == 0.0
        assert copies == 2          # frozen_bias's autograd-made gradient, once per step; the four sink-aware gradients never
        assert twice_before, "two gradients of one parameter in one backward: .grad must be their sum"
        assert twice_after, "... and after the exchange the two-rank mean of that sum, living in the bucket"
        assert launched_before_last_grad and n_launch >= 1, "the first bucket's all-reduce must be issued before the last gradient of backward arrives"
        assert direct_is_fresh, "outside a backward _gout must not hand out bucket memory"


def _routed_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from mdie_amd import engine as E
    labels = bench.routed_labels(3, 64)                  # the same list on every rank
    labels[5] = labels[17] = labels[40] = None           # images the router left alone
    mine = E.routed_shard(labels, rank, world, bench.ROUTED_TASKS)
    tasks = sorted({labels[i] for i in mine if labels[i] is not None})
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, tasks, E.routed_slice(65, rank, world)))
    t = bench.max_over_ranks(0.5 + rank, dist, "cpu")
    q.put((rank, gathered, t))
    dist.barrier()
    dist.destroy_process_group()


def test_routed_batch_is_dealt_by_task_two_ranks():
    """bench.py --workload routed (BASELINE configs[3]; classification/train_multilabel_classifier.py:251-253 labels the images): every
    image of the global batch runs on exactly one rank, a task's images all on the rank that owns the task (so a rank binds only its
    own weight sets), un-routed images are spread; the step time is the max over ranks.  No collective on the data path."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_routed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, gathered, t in res:
        (m0, t0, s0), (m1, t1, s1) = gathered
        assert sorted(m0 + m1) == list(range(64)) and not set(m0) & set(m1)
        assert not set(t0) & set(t1) and len(t0) + len(t1) == 9
        assert s0 + s1 == list(range(65)) and abs(len(s0) - len(s1)) <= 1     # chain mode: equal contiguous slices, labels play no part
        assert t == 1.5


def _run_bench(extra_args, env_drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, env=env, capture_output=True, text=True, timeout=300)


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (how the driver runs `--gpus 1`): the parent starts two ranks itself
    (mdie_amd/launch.py), exactly one JSON line comes out, it says n_gpus 2, and the ranks held different batches.  `--rehearse` swaps
    RCCL for gloo and the engine for a stub step; every line of the launch / rendezvous / fence / max-over-ranks plumbing is the real one."""
    import json
    r = _run_bench(["--gpus", "2", "--rehearse", "--steps", "3", "--warmup", "1", "--batch", "2", "--size", "16"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["rehearsal"] is True and rec["scaling"] == "weak"
    assert rec["config"]["launcher"] == "self-launched children" and rec["config"]["distinct_rank_batches"] == 2
    assert rec["config"]["global_batch"] == 4


def test_self_launch_returns_the_worst_rank_exit_code(tmp_path):
    """a rank that dies takes the job's exit code with it (and the surviving rank is not left behind)"""
    sys.path.insert(0, ROOT)
    from mdie_amd import launch as LA
    script = tmp_path / "rank.py"
    script.write_text("import os, sys\nsys.exit(7 if os.environ['RANK'] == '1' else 0)\n")
    assert LA.self_launch([str(script)], 2) == 7
    assert LA.needs_self_launch(2, env={}) and not LA.needs_self_launch(1, env={}) and not LA.needs_self_launch(8, env={"WORLD_SIZE": "8"})
    e = LA.child_env(3, 8, 1234, base={})
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"], e["MASTER_PORT"]) == ("3", "3", "8", "127.0.0.1", "1234")


def test_self_launch_kills_a_rank_that_ignores_sigterm_and_retries_a_taken_port(tmp_path):
    """(a) one rank fails, the other is stuck where SIGTERM does not reach it (a rank blocked inside a collective behaves like that): after
    the grace period it is terminated, then KILLED, and the job reports failure instead of hanging; (b) every failed rank reporting
    EADDRINUSE_EXIT (the rendezvous port was taken between free_port() and the bind) repeats the launch once on a fresh port."""
    import time
    sys.path.insert(0, ROOT)
    from mdie_amd import launch as LA
    stuck = tmp_path / "stuck.py"
    stuck.write_text("import os, signal, sys, time\n"
                     "if os.environ['RANK'] == '1':\n    sys.exit(5)\n"
                     "signal.signal(signal.SIGTERM, signal.SIG_IGN)\ntime.sleep(600)\n")
    t0 = time.monotonic()
    rc = LA.self_launch([str(stuck)], 2, grace_s=0.5, kill_after_s=0.5)
    assert rc != 0 and time.monotonic() - t0 < 30
    marker = tmp_path / "seen"
    retry = tmp_path / "retry.py"
    retry.write_text("import os, sys\n"
                     f"m = {str(marker)!r} + os.environ['RANK']\n"
                     "if not os.path.exists(m):\n    open(m, 'w').write(os.environ['MASTER_PORT'])\n"
                     f"    sys.exit({LA.EADDRINUSE_EXIT} if os.environ['RANK'] == '0' else 0)\n"
                     "sys.exit(0)\n")
    assert LA.self_launch([str(retry)], 2, grace_s=0.5) == 0


def test_bench_train_rehearsal_launches_exchanges_and_tears_down():
    """`tools/bench_train.py --gpus 2 --rehearse`: the training bench's own launch, rendezvous, GradBuckets exchange (hooks -> bucketed
    all-reduce -> finish) and host.shutdown_distributed teardown on gloo with a stand-in network -- so that the first multi-GPU run of
    the training curve cannot fail on argument plumbing (the step itself needs the GPU and is covered by the GPU tests)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_train.py"), "--gpus", "2", "--rehearse", "bf16", "2", "64"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("train[rehearse")]
    assert len(lines) == 1 and "B=2x2 64x64" in lines[0] and "exchange: hooks, 2 buckets, ranks agree: True" in lines[0] and "teardown: ok" in lines[0], r.stdout


def test_shutdown_distributed_order_two_ranks(tmp_path):
    """host.shutdown_distributed: graphs dropped, buckets closed, collected, synchronised, THEN the group destroyed -- and callable without
    a group.  Two gloo ranks: afterwards no process group, no gradient sink, the holder of captured steps empty."""
    import subprocess
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist\n"
        "from mdie_amd import host as H, train as T\n"
        "H.shutdown_distributed()            # no group: a no-op\n"
        "dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))\n"
        "net = torch.nn.Linear(4, 4)\n"
        "bk = T.GradBuckets(net.parameters(), n_buckets=1)\n"
        "net(torch.ones(2, 4)).sum().backward(); bk.finish()\n"
        "held = {'k': object()}\n"
        "H.shutdown_distributed(held, bk)\n"
        "assert not dist.is_initialized() and T._GRAD_SINK is None and held == {} and not bk._hooks\n"
        "print('RANK-OK', os.environ['RANK'])\n")
    sys.path.insert(0, ROOT)
    from mdie_amd import launch as LA
    assert LA.self_launch([str(script)], 2) == 0
