#!/usr/bin/env python3
"""One layer shape through TWO builds of the library (MDIE_LIB of the child process): bitwise comparison of the outputs + timing.
  python tools/ab_layer.py <variant .so> [bf16|fp16|fp32] name ...        names as tools/bench_conv.py
Each build runs in its own child process (a process binds one library); outputs are exchanged through gpurun_out/."""
import os, sys, subprocess, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    import mdie_amd.engine as E
    import mdie_amd.lib as L
    sys.argv = [sys.argv[0]] + sys.argv[2:]
    prec, names = sys.argv[1], sys.argv[2:]
    dt = E.dtype_id(prec)
    td = E.TORCH_DTYPE[dt]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_conv
    ns = {"SHAPES": bench_conv.SHAPES}
    B = 32
    for name in names:
        segc, cout, H, ks, pool, pre = ns["SHAPES"][name]
        g = torch.Generator(device="cuda").manual_seed(7)
        segs = [torch.randn(B, H, H, c, device="cuda", generator=g).to(td) for c in segc]
        cin = sum(segc)
        gw = torch.Generator().manual_seed(8)
        w = E.pack_conv_weight(torch.randn(cout, cin, ks, ks, generator=gw) * 0.05, dt).cuda()
        s, t = torch.rand(cout, device="cuda", generator=g) + 0.5, torch.randn(cout, device="cuda", generator=g)
        ps = (torch.rand(cin, device="cuda", generator=g) + 0.5) if pre else None
        pt = torch.randn(cin, device="cuda", generator=g) if pre else None
        Ho = H // 2 if pool else H
        out = torch.empty(B, Ho, Ho, cout, device="cuda", dtype=td)
        fn = lambda: E.conv_fwd(segs, w, s, t, dtype=dt, ksize=ks, cout=cout, act=L.ACT_NONE if pre else L.ACT_RELU, pool=pool, pre_scale=ps, pre_shift=pt, out=out)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # (a CUDA graph of 20 launches: the Python wrapper costs more than a small kernel)
        gr = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            fn()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(20):
                    fn()
        torch.cuda.synchronize()
        gr.replay()
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        h = hashlib.sha256(out.cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]
        print(f"{name:6s} {us:8.1f} us  sha {h}", flush=True)
    sys.exit(0)

variant, prec, names = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for tag, lib in (("shipped", None), ("variant", os.path.abspath(variant))):
    env = dict(os.environ)
    if lib:
        env["MDIE_LIB"] = lib
    else:
        env.pop("MDIE_LIB", None)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", prec] + names, env=env, capture_output=True, text=True)
    if out.returncode != 0:
        print(out.stderr[-2000:])
        sys.exit(1)
    for line in out.stdout.splitlines():
        f = line.split()
        if len(f) == 5 and f[3] == "sha":
            res.setdefault(f[0], {})[tag] = (float(f[1]), f[4])
print(f"# {prec}, B = 32; shipped library vs {os.path.basename(variant)}; times: 100 launches inside a hipGraph")
for n in names:
    a, b = res[n]["shipped"], res[n]["variant"]
    print(f"{n:6s} shipped {a[0]:7.1f} us   variant {b[0]:7.1f} us   outputs {'bit-identical' if a[1] == b[1] else 'DIFFER'}")
