#!/usr/bin/env python3
"""k engines on k streams, each a sub-batch of one batch, run concurrently; every tap and the output compared with the same
sub-batch run alone.  python tools/probe_concurrent.py [prec] [k] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdie_amd import engine as EG
from mdie_amd import synthetic as P

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 20
B, S = 32, 256
dev = torch.device("cuda", 0)
sd = P.make_state_dict(42)
x = P.lowlight_batch(1, B, S, S)[0].to(dev)
xs = [c.contiguous() for c in x.chunk(k)]
engs = [EG.CdanEngine(dev, prec).load(sd) for _ in range(k)]
solo = []
for i in range(k):
    y, ex = engs[i].forward(xs[i], want_taps=True)
    torch.cuda.synchronize()
    solo.append((y.clone(), {n: t.clone() for n, t in ex["taps"].items()}))
streams = [torch.cuda.Stream(dev) for _ in range(k)]
bad_total = {}
for r in range(rounds):
    outs = [None] * k
    for i in range(k):
        with torch.cuda.stream(streams[i]):
            outs[i] = engs[i].forward(xs[i], out=torch.empty_like(xs[i]), want_taps=True)
    torch.cuda.synchronize()
    for i in range(k):
        y, ex = outs[i]
        bad = [n for n in ex["taps"] if not torch.equal(ex["taps"][n], solo[i][1][n])]
        if bad or not torch.equal(y, solo[i][0]):
            first = bad[0] if bad else "output"
            a = ex["taps"][first] if bad else y
            b = solo[i][1][first] if bad else solo[i][0]
            d = (a - b).abs()
            idx = (d > 0).nonzero()
            print(f"round {r} engine {i}: differing taps {bad} | first '{first}': {idx.shape[0]} elements, max |d| {d.max().item():.3e}, "
                  f"images {sorted(set(idx[:, 0].tolist()))[:8]}, channels {sorted(set(idx[:, 1].tolist()))[:12]}, rows {sorted(set(idx[:, 2].tolist()))[:12]}, cols {sorted(set(idx[:, 3].tolist()))[:16]}", flush=True)
            bad_total[first] = bad_total.get(first, 0) + 1
print("summary:", bad_total if bad_total else f"all {rounds} rounds x {k} engines identical to the solo runs")

# the same sub-batches against the WHOLE batch through one engine (bitwise batch independence at every position)
whole = EG.CdanEngine(dev, prec).load(sd)
yw, exw = whole.forward(x, want_taps=True)
torch.cuda.synchronize()
nb = B // k
for i in range(k):
    bad = [n for n in solo[i][1] if not torch.equal(solo[i][1][n], exw["taps"][n][i * nb:(i + 1) * nb])]
    if bad or not torch.equal(solo[i][0], yw[i * nb:(i + 1) * nb]):
        first = bad[0] if bad else "output"
        a = solo[i][1][first] if bad else solo[i][0]
        b = (exw["taps"][first] if bad else yw)[i * nb:(i + 1) * nb]
        d = (a - b).abs()
        idx = (d > 0).nonzero()
        print(f"sub-batch {i} alone vs in the batch: differing taps {bad} | first '{first}': {idx.shape[0]} elements, max |d| {d.max().item():.3e}, images {sorted(set(idx[:, 0].tolist()))}, "
              f"channels {sorted(set(idx[:, 1].tolist()))[:10]}.. ({len(set(idx[:, 1].tolist()))}), rows {sorted(set(idx[:, 2].tolist()))[:8]}.. ({len(set(idx[:, 2].tolist()))}), cols ({len(set(idx[:, 3].tolist()))})")
    else:
        print(f"sub-batch {i} alone == its images in the batch, every tap")
