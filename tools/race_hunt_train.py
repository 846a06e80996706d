#!/usr/bin/env python3
"""Race hunt for train.WGRAD_STREAM (weight gradients on a side stream): N rounds of three optimizer steps with the side stream against
the single-stream schedule, bitwise, in ONE process; first, the single-stream step with the CBAM workspaces pre-filled with three
byte patterns (an uninitialised read would show).   python tools/race_hunt_train.py [bf16|fp16|fp32] [rounds] [B] [S]

What it found (round 3): with packed-f32 `op_sel` forms in cbam_train.hip's MFMA-free backward kernels, 4 of 12 rounds at 8x512x512
differed from the single-stream result -- always the bottleneck CBAM's channel-gate MLP gradients and everything downstream of them
(encoder.conv1-4) -- and 0 of 12 with that file built without packed f32 (csrc/Makefile NOPK; tools/isa_guard.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdie_amd.train as T
from models.cdan import CDAN
from oracle import params as P
PREC = sys.argv[1] if len(sys.argv) > 1 else "bf16"
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
S = int(sys.argv[4]) if len(sys.argv) > 4 else 512
sd = P.make_state_dict(42)
batches = [tuple(v.cuda() for v in P.lowlight_batch(5 + i, B, S, S)) for i in range(3)]
T.WGRAD_STREAM_MIN_PIXELS = 0


def run(side, bnred):
    T.WGRAD_STREAM = side
    T.BN_REDUCE_IN_DGRAD = bnred
    torch.manual_seed(123)
    net = CDAN(precision=PREC)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    out = []
    for x, t in batches:
        opt.zero_grad(set_to_none=True)
        loss = torch.sqrt((net(x) - t) ** 2 + 1e-6).mean()
        loss.backward()
        out.append((loss.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters()}))
        opt.step()
    torch.cuda.synchronize()
    return out


def diff(a, b):
    for i, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        bad = [n for n in ga if not torch.equal(ga[n], gb[n])]
        if bad or not torch.equal(la, lb):
            return f"step {i}: loss equal {torch.equal(la, lb)}, {len(bad)} gradients differ: {bad}"
    return None


for poison in (False, True):      # every uninitialised allocation of the step NaN / 0xFF-filled before use (train.POISON): a read of unwritten memory shows as NaN
    T.POISON = poison
    r = run(False, True)
    if not poison:
        base = r
    print(f"main stream, allocations {'poisoned' if poison else 'as allocated'}: {'identical' if diff(base, r) is None else diff(base, r)}", flush=True)
T.POISON = False
for bnred in (True, False):
    ref = run(False, bnred)
    for r in range(ROUNDS):
        d = diff(ref, run(True, bnred))
        print(f"bnred={bnred} round {r}: {'identical' if d is None else d}", flush=True)
    d = diff(ref, run(False, bnred))
    print(f"bnred={bnred} main stream again: {'identical' if d is None else d}", flush=True)
