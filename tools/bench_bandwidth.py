#!/usr/bin/env python3
"""What the memory system of this box delivers to plain streaming kernels (torch copy_ / zero_ / add over 64 MB ... 1 GB):
the denominator to read the per-kernel TB/s figures of DESIGN.md against (profiles/r02s_achievable_bandwidth_torch_streaming.txt).
`sum` goes through a slow reduction kernel and says nothing about bandwidth."""
import torch, time
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for mb in (64, 256, 1024):
    x = torch.empty(mb * 1024 * 1024 // 2, dtype=torch.bfloat16, device="cuda").normal_()
    y = torch.empty_like(x)
    us = t(lambda: y.copy_(x)); print(f"copy {mb} MB -> {mb} MB: {us:.1f} us = {2 * mb * 1.048576 / us * 1e3:.0f} GB/s total")
    us = t(lambda: torch.sum(x.view(torch.int16), dtype=torch.int64)); print(f"read {mb} MB (sum): {us:.1f} us = {mb * 1.048576 / us * 1e3:.0f} GB/s")
    us = t(lambda: y.zero_()); print(f"write {mb} MB (zero): {us:.1f} us = {mb * 1.048576 / us * 1e3:.0f} GB/s")
    z = torch.empty(mb * 1024 * 1024 // 2 // 4, dtype=torch.bfloat16, device="cuda")
    us = t(lambda: torch.add(x[: z.numel()], x[z.numel(): 2 * z.numel()], out=z)); print(f"read {mb//2} MB write {mb//4} MB (add): {us:.1f} us = {0.75 * mb * 1.048576 / us * 1e3:.0f} GB/s")
