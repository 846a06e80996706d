#!/usr/bin/env python3
"""What the GPU's clocks and power do under the bench workload: rocm-smi sampled from a side thread while the forward (or one layer shape) loops.
  python tools/power_probe.py [seconds]     -> one line per sample: socket power (W), sclk, mclk, temperature, busy %"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdie_amd import engine as E
from mdie_amd import synthetic as P

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
eng = E.CdanEngine("cuda", "bf16").load(P.make_state_dict(42))
x = P.lowlight_batch(1000, 32, 256, 256)[0].cuda()
y = torch.empty_like(x)
samples, stop = [], False


def smi():
    while not stop:
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showuse", "--showtemp", "--showmaxpower", "--json"],
                                 capture_output=True, text=True, timeout=10).stdout
            samples.append((time.perf_counter(), out))
        except Exception as e:     # (the tool may be missing or refused on a box: the probe then reports nothing)
            samples.append((time.perf_counter(), f"ERR {e}"))
        time.sleep(0.5)


def sample_idle():
    try:
        return subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:
        return f"ERR {e}"


print("idle:", sample_idle().strip()[:1500])
th = threading.Thread(target=smi, daemon=True)
th.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(200):
        eng.forward(x, out=y)
    torch.cuda.synchronize()
    n += 200
dt = time.perf_counter() - t0
stop = True
th.join(timeout=5)
print(f"{n} forwards of 32 x 256x256 bf16 in {dt:.2f} s = {32 * n / dt:.0f} img/s")
for t, s in samples:
    print(f"t={t - t0:5.1f}s  {s.strip()[:1500]}")
