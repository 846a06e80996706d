#!/usr/bin/env python3
"""Copy the summaries of one tools/measure_all.sh pass (gpurun_out/<tag>/) into profiles/ under round-prefixed names.
  python tools/collect_profiles.py <tag> <prefix>      e.g.  collect_profiles.py r02n r02n"""
import glob, json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pre = sys.argv[1], sys.argv[2]
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")


def cp(name, out):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{pre}_{out}"))
        return True
    print("missing:", name)
    return False


for prec in ("bf16", "fp16", "fp32"):
    cp(f"bench_{prec}.txt", f"bench_{prec}.json")
cp("bench_bf16_graph.txt", "bench_bf16_graph.json")
cp("layers_bf16.txt", "layers_bf16.txt")
cp("layers_fp16.txt", "layers_fp16.txt")
cp("mfma_busy.txt", "mfma_busy_bf16_b32.txt")
cp("conv_microbench.txt", "conv_microbench_bf16_b32.txt")
cp("achievable_bandwidth_hip.txt", "achievable_bandwidth_hip.txt")
cp("tail_fold_ab_bf16.txt", "tail_fold_ab_bf16.txt")
cp("tail_fold_ab_fp16.txt", "tail_fold_ab_fp16.txt")
cp("layers_bf16_general_tail.txt", "layers_bf16_general_tail.txt")
cp("batch_independence_probe.txt", "batch_independence_probe.txt")
cp("train_timeline.txt", "train_timeline_bf16_b8_512.txt")
cp("infer_timeline.txt", "infer_timeline_bf16_b32.txt")
cp("ablate_bf16.txt", "ablate_bf16.txt")
cp("bench_routed.txt", "bench_routed.json")
with open(os.path.join(dst, f"{pre}_configs.txt"), "w") as f:
    f.write("# BASELINE configs[2] (training step shape), [3] (routed, 9 weight sets) and [4] (1024x1024, its stated dtype fp16), plus the\n"
            "# PCIe-inclusive and batches-in-flight serving rates: output lines of tools/bench_train.py, bench_configs.py, bench_e2e.py,\n"
            "# bench_inflight.py on one MI355X (tools/measure_all.sh)\n")
    for name in ("configs_large_fp16", "configs_large_bf16", "configs_routed_bf16", "train_bf16_b8_512", "train_bf16_b8_256", "train_fp16_b8_512", "train_fp16_b8_256",
                 "e2e_bf16", "inflight_bf16", "train_ddp1_bf16_b8_512", "train_ddp1_bf16_b8_256", "train_host_enqueue"):
        p = os.path.join(src, name + ".txt")
        if os.path.exists(p):
            f.write("".join(l for l in open(p) if "amdgpu.ids" not in l))
for sub, out in (("stats", "kernel_stats_bf16_b32.csv"), ("train_stats", "train_bf16_b8_512_kernel_stats.csv")):
    fs = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if fs:
        shutil.copy(fs[0], os.path.join(dst, f"{pre}_{out}"))
    else:
        print("missing kernel stats:", sub)
t = os.path.join(src, "traffic_bf16_b32_256.json")
if os.path.exists(t):
    shutil.copy(t, os.path.join(dst, f"{pre}_traffic_bf16_b32.json"))
    shutil.copy(t, os.path.join(dst, "traffic_bf16_b32_256.json"))     # what bench.py echoes (only while the kernel sources match)
    print("traffic:", json.load(open(t)).get("hbm_bytes_per_step"), json.load(open(t)).get("kernel_source_sha16"))
