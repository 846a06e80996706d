#!/bin/bash
# Everything the numbers in DESIGN.md and profiles/LEDGER.md come from, in one pass on a one-GPU MI355X box:  tools/measure_all.sh <tag>
# writes gpurun_out/<tag>/ ; copy what is to be kept into profiles/ afterwards (tools/collect_profiles.py <tag>).
# rocprofv3 runs take the program itself after `--` (python3 ...), PMC passes are separate from each other and from --stats.
set -u
TAG=${1:-final}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
PART=${2:-all}     # benches | profiler | all   (a gpurun call is limited to 20 minutes: the two halves fit one call each)
run() { local name=$1; shift; echo "== $name"; timeout -k 10 600 "$@" > "$OUT/$name.txt" 2> "$OUT/$name.err" || echo "   ($name: exit $?)"; }

if [ "$PART" != profiler ]; then
run bench_bf16 python bench.py
run bench_bf16_graph python bench.py --launch graph --no-cpu --no-extra
run bench_fp16 python bench.py --precision fp16 --no-cpu --no-extra
run bench_fp32 python bench.py --precision fp32 --no-cpu --no-extra
run layers_bf16 python tools/profile_layers.py bf16
run layers_fp16 python tools/profile_layers.py fp16
run configs_large_fp16 python tools/bench_configs.py large fp16
run configs_large_bf16 python tools/bench_configs.py large bf16
run configs_routed_bf16 python tools/bench_configs.py routed bf16
run train_bf16_b8_512 python tools/bench_train.py bf16 8 512
run train_bf16_b8_256 python tools/bench_train.py bf16 8 256
run train_fp16_b8_512 python tools/bench_train.py fp16 8 512
run train_fp16_b8_256 python tools/bench_train.py fp16 8 256
run e2e_bf16 python tools/bench_e2e.py bf16
run inflight_bf16 python tools/bench_inflight.py 1 2 3
mkdir -p build && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/bandwidth.hip -o build/bandwidth && run achievable_bandwidth_hip build/bandwidth
run tail_fold_ab_bf16 python tools/bench_tail_ab.py bf16
run tail_fold_ab_fp16 python tools/bench_tail_ab.py fp16
run layers_bf16_general_tail env GENERAL_TAIL=1 python tools/profile_layers.py bf16
run batch_independence_probe python tools/probe_concurrent.py bf16 4 5
run conv_microbench python tools/bench_conv.py bf16 conv2 conv3 conv4 dec1 dec2 dec3 first d1l0 d1l3 d2l3 d3l3 fl3 ftr
# what each stage costs the step (needs libmdie_hip_ablate.so: `tools/ablate.sh build` before the gpurun call)
[ -f "$ROOT/multi-degradation-image-enhancement_amd/libmdie_hip_ablate.so" ] && run ablate_bf16 python tools/ablate.py bf16 3
run train_host_enqueue python tools/host_time_train.py bf16 8 512
run train_ddp1_bf16_b8_512 env MDIE_DDP_SINGLE=1 python tools/bench_train.py bf16 8 512 charbonnier:1,ssim:0.5 both
run train_ddp1_bf16_b8_256 env MDIE_DDP_SINGLE=1 python tools/bench_train.py bf16 8 256 charbonnier:1,ssim:0.5 both
run bench_routed python bench.py --workload routed
fi
if [ "$PART" = benches ]; then echo "done (benches): $(ls "$OUT" | wc -l) files"; exit 0; fi

cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 kernel stats"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --launch eager --steps 20 --warmup 5 --no-cpu --no-extra > "$OUT/stats_run.log" 2>&1
echo "== rocprofv3 pmc FETCH_SIZE"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc/fetch" -- python3 "$ROOT/bench.py" --launch eager --steps 10 --warmup 2 --no-cpu --no-extra > "$OUT/pmc_fetch.log" 2>&1
echo "== rocprofv3 pmc WRITE_SIZE"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc/write" -- python3 "$ROOT/bench.py" --launch eager --steps 10 --warmup 2 --no-cpu --no-extra > "$OUT/pmc_write.log" 2>&1
echo "== rocprofv3 pmc MFMA busy"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- python3 "$ROOT/bench.py" --launch eager --steps 10 --warmup 2 --no-cpu --no-extra > "$OUT/pmc_mfma.log" 2>&1
echo "== rocprofv3 training step stats"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_stats" -- python3 "$ROOT/tools/bench_train.py" bf16 8 512 charbonnier:1,ssim:0.5 eager > "$OUT/train_stats_run.log" 2>&1
cd "$ROOT"
python tools/traffic_from_pmc.py "$OUT/pmc" "$OUT/traffic_bf16_b32_256.json" > "$OUT/traffic_summary.txt" 2>&1
python tools/mfma_busy_from_pmc.py "$OUT/pmc_mfma" "$OUT/mfma_busy.txt" > /dev/null 2>&1
python tools/train_timeline.py "$OUT/train_stats" "$OUT/train_timeline.txt" > /dev/null 2>&1
python tools/infer_timeline.py "$OUT/stats" "$OUT/infer_timeline.txt" > /dev/null 2>&1
echo "done: $(ls "$OUT" | wc -l) files"
