#!/bin/bash
# Regenerates the artefacts kept under profiles/ on a GPU box, into gpurun_out/<tag>/: the bench line
# alone and under rocprofv3 --kernel-trace --stats, the per-level report, the kernel microbenchmark,
# separate FETCH_SIZE / WRITE_SIZE passes over it (-> pmc_traffic.json, which the bench lines then read), the other configurations, the
# solve to tolerance, the one-GPU slab proxy, the bench lines of the other workloads and the 2- and
# 8-rank rehearsals.  Every GPU step runs under its own time limit; the script stops at the
# first failure.
# Usage: tools/refresh_profiles.sh <tag>
set -u
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$root" || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python3 "$root/bench.py" --steps 20 --warmup 3 > "$out/bench_under_rocprof.json" 2> "$out/rocprof.err" || exit 1
cd "$root" || exit 1
cp "$out"/trace/bench_kernel_stats.csv "$out/bench_kernel_stats.csv" || exit 1
python3 tools/trace_summary.py "$out/trace" > "$out/bench_kernel_trace_summary.txt" || exit 1
python3 tools/level_report.py "$out/trace" 513 4 > "$out/per_level.txt" || exit 1
# the same per-level report for the double-precision grids
for n in 513 1025; do
  cd /tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out/trace${n}d" -o bench -- python3 "$root/bench.py" --config 3d${n}f64 --steps 6 --warmup 2 --no-cpu > "$out/bench_3d${n}f64_under_rocprof.json" 2>> "$out/rocprof.err" || exit 1
  cd "$root" || exit 1
  python3 tools/level_report.py "$out/trace${n}d" $n 8 > "$out/per_level_${n}f64.txt" || exit 1
done
timeout -k 10 200 python3 tools/kbench.py > "$out/kbench513.txt" 2>&1 || exit 1
rm -f "$out/pmc_traffic.json"
bash tools/pmc.sh "$out/pmc" fetch,write || exit 1
python3 tools/pmc_summary.py "$out/pmc" > "$out/pmc_fetch_write_kbench513.txt" || exit 1
python3 tools/pmc_traffic.py "$out/pmc" 513 f32 "$out/pmc_traffic.json" || exit 1
# the dominant kernels of the other workloads: the same two counter passes at their sizes
bash tools/pmc.sh "$out/pmc257" fetch,write --n 257 --only "smooth_sweep x2" || exit 1
python3 tools/pmc_traffic.py "$out/pmc257" 257 f32 "$out/pmc_traffic.json" || exit 1
bash tools/pmc.sh "$out/pmc513d" fetch,write --f64 --only "smooth_sweep x2" || exit 1
python3 tools/pmc_traffic.py "$out/pmc513d" 513 f64 "$out/pmc_traffic.json" || exit 1
bash tools/pmc.sh "$out/pmc1025d" fetch,write --f64 --n 1025 --only "smooth_sweep x2" || exit 1
python3 tools/pmc_traffic.py "$out/pmc1025d" 1025 f64 "$out/pmc_traffic.json" || exit 1
bash tools/pmc.sh "$out/pmc2d4097" fetch,write --dim 2 --n 4097 || exit 1
python3 tools/pmc_traffic.py "$out/pmc2d4097" 4097 f32 "$out/pmc_traffic.json" || exit 1
bash tools/pmc.sh "$out/pmc2d257" fetch,write --dim 2 --n 257 || exit 1
python3 tools/pmc_traffic.py "$out/pmc2d257" 257 f32 "$out/pmc_traffic.json" || exit 1
# the counter passes above are what the bench lines take their traffic from
cp "$out/pmc_traffic.json" profiles/pmc_traffic.json || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err" || exit 1
timeout -k 10 400 python3 tools/bench_configs.py > "$out/configs.txt" 2> "$out/configs.err" || exit 1
timeout -k 10 400 python3 tools/solve_to_tol.py > "$out/solve_to_tolerance.txt" 2>&1 || exit 1
timeout -k 10 300 python3 tools/slab_proxy.py > "$out/slab_proxy_513.txt" 2> "$out/slab_proxy.err" || exit 1
timeout -k 10 300 python3 tools/slab_proxy.py --weak > "$out/slab_proxy_weak.txt" 2>> "$out/slab_proxy.err" || exit 1
timeout -k 10 600 python3 tools/slab_proxy.py --n 1025 --f64 --cycles 4 > "$out/slab_proxy_1025f64.txt" 2>> "$out/slab_proxy.err" || exit 1
for c in 2d257 2d4097mixed 3d257neumann 3d513f64 3d1025f64; do
  timeout -k 10 400 python3 bench.py --config $c > "$out/bench_$c.json" 2> "$out/bench_$c.err" || exit 1
done
timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu > "$out/rehearsal_gpus2.json" 2> "$out/rehearsal_gpus2.err" || exit 1
timeout -k 10 400 python3 bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu > "$out/rehearsal_gpus8.json" 2> "$out/rehearsal_gpus8.err" || exit 1
timeout -k 10 60 tools/ubench/valu.bin > "$out/valu_rates.txt" 2>&1 || exit 1
rm -rf "$out"/trace*/*.db "$out"/trace*/*/*.db "$out"/pmc*/*/*.db
echo refreshed
