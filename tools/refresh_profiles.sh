#!/bin/bash
# Regenerates the measured files under profiles/ for round $1 (default r03) on the GPU box:  bash tools/refresh_profiles.sh r03
#   <rn>_bench_<w>.json          bench line of workload w (c4 = the default line with every extra; c2 / ds on their own)
#   <rn>_<w>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same bench command (extras run in a profiler-free child)
#   <rn>_pmc_<w>.json / .txt     PMC passes (tools/pmc_collect.sh)
#   <rn>_train_step_*.json       DAS3R-shaped optimisation step, unfused / fused / fused on spatially coherent depth maps
# Everything lands in gpurun_out/profiles/ (copied into profiles/ by hand after a look).  Every child is time-bounded.
R=$PWD; RN=${1:-r05}; OUT=$R/gpurun_out/profiles; mkdir -p $OUT
export PYTHONPATH=$R
DAS3R_BENCH_JOBS=full timeout 600 python bench.py --full-line 2>$OUT/${RN}_bench_c4.stderr | tail -1 > $OUT/${RN}_bench_c4.json
for w in c2 ds dsc; do
  timeout 200 python bench.py --workload $w --no-extras 2>/dev/null | tail -1 > $OUT/${RN}_bench_$w.json
done
cd /tmp; export TMPDIR=/tmp
for w in c4 c2 ds dsc; do
  rm -rf /tmp/kt_$w
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$w -o kt --output-format csv -- python $R/bench.py --workload $w --no-cpu-baseline --no-pmc > /dev/null 2>&1
  f=$(find /tmp/kt_$w -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${RN}_${w}_kernel_stats.csv
done
cd $R
for w in c4 c2 ds; do
  bash tools/pmc_collect.sh $w gpurun_out/pmc_$w > $OUT/${RN}_pmc_$w.txt 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_$w --json $OUT/${RN}_pmc_$w.json > /dev/null
done
timeout 200 python tools/train_bench.py --breakdown 2>/dev/null | tail -1 > $OUT/${RN}_train_step_unfused.json
timeout 200 python tools/train_bench.py --fused-adam --fused-loss --fused-pre --breakdown 2>/dev/null | tail -1 > $OUT/${RN}_train_step_fused.json
timeout 200 python tools/train_bench.py --fused-adam --fused-loss --fused-pre --breakdown --depth smooth 2>/dev/null | tail -1 > $OUT/${RN}_train_step_fused_smooth_depth.json
cd /tmp
for d in noise smooth; do   # the same train step under rocprofv3
  rm -rf /tmp/kt_ts_$d
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt_ts_$d -o kt --output-format csv -- python $R/tools/train_bench.py --fused-adam --fused-loss --fused-pre --depth $d --iters 200 > /dev/null 2>&1
  f=$(find /tmp/kt_ts_$d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${RN}_train_step_${d}_depth_kernel_stats.csv
done
cd $R
timeout 300 python tools/knn_bench.py 2>/dev/null > $OUT/${RN}_knn.json
ls -la $OUT
