#!/bin/bash
# Regenerates everything under profiles/ for round $1 (default r01) on the GPU box:  bash tools/refresh_profiles.sh r01
# (bench lines, rocprofv3 --kernel-trace --stats summaries of the same bench command, PMC passes, train-step lines).
R=$PWD; RN=${1:-r01}; OUT=$R/gpurun_out/profiles; mkdir -p $OUT
export PYTHONPATH=$R
for w in c2 c4 ds; do
  python bench.py --workload $w $( [ $w = c2 ] || echo --no-cpu-baseline ) 2>/dev/null | tail -1 > $OUT/${RN}_bench_$w.json
done
cd /tmp; export TMPDIR=/tmp
for w in c2 c4; do
  rm -rf /tmp/kt_$w
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$w -o kt --output-format csv -- python $R/bench.py --workload $w --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/kt_$w -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${RN}_${w}_kernel_stats.csv
done
cd $R
for w in c2 c4; do
  bash tools/pmc_collect.sh $w gpurun_out/pmc_$w > $OUT/${RN}_pmc_$w.txt 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_$w --json $OUT/${RN}_pmc_$w.json > /dev/null
done
python tools/train_bench.py --breakdown 2>/dev/null | tail -1 > $OUT/${RN}_train_step_unfused.json
python tools/train_bench.py --fused-adam --fused-loss --fused-pre --breakdown 2>/dev/null | tail -1 > $OUT/${RN}_train_step_fused.json
ls -la $OUT
