#!/bin/bash
# VERDICT r4 item 2, the measurement behind the answer: what do larger rounds of the block-walk backward buy at C4?  The same bench
# workload with the staged entries per round forced to 128 (shipped: constants in LDS, five workgroups per CU), 192 and 256 (fewer
# workgroups per CU: the four wave-private record sets are 36 B x 4 per entry) — pairs evaluated per instance (the padding: batches of
# 16, four rows in lockstep), live pairs per instance, kernel time.  Output: one line per variant in gpurun_out/blk_rounds_ablation.txt
#   tools/blk_rounds_ablation.sh [workload=c4]
W=${1:-c4}
OUT=gpurun_out/blk_rounds_ablation_$W.txt
mkdir -p gpurun_out
: > $OUT
for V in "" blk128p1 blk128p0 blk160p1 blk192p1 blk192p0 blk256p1 blk256p0; do
  DAS3R_RENDER_BWD=$V python bench.py --workload $W --steps 100 --warmup 20 --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']['render_backward_kernel']
print('variant=%-10s step_ms=%.4f bwd_ms=%.4f pairs_per_instance=%s live=%s padded_work=%s valu_roofline=%s' % ('${V:-default}', d['ms_per_step'], k['ms_per_step'], k.get('pairs_per_instance'), k.get('live_pairs_per_instance'), k.get('padded_work'), (k.get('valu_roofline') or {}).get('frac')))" >> $OUT
done
cat $OUT
