# One GPU-box session of the round's standing checks: full GPU test suite, smoke, default bench line.   bash tools/gpu_round.sh [tag]
TAG=${1:-run}; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/pytest_$TAG.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5) > gpurun_out/smoke_$TAG.log
(timeout 900 python bench.py 2>&1 | tail -3) > gpurun_out/bench_$TAG.log
tail -6 gpurun_out/pytest_$TAG.log; cat gpurun_out/smoke_$TAG.log; cat gpurun_out/bench_$TAG.log
