mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_raster.py -m gpu -q -x -k "backward_kernel or bucket_parallel or kernels_agree" 2>&1 | tail -15) > gpurun_out/pytest_r3b.log
tail -6 gpurun_out/pytest_r3b.log
timeout 900 python tools/gpu_perf.py --workloads c4,ds,c2 --steps 20 --env DAS3R_RENDER_BWD=blk128 --env DAS3R_RENDER_BWD=blk64 --env DAS3R_RENDER_BWD=blk256 --env DAS3R_RENDER_BWD=scan128 --env DAS3R_RENDER_BWD=dpp 2>&1 | grep -v "amdgpu.ids" | grep "==\|render_backward" > gpurun_out/perf_r3b.txt
cat gpurun_out/perf_r3b.txt
