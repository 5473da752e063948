mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I das3r_amd/csrc -o /tmp/blk_loop_probe tools/probes/blk_loop_probe.hip 2>/dev/null && timeout 60 /tmp/blk_loop_probe > gpurun_out/blk_loop_probe.txt 2>&1; cat gpurun_out/blk_loop_probe.txt
(timeout 300 python -m pytest tests/test_gpu_raster.py -m gpu -q -x -k "every_backward_kernel and blk" 2>&1 | tail -3)
for k in blk128p1 blk120p2 blk128p2 blk120p1; do
  DAS3R_RENDER_BWD=$k timeout 300 python -m pytest tests/test_gpu_raster.py -m gpu -q -x -k "test_forward_backward_vs_oracle" 2>&1 | tail -1
done
timeout 600 python tools/gpu_perf.py --workloads c4,ds --steps 20 --env DAS3R_RENDER_BWD=blk128 --env DAS3R_RENDER_BWD=blk128p1 --env DAS3R_RENDER_BWD=blk120p2 --env DAS3R_RENDER_BWD=blk128p2 --env DAS3R_RENDER_BWD=blk120 2>&1 | grep "==\|render_backward" > gpurun_out/perf_r3d.txt
cat gpurun_out/perf_r3d.txt
