mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_probe tools/probes/dpp_probe.hip 2>/dev/null && /tmp/dpp_probe > gpurun_out/dpp_probe.txt 2>&1
cat gpurun_out/dpp_probe.txt
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > gpurun_out/pytest_r3a.log
tail -5 gpurun_out/pytest_r3a.log
timeout 600 python tools/grad_error.py --workload c4 --kinds default,dpp,scan128 --out gpurun_out/grad_error_c4.json 2>&1 | tail -5
timeout 900 python tools/grad_error.py --workload ds --kinds default --out gpurun_out/grad_error_ds.json 2>&1 | tail -3
(timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/bench_r3a.err | tail -1) > gpurun_out/bench_r3a.json
python -c "
import json; d=json.load(open('gpurun_out/bench_r3a.json')); print(d['ms_per_step'], d.get('kernel_ms'))"
