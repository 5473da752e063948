mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > gpurun_out/pytest_r3h.log
tail -4 gpurun_out/pytest_r3h.log
timeout 600 python tools/grad_error.py --workload c4 --kinds default,dpp,scan128 --out gpurun_out/grad_error_c4.json 2>&1 | grep -v amdgpu.ids | cut -c1-400
timeout 900 python tools/grad_error.py --workload ds --kinds default,scan128 --out gpurun_out/grad_error_ds.json 2>&1 | grep -v amdgpu.ids | cut -c1-400
