mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6) > gpurun_out/pytest_r3n.log
tail -3 gpurun_out/pytest_r3n.log
timeout 900 python tools/farm_davis_shape.py --out gpurun_out/farm_davis_shape.json 2>&1 | grep -v amdgpu.ids | tail -32
