"""CPU tests of bench.py's launch contract (`--gpus N` was parsed and ignored): `python bench.py --gpus N` without a
torch.distributed environment must start N ranks by itself, a torchrun launch must agree with --gpus, and rank 0 prints one JSON
line whose n_gpus is the size of the process group.  Run with a stand-in step over gloo (--stub): no GPU here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_flag_spawns_that_many_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--stub"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    out = _json_line(p.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["scaling"] == "weak"
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert key in out


def test_single_rank_stub_and_world_size_mismatch():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--stub"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _json_line(p.stdout)["n_gpus"] == 1
    env = _env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")   # a torchrun environment of ONE rank, but --gpus 2 asked for
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
