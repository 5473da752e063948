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


def test_n_ranks_report_jobs_in_flight_over_the_slowest_rank():
    """`bench.py --gpus N` carries `jobs_in_flight` for N > 1 — every rank runs K = 2 whole jobs at once behind one
    barrier (farm.run_jobs), the rate is N x K jobs over the slowest rank's wall time.  Stub jobs: rank r's take 0.05 (r + 1) s."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--stub"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _json_line(p.stdout)["jobs_in_flight"]
    assert j["jobs_per_gpu"] == 2 and j["jobs_ok"] == 4 and len(j["per_rank_wall_s"]) == 2 and len(j["per_rank_scenes_per_hour"]) == 2
    assert j["wall_s"] == max(j["per_rank_wall_s"]) and 0.1 <= j["per_rank_wall_s"][1] < 2.0
    assert j["per_rank_wall_s"][1] > j["per_rank_wall_s"][0] - 0.02, "rank 1's stand-in jobs are the slower ones"
    assert abs(j["scenes_per_hour"] - 4 * 3600.0 / j["wall_s"]) < 0.01 * j["scenes_per_hour"] + 1


def test_a_failing_rank_still_yields_the_line():
    """a rank whose job dies keeps its appointments (barriers, reductions); the line comes out with ok = 0 for it,
    its units do not count, and per_rank_ms shows every rank's own time."""
    env = _env()
    env["DAS3R_BENCH_STUB_FAIL_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--stub"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    out = _json_line(p.stdout)
    assert out["n_gpus"] == 2 and out["ranks_ok"] == [1, 0] and len(out["per_rank_ms"]) == 2
    assert "rank 1 failed" in p.stderr
    ok = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--stub"],
                        capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert _json_line(ok.stdout)["ranks_ok"] == [1, 1]


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


def test_counter_csv_rows_are_grouped_and_converted(tmp_path):
    """bench.py pmc_rows: rocprofv3 counter_collection CSV -> bytes per launch and kernel group (KiB -> bytes, FETCH_SIZE doubled,
    foreign kernels dropped, template arguments and namespaces stripped, binning kernels folded into one group)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    head = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
            '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value",'
            '"Start_Timestamp","End_Timestamp"\n')
    def row(i, name, counter, value):
        return f'{i},{i},"Agent 2",1,5,5,512,8,"{name}",256,0,0,8,0,32,"{counter}",{value:.8e},1,2\n'
    f = tmp_path / "pmc_counter_collection.csv"
    f.write_text(head
                 + row(1, "__amd_rocclr_copyBuffer", "FETCH_SIZE", 64.0)
                 + row(2, "void das3r::render_backward_scan_kernel<128, false, 0>(HIP_vector_type<unsigned int, 2u> const*, int)", "FETCH_SIZE", 100.0)
                 + row(3, "das3r::render_forward_rows_kernel<true>(int)", "FETCH_SIZE", 10.0)
                 + row(4, "void das3r::onesweep_pass_kernel<16, true>(unsigned int const*)", "FETCH_SIZE", 3.0)
                 + row(5, "das3r::scan_emit_kernel<true, 8>(int)", "FETCH_SIZE", 1.0)
                 + row(6, "void das3r::preprocess_kernel<true, false, true>(int)", "WRITE_SIZE", 7.0)
                 + row(7, "void at::native::vectorized_elementwise_kernel<4>(int)", "FETCH_SIZE", 9.0))
    got = list(bench.pmc_rows([str(f)], "FETCH_SIZE"))
    assert got == [("render_backward_kernel", "render_backward_scan_kernel", 100.0 * 1024 * 2),
                   ("render_forward_kernel", "render_forward_rows_kernel", 10.0 * 1024 * 2),
                   ("binning", "onesweep_pass_kernel", 3.0 * 1024 * 2), ("binning", "scan_emit_kernel", 1.0 * 1024 * 2)]
    assert list(bench.pmc_rows([str(f)], "WRITE_SIZE")) == [("preprocess_kernel", "preprocess_kernel", 7.0 * 1024)]
    g = tmp_path / "valu_counter_collection.csv"
    g.write_text(head + row(1, "void das3r::render_backward_blk_kernel<128, 1, 0, 5>(int)", "SQ_INSTS_VALU", 2.5e8))
    assert list(bench.pmc_rows([str(g)], "SQ_INSTS_VALU")) == [("render_backward_kernel", "render_backward_blk_kernel", 2.5e8)]   # a count, not KiB
