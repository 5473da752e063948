"""SURVEY.md section 5 (aux subsystems; VERDICT r2 "missing" 3): the library's HOST code under AddressSanitizer.  Builds the C-ABI
sources with `hipcc -fsanitize=address -fno-gpu-sanitize` (host instrumented, device code as shipped) together with
tests/host_asan_main.cpp, which drives every entry point that works without a GPU (layout arithmetic over a sweep of shapes,
argument validation of forward / backward / check / mark_visible with the upstream error strings, profiler and counter
bookkeeping), and runs it: ASan aborts on any bad access, the program exits non-zero on a wrong answer.  ~2 minutes to build:
only api.hip (where all the host logic lives) and the launchers it links against are compiled."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_host_entry_points_under_address_sanitizer(tmp_path):
    src = os.path.join(ROOT, "das3r_amd", "csrc")
    hips = sorted(f for f in os.listdir(src) if f.endswith(".hip") and f not in ("render_bwd_mfma.hip", "render_bwd_stream.hip"))
    exe = str(tmp_path / "host_asan")
    cmd = [HIPCC, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-fsanitize=address", "-fno-gpu-sanitize", "-fno-omit-frame-pointer",
           "-Wno-unused-value", "-o", exe, os.path.join(ROOT, "tests", "host_asan_main.cpp")] + [os.path.join(src, f) for f in hips]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "all checks passed" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    shutil.rmtree(tmp_path, ignore_errors=True)
