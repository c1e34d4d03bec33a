"""The host code of the library (planner, mesh stand-ins, multigrid hierarchy: plain C++, no HIP) compiled with
AddressSanitizer + UndefinedBehaviorSanitizer and driven over every mesh kind and degree (tests/sanitize/plan_driver.cpp).
Sanitizers run on the CPU build only; the kernels' indexing is covered by the parity tests on the GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dealii-cuda_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_planner_and_meshes_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "plan_driver")
    src = [os.path.join(ROOT, "tests", "sanitize", "plan_driver.cpp")] + [
        os.path.join(CSRC, f) for f in ("mfgpu_plan.cpp", "mfgpu_mesh.cpp", "mfgpu_mesh_adaptive.cpp", "mfgpu_mesh_ball.cpp",
                                        "mfgpu_mg_hierarchy.cpp")]
    cc = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                         "-fno-omit-frame-pointer", "-std=c++17", "-I", CSRC, "-I", os.path.join(ROOT, "include"), *src, "-o", exe],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert cc.returncode == 0, cc.stdout[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert run.returncode == 0 and "done bad=0" in run.stdout, run.stdout[-4000:]
    assert "ERROR: AddressSanitizer" not in run.stdout and "runtime error" not in run.stdout, run.stdout[-4000:]
