"""bench.py --gpus N started the way the driver starts it (no WORLD_SIZE): the parent launches N fresh rank
processes before it touches the GPU and reports their failure cleanly.  Here (no GPU) the ranks stop at bench.py's
"needs a GPU" check; what is tested is the launcher: N children, each with its own RANK, non-zero exit, no result
line invented."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_ranks_without_world_size():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MFGPU_BENCH_ECHO_RANK"] = "1"
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    err = pr.stderr
    assert "launching 2 ranks" in err, err[-2000:]
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert pr.returncode == 0, err[-2000:]
        assert '"n_gpus": 2' in pr.stdout
        return
    # no (or one) GPU: both ranks were started as children with WORLD_SIZE = 2 and their own RANK, and stopped
    # (the launcher ends the other rank as soon as one fails: both lines are there unless that happens within the
    # first milliseconds of a rank's life)
    assert "of 2 started" in err, err[-2000:]
    assert pr.returncode != 0
    assert "multi-GPU run failed" in err
    assert '"metric"' not in pr.stdout


def test_bench_refuses_mismatched_world_size():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, text=True, timeout=600)
    assert pr.returncode != 0 and "WORLD_SIZE=3" in pr.stderr
