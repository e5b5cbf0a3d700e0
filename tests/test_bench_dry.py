"""bench.py's multi-rank orchestration on CPU (YMK_BENCH_DRY=1): ranks over gloo, helper processes per rank, step
barriers, max-over-ranks clock, one JSON line from rank 0, clean teardown - with stub page workers.  The real
N-GPU run is the driver's; this is what can be verified without GPUs."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd):
    env = dict(os.environ, YMK_BENCH_DRY="1", PYTHONPATH=ROOT)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_two_ranks_with_helper_processes():
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                 "--pages", "9", "--procs", "2", "--workers", "2", "--wave", "2", "--no-cpu-baseline"])
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["config"]["pages_per_step_per_gpu"] == 9
    assert "waves of 2 pages, 2 process(es) x 2 waves in flight" in line["config"]["parallelism"]
    # 2 ranks x 9 pages x 2 steps over the max-over-ranks time
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 18) < 1e-2
    assert line["roofline"] is None and line["cpu_baseline"] is None  # nothing was measured


def test_single_rank_three_processes():
    line = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--pages", "7", "--procs", "3",
                 "--workers", "1", "--no-cpu-baseline"])
    assert line["dry_run"] is True and line["n_gpus"] == 1
    assert line["config"]["pages_per_step_per_gpu"] == 7


def test_self_spawned_ranks_strong_scaling():
    """`python bench.py --gpus 8` with no torchrun environment starts its own 8 ranks (the driver's launch form);
    --total-pages deals the job's pages round-robin to the ranks (BASELINE.json configs[4], strong scaling)."""
    line = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--total-pages", "50",
                 "--wave", "4", "--workers", "2", "--no-cpu-baseline"])
    assert line["dry_run"] is True and line["n_gpus"] == 8 and line["scaling"] == "strong"
    assert line["config"]["total_pages_per_step"] == 50
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 50) < 1e-2  # the whole job's pages over the max-over-ranks time
