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


def test_two_torchrun_ranks():
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                 "--pages", "9", "--in-flight", "2", "--wave", "2", "--no-cpu-baseline"])
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["config"]["pages_per_step_per_gpu"] == 9
    assert "one process per GPU: DocumentAnalyzer.serve, waves of 2 pages, 2 waves in flight" in line["config"]["parallelism"]
    # 2 ranks x 9 pages x 2 steps over the max-over-ranks time
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 18) < 1e-2
    assert line["roofline"] is None and line["cpu_baseline"] is None and line["secondary"] is None  # nothing was measured
    # the proof fields of the multi-GPU run: ranks the collective saw, CRCs of the received weights, per-rank rates
    assert line["rccl"]["ranks"] == 2 and line["rccl"]["backend"] == "gloo" and line["rccl"]["weights_crc_equal"] is True
    assert set(line["rccl"]["weights_crc"]) == {"det", "rec", "lay", "tab"}
    assert 0 < line["per_rank"]["pages_per_s_min"] <= line["per_rank"]["pages_per_s_max"] and line["per_rank"]["failed_pages"] == 0


def test_single_rank():
    line = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--pages", "7", "--no-cpu-baseline"])
    assert line["dry_run"] is True and line["n_gpus"] == 1
    assert line["config"]["pages_per_step_per_gpu"] == 7
    assert line["rccl"]["ranks"] == 1 and line["rccl"]["weights_crc_equal"] is True


def test_eight_self_spawned_ranks_strong_scaling_with_a_poisoned_page():
    """`python bench.py --gpus 8` with no torchrun environment starts its own 8 ranks x 1 process (the driver's launch
    form); --total-pages deals the job's pages round-robin to the ranks (BASELINE.json configs[4], strong scaling).  One
    page of rank 3 fails in every step: that rank reports it and keeps going, nobody waits at the barrier for it."""
    env_poison = "3:1003"  # seeds are 1000 + page index; page 3 belongs to rank 3 (3 % 8)
    os.environ["YMK_BENCH_POISON"] = env_poison
    try:
        line = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--total-pages", "50",
                     "--wave", "4", "--in-flight", "2", "--no-cpu-baseline"])
    finally:
        del os.environ["YMK_BENCH_POISON"]
    assert line["dry_run"] is True and line["n_gpus"] == 8 and line["scaling"] == "strong"
    assert line["config"]["total_pages_per_step"] == 50
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 50) < 1e-2  # the whole job's pages over the max-over-ranks time
    assert line["rccl"]["ranks"] == 8 and line["rccl"]["weights_crc_equal"] is True
    assert line["per_rank"]["failed_pages"] == 2  # the poisoned page, once per timed step
