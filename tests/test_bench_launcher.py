"""`python bench.py --gpus N` launches its N ranks itself (the driver's contract): without WORLD_SIZE in the
environment the script re-executes under torch.distributed.run, one process per GPU.  CPU: the launcher, the rendezvous,
the max-over-ranks timing and the gather of the direction records run with --dry-run (gloo, no GPU work).  GPU: the real
bench with two ranks on the one device of the test box (gloo: RCCL refuses two ranks on one device; the RCCL path itself
needs N devices and is what the driver's 8-GPU run exercises)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]   # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_launches_its_ranks_itself_dry_run():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"], timeout=300)
    assert r["n_gpus"] == 2 and r["gather_ok"] is True and r["dry_run"] is True
    r = _run(["--gpus", "1", "--dry-run"], timeout=300)
    assert r["n_gpus"] == 1


def test_bench_refuses_a_rank_count_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64", "--no-cpu-baseline", "--no-sqp", "--no-configs"],
             {"RTOC_BENCH_BACKEND": "gloo", "RTOC_BENCH_ONE_DEVICE": "1"})
    assert r["n_gpus"] == 2 and r["rccl_gather_ok"] is True and r["status_nonzero_instances"] == 0
    assert r["value"] > 0 and r["config"]["per_gpu_batch"] == 64
    # the kernel times come from events inside the timed loop: they cannot exceed the step
    assert r["roofline"]["kernel_ms"] + r["roofline"]["forward_kernel_ms"] <= r["ms_per_step"] * 1.001
