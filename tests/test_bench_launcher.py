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
    # BASELINE configs[4] both ways: the weak line above (64 per GPU) and the strong block (64 in total, 32 per GPU, gather timed)
    assert r["scaling"] == "weak"
    ss = r["strong_scaling"]
    assert ss["total_instances"] == 64 and ss["per_gpu_batch"] == 32 and ss["value"] > 0
    assert ss["backward_kernel_ms"] + ss["forward_kernel_ms"] <= ss["ms_per_step"] * 1.001


def test_compact_line_of_a_full_record_fits_the_drivers_tail():
    """the line the driver parses is the LAST stdout line, strict JSON, < 4 KB, and carries `roofline` and
    `cpu_baseline`.  Fed with the largest full record committed so far (round 4's 22 KB line, which the driver could not parse)."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    full["roofline"]["junk"] = float("nan")          # the emitter must not let a NaN through either
    line = bench.compact_line(bench._finite(full))
    assert len(line.encode()) < 4096 and "\n" not in line
    r = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))   # strict: no NaN / Infinity
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["config"]["workload"].startswith("anymal_trot_N40") and "model" not in r["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert k in r["roofline"], k
    assert r["roofline"]["traffic_measured_in_this_run"] is False
    assert abs(r["roofline"]["frac"] - r["roofline"]["achieved"] / r["roofline"]["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in r["cpu_baseline"], k
    assert set(r["other_configs"]) >= {"anymal_jump_sto_N40", "icub_nv32_jump_N30", "icub_nv35_jump_N30", "iiwa14_unconstr_N20"}


def test_dry_run_prints_one_short_last_line(tmp_path):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--detail-out", str(tmp_path / "d.json")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    last = p.stdout.strip().splitlines()[-1]
    assert len(last.encode()) < 4096
    assert json.loads(last)["dry_run"] is True
    assert json.load(open(tmp_path / "d.json"))["dry_run"] is True
