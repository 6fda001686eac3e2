"""bench.py's launch / collective / report path on a CPU-only machine (`--dry-run`: gloo, no device, synthetic counters).
`python bench.py --gpus N` with no launcher environment must start its N ranks itself, and rank 0 must print exactly one
JSON line that reports the real world size (round-1 VERDICT, missing 5)."""
import json
import os
import subprocess
import sys

import numpy as np

import util


def _run(args, env_extra=None, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py")] + args, capture_output=True, text=True,
                       env=env, timeout=timeout)
    return r


def test_bench_self_launches_its_ranks():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["dry_run"] is True and j["value"] is None
    assert j["instances_total"] == 2 * j["config"]["instances_per_gpu"]      # both shards were gathered
    assert j["config"]["workload"].startswith("C2:") and "roofline" in j
    pr = j["per_rank"]                                                        # a scaling run explains its own efficiency
    assert len(pr["ms_per_step"]) == 2 and len(pr["kernel_ms_per_step"]) == 2
    assert pr["ms_per_step_max"] == max(pr["ms_per_step"]) and pr["ms_per_step_min"] == min(pr["ms_per_step"])
    assert abs(j["ms_per_step"] - pr["ms_per_step_max"]) < 1e-6 * max(1.0, j["ms_per_step"]) and pr["imbalance"] >= 1.0


def test_bench_eight_rank_dry_run_line_carries_the_per_rank_fields():
    """The driver's 8-GPU form rehearsed without devices: 8 self-launched ranks over gloo, one line, 8 per-rank clocks."""
    r = _run(["--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "0"], timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    pr = j["per_rank"]
    assert j["n_gpus"] == 8 and j["instances_total"] == 8 * j["config"]["instances_per_gpu"]
    assert len(pr["ms_per_step"]) == 8 and pr["imbalance"] >= 1.0 and pr["ms_per_step_mean"] > 0


def test_bench_under_a_launcher_environment():
    """The driver's form: one process per rank started by a launcher (here: by hand), WORLD_SIZE etc. in the env."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                                       "--steps", "1", "--warmup", "0", "--workload", "c5"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    lines = [ln for o in outs for ln in o[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["workload"].startswith("C5:")


def test_bench_time_budget_reports_steps_done():
    """--max-seconds counts from process start and the line reports the steps actually timed."""
    r = _run(["--dry-run", "--steps", "500000", "--warmup", "1", "--max-seconds", "6"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert 1 <= j["steps"] < 500000 and j["steps_requested"] == 500000 and j["elapsed_s"] < 30


def test_bench_gpus_mismatch_is_an_error():
    r = _run(["--gpus", "3", "--dry-run"], env_extra=dict(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0",
                                                           MASTER_ADDR="127.0.0.1", MASTER_PORT="1"))
    assert r.returncode != 0


def test_result_records_round_trip():
    """The 16-byte record layout the gather sends equals rppk::Result {f64 path_cost, i32 n_nodes, i32 status}."""
    import importlib
    sharding = importlib.import_module("robotics-path-planning_amd.sharding")
    pc = np.array([12.5, np.inf, 3.25]); nn = np.array([10, 7, 99999], dtype=np.int32); st = np.array([3, 1, 7], dtype=np.int32)
    a = sharding.pack_records(pc, nn, st)
    assert a.dtype == np.int64 and a.shape == (3, 2) and a.nbytes == 48
    pc2, nn2, st2 = sharding.unpack_records(a)
    assert np.array_equal(pc2, pc) and np.array_equal(nn2, nn) and np.array_equal(st2, st)
    # a record without ST_PATH reads back as +inf whatever the stored cost
    pc3, _, _ = sharding.unpack_records(sharding.pack_records(np.array([5.0]), np.array([1]), np.array([1])))
    assert np.isinf(pc3[0])


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_two_shards_on_one_gpu_equal_the_single_shard_run():
    """SURVEY 8(e), "test without 8 GPUs": two rank processes (self-launched by bench.py) plan their shards on the ONE GPU
    of the box and gather their result tables (RRTX_BENCH_SHARE_GPU=1: gloo collectives -- RCCL refuses two ranks on one
    device); the gathered table must be the single-shard run's: same instances found, same mean cost and tree size to the
    last bit (rank r owns seeds r*B+1 ... (r+1)*B, the single run seeds 1 ... 2B)."""
    common = ["--steps", "1", "--warmup", "0", "--max-iter", "2000", "--no-cpu-baseline"]
    r2 = _run(["--gpus", "2", "--instances", "48"] + common, env_extra={"RRTX_BENCH_SHARE_GPU": "1"})
    assert r2.returncode == 0, r2.stderr[-2000:]
    r1 = _run(["--gpus", "1", "--instances", "96"] + common)
    assert r1.returncode == 0, r1.stderr[-2000:]
    l2 = [ln for ln in r2.stdout.splitlines() if ln.startswith("{")]
    l1 = [ln for ln in r1.stdout.splitlines() if ln.startswith("{")]
    assert len(l2) == 1 and len(l1) == 1
    j2, j1 = json.loads(l2[0]), json.loads(l1[0])
    assert j2["n_gpus"] == 2 and j2["instances_total"] == 96 and j1["instances_total"] == 96 and "rehearsal" in j2
    for k in ("paths_found", "final_path_cost_mean", "final_path_cost_min", "mean_nodes_per_tree"):
        assert j2[k] == j1[k], k
    assert j2["iterations_per_s"] > 0 and j2["roofline"]["frac"] > 0
