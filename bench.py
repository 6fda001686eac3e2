#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native batched RRT* planner.

Metric (BASELINE.json): collision-checked edge expansions per second (+ final path cost) while growing
100k-node RRT* trees (config C2 of SURVEY.md 8d: rrt_04 semantics, 50 circle obstacles on a 100x100 map,
max_iter 105 000 => ~100k nodes), many independent instances per GPU.

One "step" = one full planning pass of the hot path over one batch: `--instances` independent trees per
GPU (seeds rank*B+1 ...), each grown for `--max-iter` iterations by the HIP kernels through the C ABI
(librrtx.so).  Inputs (obstacles, RNG states) are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W

* N > 1 without a launcher: this process starts N rank processes itself (RANK / LOCAL_RANK / WORLD_SIZE /
  MASTER_ADDR=127.0.0.1 / MASTER_PORT set, before anything touches the GPU), waits for them and exits non-zero if
  one fails.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the launcher's
  environment is used as it is.  One process per GPU, instances sharded with no data-path collective; the only
  collective is the final all_gather of the 16-byte per-instance result records (RCCL).
* Time budget: a full-size C2 step takes tens of seconds.  `--max-seconds` (default 240, counted from PROCESS START:
  allocation, CPU baseline and warm-up included) bounds the run: the loop stops after the last step that fits and the
  JSON line reports the steps actually timed in `steps` (`steps_requested` keeps K).  The workload is never shrunk.
* The CPU baseline (oracle/rrt_oracle.c on the host cores, single thread and one instance per core) runs BEFORE the
  timed loop, bounded to ~12 s.
* Prints ONE JSON line (rank 0) on stdout.  If the process receives SIGTERM after at least one timed step, the line
  for the steps completed so far is printed before exiting.
"""
import argparse
import hashlib
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

T_PROCESS_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md)


def elapsed():
    return time.perf_counter() - T_PROCESS_START


def profile_file(suffix):
    """profiles/r<N>_<suffix> of the newest round that has one (the files carry the device-code hash they were measured
    on; a file of other code is reported as such, never used)."""
    for rnd in ("r3", "r2"):
        f = os.path.join(ROOT, "profiles", "%s_%s" % (rnd, suffix))
        if os.path.exists(f):
            return f
    return os.path.join(ROOT, "profiles", "r3_%s" % suffix)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--instances", type=int, default=None, help="planning instances per GPU (weak scaling)")
    ap.add_argument("--max-iter", type=int, default=None)
    ap.add_argument("--obstacles", type=int, default=None)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5", "c6"],
                    help="c2: rrt_04 RRT*, 50 obstacles, 105k iterations, 4096 instances/GPU (the headline metric); "
                         "c3: rrt_07 Informed RRT*, Sobol sampler, 200 obstacles, 20k iterations / 1024 instances; "
                         "c4: rrt_08 BIT*, driver constants, per-instance start/goal, 80 iterations / 4096 instances; "
                         "c5: rrt_05 RRT*-Dubins, driver constants, 5000 iterations / 1536 instances; "
                         "c6: rrt_06 RRT*-Reeds-Shepp, driver constants, 750 iterations / 16384 instances")
    ap.add_argument("--cpu-seconds", type=float, default=float(os.environ.get("RRTX_BENCH_CPU_SECONDS", "12")),
                    help="wall-clock bound of the CPU baseline (single-thread leg + all-cores leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--warmup-max-iter", type=int, default=3000,
                    help="iterations of a warm-up step (0 = same as a timed step); a warm-up only has to page in the "
                         "code objects and allocations, a full-size C2 pass takes tens of seconds")
    ap.add_argument("--max-seconds", type=float, default=float(os.environ.get("RRTX_BENCH_MAX_SECONDS", "480")),
                    help="time budget from process start (setup, CPU baseline and warm-up included); the JSON reports "
                         "the steps actually timed")
    ap.add_argument("--dry-run", action="store_true",
                    help="no device: every rank reports synthetic counters (gloo backend); exercises the launch, the "
                         "collectives and the report only -- the line carries \"dry_run\": true and no throughput")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ self-launch
def spawn_ranks(a):
    """--gpus N without a launcher environment: start N rank processes (fresh interpreters; this parent never touches
    the GPU and never re-execs), wait, propagate failure."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is None:
                    continue
                pending.remove(p)
                if r != 0 and rc == 0:
                    rc = r if r > 0 else 1
                    for q in pending:      # one rank failed: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.2)
    except KeyboardInterrupt:
        rc = 130
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ------------------------------------------------------------------------------------------------ workloads
def csrc_hash(workload):
    """Identity of the device code the numbers belong to (tools/csrc_hash.py; the PMC files record it)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import csrc_hash as ch
    return ch.csrc_hash(workload)


class Workload:
    """What differs between the configurations: handle construction, CPU sample, report strings."""

    def __init__(self, a, np, util, rrt_amd):
        self.a, self.np, self.util, self.rrt_amd = a, np, util, rrt_amd
        w = a.workload
        d_inst = {"c2": 4096, "c3": 1024, "c4": 4096, "c5": 1536, "c6": 16384}[w]
        d_iter = {"c2": 105000, "c3": 20000, "c4": 80, "c5": 5000, "c6": 750}[w]
        d_obst = {"c2": 50, "c3": 200, "c4": 6, "c5": 6, "c6": 7}[w]
        env_i = os.environ.get("RRTX_BENCH_INSTANCES")
        env_m = os.environ.get("RRTX_BENCH_MAX_ITER")
        self.B = a.instances if a.instances is not None else (int(env_i) if env_i and w == "c2" else d_inst)
        self.max_iter = a.max_iter if a.max_iter is not None else (int(env_m) if env_m and w == "c2" else d_iter)
        self.M = a.obstacles if a.obstacles is not None else d_obst
        if w != "c2":
            a.warmup_max_iter = 0
        if w == "c2":
            self.kw = util.c2_kwargs(self.max_iter, m=self.M)
        elif w == "c3":
            self.kw = dict(algo="informed", start=[2, 2], goal=[98, 98], obstacles=util.synth_map(11, self.M, 0.3, 1.5),
                           rand_area=[0, 100], expand_dis=0.5, goal_sample_rate=10, max_iter=self.max_iter, sobol=1)
        elif w == "c5":   # rrt_05 driver constants (rrt_05:1804-1859)
            self.kw = dict(algo="dubins", start=[0.0, 0.0, 0.0], goal=[10.0, 10.0, 0.0],
                           obstacles=[(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2)][:self.M],
                           rand_area=[-2, 15], expand_dis=3.0, goal_sample_rate=10, max_iter=self.max_iter)
        elif w == "c6":   # rrt_06 driver constants (rrt_06:2012-2083)
            self.kw = dict(algo="rs", start=[0.0, 0.0, 0.0], goal=[10.0, 9.0, 0.0],
                           obstacles=[(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2), (8, 10, 1)][:self.M],
                           rand_area=[-2, 15], expand_dis=3.0, goal_sample_rate=10, max_iter=self.max_iter)
        else:             # c4: rrt_08 driver constants (rrt_08:633-665)
            self.kw = dict(algo="bitstar", obstacles=[(5, 5, 0.5), (9, 6, 1), (7, 5, 1), (1, 5, 1), (3, 6, 1), (7, 9, 1)],
                           rand_area=[-2.0, 15.0], max_iter=self.max_iter)

    # -- per-rank problem instances
    def prepare(self, rank, sharding):
        import random
        self.seeds = sharding.shard_seeds(rank, self.B)     # rank r owns seeds r*B+1 .. (r+1)*B
        if self.a.workload == "c4":
            obst = self.kw["obstacles"]

            def free_point(rng):
                while True:
                    x, y = rng.uniform(-1, 14), rng.uniform(-1, 14)
                    if all((x - ox) ** 2 + (y - oy) ** 2 > r ** 2 for ox, oy, r in obst):
                        return [x, y]
            self.starts, self.goals, self.seeds = [], [], []
            for i in range(rank * self.B, (rank + 1) * self.B):
                rng = random.Random(2000 + i)
                self.starts.append(free_point(rng))
                self.goals.append(free_point(rng))
                self.seeds.append(1000 + i)

    def make_handle(self, max_iter, device):
        A, np, kw, B, w = self.rrt_amd._abi, self.np, self.kw, self.B, self.a.workload
        if w == "c6":
            h = A.Handle(A.ALGO_RS, kw["start"], kw["goal"], kw["rand_area"], 3.0, 0.5, 10, max_iter, robot_radius=0.6,
                         connect_circle_dist=50.0, search_until_max_iter=True, n_instances=B, device=device,
                         curvature=2.0, goal_yaw_th=float(np.deg2rad(1.0)), goal_xy_th=0.5, step_size=0.1)
        elif w == "c5":
            h = A.Handle(A.ALGO_DUBINS, kw["start"], kw["goal"], kw["rand_area"], 3.0, 0.5, 10, max_iter, robot_radius=0.0,
                         connect_circle_dist=50.0, search_until_max_iter=True, n_instances=B, device=device,
                         curvature=1.0, goal_yaw_th=float(np.deg2rad(1.0)), goal_xy_th=0.5)
        elif w == "c3":
            c_min, c = self.rrt_amd.informed_rotation(kw["start"], kw["goal"])
            h = A.Handle(A.ALGO_INFORMED, kw["start"], kw["goal"], kw["rand_area"], kw["expand_dis"], 1.0,
                         kw["goal_sample_rate"], max_iter, sampler=A.SAMPLER_SOBOL, n_instances=B, device=device,
                         informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
        elif w == "c4":
            c_min, c = self.rrt_amd.bitstar_rotation(self.starts[0], self.goals[0])
            h = A.Handle(A.ALGO_BITSTAR, self.starts[0], self.goals[0], kw["rand_area"], 2.0, 1.0, 0, max_iter,
                         n_instances=B, device=device, informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]],
                         informed_c_min=c_min)
        else:
            h = A.Handle(A.ALGO_RRT_STAR, kw["start"], kw["goal"], kw["rand_area"], kw["expand_dis"],
                         kw["path_resolution"], kw["goal_sample_rate"], max_iter, play_area=None,
                         robot_radius=kw["robot_radius"], sampler=A.SAMPLER_MT,
                         connect_circle_dist=kw["connect_circle_dist"], search_until_max_iter=True, n_instances=B,
                         device=device)
        h.set_obstacles(kw["obstacles"])
        h.seed_instances(self.seeds)
        if w == "c4":
            for i in range(B):
                cm, ci = self.rrt_amd.bitstar_rotation(self.starts[i], self.goals[i])   # numpy SVD on the host (rrt_08:189-202)
                h.set_instance(i, self.starts[i], self.goals[i])
                h.set_instance_rotation(i, [ci[0, 0], ci[0, 1], ci[1, 0], ci[1, 1]], cm)
        return h

    def step(self, h):
        h.plan()                           # blocking: returns after the last kernel of the batch has finished; every plan
                                           # starts from the staged per-instance state (seeds, starts, goals) again

    # -- report strings
    def names(self):
        w, B, mi, M = self.a.workload, self.B, self.max_iter, self.M
        if w == "c2":
            return ("RRT* edge expansions/sec + final path cost, %s tree (collision-checked edges evaluated on the device, "
                    "distinct per iteration; %d iterations)" % ("100k-node" if mi >= 100000 else "%d-iteration" % mi, mi),
                    "C2: rrt_04 RRT*, %d circle obstacles (map_seed 7) on 100x100, expand_dis 2.0, path_resolution 0.25, "
                    "max_iter %d, %d instances/GPU (seeds 1..), MT sampler" % (M, mi, B),
                    "rppk2t::rrt_star_kernel_v2<true>" if B > 2560 else "rppk2(s)::rrt_star_kernel_v2<true>")
        if w == "c3":
            return ("Informed RRT* edge expansions/sec + final path cost, %d-iteration tree" % mi,
                    "C3: rrt_07 Informed RRT*, Sobol sampler, %d circle obstacles (map_seed 11, radii U(0.3,1.5)) on "
                    "100x100, expand_dis 0.5, max_iter %d, %d instances/GPU (seeds 1..)" % (M, mi, B),
                    "rppi::rrt_informed_kernel")
        if w == "c4":
            return ("BIT* edge-queue expansions/sec (edges popped and processed, rrt_08:262-318), %d-iteration plans" % mi,
                    "C4: rrt_08 BIT*, driver constants (6 obstacles, rand_area [-2,15], maxIter %d), per-instance "
                    "start/goal from random.Random(2000+i), planner seed 1000+i, %d instances/GPU" % (mi, B),
                    "rppb::bitstar_wave_kernel")
        if w == "c5":
            return ("RRT*-Dubins edge expansions/sec + final path cost, %d-iteration tree" % mi,
                    "C5: rrt_05 RRT*-Dubins, driver constants (%d obstacles, 17x17 area, curvature 1), max_iter %d, "
                    "%d instances/GPU (seeds 1..)" % (M, mi, B), "rppd::rrt_dubins_kernel")
        return ("RRT*-Reeds-Shepp edge expansions/sec + final path cost, %d-iteration tree" % mi,
                "C6: rrt_06 RRT*-Reeds-Shepp, driver constants (%d obstacles, 17x17 area, curvature 2, step_size 0.1, "
                "robot_radius 0.6), max_iter %d, %d instances/GPU (seeds 1..)" % (M, mi, B), "rppr::rrt_rs_kernel")


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_job(args):
    """One oracle plan in a worker process (imports oracle/ -- the checker -- only here, in the cpu_baseline leg)."""
    workload, kw, iters, seed, extra = args
    import oracle
    t = time.perf_counter()
    if workload == "c2":
        kc = dict(kw)
        kc["max_iter"] = iters
        r = oracle.plan(seed=seed, exact_pow=False, **kc)
        st = r["stats"]
        out = (st["edges_unique"], st["edges_ref"], 1)
    elif workload == "c3":
        kc = dict(kw)
        kc.pop("algo")
        kc["max_iter"] = iters
        r = oracle.plan_informed(seed=seed, exact_pow=False, **kc)
        st = r["stats"]
        out = (st["edges_unique"], st["edges_ref"], 1)
    elif workload == "c5":
        r = oracle.plan_dubins(kw["start"], kw["goal"], kw["obstacles"], kw["rand_area"], iters, seed=seed)
        st = r["stats"]
        out = (st["edges_unique"], st["edges_ref"], 1)
    elif workload == "c6":   # `extra` = list of seeds
        eu = er = n = 0
        for sd in extra:
            r = oracle.plan_rrt_rs(kw["start"], kw["goal"], kw["obstacles"], kw["rand_area"], iters, seed=sd,
                                   curvature=2.0, robot_radius=0.6, step_size=0.1)
            eu += r["stats"]["edges_unique"]
            er += r["stats"]["edges_ref"]
            n += 1
        out = (eu, er, n)
    else:   # c4: `extra` = list of (start, goal, seed)
        n = 0
        for (s, g, sd) in extra:
            oracle.plan_bitstar(s, g, kw["obstacles"], kw["rand_area"], iters, seed=sd)
            n += 1
        out = (0, 0, n)
    return out + (time.perf_counter() - t,)


def cgroup_cpu_quota():
    """CPUs' worth of time the cgroup of this process may use (cpu.max of cgroup v2, cfs quota of v1), or None."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(wl, budget_s):
    """the C restatement of the reference (oracle/rrt_oracle.c, pinned to the reference's goldens) on
    the host cores of this box: (a) one instance on one thread, (b) one instance per core over all cores.  Bounded
    sample of the same workload (fewer iterations per tree for C2/C3/C5: the CPU rate at full size is lower still)."""
    import concurrent.futures as cf
    import multiprocessing as mp
    w, kw = wl.a.workload, wl.kw
    try:
        ncore = len(os.sched_getaffinity(0))
    except AttributeError:
        ncore = os.cpu_count() or 1
    # one instance per core over all cores.  "All cores" = what this process may really use: the
    # affinity mask, cut by a cgroup CPU quota when the box sets one (a GPU box hands a 1-GPU job a share of its host),
    # and by RRTX_BENCH_CPU_WORKERS (default 64: one forked worker per job, each a full oracle plan)
    quota = cgroup_cpu_quota()
    cap = int(os.environ.get("RRTX_BENCH_CPU_WORKERS", "64"))
    nwork = max(1, min(ncore, cap, int(quota) if quota and quota >= 1 else ncore))
    # sample sizes aimed at ~budget/3 per leg on one core (measured rates of the oracle on a 2.1 GHz Xeon core)
    scale = max(0.25, min(4.0, budget_s / 12.0))
    per_job = 1
    if w == "c2":
        iters = int(14000 * scale ** 0.5)    # time ~ n^2
        unit = "edge expansions/s"
    elif w == "c3":
        iters = int(min(wl.max_iter, 12000 * scale ** 0.5))
        unit = "edge expansions/s"
    elif w == "c5":
        iters = wl.max_iter
        unit = "edge expansions/s"
    elif w == "c6":
        iters = wl.max_iter
        unit = "edge expansions/s"
        per_job = max(1, int(20 * scale))
    else:
        iters = wl.max_iter
        unit = "plans/s"
        per_job = max(1, int(240 * scale))

    def job(k):
        extra = None
        if w == "c4":
            idx = [(k * per_job + j) % wl.B for j in range(per_job)]
            extra = [(wl.starts[i], wl.goals[i], wl.seeds[i]) for i in idx]
        if w == "c6":
            extra = [1 + k * per_job + j for j in range(per_job)]
        return (w, kw, iters, 1 + k, extra)
    ctx = mp.get_context("fork")     # forked before this process has touched the GPU
    t0 = time.perf_counter()
    with cf.ProcessPoolExecutor(max_workers=nwork, mp_context=ctx) as ex:
        # (a) single thread: one job, alone on the machine
        tA = time.perf_counter()
        r1 = [ex.submit(_cpu_job, job(0)).result()]
        tA = time.perf_counter() - tA
        # (b) all cores: one job per worker, concurrently
        tB = time.perf_counter()
        rN = list(ex.map(_cpu_job, [job(k) for k in range(nwork)]))
        tB = time.perf_counter() - tB
    eu1, er1, pl1 = sum(r[0] for r in r1), sum(r[1] for r in r1), sum(r[2] for r in r1)
    euN, erN, plN = sum(r[0] for r in rN), sum(r[1] for r in rN), sum(r[2] for r in rN)
    plans = unit == "plans/s"
    out = {"value": (plN if plans else euN) / tB, "unit": unit, "cores": nwork, "kind": "port",
           "single_thread_value": (pl1 if plans else eu1) / tA,
           "nproc": ncore, "cgroup_cpu_quota": quota, "cpu_model": cpu_model(),
           "sample": "oracle/rrt_oracle.c (C restatement pinned to the reference's goldens); same workload, %s; "
                     "single thread: %d plan(s) in %.1f s; all cores: one job per core on %d cores in %.1f s (of %d "
                     "visible cores); every near candidate steered as the reference does"
                     % ("%d iterations per tree" % iters + ("" if per_job == 1 else ", %d plans per core" % per_job),
                        pl1, tA, nwork, tB, ncore),
           "seconds": time.perf_counter() - t0}
    if not plans:
        out["reference_equivalent_value"] = erN / tB
        out["plans_per_s"] = plN / tB
    # the full-size single-thread rate (one whole plan at the workload's own max_iter: minutes of CPU time, so it is
    # measured once per CPU model by tools/cpu_fullsize.py and kept under profiles/, not re-run by every bench)
    try:
        fj = json.load(open(os.path.join(ROOT, "profiles", "cpu_fullsize_%s.json" % w)))
        for rec in fj["runs"]:
            if rec["max_iter"] == wl.max_iter and rec["obstacles"] == wl.M:
                out["full_size_single_thread"] = dict(rec, same_cpu_model=(rec.get("cpu_model") == out["cpu_model"]))
                break
    except (OSError, ValueError, KeyError):
        pass
    return out


# ------------------------------------------------------------------------------------------------ dry run
class DryHandle:
    """--dry-run: no device, no planning.  Fixed synthetic counters so that the launch / collective / report path can
    be exercised on a CPU-only machine (tests/test_bench_launch.py).  Never used for a measurement."""

    def __init__(self, B, max_iter):
        self.B, self.max_iter = B, max_iter

    def plan(self):
        time.sleep(0.02)

    def get_stats(self):
        n = self.B * self.max_iter
        return dict(iterations=n, edges_unique=10 * n, edges_ref=100 * n, algorithmic_bytes=1000 * n,
                    algorithmic_bytes_two_scan=8000 * n, kernel_ms=20.0, launches=1, near_unique_max=0, f32_fallbacks=0,
                    q16_fallbacks=0, exact_rescans=0, replanned=0, main_shape=64, main_f32=1)

    def get_results(self):
        import numpy as np
        return np.full(self.B, 1.0), np.full(self.B, self.max_iter, dtype=np.int32), np.full(self.B, 3, dtype=np.int32)

    def close(self):
        pass


# ------------------------------------------------------------------------------------------------ main
def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))

    import numpy as np
    import util
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and not (a.gpus == 1 and world == 1):
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world), file=sys.stderr)
        sys.exit(2)
    ngpu = world

    import importlib
    import rrt_amd
    sharding = importlib.import_module("robotics-path-planning_amd.sharding")
    wl = Workload(a, np, util, rrt_amd)
    wl.prepare(rank, sharding)

    # ---- CPU baseline first (rank 0 of a single-GPU run only): forks its workers before the GPU is touched
    cpu = None
    if not a.no_cpu_baseline and ngpu == 1 and not a.dry_run:
        try:
            cpu = cpu_baseline(wl, a.cpu_seconds)
        except Exception as e:  # noqa: BLE001  (a missing checker must not lose the GPU measurement)
            cpu = {"error": "%s: %s" % (type(e).__name__, e)}

    dist = None
    torch = None
    cuda = None
    if ngpu > 1 or os.environ.get("RRTX_BENCH_FORCE_DIST"):
        import torch
        import torch.distributed as dist
        if a.dry_run:
            dist.init_process_group(backend="gloo")
        elif os.environ.get("RRTX_BENCH_SHARE_GPU"):
            # rehearsal of the N-rank path on a box with ONE GPU (RCCL refuses two ranks on one device): every rank plans on
            # device 0, the collectives run over gloo on host tensors.  The line carries "rehearsal": true; never a headline
            dist.init_process_group(backend="gloo")
            local_rank = 0
        else:
            torch.cuda.set_device(local_rank)
            cuda = torch.device("cuda", local_rank)
            dist.init_process_group(backend="nccl", device_id=cuda)

    def sync_all():
        if dist is not None:
            dist.barrier()
            if cuda is not None:
                torch.cuda.synchronize()

    state = dict(steps_done=0, kernel_ms=0.0, kms_main=0.0, launches_main=0, alg=0, alg2=0, eu=0, er=0, iters=0, launches=0,
                 t0=None, dt=0.0,
                 last_stats={}, printed=False, results=None)
    metric, workload_name, kernel_name = wl.names()

    def kernel_of(stats):
        """Name of the dominant kernel as rrtx_plan really launched it (C2: the workgroup shape is picked from the
        instance count, the obstacle count, the estimated near-set size and RRTX_TPB -- rrtx_stats.main_shape)."""
        if a.workload != "c2":
            return kernel_name
        ns = {64: "rppk2t", 128: "rppk2s", 256: "rppk2"}.get(int(stats.get("main_shape", 0) or 0))
        return "%s::rrt_star_kernel_v2<%s>" % (ns, "true" if stats.get("main_f32", 1) else "false") if ns else kernel_name

    def build_line(h, final):
        """The JSON line for the steps completed so far (cross-rank reductions only when `final`)."""
        sd = max(state["steps_done"], 1)
        dt = state["dt"]
        pc, nn, st = state["results"]     # table of the last completed step (never read while a plan is running)
        per_rank = None
        if final:
            tmax = sharding.reduce_max(dist, dt, cuda)
            # per-rank clocks, so that a scaling run explains its own efficiency: wall time of the timed region and the
            # dominant kernel's HIP-event time on every rank (the job's clock is the max)
            per_rank = sharding.gather_floats(dist, [dt, state["kms_main"] / 1e3], cuda)
            tot_eu, tot_er, tot_it = sharding.reduce_sum_int(dist, [state["eu"], state["er"], state["iters"]], cuda)
            if state.get("native_rccl"):
                all_pc, all_nn, all_st = h.rccl_gather_results()
                all_pc, all_nn, all_st = np.asarray(all_pc), np.asarray(all_nn, dtype=np.int64), np.asarray(all_st, dtype=np.int64)
            else:
                all_pc, all_nn, all_st = sharding.gather_results(dist, pc, nn, st, cuda,
                                                                 device_table=None if a.dry_run else h)
        else:
            tmax, tot_eu, tot_er, tot_it = dt, state["eu"], state["er"], state["iters"]
            all_pc, all_nn, all_st = np.asarray(pc), np.asarray(nn), np.asarray(st)
        if rank != 0:
            return None
        # roofline of the DOMINANT kernel: its own launches and HIP-event time (C2: the iteration kernel; the final
        # goal-search launch of each plan -- a few ms -- is reported in kernel_ms_all_per_step only)
        kms, alg, alg2 = state["kms_main"], state["alg"], state["alg2"]
        nl_main = max(state["launches_main"], 1)
        c4 = a.workload == "c4"
        found = ((all_st & 2) != 0) if c4 else np.isfinite(all_pc)
        edges = tot_eu
        value = edges / tmax if tmax > 0 else None
        achieved = (alg / 1e9) / (kms / 1e3) if kms > 0 and alg > 0 else None
        achieved_2s = (alg2 / 1e9) / (kms / 1e3) if kms > 0 and alg2 > 0 else None
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS if achieved else None, "traffic": None, "traffic_frac": None,
                "kernel": kernel_of(state["last_stats"]), "launches": state["launches_main"],
                "kernel_ms_per_step": kms / sd, "kernel_ms_per_launch": kms / nl_main,
                "kernel_ms_all_per_step": state["kernel_ms"] / sd,
                "algorithmic_bytes_per_step": alg / sd, "algorithmic_bytes_per_launch": alg / nl_main,
                "note": "achieved = bytes the algorithm as implemented must move (rrtx_stats.algorithmic_bytes, "
                        "accumulated on the device; DESIGN.md 5.1) / HIP-event time of the planner-kernel launches on "
                        "the handle's stream, measured in this run"}
        ls0 = state["last_stats"]
        if a.workload == "c2" and ls0.get("passes_shared") and ls0.get("iterations"):
            # iterations whose near query was answered by the streaming pass of an earlier iteration:
            # they move no mirror bytes, so the algorithmic bytes above are those of the passes that did run
            roof["passes_shared_frac"] = ls0["passes_shared"] / ls0["iterations"]
            roof["passes_shared_note"] = ("the pass of iteration i also answers the near balls of iterations i+1 .. i+3 (speculated "
                                          "about their samples) and the nearest queries of the samples after them: that share of the "
                                          "iterations ran without a pass of their own; RRTX_SPEC2=0 runs one pass per iteration (three "
                                          "times the bytes, a higher fraction of the HBM peak, a lower edge rate: DESIGN.md 6.0)")
        if achieved_2s:
            roof["survey_8d_two_scan_equivalent_GBps"] = achieved_2s
            roof["survey_8d_two_scan_note"] = ("SURVEY 8(d)'s two-f64-scan formula (32*n + 48*k + 24*M + 28 per "
                                               "iteration) applied to the same run: an EQUIVALENT rate, not bytes moved")
        # HBM traffic from separate rocprofv3 --pmc passes of this exact device code (tools/profile_headline.sh writes
        # profiles/r2_*_traffic.json with the hash of csrc/ it measured)
        tfile = profile_file("%s_traffic.json" % a.workload)
        try:
            tj = json.load(open(tfile))
            tc = tj["config"]
            if tc["instances_per_gpu"] == wl.B and tc["max_iter"] == wl.max_iter and tc["obstacles"] == wl.M:
                knobs = [k for k in ("RRTX_SPEC2", "RRTX_Q16", "RRTX_F32", "RRTX_TPB", "RRTX_KERNEL") if os.environ.get(k) is not None]
                if knobs:
                    # the PMC passes ran the default configuration: another pass structure moves other bytes
                    roof["traffic_note"] = "profiles/%s belongs to the default configuration (%s set here): not used" % (
                        os.path.basename(tfile), ", ".join(knobs))
                elif tj.get("csrc_hash") == csrc_hash(a.workload):
                    roof["traffic"] = tj["hbm_bytes_per_launch"]
                    roof["traffic_per_step"] = tj["hbm_bytes_per_step"]
                    t_gbps = tj["hbm_bytes_per_step"] / 1e9 / (kms / 1e3 / sd)
                    roof["traffic_GBps"] = t_gbps
                    roof["traffic_frac"] = t_gbps / HBM_PEAK_GBPS
                    roof["traffic_note"] = ("HBM bytes per kernel launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                            "passes of this device code (%s, commit %s); traffic_frac = those bytes / "
                                            "this run's kernel time / peak" % (os.path.basename(tfile), tj.get("commit")))
                else:
                    roof["traffic_note"] = "profiles/%s was measured on other device code (hash %s != %s): not used" % (
                        os.path.basename(tfile), tj.get("csrc_hash"), csrc_hash(a.workload))
        except (OSError, ValueError, KeyError):
            pass
        # C3..C6: closed-form f64 geometry per candidate edge, small trees -- the kernels are bound by VALU issue and
        # dependent-instruction latency, not by HBM.  Their roof is the VALU issue rate, measured by a rocprofv3 --pmc
        # pass of this exact device code (tools/valu_pass.sh -> profiles/r2_<workload>_valu.json); the HBM figures stay
        # next to it under "hbm".
        vfile = profile_file("%s_valu.json" % a.workload)
        if a.workload != "c2":
            try:
                vj = json.load(open(vfile))
            except (OSError, ValueError):
                vj = None
            if vj is not None and vj.get("csrc_hash") == csrc_hash(a.workload):
                # issue cycles per launch from the counters (a property of the code and the workload), over THIS run's
                # kernel time: the profiled run's own time is inflated by counter collection (its figure stays beside it)
                cn = vj.get("counters", {})
                f64i = sum(cn.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64",
                                                    "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64"))
                disp = max(vj.get("dispatches", 1), 1)
                issue = (2.0 * (cn.get("SQ_INSTS_VALU", 0.0) - f64i) + 4.0 * f64i) / disp
                vfrac = vj["valu_busy_frac"]
                if issue > 0 and kms > 0:
                    vfrac = issue * nl_main / (kms * 1e-3 * 2.4e9 * 1024.0)
                roof = {"bound": "valu", "achieved": 100.0 * vfrac, "peak": 100.0,
                        "unit": "% of VALU issue slots", "frac": vfrac, "traffic": None,
                        "frac_in_profiled_run": vj["valu_busy_frac"], "issue_cycles_per_launch": issue,
                        "kernel": kernel_name, "launches": state["launches_main"], "kernel_ms_per_step": kms / sd,
                        "kernel_ms_per_launch": kms / nl_main,
                        "valu_insts_per_launch": vj.get("valu_insts_per_launch"),
                        "active_lane_frac": vj.get("active_lane_frac"),
                        "f64_flops_upper_TFLOPs": vj.get("f64_flops_upper_TFLOPs"), "f64_peak_TFLOPs": 78.6,
                        "note": "VALU-issue roofline from rocprofv3 --pmc SQ_INSTS_VALU / _FMA_F64 / _ADD_F64 / _MUL_F64 / "
                                "_TRANS_F64 of this device code (profiles/%s, commit %s): frac = (2 cycles x non-f64 + 4 cycles x "
                                "f64 VALU instructions per launch) x launches / (THIS run's kernel time x 2.4 GHz x 1024 SIMDs).  The kernel is latency bound "
                                "(dependent f64 chains of the libm replicas on few lanes: active_lane_frac), HBM is idle"
                                % (os.path.basename(vfile), vj.get("commit")),
                        "hbm": roof}
                # the primary roof is the one the kernel sits closer to (C3: two f32-mirror passes per iteration put its
                # algorithmic byte rate above its VALU share); the other stays next to it
                hb = roof["hbm"]
                h_frac = max(hb.get("frac") or 0.0, hb.get("traffic_frac") or 0.0)
                if h_frac > roof["frac"]:
                    valu = {k: v for k, v in roof.items() if k != "hbm"}
                    valu["note"] = valu["note"].replace(", HBM is idle", "")
                    roof = dict(hb)
                    roof["valu"] = valu
                    roof["note"] += ("; the working set of this workload (instances x mirror bytes) fits the 256 MB "
                                     "Infinity Cache, so the algorithmic byte rate bounds the HBM traffic from above "
                                     "(traffic, when a PMC pass of this code exists, is the measured figure)")
            else:
                roof["note"] += ("; no VALU counters for this device code (profiles/%s missing or measured on other code): "
                                 "only the HBM side is reported" % os.path.basename(vfile))
        line = {
            "metric": metric, "value": value, "unit": "edge expansions/s", "n_gpus": ngpu,
            "steps": state["steps_done"], "warmup": a.warmup,
            "ms_per_step": 1e3 * tmax / sd, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_name, "instances_per_gpu": wl.B, "max_iter": wl.max_iter,
                       "obstacles": wl.M, "parallelism": "instances x%d" % ngpu},
            "edge_expansions_reference_equivalent_per_s": tot_er / tmax if tmax > 0 else None,
            "iterations_per_s": tot_it / tmax if tmax > 0 else None,
            "plans_per_s": len(all_pc) * state["steps_done"] / tmax if tmax > 0 else None,
            "mean_nodes_per_tree": float(np.mean(all_nn)),
            "final_path_cost_mean": float(np.mean(all_pc[found])) if found.any() else None,
            "final_path_cost_min": float(np.min(all_pc[found])) if found.any() else None,
            "paths_found": int(found.sum()), "instances_total": int(len(all_pc)),
            "instances_with_status_bits": {"overflow": int(((all_st & 4) != 0).sum()),
                                           "unsupported": int(((all_st & 16) != 0).sum()),
                                           "ref_raises": int(((all_st & 32) != 0).sum())},
            "roofline": roof,
            "steps_requested": a.steps, "warmup_max_iter": a.warmup_max_iter or wl.max_iter,
            "time_budget_s": a.max_seconds, "elapsed_s": elapsed(), "csrc_hash": csrc_hash(a.workload),
            "value_note": "`value` counts the distinct collision-checked edges the device evaluates; "
                          "edge_expansions_reference_equivalent_per_s counts check_collision calls as the reference "
                          "makes them for the same trees (repeated near indices included, SURVEY R6)",
        }
        ls = state["last_stats"]
        for k in ("near_unique_max", "f32_fallbacks", "q16_fallbacks", "exact_rescans", "passes_shared"):
            if k in ls:
                line[k + "_last_step"] = ls[k]
        if a.workload in ("c5", "c6"):
            line["edges_steered_per_iteration"] = tot_eu / max(tot_it, 1)
            # what the reference steers for the same trees: the extension plus every near-list entry twice (choose_parent,
            # rewire); from this rank's counters of the last step (the kernels count only what the device steers)
            if ls.get("iterations") and tmax > 0:
                per_it = 1.0 + 2.0 * ls.get("near_hits", 0) / ls["iterations"]
                line["edge_expansions_reference_equivalent_per_s"] = per_it * tot_it / tmax
                line["reference_edges_per_iteration_estimate"] = per_it
            line["value_note"] = ("`value` counts the edges the device steers.  The default build steers only the "
                                  "candidates that can change the result (DESIGN.md 5.5 / 5.7: same trees, several times "
                                  "fewer edges); the reference and the CPU baseline steer every near candidate: "
                                  "edge_expansions_reference_equivalent_per_s = (1 + 2 x near-list entries per iteration) x "
                                  "iterations/s, an upper estimate (an entry whose steer returns None is not collision-"
                                  "checked by the reference; rrt_06's try_goal_path edge is not counted) -- compare "
                                  "plans_per_s / iterations_per_s with the CPU baseline, not the device's edge rate")
        if a.workload == "c3":
            line["edges_tested_per_iteration"] = tot_eu / max(tot_it, 1)
            line["value_note"] = ("`value` counts the segments the device collision-tests: choose_parent tests the near "
                                  "candidates cheapest first and stops at the first free one, rewire tests only candidates "
                                  "whose cost improves (DESIGN.md 5.8: same trees; RRTX_INFORMED_EAGER=1 tests every candidate); "
                                  "edge_expansions_reference_equivalent_per_s counts check_collision calls as the reference "
                                  "makes them for the same trees")
        if c4:
            line["unit"] = "edge expansions/s"
            roof["note"] += "; BIT* is an instance-parallel sequential search (one wave per instance, state in LDS)"
        if per_rank is not None:
            wall = [1e3 * r[0] / sd for r in per_rank]
            kern = [1e3 * r[1] / sd for r in per_rank]
            line["per_rank"] = {"ms_per_step": wall, "kernel_ms_per_step": kern,
                                "ms_per_step_min": min(wall), "ms_per_step_max": max(wall),
                                "ms_per_step_mean": sum(wall) / len(wall),
                                "imbalance": max(wall) / (sum(wall) / len(wall)) if sum(wall) > 0 else None,
                                "note": "rank r plans instances r*B .. (r+1)*B-1; the job's ms_per_step is the max over "
                                        "ranks; imbalance = max / mean (1.0 = every GPU finished its shard together)"}
        line["replanned_last_step"] = ls.get("replanned", 0)
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if os.environ.get("RRTX_BENCH_SHARE_GPU"):
            line["rehearsal"] = "all ranks share GPU 0, gloo collectives: exercises the launch / sharding / report path only"
        if a.dry_run:
            line["dry_run"] = True
            line["value"] = None
        if not final:
            line["partial"] = True
        return line

    holder = {}

    def on_term(signum, frame):
        # killed from outside (e.g. the driver's time limit): report what has been measured, then go
        h = holder.get("h")
        if h is not None and state["steps_done"] > 0 and not state["printed"] and rank == 0 and ngpu == 1:
            try:
                print(json.dumps(build_line(h, final=False)), flush=True)
            except Exception:  # noqa: BLE001
                pass
        os._exit(143)
    signal.signal(signal.SIGTERM, on_term)

    def run_blocking(fn):
        """Run a blocking ABI call on a worker thread so that this (main) thread keeps serving signals."""
        box = {}

        def target():
            try:
                fn()
            except BaseException as e:  # noqa: BLE001
                box["e"] = e
        t = threading.Thread(target=target, daemon=True)
        t.start()
        while t.is_alive():
            t.join(0.25)
        return box.get("e")

    def all_ok(err):
        ok = sharding.all_agree_min(dist, 0 if err is not None else 1, cuda)
        if not ok:
            if err is not None:
                print("bench.py rank %d: %s: %s" % (rank, type(err).__name__, err), file=sys.stderr, flush=True)
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(1)

    # ---- handles, warm-up
    device = local_rank
    if a.dry_run:
        h = DryHandle(wl.B, wl.max_iter)
        hw = h
    else:
        h = wl.make_handle(wl.max_iter, device)
        hw = h
        if a.warmup_max_iter and a.warmup_max_iter != wl.max_iter:
            hw = wl.make_handle(a.warmup_max_iter, device)
    holder["h"] = h
    warm_done = 0
    for _ in range(a.warmup):
        tw = time.perf_counter()
        all_ok(run_blocking(lambda: wl.step(hw) if not a.dry_run else hw.plan()))
        warm_done += 1
        tw = time.perf_counter() - tw
        # a warm-up must leave room for at least one timed step (a full-size warm-up costs as much as one)
        if not sharding.all_agree_min(dist, 1 if elapsed() + 2.5 * tw <= a.max_seconds else 0, cuda):
            break
    if hw is not h:
        hw.close()

    # ---- optional: the result table gathered by the library's own RCCL all_gather (rrtx_rccl_*: no framework in the data
    # path) instead of torch.distributed's; rank 0's ncclUniqueId travels over the process group that is up anyway.  Off by
    # default (the torch path is the one rehearsed at N > 1); any failure falls back to it.
    native_rccl = False
    if (os.environ.get("RRTX_BENCH_NATIVE_RCCL") and dist is not None and cuda is not None and not a.dry_run):
        try:
            box = [rrt_amd._abi.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            h.rccl_init(box[0], rank, ngpu)
            native_rccl = True
        except Exception as e:  # noqa: BLE001
            print("bench.py rank %d: native RCCL gather unavailable (%s: %s), using torch.distributed" %
                  (rank, type(e).__name__, e), file=sys.stderr, flush=True)
        native_rccl = bool(sharding.all_agree_min(dist, 1 if native_rccl else 0, cuda))

    # ---- timed region
    sync_all()
    state["t0"] = time.perf_counter()
    for _ in range(a.steps):
        ts = time.perf_counter()
        all_ok(run_blocking(lambda: wl.step(h) if not a.dry_run else h.plan()))
        s = h.get_stats()
        state["last_stats"] = s
        state["kernel_ms"] += s["kernel_ms"]
        state["alg"] += s["algorithmic_bytes"]
        state["alg2"] += s["algorithmic_bytes_two_scan"]
        state["eu"] += s["edges_unique"]
        state["er"] += s["edges_ref"]
        state["iters"] += s["iterations"]
        state["launches"] += s["launches"]
        state["kms_main"] += s.get("kernel_ms_main", s["kernel_ms"])
        state["launches_main"] += s.get("launches_main", s["launches"])
        state["results"] = h.get_results()
        state["steps_done"] += 1
        state["dt"] = time.perf_counter() - state["t0"]
        last = time.perf_counter() - ts
        # time guard (all ranks take the same decision: the slowest rank's clock decides)
        if not sharding.all_agree_min(dist, 1 if elapsed() + 1.1 * last <= a.max_seconds else 0, cuda):
            break
    sync_all()
    state["dt"] = time.perf_counter() - state["t0"]
    state["warmup_done"] = warm_done

    state["native_rccl"] = native_rccl
    line = build_line(h, final=True)
    if rank == 0:
        line["warmup"] = warm_done
        line["result_gather"] = ("rrtx_rccl_gather_results (ncclAllGather from librrtx.so)" if native_rccl else
                                 ("torch.distributed all_gather" if dist is not None else "single process"))
        state["printed"] = True
        print(json.dumps(line), flush=True)
    h.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
