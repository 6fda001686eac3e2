#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native batched RRT* planner.

Metric (BASELINE.json): collision-checked edge expansions per second (+ final path cost) while growing
100k-node RRT* trees (config C2 of SURVEY.md 8d: rrt_04 semantics, 50 circle obstacles on a 100x100 map,
max_iter 105 000 => ~100k nodes), many independent instances per GPU.

One "step" = one full planning pass of the hot path over one batch: `--instances` independent trees per
GPU (seeds rank*B+1 ...), each grown for `--max-iter` iterations by the HIP kernels through the C ABI
(librrtx.so).  Inputs (obstacles, RNG states) are resident in HBM before the timed region.  N GPUs = N
processes (torch.distributed / RCCL), instances sharded with no data-path collective; the only collective
is the final all_gather of the 16-byte per-instance result records.  `value` = edge expansions of all ranks
per step / max-over-ranks step time.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes (SURVEY.md 8d:
sum over iterations of 32*n + 48*k + 24*M + 28) / HIP-event time of the planner kernel launches; the fused
single-pass figure (16*n per iteration) is reported next to it.
`cpu_baseline` = the CPU oracle (oracle/rrt_oracle.c, "port") on one host core on a bounded sample.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--instances", type=int, default=int(os.environ.get("RRTX_BENCH_INSTANCES", "4096")),
                    help="planning instances per GPU (weak scaling)")
    ap.add_argument("--max-iter", type=int, default=int(os.environ.get("RRTX_BENCH_MAX_ITER", "105000")))
    ap.add_argument("--obstacles", type=int, default=None)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5", "c6"],
                    help="c2: rrt_04 RRT*, 50 obstacles, 105k iterations (the headline metric); c3: rrt_07 Informed RRT* "
                         "with the Sobol sampler, 200 obstacles (SURVEY 8d), default 20k iterations / 1024 instances; "
                         "c5: rrt_05 RRT*-Dubins, driver constants, default 5000 iterations / 1536 instances; "
                         "c4: rrt_08 BIT*, driver constants, per-instance start/goal (SURVEY 8d), 80 iterations; "
                         "c6: rrt_06 RRT*-Reeds-Shepp, driver constants, 750 iterations / 16384 instances")
    ap.add_argument("--cpu-iters", type=int, default=40000, help="iterations of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--warmup-max-iter", type=int, default=3000,
                    help="iterations of a warm-up step (0 = same as a timed step); a warm-up only has to page in the "
                         "code objects and allocations, a full-size pass takes ~40 s")
    ap.add_argument("--max-seconds", type=float, default=float(os.environ.get("RRTX_BENCH_MAX_SECONDS", "1500")),
                    help="stop timing further steps once this much time has been spent (the JSON reports the steps "
                         "actually timed)")
    a = ap.parse_args()

    import numpy as np
    import util
    c3 = a.workload == "c3"
    c6 = a.workload == "c6"
    c5 = a.workload == "c5" or c6     # c6 shares c5's reporting; only the planner and its constants differ
    if a.obstacles is None:
        a.obstacles = 200 if c3 else ((7 if c6 else 6) if c5 else 50)
    if c6:
        if "--max-iter" not in sys.argv:
            a.max_iter = 750
        if "--instances" not in sys.argv:
            a.instances = 16384     # four rounds of 16 waves per CU
        a.warmup_max_iter = 0
    if c5 and not c6:
        if "--max-iter" not in sys.argv:
            a.max_iter = 5000
        if "--instances" not in sys.argv:
            a.instances = 1536      # two full rounds of 3 workgroups per CU
    if c3:
        if "--max-iter" not in sys.argv:
            a.max_iter = 20000
        if "--instances" not in sys.argv:
            a.instances = 1024     # 4 workgroups of 256 threads per CU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if a.gpus > 1 or world > 1 or os.environ.get("RRTX_BENCH_FORCE_DIST"):
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    ngpu = max(world, 1)
    device = local_rank

    import importlib
    import rrt_amd
    sharding = importlib.import_module("robotics-path-planning_amd.sharding")
    A = rrt_amd._abi
    if a.workload == "c4":
        bench_c4(a, A, sharding, rrt_amd, dist, torch, rank, local_rank, ngpu)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    kw = util.c2_kwargs(a.max_iter, m=a.obstacles)
    if c3:
        kw = dict(algo="informed", start=[2, 2], goal=[98, 98], obstacles=util.synth_map(11, a.obstacles, 0.3, 1.5),
                  rand_area=[0, 100], expand_dis=0.5, goal_sample_rate=10, max_iter=a.max_iter, sobol=1)
    if c5:   # rrt_05 driver constants (rrt_05:1804-1859)
        kw = dict(algo="dubins", start=[0.0, 0.0, 0.0], goal=[10.0, 10.0, 0.0],
                  obstacles=[(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2)], rand_area=[-2, 15],
                  expand_dis=3.0, goal_sample_rate=10, max_iter=a.max_iter)
    if c6:   # rrt_06 driver constants (rrt_06:2012-2083)
        kw = dict(algo="rs", start=[0.0, 0.0, 0.0], goal=[10.0, 9.0, 0.0],
                  obstacles=[(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2), (8, 10, 1)],
                  rand_area=[-2, 15], expand_dis=3.0, goal_sample_rate=10, max_iter=a.max_iter)
    B = a.instances
    seeds = sharding.shard_seeds(rank, B)          # rank r owns seeds r*B+1 .. (r+1)*B, no exchange while planning
    cuda = torch.device("cuda", local_rank) if dist is not None else None

    def make_handle(max_iter):
        if c6:
            h = A.Handle(A.ALGO_RS, kw["start"], kw["goal"], kw["rand_area"], 3.0, 0.5, 10, max_iter, robot_radius=0.6,
                         connect_circle_dist=50.0, search_until_max_iter=True, n_instances=B, device=device,
                         curvature=2.0, goal_yaw_th=float(np.deg2rad(1.0)), goal_xy_th=0.5, step_size=0.1)
        elif c5:
            h = A.Handle(A.ALGO_DUBINS, kw["start"], kw["goal"], kw["rand_area"], 3.0, 0.5, 10, max_iter, robot_radius=0.0,
                         connect_circle_dist=50.0, search_until_max_iter=True, n_instances=B, device=device,
                         curvature=1.0, goal_yaw_th=float(np.deg2rad(1.0)), goal_xy_th=0.5)
        elif c3:
            c_min, c = rrt_amd.informed_rotation(kw["start"], kw["goal"])
            h = A.Handle(A.ALGO_INFORMED, kw["start"], kw["goal"], kw["rand_area"], kw["expand_dis"], 1.0,
                         kw["goal_sample_rate"], max_iter, sampler=A.SAMPLER_SOBOL, n_instances=B, device=device,
                         informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
        else:
            h = A.Handle(A.ALGO_RRT_STAR, kw["start"], kw["goal"], kw["rand_area"], kw["expand_dis"],
                         kw["path_resolution"], kw["goal_sample_rate"], max_iter, play_area=None,
                         robot_radius=kw["robot_radius"], sampler=A.SAMPLER_MT,
                         connect_circle_dist=kw["connect_circle_dist"], search_until_max_iter=True, n_instances=B,
                         device=device)
        h.set_obstacles(kw["obstacles"])
        h.seed_instances(seeds)
        return h

    def sync_all():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    h = make_handle(a.max_iter)
    hw = h
    if a.warmup_max_iter and a.warmup_max_iter != a.max_iter:
        hw = make_handle(a.warmup_max_iter)
    for _ in range(a.warmup):
        hw.plan()
    if hw is not h:
        hw.close()

    sync_all()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    alg_bytes = alg_bytes2 = 0
    edges_u = edges_r = iters = launches = 0
    steps_done = 0
    for _ in range(a.steps):
        ts = time.perf_counter()
        h.plan()                      # blocking: returns after the last kernel of the batch has finished
        steps_done += 1
        s = h.get_stats()
        kernel_ms += s["kernel_ms"]
        alg_bytes += s["algorithmic_bytes"]
        alg_bytes2 += s["algorithmic_bytes_two_scan"]
        edges_u += s["edges_unique"]
        edges_r += s["edges_ref"]
        iters += s["iterations"]
        launches += s["launches"]
        # time guard (all ranks take the same decision: the slowest rank's clock decides)
        spent, last = time.perf_counter() - t0, time.perf_counter() - ts
        go = sharding.all_agree_min(dist, 1 if (spent + last <= a.max_seconds) else 0, cuda)
        if not go:
            break
    sync_all()
    dt = time.perf_counter() - t0
    pc, nn, st = h.get_results()
    stats = h.get_stats()

    # ---- cross-rank: max time, summed work, RCCL gather of the result table (the only data collective)
    tmax = sharding.reduce_max(dist, dt, cuda)
    tot_edges_u, tot_edges_r = sharding.reduce_sum_int(dist, [edges_u, edges_r], cuda)
    all_pc, all_nn, all_st = sharding.gather_results(dist, pc, nn, st, cuda)

    if rank == 0:
        finite = np.isfinite(all_pc)
        value = tot_edges_u / tmax
        # roofline.achieved = bytes the algorithm as implemented must move (ONE pass over the node arrays per iteration
        # serves both the near-ball query of iteration i and the nearest query of i+1: 16*n + 48*k + 24*M + 28 per
        # iteration) / HIP-event time of the planner kernel.  SURVEY.md 8(d) wrote the formula for two separate scans
        # (32*n + ...): that figure is reported next to it as `survey_8d_two_scan_*`.
        achieved = (alg_bytes / 1e9) / (kernel_ms / 1e3) if kernel_ms > 0 else 0.0
        achieved_2s = (alg_bytes2 / 1e9) / (kernel_ms / 1e3) if kernel_ms > 0 else 0.0
        traffic = None
        traffic_note = None
        # PMC traffic of the same workload and kernel variant, measured in separate rocprofv3 --pmc passes
        # (tools/profile_headline.sh -> profiles/*_traffic.json)
        variant = "f64" if os.environ.get("RRTX_F32", "1") == "0" else (
            "f32_mirror" if os.environ.get("RRTX_Q16", "1") == "0" else "q16_mirror")
        for tf in ("r1_q16_traffic.json", "r1_t64_traffic.json", "r1_f32_traffic.json", "r1_traffic.json"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", tf)))
            except Exception:  # noqa: BLE001
                continue
            tc = tj["config"]
            if tc["instances_per_gpu"] == B and tc["max_iter"] == a.max_iter and tc["obstacles"] == a.obstacles \
                    and tc.get("variant", "f64") == variant and not (c3 or c5):
                traffic = tj["hbm_bytes_per_step"] / 1e9 / (kernel_ms / 1e3 / max(steps_done, 1))
                traffic_note = "HBM bytes/step %.4g from profiles/%s" % (tj["hbm_bytes_per_step"], tf)
                break
        line = {
            # BASELINE.json: "RRT* edge expansions/sec + final path cost, 100k-node tree"; the path cost is reported in
            # final_path_cost_mean / _min below
            "metric": "%s edge expansions/sec + final path cost, %s tree (collision-checked edges evaluated on the "
                      "device, distinct per iteration; %d iterations)"
                      % ("RRT*-Reeds-Shepp" if c6 else "RRT*-Dubins" if c5 else ("Informed RRT*" if c3 else "RRT*"),
                         "100k-node" if (not c3 and not c5 and a.max_iter >= 100000) else "%d-iteration" % a.max_iter,
                         a.max_iter),
            "value": value, "unit": "edge expansions/s", "n_gpus": ngpu, "steps": steps_done, "warmup": a.warmup,
            "ms_per_step": 1e3 * tmax / max(steps_done, 1), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("C6: rrt_06 RRT*-Reeds-Shepp, driver constants (%d obstacles, 17x17 area, curvature 2, "
                                    "step_size 0.1, robot_radius 0.6), max_iter %d, %d instances/GPU (seeds 1..)" if c6 else
                                    "C5: rrt_05 RRT*-Dubins, driver constants (%d obstacles, 17x17 area, curvature 1), "
                                    "max_iter %d, %d instances/GPU (seeds 1..)" if c5 else
                                    "C3: rrt_07 Informed RRT*, Sobol sampler, %d circle obstacles (map_seed 11, radii "
                                    "U(0.3,1.5)) on 100x100, expand_dis 0.5, max_iter %d, %d instances/GPU (seeds 1..)"
                                    if c3 else
                                    "C2: rrt_04 RRT*, %d circle obstacles (map_seed 7) on 100x100, expand_dis 2.0, "
                                    "path_resolution 0.25, max_iter %d, %d instances/GPU (seeds 1..), MT sampler")
                                   % (a.obstacles, a.max_iter, B),
                       "instances_per_gpu": B, "max_iter": a.max_iter, "parallelism": "instances x%d" % ngpu},
            "edge_expansions_reference_equivalent_per_s": tot_edges_r / tmax,
            "mean_nodes_per_tree": float(np.mean(all_nn)),
            "final_path_cost_mean": float(np.mean(all_pc[finite])) if finite.any() else None,
            "final_path_cost_min": float(np.min(all_pc[finite])) if finite.any() else None,
            "paths_found": int(finite.sum()), "instances_total": int(len(all_pc)),
            "iterations_per_s": iters * ngpu / tmax,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": "rppr::rrt_rs_kernel" if c6 else "rppd::rrt_dubins_kernel" if c5 else (
                             "rppi::rrt_informed_kernel" if c3 else "rppk2(s)::rrt_star_kernel_v2"),
                         "launches": launches,
                         "algorithmic_bytes_per_step": alg_bytes / max(steps_done, 1),
                         "traffic_note": traffic_note,
                         "survey_8d_two_scan_GBps": achieved_2s, "survey_8d_two_scan_frac": achieved_2s / 8000.0,
                         "note": "achieved = single-pass algorithmic bytes (4*n per iteration from the 16-bit coordinate mirror, 8*n with RRTX_Q16=0 (f32 mirror), 16*n with RRTX_F32=0: the near pass of "
                                 "iteration i also answers the nearest query of i+1) / kernel time; "
                                 "survey_8d_two_scan_* applies SURVEY 8(d)'s two-scan formula (32*n) to the same run",
                         "kernel_ms_per_step": kernel_ms / max(steps_done, 1)},
            "steps_requested": a.steps, "warmup_max_iter": a.warmup_max_iter or a.max_iter,
            "near_unique_max": stats.get("near_unique_max"), "f32_fallbacks_last_step": stats.get("f32_fallbacks"),
            "q16_fallbacks_last_step": stats.get("q16_fallbacks"),
            "exact_rescans_last_step": stats.get("exact_rescans"),
        }
        if c5 and not c6:
            line["plans_per_s"] = len(all_pc) * steps_done / tmax
            line["edges_steered_per_iteration"] = tot_edges_u / max(iters * ngpu, 1)
            line["value_note"] = ("`value` counts the Dubins edges the device steers.  The default build steers only the "
                                  "candidates that can change the result (DESIGN.md 5.5, filtered candidate stages: same "
                                  "trees, ~8x fewer edges); RRTX_DUBINS_FILTER=0 steers every near candidate as the "
                                  "reference and the CPU baseline do -- compare iterations_per_s / plans_per_s across "
                                  "builds, not the edge rates")
        if c6:
            line["plans_per_s"] = len(all_pc) * steps_done / tmax
            line["edges_steered_per_iteration"] = tot_edges_u / max(iters * ngpu, 1)
            line["roofline"]["note"] = (
                "rrt_06's iteration is closed-form f64 trigonometry (48 Reeds-Shepp word evaluations per steer, glibc-exact "
                "sin/cos/atan2/acos/asin/pow replicas), not a stream: the kernel is f64-VALU / latency bound and its HBM "
                "figure (16 B per node and scan + 24 B per polyline point written) is far below the HBM roof by "
                "construction.  `value` counts the edges the device steers (lazy candidate order, DESIGN.md 5.7); the "
                "reference and the oracle steer every near candidate (RRTX_RS_EAGER=1 does the same on the device), so "
                "compare plans_per_s with cpu_baseline.plans_per_s, not the edge rates")
        if not a.no_cpu_baseline and ngpu == 1:   # the CPU baseline is reported by the single-GPU run only
            import oracle
            tc = time.perf_counter()
            if c6:
                kc = dict(max_iter=a.max_iter)
                ncpu = 64
                r = None
                eu = er = 0
                for sd in range(1, ncpu + 1):
                    r = oracle.plan_rrt_rs(kw["start"], kw["goal"], kw["obstacles"], kw["rand_area"], a.max_iter, seed=sd,
                                           curvature=2.0, robot_radius=0.6, step_size=0.1)
                    eu += r["stats"]["edges_unique"]
                    er += r["stats"]["edges_ref"]
                r["stats"]["edges_unique"], r["stats"]["edges_ref"] = eu, er
            elif c5:
                kc = dict(max_iter=min(a.cpu_iters, a.max_iter))
                r = oracle.plan_dubins(kw["start"], kw["goal"], kw["obstacles"], kw["rand_area"], kc["max_iter"], seed=1)
            elif c3:
                kc = dict(kw)
                kc.pop("algo")
                kc["max_iter"] = min(a.cpu_iters, a.max_iter)
                r = oracle.plan_informed(seed=1, exact_pow=False, **kc)
            else:
                kc = util.c2_kwargs(a.cpu_iters, m=a.obstacles)
                r = oracle.plan(seed=1, exact_pow=False, **kc)
            tc = time.perf_counter() - tc
            if c6:
                line["cpu_baseline"] = {"value": r["stats"]["edges_unique"] / tc, "unit": "edge expansions/s", "cores": 1,
                                        "kind": "port", "plans_per_s": ncpu / tc,
                                        "sample": "oracle/rrt_oracle.c (C restatement pinned to the reference), the first %d "
                                                  "instances (seeds 1..%d) of the same workload one after the other, %.1f s; "
                                                  "every near candidate steered as the reference does"
                                                  % (ncpu, ncpu, tc)}
                print(json.dumps(line), flush=True)
                h.close()
                if dist is not None:
                    dist.barrier()
                    dist.destroy_process_group()
                return
            line["cpu_baseline"] = {"value": r["stats"]["edges_unique"] / tc, "unit": "edge expansions/s", "cores": 1,
                                    "kind": "port",
                                    "sample": "oracle/rrt_oracle.c (C restatement pinned to the reference), 1 instance, "
                                              "seed 1, %d iterations (%d nodes) of the same workload, %.1f s; "
                                              "reference-equivalent rate %.0f/s"
                                              % (kc["max_iter"], len(r["x"]), tc, r["stats"]["edges_ref"] / tc)}
        print(json.dumps(line), flush=True)
    h.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_c4(a, A, sharding, rrt_amd, dist, torch, rank, local_rank, ngpu):
    """C4: rrt_08 BIT*, driver constants (rrt_08:633-665); instance i has start/goal drawn from
    random.Random(2000+i) in [-1,14]^2 outside the obstacles and planner seed 1000+i; rank r owns instances
    r*B .. (r+1)*B-1.  One step = every instance planned (maxIter 80).  Unit of work: edges popped from the edge queue."""
    import random
    import numpy as np
    obst = [(5, 5, 0.5), (9, 6, 1), (7, 5, 1), (1, 5, 1), (3, 6, 1), (7, 9, 1)]
    B = a.instances if "--instances" in sys.argv else 4096
    max_iter = a.max_iter if "--max-iter" in sys.argv else 80
    cuda = torch.device("cuda", local_rank) if dist is not None else None

    def free_point(rng):
        while True:
            x, y = rng.uniform(-1, 14), rng.uniform(-1, 14)
            if all((x - ox) ** 2 + (y - oy) ** 2 > r ** 2 for ox, oy, r in obst):
                return [x, y]
    starts, goals, seeds = [], [], []
    for i in range(rank * B, (rank + 1) * B):
        rng = random.Random(2000 + i)
        starts.append(free_point(rng))
        goals.append(free_point(rng))
        seeds.append(1000 + i)
    c_min, c = rrt_amd.bitstar_rotation(starts[0], goals[0])
    h = A.Handle(A.ALGO_BITSTAR, starts[0], goals[0], [-2.0, 15.0], 2.0, 1.0, 0, max_iter, n_instances=B,
                 device=local_rank, informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
    h.set_obstacles(obst)
    h.seed_instances(seeds)
    for i in range(B):
        cm, ci = rrt_amd.bitstar_rotation(starts[i], goals[i])   # numpy SVD on the host, as the reference does
        h.set_instance(i, starts[i], goals[i])
        h.set_instance_rotation(i, [ci[0, 0], ci[0, 1], ci[1, 0], ci[1, 1]], cm)
    def plan_tolerant():
        # an instance whose start is walled in keeps drawing sample batches without bound (the reference would too);
        # it ends with RRTX_E_OVERFLOW in its status word and is reported below, the other instances are unaffected
        try:
            h.plan()
        except A.RrtxError as e:
            if "OVERFLOW" not in str(e):
                raise

    for _ in range(a.warmup):
        h.seed_instances(seeds)
        plan_tolerant()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    edges = 0
    for _ in range(a.steps):
        h.seed_instances(seeds)
        plan_tolerant()
        s = h.get_stats()
        kernel_ms += s["kernel_ms"]
        edges += s["edges_unique"]
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pc, nn, st = h.get_results()
    tmax = sharding.reduce_max(dist, dt, cuda)
    (tot_edges,) = sharding.reduce_sum_int(dist, [edges], cuda)
    all_pc, all_nn, all_st = sharding.gather_results(dist, pc, nn, st, cuda)
    if rank == 0:
        found = (all_st & 2) != 0
        line = {"metric": "BIT* edge-queue expansions/sec (edges popped and processed, rrt_08:262-318), %d-iteration plans"
                          % max_iter,
                "value": tot_edges / tmax, "unit": "edge expansions/s", "n_gpus": ngpu, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": 1e3 * tmax / max(a.steps, 1), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "C4: rrt_08 BIT*, driver constants (6 obstacles, rand_area [-2,15], maxIter %d), "
                                       "per-instance start/goal from random.Random(2000+i), planner seed 1000+i, "
                                       "%d instances/GPU" % (max_iter, B),
                           "instances_per_gpu": B, "max_iter": max_iter, "parallelism": "instances x%d" % ngpu},
                "plans_per_s": B * ngpu * a.steps / tmax, "paths_found": int(found.sum()),
                "instances_capacity_exceeded": int(((all_st & 4) != 0).sum()),
                "instances_total": int(len(all_pc)), "mean_vertices_per_tree": float(np.mean(all_nn)),
                "roofline": {"bound": "hbm", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None,
                             "traffic": None, "kernel": "rppb::bitstar_kernel",
                             "note": "instance-parallel sequential search (one lane per instance): latency-bound, "
                                     "no streaming pass to price against HBM",
                             "kernel_ms_per_step": kernel_ms / max(a.steps, 1)}}
        if not a.no_cpu_baseline and ngpu == 1:
            import oracle
            nsmp = min(B, 48)
            tc = time.perf_counter()
            ce = 0
            for i in range(nsmp):
                r = oracle.plan_bitstar(starts[i], goals[i], obst, [-2, 15], max_iter, seed=seeds[i])
                ce += int(r.get("n_trace", 0)) if isinstance(r, dict) else 0
            tc = time.perf_counter() - tc
            line["cpu_baseline"] = {"value": nsmp / tc, "unit": "plans/s", "cores": 1, "kind": "port",
                                    "sample": "oracle/rrt_oracle.c, the first %d instances of the same workload, %.1f s "
                                              "(compare with plans_per_s)" % (nsmp, tc)}
        print(json.dumps(line), flush=True)
    h.close()


if __name__ == "__main__":
    main()
